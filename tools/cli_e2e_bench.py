#!/usr/bin/env python3
"""End-to-end CLI timing helper (not a test): synthetic c3 VCF on disk -> bystro-vcf -> /dev/null,
next to the oracle CLI on the same file.  Usage: python tests/cli_e2e_bench.py [rows] [profile]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benchgen as bg  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
profile = sys.argv[2] if len(sys.argv) > 2 else "c3"
cfg = bg.make_cfg(profile)
path = "/tmp/bvcf_e2e_%s_%d.vcf" % (profile, rows)
if not os.path.exists(path):
    with open(path, "wb") as f:
        f.write(bg.header(cfg))
        for first in range(0, rows, 10_000):
            f.write(bg.rows_host(cfg, first, min(10_000, rows - first)))
size = os.path.getsize(path)
subprocess.run(["cat", path], stdout=subprocess.DEVNULL)  # page cache
for name, cmd in (("hip ", [os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf"), "--in", path]),
                  ("orc ", [os.path.join(ROOT, "oracle", "bvcf_oracle"), "--in", path, "--threads", str(os.cpu_count())])):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print("%s rc=%d %.2fs  %.0f variants/s  %.2f GB/s  out=%d bytes md5=%s" % (
        name, p.returncode, dt, rows / dt, size / dt / 1e9, len(p.stdout),
        __import__("hashlib").md5(p.stdout).hexdigest()))
