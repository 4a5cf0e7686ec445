#!/usr/bin/env python3
"""End-to-end CLI timing on compressed input (not a test): text vs gzip vs BGZF of the same rows."""
import gzip
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benchgen as bg  # noqa: E402
import bgzf  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
cfg = bg.make_cfg("c3")
base = "/tmp/bvcf_gz_%d" % rows
if not os.path.exists(base + ".vcf"):
    with open(base + ".vcf", "wb") as f, open(base + ".bgzf", "wb") as fb, gzip.open(base + ".gz", "wb", 1) as fg:
        hdr = bg.header(cfg)
        f.write(hdr)
        fg.write(hdr)
        fb.write(bgzf.bgzf_compress(hdr, eof_marker=False, level=1))
        for first in range(0, rows, 5_000):
            chunk = bg.rows_host(cfg, first, min(5_000, rows - first))
            f.write(chunk)
            fg.write(chunk)
            fb.write(bgzf.bgzf_compress(chunk, eof_marker=False, level=1))
        fb.write(bgzf.bgzf_block(b""))
exe = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
md5s = set()
for ext in (".vcf", ".gz", ".bgzf"):
    path = base + ext
    subprocess.run(["cat", path], stdout=subprocess.DEVNULL)
    t0 = time.perf_counter()
    p = subprocess.run([exe, "--in", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    import hashlib
    md5s.add(hashlib.md5(p.stdout).hexdigest())
    print("%-5s %8.1f MB  rc=%d  %.2f s  %.0f variants/s" % (ext, os.path.getsize(path) / 1e6, p.returncode, dt, rows / dt))
t0 = time.perf_counter()
subprocess.run("gzip -dc %s.gz > /dev/null" % base, shell=True)
print("gzip -dc alone: %.2f s" % (time.perf_counter() - t0))
print("outputs identical:", len(md5s) == 1)
