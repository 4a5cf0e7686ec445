#!/usr/bin/env python3
"""End-to-end CLI timing: synthetic rows -> file -> bystro-vcf (HIP) -> /dev/null, with the stage times of
BVCF_TIMING, and (optionally) an md5 comparison with the oracle CLI on the same file.

    python tools/e2e_cli.py [rows=400000] [profile=c3] [--check] [--dosage] [--runs=N] [--keep=PATH] [--bgzf] [--samples=N] [--pipe] [--json] [--devices=LIST] [--batchMB=N] [--threads=N]

Needs a GPU box.  The file is written to /dev/shm when it fits there, else /tmp, and removed afterwards.
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benchgen as bg  # noqa: E402


# zlib level of --bgzf: 6 is what bgzip writes by default (--level=1: half the ratio, twice the symbols)
LEVEL = int(([a.split('=')[1] for a in sys.argv if a.startswith('--level=')] or ['6'])[0])


def _bgzf_part(args):
    import bgzf
    path, off, n = args
    with open(path, "rb") as f:
        f.seek(off)
        return bgzf.bgzf_compress(f.read(n), level=LEVEL, eof_marker=False)


def bgzf_file(src, dst, part=0xFF00 * 256):
    """BGZF-compress src into dst with a process pool (test helper speed, not bgzip's)"""
    import multiprocessing as mp
    import bgzf
    size = os.path.getsize(src)
    jobs = [(src, off, min(part, size - off)) for off in range(0, size, part)]
    with mp.Pool(min(16, os.cpu_count() or 1)) as pool, open(dst, "wb") as out:
        for blob in pool.imap(_bgzf_part, jobs):
            out.write(blob)
        out.write(bgzf.bgzf_block(b""))

CLI = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
ORACLE = os.path.join(ROOT, "oracle", "bvcf_oracle")


def md5_of(cmd, path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        p = subprocess.Popen(cmd, stdin=f, stdout=subprocess.PIPE)
        for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
            h.update(chunk)
        p.wait()
    return p.returncode, h.hexdigest()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = int(args[0]) if args else 400_000
    profile = args[1] if len(args) > 1 else "c3"
    check = "--check" in sys.argv
    over = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--samples=")]
    cfg = bg.make_cfg(profile, **({"n_samples": over[0]} if over else {}))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(base, "bvcf_e2e_%d.vcf" % os.getpid())
    keep = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--keep=")]
    if keep:
        path = keep[0]
    try:
        t0 = time.perf_counter()
        if not (keep and os.path.exists(path)):
            import torch  # rows come from the device generator (the host one is a single thread: 50 MB/s)
            with open(path, "wb") as f:
                f.write(bg.header(cfg))
                step = max(1, min(200_000, (2 << 30) // (4 * max(cfg.n_samples, 1) + 200)))  # ~2 GB at a time
                for first in range(0, rows, step):
                    t, nb = bg.rows_device(cfg, first, min(step, rows - first))
                    f.write(memoryview(t[:nb].cpu().numpy()))
                    del t
        size = os.path.getsize(path)
        text_path = path
        if "--bgzf" in sys.argv:
            t1 = time.perf_counter()
            bgzf_file(path, path + ".gz")
            print("bgzf: %.3f GB (%.1f s to compress)" % (os.path.getsize(path + ".gz") / 1e9, time.perf_counter() - t1), flush=True)
            path = path + ".gz"
        print("file: %d rows, %.2f GB in %s (%.1f s to generate)" % (rows, size / 1e9, base, time.perf_counter() - t0), flush=True)
        env = dict(os.environ, BVCF_TIMING="json" if "--json" in sys.argv else "1")
        extra = []
        for a in sys.argv:
            if a.startswith("--devices="):
                extra += ["--devices", a.split("=", 1)[1]]
            if a.startswith("--batchMB="):
                extra += ["--batchMB", a.split("=", 1)[1]]
        if "--dosage" in sys.argv:
            extra = ["--dosageOutput", path + ".arrow"]
        runs = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--runs=")]
        for it in range(runs[0] if runs else 3):
            t0 = time.perf_counter()
            with open(path, "rb") as f, open("/dev/null", "wb") as out:
                if "--pipe" in sys.argv:  # stdin is a pipe, as in `pigz -dc in.vcf.gz | bystro-vcf`
                    cat = subprocess.Popen(["cat", path], stdout=subprocess.PIPE)
                    p = subprocess.run([CLI] + extra, stdin=cat.stdout, stdout=out, stderr=subprocess.PIPE, env=env)
                    cat.wait()
                else:
                    p = subprocess.run([CLI] + extra, stdin=f, stdout=out, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t0
            tl = [l for l in p.stderr.decode().splitlines() if "timing" in l]
            print("run %d: rc %d, %.3f s wall = %.2f M variants/s, %.1f GB/s   %s" %
                  (it, p.returncode, dt, rows / dt / 1e6, size / dt / 1e9, tl[-1] if tl else ""), flush=True)
        if check:
            rc_g, m_g = md5_of([CLI], path)
            rc_o, m_o = md5_of([ORACLE], text_path)
            print("md5 hip %s (rc %d)  oracle %s (rc %d)  %s" % (m_g, rc_g, m_o, rc_o, "IDENTICAL" if m_g == m_o else "DIFFERENT"))
            if m_g != m_o:
                sys.exit(1)
    finally:
        for q in (text_path, text_path + ".gz", path + ".arrow"):
            if os.path.exists(q) and not (keep and q in (text_path, text_path + ".gz")):
                os.unlink(q)


if __name__ == "__main__":
    main()
