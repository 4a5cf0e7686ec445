#!/usr/bin/env python3
"""Where does the first CLI run after the e2e file is written spend its extra seconds?  (not a test)

Round 3's bench saw `bystro-vcf --in /dev/shm/file` take 4.3-5.7 s the first time and 1.9-2.0 s the second, with the
bench process holding its ctx and 25 GB of HBM both times.  This writes the same file the same way and runs the CLI
under a few conditions, printing every run's BVCF_TIMING stage split and what /proc/vmstat counted meanwhile:

    A  right after the file is written, this process still holding its blocks in HBM   (round 3's first run)
    A2 again
    B  after `cat file > /dev/null`
    C  after this process has freed its device memory

    python tools/cold_start.py [rows=6200000]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

CLI = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
KEYS = ("pgfault", "pgmajfault", "numa_pages_migrated", "pgmigrate_success", "thp_fault_alloc", "thp_collapse_alloc",
        "compact_stall", "pgscan_direct", "pgsteal_direct", "pgscan_kswapd", "pgsteal_kswapd", "allocstall_normal",
        "allocstall_movable", "pgalloc_normal", "pgfree", "numa_hint_faults", "thp_file_alloc", "pswpout")


def vmstat():
    out = {}
    for ln in open("/proc/vmstat"):
        k, v = ln.split()
        if k in KEYS:
            out[k] = int(v)
    return out


def meminfo():
    want = ("MemFree", "MemAvailable", "Cached", "Shmem", "ShmemHugePages", "AnonHugePages")
    return {ln.split(":")[0]: ln.split()[1] for ln in open("/proc/meminfo") if ln.split(":")[0] in want}


def run(tag, path):
    v0 = vmstat()
    t0 = time.perf_counter()
    with open(os.devnull, "wb") as out:
        p = subprocess.run([CLI, "--in", path], stdout=out, stderr=subprocess.PIPE, env=dict(os.environ, BVCF_TIMING="json"))
    wall = time.perf_counter() - t0
    v1 = vmstat()
    st = {}
    for ln in p.stderr.decode(errors="replace").splitlines():
        if ln.startswith("[bvcf timing-json] "):
            st = json.loads(ln[len("[bvcf timing-json] "):])
    keep = ("total_s", "init_s", "warmup_max_s", "ctx_create_max_s", "first_submit_at_s", "steady_s", "wait_for_reader_s", "reader_busy_max_s",
            "gpu_wait_max_s", "formatter_busy_s", "major_faults", "minor_faults", "user_cpu_s", "system_cpu_s")
    print("%-3s rc %d wall %.3f s  %s" % (tag, p.returncode, wall, {k: round(st.get(k, -1), 3) for k in keep}), flush=True)
    print("    vmstat delta:", {k: v1[k] - v0[k] for k in v0 if v1[k] != v0[k]}, flush=True)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_200_000
    cfg = bg.make_cfg("c3")
    per = 311_296
    n_blk = -(-rows // per)
    blocks = []
    path = "/dev/shm/bvcf_cold_%d.vcf" % os.getpid()
    stage = torch.empty(1 << 28, dtype=torch.uint8, pin_memory=True)
    print("meminfo before:", meminfo(), flush=True)
    t0 = time.perf_counter()
    try:
        with open(path, "wb") as f:
            f.write(bg.header(cfg))
            for b in range(n_blk):
                t, nb = bg.rows_device(cfg, b * per, per, pad=bv.DEVICE_PAD)
                if b < 8:
                    blocks.append(t)  # (as the bench: eight resident blocks, 25 GB of HBM)
                for off in range(0, nb, stage.numel()):
                    n = min(stage.numel(), nb - off)
                    stage[:n].copy_(t[off:off + n])
                    f.write(memoryview(stage[:n].numpy()))
        print("file: %d rows, %.1f GB written in %.1f s" % (n_blk * per, os.path.getsize(path) / 1e9, time.perf_counter() - t0), flush=True)
        print("meminfo after writing:", meminfo(), flush=True)
        run("A", path)
        run("A2", path)
        subprocess.run(["cat", path], stdout=subprocess.DEVNULL)
        run("B", path)
        blocks.clear()
        del stage
        torch.cuda.empty_cache()
        run("C", path)
        run("C2", path)
        print("meminfo at the end:", meminfo(), flush=True)
        # the control: is it the CLI, or the first read of a freshly written tmpfs file by anybody?  A second file, read
        # twice by `cat` (one thread, read() into a 128 KiB buffer)
        os.unlink(path)
        with open(path, "wb") as f:
            for b in range(6):
                t, nb = bg.rows_device(cfg, b * per, per, pad=bv.DEVICE_PAD)
                host = t[:nb].cpu().numpy()
                f.write(memoryview(host))
        for tag in ("cat 1st read", "cat 2nd read"):
            v0 = vmstat()
            r0 = os.times()
            t0 = time.perf_counter()
            subprocess.run(["cat", path], stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            r1 = os.times()
            v1 = vmstat()
            print("%s: %.1f GB in %.3f s = %.2f GB/s, children system CPU %.2f s   vmstat delta: %s" % (
                tag, os.path.getsize(path) / 1e9, dt, os.path.getsize(path) / dt / 1e9, r1.children_system - r0.children_system,
                {k: v1[k] - v0[k] for k in v0 if v1[k] != v0[k]}), flush=True)
    finally:
        if os.path.exists(path):
            os.unlink(path)


if __name__ == "__main__":
    main()
