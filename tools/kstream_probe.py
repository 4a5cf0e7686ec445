#!/usr/bin/env python3
"""k_stream alone over one resident block, for several builds of libbvcf (probe builds whose results are not valid:
-DBVCF_EXP_NOSTORE, -DBVCF_EXP_HV2, ... -- the chain's other kernels are not what is measured, and NOSTORE builds stop the
chain behind k_stream).  Not a test.
    python tools/kstream_probe.py [c3|c4] libA.so libB.so ...   (three rounds, interleaved; medians of the kernel's HIP-event times)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = {"c3": 311_296, "c4": 262_144, "c5": 98_304}

if os.environ.get("KPROBE_CHILD"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import benchgen as bg
    import bystro_vcf_amd as bv
    prof = os.environ["KPROBE_CHILD"]
    cfg = bg.make_cfg(prof)
    rows = ROWS[prof]
    ns = cfg.n_samples
    blocks = [bg.rows_device(cfg, b * rows, rows, pad=bv.DEVICE_PAD) for b in range(4)]
    stride = ((ns + 3) // 4 + 15) & ~15
    n_alt = rows * 4 + 1024
    nbytes = max(n for _, n in blocks)
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=n_alt,
                 cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
    ptrs, sizes = [t.data_ptr() for t, _ in blocks], [n for _, n in blocks]
    bv.lib.bvcf_bench_device_slots.restype = int
    import ctypes as C
    def run(iters):
        n = len(ptrs)
        chain, scan, counts = (C.c_float * iters)(), (C.c_float * iters)(), (C.c_uint64 * 5)()
        bv.lib.bvcf_bench_device_slots(ctx.h, (C.c_void_p * n)(*ptrs), (C.c_size_t * n)(*sizes), n, iters, 1, chain, scan, counts)  # (rc ignored: counts may be garbage)
        return list(scan)
    run(4)
    s = run(12)
    print(json.dumps({"ms": statistics.median(s), "min": min(s), "GB": sum(sizes) / len(sizes) / 1e9}))
    sys.exit(0)

args = sys.argv[1:]
prof = args.pop(0) if args and args[0] in ROWS else "c3"
res = {l: [] for l in args}
for rnd in range(3):
    for l in args:
        env = dict(os.environ, BVCF_LIB=os.path.abspath(l), KPROBE_CHILD=prof)
        out = subprocess.check_output([sys.executable, os.path.abspath(__file__)], env=env, stderr=subprocess.DEVNULL)
        res[l].append(json.loads(out.decode().strip().splitlines()[-1]))
for l in args:
    ms = statistics.median(r["ms"] for r in res[l])
    print("%-36s %s k_stream median %.4f ms (min %.4f) = %.2f TB/s" % (os.path.basename(l), prof, ms, min(r["min"] for r in res[l]), res[l][0]["GB"] / ms))
