#!/usr/bin/env python3
"""Where do k_stream_gen's cycles go, tier by tier?  (needs a -DBVCF_EXP_TIMES build: BVCF_LIB=...; not a test)
One block of the c5 / c5h row model through the chain, then the per-wave cycle sums the kernel left behind:
    BVCF_LIB=bystro-vcf_amd/libbvcf_times.so python tools/gen_tiers.py c5 c5h"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["BVCF_GEN_STREAM"] = "1"
import numpy as np  # noqa: E402
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

names = ["loop top: wait, ring slot, flags", "packed-flag tier", "medium tier", "exact handler (without line ends)", "line ends (finish_line)"]
for prof in sys.argv[1:] or ["c5", "c5h"]:
    cfg = bg.make_cfg(prof)
    rows = 98_304
    t, nbytes = bg.rows_device(cfg, 0, rows, pad=bv.DEVICE_PAD)
    ns = cfg.n_samples
    stride = ((ns + 3) // 4 + 15) & ~15
    n_alt = rows * 4 + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=n_alt,
                 cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
    ctx.bench_device([t.data_ptr()], [nbytes], 3, slots=1)
    chain, scan, counts = ctx.bench_device([t.data_ptr()], [nbytes], 1, slots=1)
    pb = (C.c_ulonglong * (8 * 32768))()
    bv.lib.bvcf_debug_phase_times.argtypes = [C.c_void_p, C.c_int]
    bv.lib.bvcf_debug_phase_times(pb, 8 * 32768)
    ph = np.frombuffer(pb, dtype=np.uint64).reshape(8, 32768).astype(np.float64)
    busy = ph[5:8].sum(0) > 0
    tot = ph[:5, busy].sum()
    n_chunks = ph[5:8, busy].sum()
    print("== %s: %d waves with work, k_stream_gen %.4f ms (stamped build), %.0f chunks, %.1f cycles per chunk and wave" % (
        prof, busy.sum(), scan[0], n_chunks, tot / n_chunks))
    for k in range(5):
        print("  %-36s %5.1f %% of the cycles" % (names[k], 100 * ph[k, busy].sum() / tot))
    for k, nm in ((5, "packed-flag tier"), (6, "medium tier"), (7, "exact handler")):
        n = ph[k, busy].sum()
        cyc = ph[k - 4, busy].sum() + (ph[4, busy].sum() if k == 7 else 0)
        print("  chunks through the %-16s %5.1f %%, %7.0f cycles each (its own part)" % (nm, 100 * n / n_chunks, cyc / max(n, 1)))
    ctx.close()
