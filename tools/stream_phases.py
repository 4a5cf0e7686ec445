#!/usr/bin/env python3
"""Where do k_stream's waves spend their cycles, phase by phase?  (needs a -DBVCF_EXP_TIMES build: BVCF_LIB=...; not a test)
    make -C bystro-vcf_amd/csrc OUT=/tmp/t1 OBJ=/tmp/t1/obj EXTRA=-DBVCF_EXP_TIMES && BVCF_LIB=/tmp/t1/libbvcf.so python tools/stream_phases.py c3 c4
(the stamps cost registers and ~10 % of the kernel: shares, not times)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

ROWS = {"c3": 311_296, "c4": 262_144}
NAMES = ["loop back-edge (after commit -> top)", "B's head parse + C's head load", "map slot, separator, line constants",
         "the chunks: scan + re-issue", "epilogue: finish_list / finish_dense, wave sums, commit", "-", "-", "-"]
for prof in sys.argv[1:] or ["c3"]:
    cfg = bg.make_cfg(prof)
    rows = ROWS[prof]
    t, nbytes = bg.rows_device(cfg, 0, rows, pad=bv.DEVICE_PAD)
    ns = cfg.n_samples
    stride = ((ns + 3) // 4 + 15) & ~15
    n_alt = rows * 4 + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=n_alt,
                 cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00), path=2)
    ctx.bench_device([t.data_ptr()], [nbytes], 4, slots=1)
    chain, scan, counts = ctx.bench_device([t.data_ptr()], [nbytes], 1, slots=1)
    pb = (C.c_ulonglong * (8 * 32768))()
    bv.lib.bvcf_debug_phase_times.argtypes = [C.c_void_p, C.c_int]
    assert bv.lib.bvcf_debug_phase_times(pb, 8 * 32768) == 0
    ph = np.frombuffer(pb, dtype=np.uint64).reshape(8, 32768).astype(np.float64)
    busy = ph[3] > 0
    tot = ph[:, busy].sum()
    print("== %s: %d waves with work, k_stream %.4f ms (stamped build), %.0f cycles per line and wave" % (
        prof, busy.sum(), scan[0], tot / rows))
    for k in range(5):
        print("  %-58s %5.1f %%  %7.0f cycles per line" % (NAMES[k], 100 * ph[k, busy].sum() / tot, ph[k, busy].sum() / rows))
    ctx.close()
    del t
