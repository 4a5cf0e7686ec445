#!/usr/bin/env python3
"""Where do k_sites1's waves spend their time?  Needs a build with -DBVCF_EXP_TIMES (BVCF_LIB=...):
    make -C bystro-vcf_amd/csrc OUT=/tmp/t EXTRA=-DBVCF_EXP_TIMES && BVCF_LIB=/tmp/t/libbvcf.so python tools/s1_times.py
Per tile: shader cycles of  0 issue of the eight chunk loads | 1 masks + staging (the loads arrive here) | 2 bitmaps,
counts, publication | 3 first line's start, round FIFO, common-line work | 4 look-back | 5 stores, general lines, next rounds;
and the wall clock (100 MHz) at which each tile started / ended."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = bg.make_cfg("c2")
t, nbytes = bg.rows_device(cfg, 0, rows, pad=bv.DEVICE_PAD)
ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16)
ptrs = [t.data_ptr()]
for _ in range(3):
    ctx.bench_device(ptrs, [nbytes], 1, slots=1)
tile = 7680 if os.environ.get("BVCF_SITES") == "3" else 7168
n_tiles = (nbytes + tile - 1) // tile
n = min(n_tiles, 32768)
ph = np.zeros((8, 32768), dtype=np.uint64)
wt = np.zeros((2, 32768), dtype=np.uint64)
bv.lib.bvcf_debug_phase_times(ph.ctypes.data_as(C.c_void_p), 8 * 32768)
bv.lib.bvcf_debug_wave_times(wt.ctypes.data_as(C.c_void_p), 2 * 32768)
ph = ph[:, :n].astype(np.float64)
names = ["issue loads", "EOL masks, count, publish (loads land)", "TAB masks, staging, bitmaps, ranks", "first start, fifo, common lines", "look-back", "stores, general, more rounds"]
tot = ph[:6].sum(axis=0)
print("tiles %d; cycles per tile: mean %.0f median %.0f p95 %.0f" % (n, tot.mean(), np.median(tot), np.percentile(tot, 95)))
for k in range(6):
    print("  %-34s mean %7.0f  median %7.0f  p95 %7.0f  (%.0f %%)" % (names[k], ph[k].mean(), np.median(ph[k]), np.percentile(ph[k], 95), 100 * ph[k].sum() / tot.sum()))
st, en = wt[0, :n].astype(np.float64), wt[1, :n].astype(np.float64)
t0 = st.min()
print("wall clock (us): first start 0, last end %.1f; tile lifetime mean %.2f median %.2f; start of tile k: %s" % (
    (en.max() - t0) / 100.0, ((en - st) / 100.0).mean(), np.median((en - st) / 100.0),
    ", ".join("%d:%.1f" % (k, (st[k] - t0) / 100.0) for k in range(0, n, max(1, n // 12)))))
ctx.close()
