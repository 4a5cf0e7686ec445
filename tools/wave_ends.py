#!/usr/bin/env python3
"""When do the waves of the streaming kernel end, one launch alone?  (needs a -DBVCF_EXP_TIMES=2 build: BVCF_LIB=...; not a test)
    make -C bystro-vcf_amd/csrc OUT=/tmp/t OBJ=/tmp/t/obj EXTRA=-DBVCF_EXP_TIMES=2 && BVCF_LIB=/tmp/t/libbvcf.so python tools/wave_ends.py c5 c3
Verdict r04 item 3: three blocks in flight take 0.52 ms per c5 block, one alone 0.635 -- if the last waves end well after the
median, the static run of tiles per wave is what overlap hides, and tiles should be dealt dynamically."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

ROWS = {"c5": 98_304, "c5h": 98_304, "c3": 311_296, "c4": 262_144}
for prof in sys.argv[1:] or ["c5", "c3"]:
    cfg = bg.make_cfg(prof)
    rows = ROWS[prof]
    t, nbytes = bg.rows_device(cfg, 0, rows, pad=bv.DEVICE_PAD)
    ns = cfg.n_samples
    stride = ((ns + 3) // 4 + 15) & ~15
    n_alt = rows * 4 + 1024
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=n_alt,
                 cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
    ctx.bench_device([t.data_ptr()], [nbytes], 4, slots=1)
    chain, scan, counts = ctx.bench_device([t.data_ptr()], [nbytes], 1, slots=1)
    n = 2 * 32768
    buf = (C.c_ulonglong * n)()
    bv.lib.bvcf_debug_wave_times.argtypes = [C.c_void_p, C.c_int]
    assert bv.lib.bvcf_debug_wave_times(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 32768).astype(np.int64)
    # (the stamps of an earlier launch with more waves stay behind this one's: keep the waves that started with the latest)
    ok = (a[1] != 0) & (a[0] > a[0].max() - 100_000)
    nw = int(ok.sum())
    t0 = a[0][ok].min()
    st = (a[0][ok] - t0) / 100.0  # 100 MHz -> us
    en = (a[1][ok] - t0) / 100.0
    d = en - st
    pc = lambda x, q: np.percentile(x, q)
    print("== %s: %s, %d waves, kernel %.1f us by HIP events (stamped build), %.2f GB" % (prof, ctx.stream_kernel(), nw, scan[0] * 1e3, nbytes / 1e9))
    print("  start us: min %.1f  p50 %.1f  p99 %.1f  max %.1f" % (st.min(), pc(st, 50), pc(st, 99), st.max()))
    print("  end   us: min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (en.min(), pc(en, 10), pc(en, 50), pc(en, 90), pc(en, 99), en.max()))
    print("  dur   us: min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f   (mean %.1f: a perfectly even kernel would end at start + %.1f)"
          % (d.min(), pc(d, 10), pc(d, 50), pc(d, 90), d.max(), d.mean(), d.mean()))
    print("  last wave ends %.1f %% after the median end, %.1f %% after the mean duration" % (100 * (en.max() / pc(en, 50) - 1), 100 * (en.max() / d.mean() - 1)))
    hist, edges = np.histogram(en, bins=20, range=(0, en.max()))
    print("  ends per 5 %% of the kernel's span:", " ".join("%d" % h for h in hist))
    # waves resident over time: how much of the kernel runs with fewer than all waves
    order = np.sort(en)
    for frac in (0.5, 0.75, 0.9, 0.97):
        k = int(frac * nw)
        print("    %2.0f %% of the waves are done at %.1f us (%.0f %% of the span)" % (100 * frac, order[k], 100 * order[k] / en.max()))
    ctx.close()
    del t
