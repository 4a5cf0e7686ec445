#!/usr/bin/env python3
"""What do k_stream's own stores cost it?  (experiment, needs the three libbvcf_exp_{a,b,c}.so builds whose chain stops after
k_stream: a = all stores, b = no class-list / class-map stores, c = no stores at all; results are not valid)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import benchgen as bg
    import bystro_vcf_amd as bv
    cfg = bg.make_cfg("c3")
    rows = 311_296
    blocks = [bg.rows_device(cfg, b * rows, rows, pad=bv.DEVICE_PAD) for b in range(4)]
    ns = cfg.n_samples
    stride = ((ns + 3) // 4 + 15) & ~15
    nbytes = max(n for _, n in blocks)
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=1, max_lines=rows + 16, max_alleles=rows + 1024,
                 cmap_bytes=min((rows + 1024 + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
    ptrs, sizes = [t.data_ptr() for t, _ in blocks], [n for _, n in blocks]
    ctx.bench_device(ptrs, sizes, 4, slots=1)
    chain, scan, _ = ctx.bench_device(ptrs, sizes, 16, slots=1)
    ms = sorted(scan)[len(scan) // 2]
    print("%s: k_stream median %.4f ms = %.2f TB/s" % (os.path.basename(os.environ.get("BVCF_LIB", "libbvcf.so")), ms, sizes[0] / ms / 1e9), flush=True)
    sys.exit(0)
for rnd in range(3):
    for t in (sys.argv[1] if len(sys.argv) > 1 else "abcj"):
        lib = os.path.join(ROOT, "bystro-vcf_amd", "libbvcf_exp_%s.so" % t)
        try:
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, BVCF_LIB=lib), timeout=120)
        except subprocess.TimeoutExpired:
            print("%s: no answer within 120 s" % os.path.basename(lib), flush=True)
            sys.exit(1)
