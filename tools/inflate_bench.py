#!/usr/bin/env python3
"""k_inflate / k_inflate_w16 alone: N BGZF blocks of synthetic configs[2] rows (zlib level 6, bgzip's default), inflated
on the device `reps` times through bvcf_bgzf_inflate_device.  Run it under rocprofv3 --kernel-trace --stats for the
kernels' durations (the call itself includes the copies over PCIe):
    python tools/inflate_bench.py [blocks=2000] [reps=5] [level=6] [profile=c3]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benchgen as bg  # noqa: E402
import bgzf  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
profile = sys.argv[4] if len(sys.argv) > 4 else "c3"
cfg = bg.make_cfg(profile)
rows = n_blocks * 0xFF00 // {"c3": 10167, "c4": 10190, "c5": 24361, "c5h": 24000, "c2": 142}.get(profile, 10167) + 64
t, nbytes = bg.rows_device(cfg, 0, rows)
text = bytes(memoryview(t[:n_blocks * 0xFF00].cpu().numpy()))
import multiprocessing as mp


def _part(i):
    return bgzf.bgzf_compress(text[i:i + 0xFF00 * 64], level=level, eof_marker=False)


with mp.Pool(16) as pool:
    comp = b"".join(pool.map(_part, range(0, len(text), 0xFF00 * 64)))
print("text %.1f MB in %d blocks -> %.2f MB compressed (level %d, ratio %.1f)" % (len(text) / 1e6, n_blocks, len(comp) / 1e6, level, len(text) / len(comp)))
out = (C.c_uint8 * len(text))()
n_out = C.c_size_t()
bv.lib.bvcf_bgzf_inflate_device.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
for r in range(reps):
    t0 = time.perf_counter()
    rc = bv.lib.bvcf_bgzf_inflate_device(0, comp, len(comp), out, len(text), C.byref(n_out))
    dt = time.perf_counter() - t0
    assert rc == 0 and n_out.value == len(text), (rc, n_out.value)
    print("call %d: %.2f ms (with copies)" % (r, dt * 1e3))
assert bytes(out) == text
print("ok")
