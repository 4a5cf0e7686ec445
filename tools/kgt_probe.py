#!/usr/bin/env python3
"""What is k_gt's time on configs[3] made of?  One block of the c4 row model with some of its knobs changed, through
the chain one block at a time, under rocprofv3 (the kernel stats are the answer):
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python3 tools/kgt_probe.py p_multi=0 p_indel=1500"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import benchgen as bg  # noqa: E402
import bystro_vcf_amd as bv  # noqa: E402

over = dict(kv.split("=") for kv in sys.argv[1:])
over = {k: int(v) for k, v in over.items()}
rows = 262_144
cfg = bg.make_cfg("c4", **over)
t, nbytes = bg.rows_device(cfg, 0, rows, pad=bv.DEVICE_PAD)
ns = cfg.n_samples
stride = ((ns + 3) // 4 + 15) & ~15
n_alt = rows * 4 + 1024
# KGT_SLOTS=2: a ctx with two slots takes k_head_lean (the kernel of the default chain); the blocks still run one at a time
ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=int(os.environ.get("KGT_SLOTS", "1")), max_lines=rows + 16, max_alleles=n_alt,
             cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
chain, scan, counts = ctx.bench_device([t.data_ptr()], [nbytes], 12, slots=1)
print(over, "chain ms", sum(chain) / len(chain), "counts", list(counts)[:8])
if hasattr(bv.lib, "bvcf_debug_gt_kinds"):  # an experiments build: k_gt's tasks by kind, per block
    import ctypes as C
    kinds = (C.c_uint * 4)()
    bv.lib.bvcf_debug_gt_kinds(kinds)
    print("k_gt tasks per block: raw list %.0f, regular text %.0f, general text %.0f" % tuple(k / 12 for k in list(kinds)[:3]))
ctx.close()
if os.environ.get("KGT_HEAD_PHASES"):  # needs a -DBVCF_EXP_TIMES build (BVCF_LIB=...): k_head's time per phase
    import ctypes as C
    import numpy as np
    # (KGT_SLOTS=2: k_head_lean, the kernel of chains with blocks in flight; the blocks still run one at a time)
    ctx = bv.Ctx(bg.n_header_fields(cfg), max_batch_bytes=nbytes, n_slots=int(os.environ.get("KGT_SLOTS", "1")), max_lines=rows + 16, max_alleles=n_alt,
                 cmap_bytes=min((n_alt + nbytes // (4 * ns + 8) + 16 * 8192) * stride + 4096, 0xFFFFFF00))
    ctx.bench_device([t.data_ptr()], [nbytes], 2, slots=1)
    hbuf = (C.c_ulonglong * (6 * 8192))()
    bv.lib.bvcf_debug_head_times.argtypes = [C.c_void_p, C.c_int]
    bv.lib.bvcf_debug_head_times(hbuf, 6 * 8192)
    hp = np.frombuffer(hbuf, dtype=np.uint64).reshape(6, 8192).astype(np.float64)
    busy = hp[1] > 0
    names = ["loop top / barrier", "phase T: tokenise", "part 1: gate, ALT shape", "slot reservation", "part 2: alleles, records, tasks", "line record"]
    if os.environ["KGT_HEAD_PHASES"] == "3":  # a -DBVCF_EXP_TIMES=3 build: part 2 split
        names = ["everything before part 2", "part 2: owners' values (ds_bpermute)", "part 2: the token", "part 2: sums over the line's lanes",
                 "part 2: lists, tasks, records", "back to the lines, line record"]
    print("k_head phases (shader clock ticks per workgroup, mean / p95 over %d workgroups):" % busy.sum())
    for k in range(6):
        print("  %-34s %9.0f %9.0f" % (names[k], hp[k, busy].mean(), np.percentile(hp[k, busy], 95)))
    ctx.close()
