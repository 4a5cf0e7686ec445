import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import benchgen as bg
import bystro_vcf_amd as bv
cfg = bg.make_cfg("c3")
rows = 131072
blocks, sizes = [], []
for b in range(2):
    t, n = bg.rows_device(cfg, b * rows, rows, pad=bv.DEVICE_PAD)
    blocks.append(t); sizes.append(n)
stride = ((cfg.n_samples + 3) // 4 + 15) & ~15
ctx = bv.Ctx(bg.n_header_fields(cfg), device=0, max_batch_bytes=max(sizes), n_slots=1, max_lines=rows + 16,
             max_alleles=rows + 1024, cmap_bytes=(rows + 1024 + 16 * 8192) * stride + 4096, path=2)
ptrs = [t.data_ptr() for t in blocks]
ctx.bench_device(ptrs, sizes, 3)
chain, gt, counts = ctx.bench_device(ptrs[:1], sizes[:1], 1)
lib = bv._lib if hasattr(bv, "_lib") else bv.lib
n = 2 * 32768
buf = (C.c_ulonglong * n)()
lib.bvcf_debug_wave_times.argtypes = [C.c_void_p, C.c_int]
print("rc", lib.bvcf_debug_wave_times(buf, n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 32768).astype(np.int64)
nw = int((a[1] != 0).sum())
t0 = a[0][:nw].min()
st = (a[0][:nw] - t0) / 100.0  # wall_clock64 ticks at 100 MHz -> us
en = (a[1][:nw] - t0) / 100.0
print("waves", nw, "k_stream ms", gt)
print("start us: min %.1f med %.1f max %.1f" % (st.min(), np.median(st), st.max()))
print("end   us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
d = en - st
print("dur   us: min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
# per-XCD (workgroup id mod 8) end time
wg = np.arange(nw) // 4
for x in range(8):
    m = (wg % 8) == x
    print("xcd %d: end med %.1f max %.1f" % (x, np.median(en[m]), en[m].max()))
print("--- structure of the variance")
dw = d.reshape(-1, 4)
print("within-WG spread (max-min) med %.1f ; across-WG std of WG means %.1f" % (np.median(dw.max(1) - dw.min(1)), dw.mean(1).std()))
bins = d[: (nw // 128) * 128].reshape(-1, 128).mean(1)
print("mean dur per 128-wave band:", " ".join("%.0f" % x for x in bins))
# waves of a WG land on the 4 SIMDs of one CU; WGs i, i+8, i+16.. share an XCD; print by WG slot within XCD
wgm = dw.mean(1)
x0 = wgm[0::8]
print("xcd0 WG means in dispatch order:", " ".join("%.0f" % x for x in x0[:96]))
print("--- phases (s_memtime ticks per wave, mean over waves with work; share of the sum)")
pb = (C.c_ulonglong * (8 * 32768))()
lib.bvcf_debug_phase_times.argtypes = [C.c_void_p, C.c_int]
lib.bvcf_debug_phase_times(pb, 8 * 32768)
ph = np.frombuffer(pb, dtype=np.uint64).reshape(8, 32768).astype(np.float64)[:, :nw]
busy = d > 50
names = ["4->0 loop back-edge", "head parse + next head load", "map slot, separator, line constants", "10 chunks: scan + re-issue", "verdict, wave sums, commit", "-", "-", "-"]
tot = ph[:, busy].mean(1).sum()
for k in range(8):
    print("  %-26s %10.0f  %5.1f%%" % (names[k], ph[k, busy].mean(), 100 * ph[k, busy].mean() / tot))
print("  sum %.0f ticks; wave dur mean %.1f us -> %.1f ticks/us" % (tot, d[busy].mean(), tot / d[busy].mean()))
hb = (C.c_uint * (2 * 32768))()
lib.bvcf_debug_wave_hw.argtypes = [C.c_void_p, C.c_int]
lib.bvcf_debug_wave_hw(hb, 2 * 32768)
hw = np.frombuffer(hb, dtype=np.uint32).reshape(2, 32768)[:, :nw]
cu = (hw[0] >> 8) & 0xF; sh = (hw[0] >> 12) & 1; se = (hw[0] >> 13) & 7; simd = (hw[0] >> 4) & 3; xcc = hw[1] & 0xF
key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu
u, cnt = np.unique(key, return_counts=True)
print("--- placement: %d distinct CUs hold the %d waves; waves per CU histogram:" % (len(u), nw), dict(zip(*np.unique(cnt, return_counts=True))))
ks = key.astype(np.int64) * 4 + simd
u2, cnt2 = np.unique(ks, return_counts=True)
print("    waves per SIMD histogram:", dict(zip(*np.unique(cnt2, return_counts=True))))
# duration vs how many waves share the SIMD
share = dict(zip(u2, cnt2))
per = np.array([share[k] for k in ks])
for c in np.unique(per):
    print("    waves on a SIMD with %d waves: mean dur %.1f us (n=%d)" % (c, d[per == c].mean(), (per == c).sum()))
print("--- k_head phases per workgroup (s_memtime ticks, mean over workgroups with work)")
hbuf = (C.c_ulonglong * (6 * 8192))()
lib.bvcf_debug_head_times.argtypes = [C.c_void_p, C.c_int]
lib.bvcf_debug_head_times(hbuf, 6 * 8192)
hp = np.frombuffer(hbuf, dtype=np.uint64).reshape(6, 8192).astype(np.float64)
busy_h = hp[1] > 0
hn = ["loop top / barrier", "phase T: tokenise 16 rounds", "phase S part 1: gate, ALT shape", "slot reservation", "part 2: alleles, records, tasks", "line record"]
tot_h = hp[:, busy_h].mean(1).sum()
for k in range(6):
    print("  %-36s %10.0f  %5.1f%%" % (hn[k], hp[k, busy_h].mean(), 100 * hp[k, busy_h].mean() / tot_h))
print("  workgroups with work: %d; sum %.0f ticks = %.1f us at 2.28 ticks/ns" % (busy_h.sum(), tot_h, tot_h / 2280))
