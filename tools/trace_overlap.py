#!/usr/bin/env python3
"""Kernel begin/end stamps of a bench run with blocks in flight (tools/r05_trace.sh) -> who runs beside whom.
usage: trace_overlap.py trace.csv [first_chain last_chain]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["start"], r["end"] = int(r["start"]), int(r["end"])
rows.sort(key=lambda r: r["start"])
t0 = rows[0]["start"]
# chains: the k-th k_stream and what follows on its queue until the next k_stream there
heads = [r for r in rows if r["name"].startswith("k_stream")]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 28)
print("kernel               queue        start us      end us   duration")
for r in rows:
    if heads[lo]["start"] <= r["start"] <= heads[hi]["start"]:
        print("%-20s %5s %12.1f %12.1f %8.1f" % (r["name"], r["queue"], (r["start"] - t0) / 1e3, (r["end"] - t0) / 1e3, (r["end"] - r["start"]) / 1e3))
# in the window: time with n kernels of each kind active
ev = []
w0, w1 = heads[lo]["start"], heads[hi]["start"]
for r in rows:
    if r["end"] > w0 and r["start"] < w1:
        ev.append((max(r["start"], w0), 1, r["name"]))
        ev.append((min(r["end"], w1), -1, r["name"]))
ev.sort()
act = defaultdict(int)
tally = defaultdict(int)
last = w0
for t, d, n in ev:
    key = (act["k_stream"] + act["k_stream_gen"], sum(v for k, v in act.items() if not k.startswith("k_stream")))
    tally[key] += t - last
    last = t
    act[n] += d
print("window %.1f us = %d chains: %.1f us per chain" % ((w1 - w0) / 1e3, hi - lo, (w1 - w0) / 1e3 / (hi - lo)))
for k in sorted(tally):
    print("  %d one-pass kernels + %d followers active: %6.1f us per chain (%4.1f %%)" % (k[0], k[1], tally[k] / 1e3 / (hi - lo), 100.0 * tally[k] / (w1 - w0)))
