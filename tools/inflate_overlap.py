#!/usr/bin/env python3
"""from a rocprofv3 kernel trace of the CLI over a BGZF file: do the inflate kernels of consecutive batches overlap?
usage: python tools/inflate_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0 = rows[0][0]
inf = [r for r in rows if r[2].startswith("k_inflate")]
print("%d kernels, %d inflate launches; first 12 inflate launches (start, end in us from the first kernel; queue/stream):" % (len(rows), len(inf)))
for s, e, n, q in inf[:12]:
    print("  %-14s %9.1f %9.1f  dur %7.1f  %s" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
busy = 0
cur_s, cur_e = None, None
for s, e, n, q in inf:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("sum of inflate durations %.1f ms, union %.1f ms, span %.1f ms" % (sum(e - s for s, e, _, _ in inf) / 1e6, busy / 1e6, (inf[-1][1] - inf[0][0]) / 1e6))
