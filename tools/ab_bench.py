#!/usr/bin/env python3
"""Same-box A/B of two libbvcf builds (not a test): alternates BVCF_LIB between the given .so files
and prints per-build medians of the dominant kernel and of the chain.
usage: python tools/ab_bench.py libA.so libB.so [libC.so ...] [rounds] [-- extra bench.py args]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
libs = [x for x in args if x.endswith(".so")]
rest = [x for x in args if not x.endswith(".so")]
rounds = int(rest[0]) if rest else 4
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, BVCF_LIB=os.path.abspath(l))
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                                       "--no-cpu-baseline", "--no-e2e", "--slots", "1", "--blocks", "4"] + extra, env=env, stderr=subprocess.DEVNULL)
        d = json.loads(out.decode().strip().splitlines()[-1])
        res[l].append((d["roofline"]["mean_launch_ms"], d["roofline"]["chain_ms_one_block_at_a_time"]))
for l in libs:
    k = [x[0] for x in res[l]]
    c = [x[1] for x in res[l]]
    print("%-40s kernel med %.4f ms (min %.4f)  chain med %.4f ms (min %.4f)" % (
        os.path.basename(l), statistics.median(k), min(k), statistics.median(c), min(c)))
