#!/bin/bash
# same-box A/B with blocks in flight (un-profiled bench lines): usage r05_ab_inflight.sh "<profiles>" libA.so libB.so ...
set -o pipefail
O=gpurun_out/abf; mkdir -p $O
profs=$1; shift
libs=("$@")
for rep in 1 2 3; do
# (the boxes drift within a call: every round starts with another build)
rot=("${libs[@]:$(( (rep - 1) % ${#libs[@]} ))}" "${libs[@]:0:$(( (rep - 1) % ${#libs[@]} ))}")
for prof in $profs; do
for lib in "${rot[@]}"; do
  name=${prof}_$(basename $(dirname $lib))_$rep
  BVCF_LIB=$PWD/$lib python3 bench.py --no-e2e --no-cpu-baseline --no-real-data --profile $prof > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; continue; }
  python3 - $O/$name.out "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-34s %7.1f M variants/s  per block %.1f us  kernel alone %.1f us  chain alone %.1f us" % (sys.argv[2], d["value"]/1e6, 1e3*d["ms_per_step"]/d["config"]["resident_blocks_per_gpu"], 1e3*d["roofline"]["mean_launch_ms"], 1e3*d["roofline"].get("chain_ms_one_block_at_a_time",0)))
PY
done; done; done 2>&1 | tee $O/summary.txt
