#!/bin/bash
# configs[3] in flight, what the 10 % against configs[2] are made of: the row model's knobs changed one at a time, and the block size
# of configs[2] for both; un-profiled bench lines, one box.  Output: gpurun_out/c4split/
set -o pipefail
O=gpurun_out/c4split; mkdir -p $O
run() { name=$1; shift; python3 bench.py --no-e2e --no-cpu-baseline --no-real-data "$@" > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 1; }
  python3 - $O/$name.out $name <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-28s %7.1f M variants/s  ms/step %.4f  rows/block %d  kernel %s %.4f ms frac %.3f chain_frac %.3f" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"], d["config"]["rows_per_block"], d["roofline"]["kernel"], d["roofline"]["mean_launch_ms"], d["roofline"]["frac"], d["roofline"]["chain_frac"]))
PY
}
for rep in 1 2; do
run c3_$rep --profile c3 &&
run c3_r262144_$rep --profile c3 --rows 262144 &&
run c4_$rep --profile c4 &&
run c4_r311296_$rep --profile c4 --rows 311296 &&
run c4_nomulti_$rep --profile c4 --over p_multi=0 &&
run c4_noindel_$rep --profile c4 --over p_indel=0 &&
run c4_nobad_$rep --profile c4 --over p_bad=0 &&
run c4_plain_$rep --profile c4 --over p_multi=0,p_indel=0,p_bad=0 || exit 1
done 2>&1 | tee $O/summary.txt
