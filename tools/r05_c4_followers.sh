#!/bin/bash
# configs[3] with three blocks in flight: which follower of k_stream costs what (an experiments build, bystro-vcf_amd/exp_out:
# BVCF_EXP_SKIP leaves kernels out -- the results are wrong then, only the timed region is read), un-profiled, one box
set -o pipefail
O=gpurun_out/c4f; mkdir -p $O
export BVCF_LIB=$PWD/bystro-vcf_amd/exp_out/libbvcf.so
run() { name=$1; shift; python3 bench.py --no-e2e --no-cpu-baseline --no-real-data --profile c4 "$@" > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 0; }
  python3 - $O/$name.out "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-34s %7.1f M variants/s  per block %.1f us  k_stream alone %.1f us  chain alone %.1f us" % (sys.argv[2], d["value"]/1e6, 1e3*d["ms_per_step"]/d["config"]["resident_blocks_per_gpu"], 1e3*d["roofline"]["mean_launch_ms"], 1e3*d["roofline"].get("chain_ms_one_block_at_a_time",0)))
PY
}
for rep in 1 2; do
run base_$rep
BVCF_EXP_SKIP=2 run no_k_gt_$rep
BVCF_EXP_SKIP=4 run no_k_finish_$rep
BVCF_EXP_SKIP=6 run no_k_gt_no_k_finish_$rep
BVCF_EXP_SKIP=7 run no_followers_$rep
BVCF_EXP_HEAD_WGS=2 run k_head_2_wgs_per_cu_$rep
BVCF_EXP_HEAD_WGS=1 run k_head_1_wg_per_cu_$rep
BVCF_EXP_GT_DIV=2 run k_gt_half_grid_$rep
BVCF_EXP_GT_DIV=4 run k_gt_quarter_grid_$rep
BVCF_STREAM_WGS=3 run k_stream_3_wgs_$rep
run nomulti_$rep --over p_multi=0
BVCF_EXP_SKIP=7 run nomulti_no_followers_$rep --over p_multi=0
done 2>&1 | tee $O/summary.txt
