#!/bin/bash
# blocks in flight: k_stream launched with unused dynamic LDS so that at most two (or one) of its workgroups -- of ANY batch -- fit a CU
# and the followers of the batches before always find room (experiments build).  Un-profiled, one box.
set -o pipefail
O=gpurun_out/c4lds; mkdir -p $O
export BVCF_LIB=$PWD/bystro-vcf_amd/exp_out/libbvcf.so
run() { name=$1; shift; python3 bench.py --no-e2e --no-cpu-baseline --no-real-data "$@" > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 0; }
  python3 - $O/$name.out "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-34s %7.1f M variants/s  per block %.1f us  k_stream alone %.1f us  chain alone %.1f us" % (sys.argv[2], d["value"]/1e6, 1e3*d["ms_per_step"]/d["config"]["resident_blocks_per_gpu"], 1e3*d["roofline"]["mean_launch_ms"], 1e3*d["roofline"].get("chain_ms_one_block_at_a_time",0)))
PY
}
for rep in 1 2; do
for prof in c4 c3; do
run ${prof}_base_$rep --profile $prof
BVCF_EXP_STREAM_LDS=36864 run ${prof}_lds36k_$rep --profile $prof
BVCF_EXP_STREAM_LDS=45056 run ${prof}_lds44k_$rep --profile $prof
BVCF_EXP_STREAM_LDS=45056 BVCF_STREAM_WGS=3 run ${prof}_lds44k_grid3_$rep --profile $prof
BVCF_EXP_STREAM_LDS=45056 BVCF_STREAM_WGS=4 run ${prof}_lds44k_grid4_$rep --profile $prof
BVCF_EXP_STREAM_LDS=45056 run ${prof}_lds44k_slots4_$rep --profile $prof --slots 4
BVCF_EXP_STREAM_LDS=45056 run ${prof}_lds44k_slots2_$rep --profile $prof --slots 2
done
done 2>&1 | tee $O/summary.txt
