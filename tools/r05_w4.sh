#!/bin/bash
# k_stream compiled for 128 registers (amdgpu_waves_per_eu 4: 8 spills) so that two follower workgroups fit beside two of its own, with and
# without the LDS request that keeps k_stream at two workgroups per CU over all blocks (experiments builds)
set -o pipefail
O=gpurun_out/w4; mkdir -p $O
run() { name=$1; shift; python3 bench.py --no-e2e --no-cpu-baseline --no-real-data "$@" > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 0; }
  python3 - $O/$name.out "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-34s %7.1f M variants/s  per block %.1f us  kernel alone %.1f us  chain alone %.1f us" % (sys.argv[2], d["value"]/1e6, 1e3*d["ms_per_step"]/d["config"]["resident_blocks_per_gpu"], 1e3*d["roofline"]["mean_launch_ms"], 1e3*d["roofline"].get("chain_ms_one_block_at_a_time",0)))
PY
}
for rep in 1 2; do
for prof in c3 c4; do
BVCF_LIB=$PWD/bystro-vcf_amd/exp_out/libbvcf.so run ${prof}_145regs_$rep --profile $prof
BVCF_LIB=$PWD/bystro-vcf_amd/w4_out/libbvcf.so run ${prof}_128regs_$rep --profile $prof
BVCF_LIB=$PWD/bystro-vcf_amd/w4_out/libbvcf.so BVCF_EXP_STREAM_LDS=45056 run ${prof}_128regs_lds44k_$rep --profile $prof
BVCF_LIB=$PWD/bystro-vcf_amd/w4_out/libbvcf.so BVCF_STREAM_WGS=3 run ${prof}_128regs_grid3_$rep --profile $prof
BVCF_LIB=$PWD/bystro-vcf_amd/w4_out/libbvcf.so BVCF_STREAM_WGS=4 run ${prof}_128regs_grid4_$rep --profile $prof
done; done 2>&1 | tee $O/summary.txt
