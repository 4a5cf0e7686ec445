#!/bin/bash
# blocks in flight and the runtime's hardware queues (GPU_MAX_HW_QUEUES, default 4: streams beyond that share a queue and serialise)
set -o pipefail
O=gpurun_out/hwq; mkdir -p $O
export BVCF_LIB=${BVCF_LIB:-$PWD/bystro-vcf_amd/exp_out/libbvcf.so}
run() { name=$1; shift; python3 bench.py --no-e2e --no-cpu-baseline --no-real-data "$@" > $O/$name.out 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 0; }
  python3 - $O/$name.out "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("%-34s %7.1f M variants/s  per block %.1f us  kernel alone %.1f us  chain alone %.1f us" % (sys.argv[2], d["value"]/1e6, 1e3*d["ms_per_step"]/d["config"]["resident_blocks_per_gpu"], 1e3*d["roofline"]["mean_launch_ms"], 1e3*d["roofline"].get("chain_ms_one_block_at_a_time",0)))
PY
}
for rep in 1 2; do
for prof in c4 c3; do
run ${prof}_base_$rep --profile $prof
for q in 2 8 16; do
GPU_MAX_HW_QUEUES=$q run ${prof}_q${q}_$rep --profile $prof
done
GPU_MAX_HW_QUEUES=8 run ${prof}_q8_slots4_$rep --profile $prof --slots 4
GPU_MAX_HW_QUEUES=8 run ${prof}_q8_slots6_$rep --profile $prof --slots 6
GPU_MAX_HW_QUEUES=8 BVCF_SCAN_STREAM=1 run ${prof}_q8_scan_stream_$rep --profile $prof
GPU_MAX_HW_QUEUES=8 BVCF_SCAN_STREAM=1 run ${prof}_q8_scan_stream_slots4_$rep --profile $prof --slots 4
done
done 2>&1 | tee $O/summary.txt
