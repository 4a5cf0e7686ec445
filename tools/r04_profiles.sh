#!/bin/bash
# round 4: everything profiles/r04_* holds that comes from the last build, in one gpurun call (one box):
#   gpurun --timeout 1150 -- bash tools/r04_profiles.sh        (outputs: gpurun_out/prof/, copied to profiles/r04_* afterwards)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-real-data "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  grep '^{' $OUT/$tag.log > $OUT/${tag}_bench_line.json || true
  echo "== $tag"; cut -d, -f1-4 $OUT/${tag}_kernel_stats.csv | cut -c1-120 | head -8
}
pmc() {  # counter, tag, bench args...
  local ctr=$1 tag=$2; shift 2
  rm -rf /tmp/q_$tag
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/q_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e --no-real-data "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/q_$tag -name '*counter_collection.csv' | head -1)" $OUT/$tag.csv
}
if [ "$ONLY" = c2 ]; then  # (after a change to the sites-only chain alone: its files again, the rest stands)
  : > $OUT/bench_unprofiled_lines_c2.json
  python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/bench_unprofiled_lines_c2.json
  python3 $R/bench.py --profile c2 --no-packed-sites --no-e2e --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/bench_unprofiled_lines_c2.json
  stats bench_c2_sites_only_packed --profile c2
  stats bench_c2_sites_only_packed_one_block_at_a_time --profile c2 --slots 1
  stats bench_c2_sites_only_full_form_one_block_at_a_time --profile c2 --slots 1 --no-packed-sites
  pmc FETCH_SIZE pmc_fetch_size_c2 --profile c2
  pmc WRITE_SIZE pmc_write_size_c2 --profile c2
  cd $R
  python3 tools/derive_traffic.py $OUT > $OUT/k_gt_hbm_traffic.json 2> $OUT/derive.err || tail -3 $OUT/derive.err
  KERNEL=k_sites2p ARGS="--profile c2" TAG=k_sites2p bash tools/pmc_sq.sh > $OUT/pmc_sq_k_sites2p.txt 2>&1
  echo "c2 done"
  exit 0
fi
if [ "$ONLY" = c5 ]; then  # (after a change to k_stream_gen alone)
  : > $OUT/bench_unprofiled_lines_c5.json
  for prof in c5 c5h; do
    python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data 2>/dev/null | grep '^{' >> $OUT/bench_unprofiled_lines_c5.json
  done
  stats bench_c5_general_stream_one_block_at_a_time --profile c5 --slots 1
  stats bench_c5h_haploid_calls_one_block_at_a_time --profile c5h --slots 1
  BVCF_GEN_STREAM=1 pmc FETCH_SIZE pmc_fetch_size_c5 --profile c5
  BVCF_GEN_STREAM=1 pmc WRITE_SIZE pmc_write_size_c5 --profile c5
  cd $R
  KERNEL=k_stream_gen ARGS="--profile c5" TAG=k_stream_gen_c5 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_stream_gen_c5.txt 2>&1
  echo "c5 done"
  exit 0
fi
# un-profiled bench lines of every profile, one call
: > $OUT/bench_unprofiled_lines.json
for prof in c3 c4 c2 c5 c5h; do
  python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data 2>/dev/null | grep '^{' >> $OUT/bench_unprofiled_lines.json
done
python3 $R/bench.py --profile c2 --no-packed-sites --no-e2e --no-cpu-baseline 2>/dev/null | grep '^{' >> $OUT/bench_unprofiled_lines.json
echo "unprofiled lines done"
stats bench_c3_streaming --path 2
stats bench_c3_streaming_one_block_at_a_time --path 2 --slots 1
stats bench_c4_auto --profile c4
stats bench_c4_auto_one_block_at_a_time --profile c4 --slots 1
stats bench_c2_sites_only_packed --profile c2
stats bench_c2_sites_only_packed_one_block_at_a_time --profile c2 --slots 1
stats bench_c2_sites_only_full_form_one_block_at_a_time --profile c2 --slots 1 --no-packed-sites
stats bench_c5_general_stream_one_block_at_a_time --profile c5 --slots 1
stats bench_c5h_haploid_calls_one_block_at_a_time --profile c5h --slots 1
pmc FETCH_SIZE pmc_fetch_size_streaming --path 2
pmc WRITE_SIZE pmc_write_size_streaming --path 2
pmc FETCH_SIZE pmc_fetch_size_census --path 1
pmc WRITE_SIZE pmc_write_size_census --path 1
BVCF_GEN_STREAM=1 pmc FETCH_SIZE pmc_fetch_size_c5 --profile c5
BVCF_GEN_STREAM=1 pmc WRITE_SIZE pmc_write_size_c5 --profile c5
pmc FETCH_SIZE pmc_fetch_size_c2 --profile c2
pmc WRITE_SIZE pmc_write_size_c2 --profile c2
pmc FETCH_SIZE pmc_fetch_size_c4 --profile c4
pmc WRITE_SIZE pmc_write_size_c4 --profile c4
echo "pmc traffic done"
cd $R
python3 tools/derive_traffic.py $OUT > $OUT/k_gt_hbm_traffic.json 2> $OUT/derive.err || tail -3 $OUT/derive.err
KERNEL=k_head_lean ARGS="--profile c4" TAG=k_head_lean_c4 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_head_lean_c4.txt 2>&1
KERNEL=k_sites2p ARGS="--profile c2" TAG=k_sites2p bash tools/pmc_sq.sh > $OUT/pmc_sq_k_sites2p.txt 2>&1
KERNEL=k_stream_gen ARGS="--profile c5" TAG=k_stream_gen_c5 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_stream_gen_c5.txt 2>&1
KERNEL=k_stream ARGS="--path 2" TAG=k_stream_c3 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_stream.txt 2>&1
echo "sq done"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
