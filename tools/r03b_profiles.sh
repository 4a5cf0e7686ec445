#!/bin/bash
# late round 3: the profiles that changed after tools/r03_profiles.sh was run (k_sites2's whole-line stores and per-tile
# census, k_stream's lists for the further alleles of dense lines); run from the repo root on the GPU box
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  grep '^{' $OUT/$tag.log > $OUT/${tag}_bench_line.json || true
  echo "== $tag"; cut -d, -f1-4 $OUT/${tag}_kernel_stats.csv | cut -c1-120 | head -8
}
pmc() {  # counter, tag, bench args...
  local ctr=$1 tag=$2; shift 2
  rm -rf /tmp/q_$tag
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/q_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/q_$tag -name '*counter_collection.csv' | head -1)" $OUT/$tag.csv
}
stats bench_c2_sites_only --profile c2
stats bench_c2_sites_only_one_block_at_a_time --profile c2 --slots 1
BVCF_S2_CENSUS=chunk stats bench_c2_k_sites2_chunk_census_one_block_at_a_time --profile c2 --slots 1
stats bench_c4_auto --profile c4
stats bench_c4_auto_one_block_at_a_time --profile c4 --slots 1
stats bench_c3_streaming --path 2
stats bench_c3_streaming_one_block_at_a_time --path 2 --slots 1
pmc FETCH_SIZE pmc_fetch_size_c2 --profile c2
pmc WRITE_SIZE pmc_write_size_c2 --profile c2
cd $R
KERNEL=k_sites2 ARGS="--profile c2" TAG=k_sites2 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_sites2.txt 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
