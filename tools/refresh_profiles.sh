#!/bin/bash
# Regenerates what profiles/ holds (run from the repo root on the GPU box; copy gpurun_out/prof/* into profiles/ as
# rNN_*).  kernel-trace stats of the chains of every bench profile; FETCH_SIZE / WRITE_SIZE passes (separate, --pmc
# only: FETCH_SIZE + WRITE_SIZE exceed the TCC slots); the default bench.py line.
#   usage: bash tools/refresh_profiles.sh [what...]   what = stats | pmc | line   (default: all)
set -e
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WHAT=${@:-stats pmc line}
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  grep '^{' $OUT/$tag.log > $OUT/${tag}_bench_line.json || true
  echo "== $tag"; cat $OUT/${tag}_kernel_stats.csv | cut -d, -f1-4 | head -12
}
pmc() {  # counter, tag, bench args...
  local ctr=$1 tag=$2; shift 2
  rm -rf /tmp/q_$tag
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/q_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/q_$tag -name '*counter_collection.csv' | head -1)" $OUT/$tag.csv
}
for w in $WHAT; do
  case $w in
    stats)
      stats bench_c3_streaming --path 2
      stats bench_c3_streaming_one_block_at_a_time --path 2 --slots 1
      stats bench_c3_census --path 1
      stats bench_c4_auto --profile c4
      stats bench_c2_sites_only --profile c2
      stats bench_c2_sites_only_one_block_at_a_time --profile c2 --slots 1
      BVCF_SITES=1 stats bench_c2_k_sites_round2_chain_one_block_at_a_time --profile c2 --slots 1
      BVCF_SITES=3 stats bench_c2_k_sites1_one_pass_one_block_at_a_time --profile c2 --slots 1
      stats bench_c5h_haploid_calls --profile c5h
      stats bench_c5h_haploid_calls_one_block_at_a_time --profile c5h --slots 1
      stats bench_c5_general_stream --profile c5
      stats bench_c5_general_stream_one_block_at_a_time --profile c5 --slots 1
      BVCF_GEN_STREAM=0 stats bench_c5_k_stream_plus_k_gt --profile c5
      ;;
    pmc)
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
      ;;
    line)
      cd $R
      python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
      cat $OUT/bench_default.json
      cd /tmp
      ;;
  esac
done
