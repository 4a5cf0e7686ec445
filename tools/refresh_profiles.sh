#!/bin/bash
# Regenerates what profiles/ holds (run from the repo root on the GPU box; copy gpurun_out/prof/* into profiles/).
#   kernel-trace stats of the streaming, census and c4 chains; FETCH_SIZE / WRITE_SIZE passes (separate, --pmc only);
#   the default bench.py line.
set -e
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  grep '^{' $OUT/$tag.log > $OUT/${tag}_bench_line.json || true
}
pmc() {  # counter, tag, bench args...
  local ctr=$1 tag=$2; shift 2
  rm -rf /tmp/q_$tag
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/q_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/q_$tag -name '*counter_collection.csv' | head -1)" $OUT/$tag.csv
}
stats bench_c3_streaming --path 2
stats bench_c3_census --path 1
stats bench_c4_auto --path 0 --profile c4
pmc FETCH_SIZE pmc_fetch_size_streaming --path 2
pmc WRITE_SIZE pmc_write_size_streaming --path 2
pmc FETCH_SIZE pmc_fetch_size_census --path 1
pmc WRITE_SIZE pmc_write_size_census --path 1
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json
