#!/bin/bash
# end of round 3: the profiles touched by the last kernel changes (raw-list tasks of k_gt on configs[3], the cheaper
# haploid-reference test of k_stream_gen); run from the repo root on the GPU box
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
  echo "== $tag"; grep bvcf_dev $OUT/${tag}_kernel_stats.csv | cut -c1-100 | head -6
}
stats bench_c3_streaming --path 2
stats bench_c4_auto --profile c4
stats bench_c4_auto_one_block_at_a_time --profile c4 --slots 1
stats bench_c5h_haploid_calls --profile c5h
stats bench_c5h_haploid_calls_one_block_at_a_time --profile c5h --slots 1
stats bench_c5_general_stream --profile c5
stats bench_c5_general_stream_one_block_at_a_time --profile c5 --slots 1
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json
