#!/bin/bash
# c2 (sites-only) A/B on one box: the packed form (ABI 6, k_sites2p) against the full form (k_sites2), un-profiled bench
# lines and rocprofv3 kernel stats one block at a time.  gpurun -- bash tools/r04_c2.sh
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for form in packed full; do
  flag=""; [ $form = full ] && flag="--no-packed-sites"
  python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline $flag > $OUT/bench_c2_${form}.json 2> $OUT/bench_c2_${form}.err || exit 1
  python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline --slots 1 $flag > $OUT/bench_c2_${form}_slots1.json 2>> $OUT/bench_c2_${form}.err || exit 1
  rm -rf /tmp/prof_c2
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline --slots 1 $flag > $OUT/bench_c2_${form}_profiled.json 2>> $OUT/bench_c2_${form}.err || exit 1
  cp $(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1) $OUT/c2_${form}_one_block_at_a_time_kernel_stats.csv
done
echo c2 done
