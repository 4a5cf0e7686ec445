#!/bin/bash
# c2 A/B on one box for several libraries (LIBS, in bystro-vcf_amd/): un-profiled bench lines (three blocks in flight, and
# one at a time) and rocprofv3 kernel stats one block at a time.  LIBS="libbvcf.so libbvcf_prev.so" TAG=r04o gpurun -- bash tools/r04_c2_census.sh
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r04o}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
for lib in ${LIBS:-libbvcf.so libbvcf_prev.so}; do
  export BVCF_LIB=$R/bystro-vcf_amd/$lib
  python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline 2>> $OUT/err.txt | grep '^{' >> $OUT/bench_c2_$lib.jsonl || exit 1
  python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline --slots 1 2>> $OUT/err.txt | grep '^{' >> $OUT/bench_c2_slots1_$lib.jsonl || exit 1
  if [ $rnd = 1 ]; then
    rm -rf /tmp/prof_c2
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python3 $R/bench.py --profile c2 --no-e2e --no-cpu-baseline --slots 1 > /dev/null 2>> $OUT/err.txt || exit 1
    cp $(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1) $OUT/c2_${lib}_one_block_at_a_time_kernel_stats.csv
    echo "== $lib"; grep "bvcf_dev" $OUT/c2_${lib}_one_block_at_a_time_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
  fi
done
done
python3 - $OUT <<'P'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_c2_*.jsonl")):
    for ln in open(f):
        d = json.loads(ln); r = d["roofline"]
        print("%-40s %6.2f G/s chain_frac %.3f kernel %.4f ms chain 1-at-a-time %.4f ms" % (f.split("/")[-1][9:-6], d["value"] / 1e9, r["chain_frac"], r["mean_launch_ms"], r["chain_ms_one_block_at_a_time"]))
P
