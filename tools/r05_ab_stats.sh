#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats), one block at a time, of several builds: usage r05_ab_stats.sh <profile> <slots> libA.so libB.so ...
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/abs; mkdir -p $OUT
prof=$1; slots=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  tag=${prof}_$(basename $(dirname $lib))_s${slots}_$rep
  rm -rf /tmp/p_$tag
  BVCF_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-real-data --profile $prof --slots $slots > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  echo "== $tag"
  python3 - $OUT/${tag}_kernel_stats.csv <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if "bvcf_dev" in r["Name"]:
        n=r["Name"].split("(")[0].replace("bvcf_dev::","")
        print("   %-16s calls %4s avg %9.1f us" % (n, r["Calls"], float(r["AverageNs"])/1e3)); tot+=float(r["AverageNs"])/1e3
print("   sum of averages %.1f us" % tot)
PY
done; done 2>&1 | tee $OUT/summary_${prof}_s${slots}.txt
