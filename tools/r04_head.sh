#!/bin/bash
# k_head A/B on one box (lane-per-pair part 2 against the one-lane-per-line loop of libbvcf_oldhead.so): kgt_probe under
# rocprofv3 at 0 / 2 / 20 % multiallelic lines and on configs[3], then c3 / c4 bench lines in one call.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r04d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libbvcf.so libbvcf_oldhead.so}; do
  [ -f $R/bystro-vcf_amd/$lib ] || continue
  for probe in "p_multi=0 p_indel=0" "p_multi=200 p_indel=0" "p_multi=2000 p_indel=0" ""; do
    tag=$(echo "$lib $probe" | tr ' =.' '___')
    rm -rf /tmp/prof_h
    BVCF_LIB=$R/bystro-vcf_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -- python3 $R/tools/kgt_probe.py $probe > $OUT/probe_$tag.log 2>&1 || exit 1
    f=$(find /tmp/prof_h -name '*kernel_stats.csv' | head -1)
    echo "== $lib [$probe]" >> $OUT/probe_summary.txt
    grep -E "k_head|k_gt|k_finish|k_stream|k_order" $f | awk -F, '{print "   ", $1, $2, $4}' >> $OUT/probe_summary.txt
  done
done
for lib in ${LIBS:-libbvcf.so libbvcf_oldhead.so}; do
  [ -f $R/bystro-vcf_amd/$lib ] || continue
  for prof in c3 c4; do
    BVCF_LIB=$R/bystro-vcf_amd/$lib python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data > $OUT/bench_${prof}_$lib.json 2>> $OUT/bench.err || exit 1
  done
done
echo head done
