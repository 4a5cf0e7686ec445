#!/bin/bash
# kernel durations of k_inflate / k_inflate_w16 on 2000 (or $1) blocks of configs[2] text, per build in $LIBS (run from the repo root on the GPU box)
R=$PWD
N=${1:-2000}
LEVEL=${2:-6}
PROFILE=${3:-c3}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-libbvcf.so}; do
  for w16 in ${VARIANTS:-0 1}; do
    rm -rf /tmp/prof_inf
    BVCF_LIB=$R/bystro-vcf_amd/$lib BVCF_INFLATE_W16=$w16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -- python3 $R/tools/inflate_bench.py $N 4 $LEVEL $PROFILE > $R/gpurun_out/inflate_bench_${lib}_$w16.log 2>&1
    f=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1)
    echo "$lib w16=$w16: $(grep -m1 '^text' $R/gpurun_out/inflate_bench_${lib}_$w16.log)"
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_inflate' in r['Name'] or 'k_crc32' in r['Name']:
        print('   %-16s calls %s  avg %.1f us' % (r['Name'].split('(')[0].split('::')[-1], r['Calls'], float(r['AverageNs']) / 1e3))
" 
  done
done
