#!/bin/bash
# round 5: k_inflate_w4 A/B between builds (LIBS="libbvcf_infcf.so libbvcf.so"): 3 840 level-6 blocks of configs[2] text under rocprofv3,
# the kernel's average duration per build; then the inflate tests on the product build
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r05w}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2; do
for lib in ${LIBS:-libbvcf_infcf.so libbvcf.so}; do
  rm -rf /tmp/pi_$lib
  BVCF_LIB=$R/bystro-vcf_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi_$lib -- python3 $R/tools/inflate_bench.py 3840 6 6 ${PROFILE:-c3} > $OUT/inflate_$lib.log 2>&1
  f=$(find /tmp/pi_$lib -name '*kernel_stats.csv' | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_inflate' in r['Name'] or 'k_crc32' in r['Name']:
        print('$lib  %-16s calls %s  avg %.1f us' % (r['Name'].split('(')[0].split('::')[-1], r['Calls'], float(r['AverageNs'])/1e3))
" | tee -a $OUT/inflate_ab.txt
done
done
cd $R
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2 | tee -a $OUT/inflate_ab.txt
fi
