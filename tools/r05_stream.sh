#!/bin/bash
# round 5, k_stream with the batched epilogue: parity first (every test that goes through the streaming path), then same-box
# A/B against the round's first build (libbvcf_base.so) on configs[2] / configs[3], one block at a time and three in flight
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r05d}
mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiallelic.py tests/test_gpu_synth.py tests/test_gpu_tables.py tests/test_gpu_streamgen.py -x -q > $OUT/pytest_stream.log 2>&1 || { tail -40 $OUT/pytest_stream.log; exit 1; }
tail -2 $OUT/pytest_stream.log
fi
LIBS=${LIBS:-"libbvcf_base.so libbvcf.so"}
for prof in ${PROFILES:-c3 c4}; do
  python tools/ab_bench.py $(for l in $LIBS; do echo bystro-vcf_amd/$l; done) 3 -- --profile $prof --no-real-data > $OUT/ab_alone_$prof.txt 2>&1
  cat $OUT/ab_alone_$prof.txt
  for rnd in 1 2; do for lib in $LIBS; do
    BVCF_LIB=$R/bystro-vcf_amd/$lib python bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data | tail -1 > $OUT/line.json
    python -c "import json,sys; d=json.load(open('$OUT/line.json')); print('$prof $lib value %.1f M/s  alone %.4f ms frac %.3f chain_frac %.3f chain alone %.4f' % (d['value']/1e6, d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['chain_frac'], d['roofline']['chain_ms_one_block_at_a_time']))" | tee -a $OUT/inflight.txt
  done; done
done
