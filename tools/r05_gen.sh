#!/bin/bash
# round 5, k_stream_gen with tiles dealt dynamically: parity first, then same-box A/B against the round's first build
# (libbvcf_base.so) on c5 / c5h one block at a time and with three in flight, then the wave end times
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r05c}
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_streamgen.py -x -q > $OUT/pytest_streamgen.log 2>&1 || { tail -30 $OUT/pytest_streamgen.log; exit 1; }
tail -2 $OUT/pytest_streamgen.log
for prof in c5 c5h; do
  python tools/ab_bench.py bystro-vcf_amd/libbvcf_base.so bystro-vcf_amd/libbvcf.so 3 -- --profile $prof --no-real-data > $OUT/ab_alone_$prof.txt 2>&1
  cat $OUT/ab_alone_$prof.txt
  for rnd in 1 2; do for lib in libbvcf_base.so libbvcf.so; do
    BVCF_LIB=$R/bystro-vcf_amd/$lib python bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data | tail -1 > $OUT/line.json
    python -c "import json,sys; d=json.load(open('$OUT/line.json')); print('$prof $lib value %.1f M/s  alone %.4f ms frac %.3f chain_frac %.3f' % (d['value']/1e6, d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['roofline']['chain_frac']))" | tee -a $OUT/inflight.txt
  done; done
done
BVCF_LIB=$R/bystro-vcf_amd/libbvcf_times2.so python tools/wave_ends.py c5 > $OUT/wave_ends_dyn.txt 2>&1; cat $OUT/wave_ends_dyn.txt
BVCF_DYN_TILES=0 BVCF_LIB=$R/bystro-vcf_amd/libbvcf_times2.so python tools/wave_ends.py c5 > $OUT/wave_ends_static.txt 2>&1; cat $OUT/wave_ends_static.txt
BVCF_LIB=$R/bystro-vcf_amd/libbvcf_times2.so python tools/wave_ends.py c3 > $OUT/wave_ends_c3.txt 2>&1; cat $OUT/wave_ends_c3.txt
