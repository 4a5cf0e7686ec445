#!/bin/bash
# k_stream_gen A/B on one box: c5 / c5h bench lines (two blocks in flight; `roofline` has the kernel one block at a time)
# for several libraries, interleaved.  LIBS="libbvcf.so libbvcf_prev.so" PROFILES="c5 c5h" TAG=r04x bash tools/r04_gen.sh
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r04i}
mkdir -p $OUT
for rnd in 1 2; do
for lib in ${LIBS:-libbvcf.so libbvcf_prevgen.so}; do
  [ -f $R/bystro-vcf_amd/$lib ] || continue
  for prof in ${PROFILES:-c5 c5h}; do
    BVCF_LIB=$R/bystro-vcf_amd/$lib python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data >> $OUT/bench_${prof}_$lib.jsonl 2>> $OUT/bench.err || exit 1
  done
done
done
echo gen done
