#!/bin/bash
# rocprofv3 kernel-trace summary of the bench chain (run from the repo root on the GPU box): $1 = path (1 census, 2 streaming), $2 = tag
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --path $1 ${@:3} > $R/gpurun_out/prof_$2.log 2>&1
f=$(find /tmp/prof_$2 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof_$2_kernel_stats.csv
tail -1 $R/gpurun_out/prof_$2.log
