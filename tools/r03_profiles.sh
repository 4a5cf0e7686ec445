#!/bin/bash
# everything profiles/r03_* holds beyond tools/refresh_profiles.sh (run from the repo root on the GPU box)
R=$PWD
OUT=$R/gpurun_out/prof
mkdir -p $OUT
bash tools/refresh_profiles.sh stats pmc > $OUT/refresh.log 2>&1
python3 tools/derive_traffic.py $OUT > $OUT/k_gt_hbm_traffic.json 2> $OUT/derive.err
KERNEL=k_sites2 ARGS="--profile c2" TAG=k_sites2 bash tools/pmc_sq.sh > $OUT/pmc_sq_k_sites2.txt 2>&1
KERNEL=k_stream ARGS="--path 2" TAG=k_stream bash tools/pmc_sq.sh > $OUT/pmc_sq_k_stream.txt 2>&1
VARIANTS="0 1 2" bash tools/inflate_prof.sh 3840 6 > $OUT/inflate_kernels_level6.txt 2>&1
VARIANTS="0 2" bash tools/inflate_prof.sh 3840 1 > $OUT/inflate_kernels_level1.txt 2>&1
bash tools/prof_cli_bgzf.sh 400000 > $OUT/cli_bgzf.txt 2>&1
cp $R/gpurun_out/prof_cli_bgzf_kernel_stats.csv $OUT/cli_bgzf_kernel_stats.csv
python3 tools/e2e_cli.py 400000 c3 --runs=3 --check > $OUT/e2e_cli_c3_400k.log 2>&1
python3 tools/e2e_cli.py 400000 c3 --bgzf --runs=3 --check > $OUT/e2e_cli_bgzf_device.log 2>&1
BVCF_DEVICE_INFLATE=0 python3 tools/e2e_cli.py 400000 c3 --bgzf --runs=2 > $OUT/e2e_cli_bgzf_host_inflate.log 2>&1
python3 tools/e2e_cli.py 200000 c5 --bgzf --runs=2 --check > $OUT/e2e_cli_c5_bgzf_device.log 2>&1
BVCF_DEVICE_INFLATE=0 python3 tools/e2e_cli.py 200000 c5 --bgzf --runs=2 > $OUT/e2e_cli_c5_bgzf_host_inflate.log 2>&1
python3 tools/e2e_cli.py 1200000 c3 --runs=2 --devices=0,0 > $OUT/e2e_cli_c3_1200k_two_workers_one_gpu.log 2>&1
./tools/hostreg_bench > $OUT/hostreg_bench.txt 2>&1
BVCF_RANGE_READ=0 python3 tools/e2e_cli.py 1200000 c3 --runs=2 > $OUT/e2e_cli_c3_1200k_single_reader.log 2>&1
python3 tools/e2e_cli.py 1200000 c3 --runs=3 > $OUT/e2e_cli_c3_1200k_range_readers.log 2>&1
ls $OUT | wc -l
