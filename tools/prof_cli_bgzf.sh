#!/bin/bash
# rocprofv3 kernel-trace summary of the CLI over a BGZF file (run from the repo root on the GPU box): $1 = rows (default 200000)
R=$PWD
ROWS=${1:-200000}
python3 tools/e2e_cli.py $ROWS c3 --bgzf --runs=1 --keep=/dev/shm/bvcf_prof.vcf > $R/gpurun_out/prof_cli_bgzf_gen.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cli_bgzf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cli_bgzf -- $R/bystro-vcf_amd/bystro-vcf --in /dev/shm/bvcf_prof.vcf.gz --out /dev/null > $R/gpurun_out/prof_cli_bgzf.log 2>&1
f=$(find /tmp/prof_cli_bgzf -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof_cli_bgzf_kernel_stats.csv
cut -d, -f1-4 $R/gpurun_out/prof_cli_bgzf_kernel_stats.csv | head -14
rm -f /dev/shm/bvcf_prof.vcf /dev/shm/bvcf_prof.vcf.gz
