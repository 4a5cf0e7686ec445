#!/bin/bash
# SQ counters of k_inflate over the CLI on a BGZF file (run from the repo root on the GPU box)
R=$PWD
python3 tools/e2e_cli.py ${1:-100000} c3 --bgzf --runs=1 --keep=/dev/shm/bvcf_prof.vcf > $R/gpurun_out/pmc_cli_gen.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pmc_cli_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_cli_$i -- $R/bystro-vcf_amd/bystro-vcf --in /dev/shm/bvcf_prof.vcf.gz --out /dev/null > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in range(1, $i + 1):
    for f in glob.glob("/tmp/pmc_cli_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "${KERNEL:-k_inflate}" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("%-28s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
rm -f /dev/shm/bvcf_prof.vcf /dev/shm/bvcf_prof.vcf.gz
