#!/bin/bash
# SQ / SQC counters of one kernel, one rocprofv3 --pmc pass per set (run from the repo root on the GPU box)
#   KERNEL=k_stream ARGS="--path 2" TAG=k_stream bash tools/pmc_sq.sh      (defaults)
R=$PWD
KERNEL=${KERNEL:-k_stream}
ARGS=${ARGS:---path 2}
TAG=${TAG:-$KERNEL}
SETS=${SETS:-"SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,SQ_ACTIVE_INST_MISC SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM SQ_IFETCH,SQ_LDS_BANK_CONFLICT,SQ_LDS_ADDR_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_BRANCH,SQ_WAIT_INST_LDS SQ_WAVES,SQ_CYCLES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE"}
cd /tmp && export TMPDIR=/tmp
i=0
for set in $SETS; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --pmc ${set//,/ } --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e --no-real-data $ARGS > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "set $i ($set) failed"; grep -m1 -i "exceeds\|error" $R/gpurun_out/pmc_${TAG}_$i.log; }
  echo "set $i done" >&2
done
python3 - <<PY
import csv, glob, collections
for i in range(1, $i + 1):
    for f in glob.glob("$R/gpurun_out/pmc_${TAG}_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "bvcf_dev::$KERNEL(" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("%-28s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
