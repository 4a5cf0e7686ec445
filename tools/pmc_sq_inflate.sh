#!/bin/bash
# SQ counters of the DEFLATE decoder on tools/inflate_bench.py's blocks, one rocprofv3 --pmc pass per set
# (run from the repo root on the GPU box):  bash tools/pmc_sq_inflate.sh [blocks=3840] [level=6]
R=$PWD
N=${1:-3840}
LEVEL=${2:-6}
SETS="SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM SQ_INSTS_BRANCH,SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_VALU"
cd /tmp && export TMPDIR=/tmp
i=0
for set in $SETS; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_inflate_$i
  rocprofv3 --pmc ${set//,/ } --output-format csv -d $R/gpurun_out/pmc_inflate_$i -- python3 $R/tools/inflate_bench.py $N 2 $LEVEL > $R/gpurun_out/pmc_inflate_$i.log 2>&1 || { echo "set $i failed"; tail -3 $R/gpurun_out/pmc_inflate_$i.log; }
done
echo "k_inflate_w4 over $N blocks (level $LEVEL): counters per dispatch, and per block"
python3 - <<PY
import csv, glob, collections
for i in range(1, $i + 1):
    for f in glob.glob("$R/gpurun_out/pmc_inflate_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_inflate" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"].split("(")[0].split("::")[-1], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (kn, k), v in sorted(acc.items()):
            m = sum(v) / len(v)
            print("%-16s %-24s %.4g  per block %.4g  (n=%d)" % (kn, k, m, m / $N, len(v)))
PY
