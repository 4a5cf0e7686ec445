#!/bin/bash
# memory-path counters of ringbench's kernels (register pipeline and LDS-DMA ring at k_stream's load), one small --pmc pass each,
# to put beside the same counters of k_stream (tools/pmc_sq.sh with SETS=...): gpurun -- bash tools/pmc_ringbench.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum,TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum,TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum,TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum" "SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_ring_$i
  timeout -k 10 120 rocprofv3 --pmc ${set//,/ } --output-format csv -d $R/gpurun_out/pmc_ring_$i -- $R/tools/ringbench > $R/gpurun_out/pmc_ring_$i.log 2>&1 || { echo "set $i failed"; grep -m1 -i "exceeds\|error" $R/gpurun_out/pmc_ring_$i.log; }
  echo "set $i done" >&2
done
python3 - <<PY
import csv, glob, collections
for i in range(1, $i + 1):
    for f in glob.glob("$R/gpurun_out/pmc_ring_%d/**/*counter_collection.csv" % i, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_regs<48>" in k or "k_ring<48, 4>" in k:
                acc[(k.split("(")[0][-16:], row["Grid_Size"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-16s grid %-8s %-34s %.4g  (n=%d)" % (k[0], k[1], k[2], sum(v) / len(v), len(v)))
PY
