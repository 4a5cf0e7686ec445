#!/bin/bash
# kernel begin/end stamps with blocks in flight (rocprofv3 --kernel-trace), for tools/trace_overlap.py
O=gpurun_out/trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prof in c4 c3; do
rocprofv3 --kernel-trace --output-format csv -d $R/$O/$prof -- python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data --steps 6 --warmup 2 > $R/$O/$prof.out 2> $R/$O/$prof.err || { tail -5 $R/$O/$prof.err; exit 1; }
f=$(find $R/$O/$prof -name "*kernel_trace.csv" | head -1)
python3 - $f $R/$O/${prof}_bvcf_trace.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if "bvcf_dev" in r["Kernel_Name"]]
w=csv.writer(open(sys.argv[2],"w"))
w.writerow(["name","queue","stream","start","end"])
for r in keep: w.writerow([r["Kernel_Name"].split("(")[0].replace("bvcf_dev::",""), r.get("Queue_Id",""), r.get("Stream_Id",""), r["Start_Timestamp"], r["End_Timestamp"]])
print(len(keep),"kernels")
PY
rm -rf $R/$O/$prof
done
