#!/bin/bash
# same-box A/B of the default bench line's kernel legs (synthetic block + real_data) between an exported older tree
# (_r03/: `git archive <rev> | tar -x -C _r03`, built there) and this one, interleaved.  gpurun -- bash tools/r04_real_ab.sh
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-real_ab}
mkdir -p $OUT
for i in 1 2; do
  for t in _r03 .; do
    n=$([ $t = . ] && echo head || echo r03)
    (cd $R/$t && python3 bench.py --no-e2e --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/${n}_$i.json) || exit 1
    python3 - $OUT/${n}_$i.json $n <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]; rd = d.get("real_data", {})
print("%-4s value %.1f M  k_stream %.4f ms (%.3f)  chain_frac %.3f | real %.4f ms  ratio %.3f  chain %.4f ms" % (
    sys.argv[2], d["value"] / 1e6, r["mean_launch_ms"], r["frac"], r["chain_frac"], rd.get("mean_launch_ms", 0),
    rd.get("ratio_to_synthetic_kernel_rate", 0), rd.get("chain_ms_one_block_at_a_time", 0)), flush=True)
P
  done
done
