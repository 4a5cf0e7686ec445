#!/bin/bash
# round 5: what profiles/r05_* holds from the last build, one gpurun call (one box):
#   gpurun --timeout 1150 -- bash tools/r05_profiles.sh        (outputs: gpurun_out/prof5/, copied to profiles/r05_* afterwards)
# the kernels of configs[2] / [3] / the GT:DP:GQ profiles did not change this round (their experiments are recorded apart):
# kernel-trace stats under the round's bench.py for the record, the HBM traffic of the headline kernel re-measured, and the
# new kernels (k_render_*) under the CLI over a sites-only file
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-real-data "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $OUT/${tag}_kernel_stats.csv
  grep '^{"metric' $OUT/$tag.log | tail -1 > $OUT/${tag}_bench_line.json || true
  echo "== $tag"; cut -d, -f1-4 $OUT/${tag}_kernel_stats.csv | cut -c1-120 | head -8
}
pmc() {  # counter, tag, bench args...
  local ctr=$1 tag=$2; shift 2
  rm -rf /tmp/q_$tag
  rocprofv3 --pmc $ctr --output-format csv -d /tmp/q_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --blocks 3 --no-cpu-baseline --no-e2e --no-real-data "$@" > $OUT/$tag.log 2>&1
  cp "$(find /tmp/q_$tag -name '*counter_collection.csv' | head -1)" $OUT/$tag.csv
}
: > $OUT/bench_unprofiled_lines.json
for prof in c3 c4 c2 c5 c5h; do
  python3 $R/bench.py --profile $prof --no-e2e --no-cpu-baseline --no-real-data 2>/dev/null | grep '^{"metric' | tail -1 >> $OUT/bench_unprofiled_lines.json
done
echo "unprofiled lines done"
stats bench_c3_streaming --path 2
stats bench_c3_streaming_one_block_at_a_time --path 2 --slots 1
stats bench_c4_auto_one_block_at_a_time --profile c4 --slots 1
stats bench_c2_sites_only_packed_one_block_at_a_time --profile c2 --slots 1
stats bench_c5_general_stream_one_block_at_a_time --profile c5 --slots 1
pmc FETCH_SIZE pmc_fetch_size_streaming --path 2
pmc WRITE_SIZE pmc_write_size_streaming --path 2
echo "pmc traffic done"
# the CLI over 8 M sites-only rows, rows rendered on the device: the render kernels beside k_census_tiles / k_sites2p
cd $R
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import benchgen as bg, bystro_vcf_amd as bv
cfg = bg.make_cfg("c2")
with open("/dev/shm/r05_c2_prof.vcf", "wb") as f:
    f.write(bg.header(cfg))
    for b in range(8):
        t, n = bg.rows_device(cfg, b * 1_000_000, 1_000_000, pad=bv.DEVICE_PAD)
        f.write(t[:n].cpu().numpy().tobytes())
PY
cd /tmp
rm -rf /tmp/p_cli_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cli_c2 -- $R/bystro-vcf_amd/bystro-vcf --in /dev/shm/r05_c2_prof.vcf --out /dev/null > $OUT/cli_c2.log 2>&1
cp "$(find /tmp/p_cli_c2 -name '*kernel_stats.csv' | head -1)" $OUT/cli_c2_sites_only_rows_on_device_kernel_stats.csv
cut -d, -f1-4 $OUT/cli_c2_sites_only_rows_on_device_kernel_stats.csv | cut -c1-120 | head -10
rm -f /dev/shm/r05_c2_prof.vcf
echo "profiles done"
