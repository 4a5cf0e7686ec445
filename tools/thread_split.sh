#!/bin/bash
# text e2e (configs[2] rows from /dev/shm) under different splits of the CPU share between the readers' copy threads and
# the formatter pool: BVCF_READ_THREADS = pread threads per reader (two readers), BVCF_FORMAT_THREADS = formatter threads.
#   gpurun -- bash tools/thread_split.sh        (writes gpurun_out/thread_split.txt)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/thread_split.txt
: > $OUT
for rt_ft in "4 8" "5 6" "6 4" "5 8" "6 6" "6 8" "8 8" "4 8"; do
  set -- $rt_ft
  BVCF_READ_THREADS=$1 BVCF_FORMAT_THREADS=$2 python3 $R/tools/e2e_cli.py ${ROWS:-2400000} c3 --runs=2 --json 2>/dev/null | python3 -c "
import json,sys
best=None
for ln in sys.stdin:
    i=ln.find('[bvcf timing-json] ')
    if i < 0: continue
    st=json.loads(ln[i+19:])
    if 'steady_s' in st and (best is None or st['steady_s']<best['steady_s']): best=st
print('copy 2x$1 format $2:', 'steady %.3f s' % best['steady_s'], 'gpu_wait %.3f' % best['gpu_wait_max_s'], 'reader_busy %.3f' % best['reader_busy_max_s'], 'formatter_busy %.3f' % best['formatter_busy_s'], 'wait_for_formatter %.3f' % best['wait_for_formatter_max_s'], 'user %.1f sys %.1f' % (best['user_cpu_s'], best['system_cpu_s']))
" | tee -a $OUT
done
