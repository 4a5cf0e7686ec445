run() { python bench.py --profile c2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.3g M/s  ms_per_block %.4f  kernel_alone_ms %.4f chain_alone_ms %.4f chain_frac %.3f frac %.3f' % (d['value']/1e6, r['ms_per_block'], r['mean_launch_ms'], r['chain_ms_one_block_at_a_time'], r['chain_frac'], r['frac']))"; }
echo "k_sites2"; run; run
echo "k_sites2 + register prefetch"; BVCF_LIB=$PWD/bystro-vcf_amd/libbvcf_pf.so run;  BVCF_LIB=$PWD/bystro-vcf_amd/libbvcf_pf.so run
for w in 2 1; do echo "k_sites2, $w wg/cu"; BVCF_SITES1_WGS=$w run; done
