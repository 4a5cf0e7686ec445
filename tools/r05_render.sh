#!/bin/bash
# round 5, sites-only rows rendered on the device: parity (every sites-only test, rendered and not), then the CLI over 20 M
# configs[1] rows with and without (BVCF_RENDER_SITES), stage split kept
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r05g}
mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests/test_gpu_sites.py tests/test_abi.py -x -q -m "gpu or not gpu" > $OUT/pytest_sites.log 2>&1 || { tail -40 $OUT/pytest_sites.log; exit 1; }
tail -2 $OUT/pytest_sites.log
fi
python - <<'PY' > $OUT/e2e_c2.txt 2>&1
import json, os, subprocess, sys, time, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, benchgen as bg, bystro_vcf_amd as bv
cfg = bg.make_cfg("c2")
path = "/dev/shm/r05_c2.vcf"
rows, per = 20_000_000, 1_000_000
with open(path, "wb") as f:
    f.write(bg.header(cfg))
    for b in range(rows // per):
        t, n = bg.rows_device(cfg, b * per, per, pad=bv.DEVICE_PAD)
        f.write(t[:n].cpu().numpy().tobytes())
        del t
torch.cuda.empty_cache()
CLI = "bystro-vcf_amd/bystro-vcf"
def run(env, args=()):
    e = dict(os.environ, BVCF_TIMING="json", **env)
    t0 = time.perf_counter()
    p = subprocess.run([CLI, "--in", path] + list(args), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
    wall = time.perf_counter() - t0
    st = [json.loads(l[len("[bvcf timing-json] "):]) for l in p.stderr.decode().splitlines() if l.startswith("[bvcf timing-json] ")]
    return wall, (st[-1] if st else {}), p.returncode
def sha(env, args=()):
    p = subprocess.Popen([CLI, "--in", path] + list(args), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
    h = hashlib.sha256()
    for c in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(c)
    p.wait()
    return h.hexdigest()
for args in ((), ("--keepId", "--keepInfo", "--keepPos")):
    hs = {}
    for name, env in (("host rows", {"BVCF_RENDER_SITES": "0"}), ("device rows", {"BVCF_RENDER_SITES": "1"})):
        hs[name] = sha(env, args)
        for rep in range(3):
            wall, st, rc = run(env, args)
            keys = ("steady_s", "gpu_wait_max_s", "wait_for_formatter_max_s", "formatter_busy_s", "wait_for_reader_max_s", "write_s")
            print("%-12s %-30s rc %d wall %.3f s  steady %.3f s = %.1f M variants/s  %s" % (
                name, " ".join(args) or "(default flags)", rc, wall, st.get("steady_s", 0), rows / max(st.get("steady_s", 1e-9), 1e-9) / 1e6,
                " ".join("%s=%.3f" % (k, st[k]) for k in keys if k in st)))
    print("whole-output sha256 equal:", hs["host rows"] == hs["device rows"], hs["device rows"][:16])
os.unlink(path)
PY
cat $OUT/e2e_c2.txt
