#!/bin/bash
# round 5: 20 M sites-only rows as a BGZF file through the CLI -- rows rendered on the device, only the lines left to the host
# come back as text (default) against the whole text coming back (BVCF_CUT_TEXT=0) and against host rows (BVCF_RENDER_SITES=0)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${TAG:-r05o}
mkdir -p $OUT
cd $R
python - <<'PY' > $OUT/e2e_c2_bgzf.txt 2>&1
import json, os, subprocess, sys, time, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, benchgen as bg, bystro_vcf_amd as bv, bgzf
import bench
cfg = bg.make_cfg("c2")
t, n = bg.rows_device(cfg, 0, 1_000_000, pad=bv.DEVICE_PAD)
b0 = t[:n].cpu().numpy(); del t
members = bench.bgzf_of(memoryview(b0), threads=16)
hdr = bg.header(cfg)
path = "/dev/shm/r05_c2.vcf.gz"
with open(path, "wb") as f:
    f.write(bench.bgzf_of(memoryview(hdr)))
    for _ in range(20):
        f.write(members)
    f.write(bgzf.bgzf_block(b""))
torch.cuda.empty_cache()
rows = 20_000_000
print("file %.3f GB for %.2f GB of text" % (os.path.getsize(path) / 1e9, (len(hdr) + 20 * n) / 1e9))
CLI = "bystro-vcf_amd/bystro-vcf"
def run(env):
    e = dict(os.environ, BVCF_TIMING="json", **env)
    t0 = time.perf_counter()
    p = subprocess.run([CLI, "--in", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
    wall = time.perf_counter() - t0
    st = [json.loads(l[len("[bvcf timing-json] "):]) for l in p.stderr.decode().splitlines() if l.startswith("[bvcf timing-json] ")]
    return wall, (st[-1] if st else {}), p.returncode
def sha(env):
    p = subprocess.Popen([CLI, "--in", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
    h = hashlib.sha256()
    for c in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(c)
    p.wait()
    return h.hexdigest()
hs = {}
for name, env in (("host rows, whole text back", {"BVCF_RENDER_SITES": "0"}), ("device rows, whole text back", {"BVCF_CUT_TEXT": "0"}), ("device rows, cut lines' text back", {})):
    hs[name] = sha(env)
    for rep in range(3):
        wall, st, rc = run(env)
        keys = ("steady_s", "gpu_wait_max_s", "wait_for_formatter_max_s", "formatter_busy_s")
        print("%-34s rc %d wall %.3f s  steady %.3f s = %.1f M variants/s  %s" % (
            name, rc, wall, st.get("steady_s", 0), rows / max(st.get("steady_s", 1e-9), 1e-9) / 1e6, " ".join("%s=%.3f" % (k, st[k]) for k in keys if k in st)))
print("whole-output sha256 equal:", len(set(hs.values())) == 1, list(hs.values())[0][:16])
os.unlink(path)
PY
cat $OUT/e2e_c2_bgzf.txt
