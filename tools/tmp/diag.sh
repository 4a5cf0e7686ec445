cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os, subprocess, sys
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
for name, env in (("host", {"BVCF_RENDER_SITES": "0"}), ("device", {"BVCF_RENDER_SITES": "1"}), ("device", {"BVCF_RENDER_SITES": "1"}), ("host", {"BVCF_RENDER_SITES": "0"})):
    p = subprocess.run(["bystro-vcf_amd/bystro-vcf", "--in", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BVCF_TIMING="json", **env))
    st = [json.loads(l[len("[bvcf timing-json] "):]) for l in p.stderr.decode().splitlines() if l.startswith("[bvcf timing-json] ")][-1]
    print(name, json.dumps(st))
os.unlink(path)
PY
