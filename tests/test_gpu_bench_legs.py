"""bench.py's N > 1 leg -- rank 0's `bystro-vcf --devices 0,..,N-1` run over text and BGZF with the whole output hashed
against the oracle -- cannot run on the one-GPU boxes with N > 1; its code path can, with the one device there is: the file
is written, the BGZF twin built, the CLI run twice over each, the hashes compared.  (A crash in there on the first 8-GPU
node would cost the scaling record.)"""
import argparse
import json

import pytest

pytestmark = pytest.mark.gpu


def test_all_devices_leg_with_the_one_device():
    import torch
    import bench
    import benchgen as bg
    import bystro_vcf_amd as bv
    cfg = bg.make_cfg("c3")
    args = argparse.Namespace(blocks=2, rows=12_000, all_devices_rows=24_000, profile="c3")
    blocks, sizes = [], []
    for first in bench.rank_blocks(0, args.blocks, args.rows):
        t, n = bg.rows_device(cfg, first, args.rows, pad=bv.DEVICE_PAD)
        blocks.append(t)
        sizes.append(n)
    line = {}
    bench.all_devices_leg(line, args, cfg, bg, bv, blocks, sizes, 1, lambda: None)
    assert "host_legs_error" not in line, line
    leg = line["e2e_all_devices"]
    assert leg["devices"] == "0" and leg["full_output_check"]["equal"], leg["full_output_check"]
    for k in ("text", "bgzf"):
        assert leg[k]["rows"] == 24_000 and leg[k]["wall_s"] > 0 and "error" not in leg[k], leg[k]
    # ... and the compact line carries its summaries
    full = {"metric": "variants/sec", "value": 1.0, "unit": "variants/s", "n_gpus": 2, "config": {"workload": "x"}, "e2e_all_devices": leg,
            "ranks_seen": 2, "per_rank_variants_per_s": [1.0, 1.0]}
    out = json.loads(bench.compact_line(full))
    assert out["e2e_all_devices_text"]["sha256_equal"] is True and out["e2e_all_devices_bgzf"]["wall_s"] > 0
    del blocks
    torch.cuda.empty_cache()


def test_bench_main_takes_the_multi_rank_path_with_one_rank():
    """bench.py's main() through every N > 1 branch -- RCCL process group, the gloo group for the host-side wait, the
    reductions, rank 0's all-devices leg, the barrier, the compact line -- with ONE rank (BVCF_BENCH_FORCE_MULTI=1 and the
    launcher's environment): what the driver's torch.distributed.run starts on an 8-GPU node, minus the other seven"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               BVCF_BENCH_FORCE_MULTI="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--blocks", "2",
                        "--rows", "12000", "--all-devices-rows", "24000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                       timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    last = p.stdout.decode().splitlines()[-1]
    assert len(last) < 3072
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and len(line["per_rank_variants_per_s"]) == 1 and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line  # (cpu_baseline: rank 0 at N == 1 only)
    assert line["e2e_all_devices_text"]["sha256_equal"] is True and line["e2e_all_devices_bgzf"]["wall_s"] > 0
