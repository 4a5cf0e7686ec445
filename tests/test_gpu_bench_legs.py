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
