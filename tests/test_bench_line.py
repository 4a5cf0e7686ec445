"""bench.py's record: the LAST stdout line must be one JSON object the driver can read (it keeps 8 KB of stdout; round 4's
53.6 KB line was cut and nothing was parsed), and `--gpus N` must launch itself.  No GPU: the line is built from stub leg
results (round 4's own full record, which is tracked under profiles/), and the N > 1 path runs --cpu-dry over gloo."""
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = ("e2e", "e2e_bgzf", "e2e_c2", "e2e_c4", "e2e_c5", "e2e_stdin_text", "e2e_stdin_bgzf")


def _stub_full():
    with open(os.path.join(ROOT, "profiles", "r04_bench_default_line.json")) as f:
        return json.load(f)


def test_compact_line_is_small_parses_and_keeps_the_headline():
    full = _stub_full()
    assert len(json.dumps(full)) > 40_000  # the stub is the record that was too big
    txt = bench.compact_line(full)
    assert "\n" not in txt and len(txt) < 3072, len(txt)
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BASELINE configs[2]")
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-4
    rf = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "mean_launch_ms", "chain_frac"):
        assert k in rf, k
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "variants/s" and cb["sample"]
    for leg in LEGS:
        sm = line[leg]
        assert set(sm) >= {"wall_s", "variants_per_min", "steady_variants_per_s", "sha256_equal"}, leg
        assert sm["wall_s"] > 0
    assert line["e2e"]["sha256_equal"] is True and line["e2e_c4"]["sha256_equal"] is True


def test_compact_line_survives_failed_and_oversized_legs():
    full = _stub_full()
    full["e2e_c5"] = {"input": "x", "runs": [], "error": "the CLI did not finish within 300 s " * 20}
    full["host_legs_error"] = "E" * 5000
    full["per_rank_variants_per_s"] = [4.6e8 + i for i in range(8)]
    full["ranks_seen"] = 8
    full["e2e_all_devices"] = {"text": dict(full["e2e"]), "bgzf": dict(full["e2e_bgzf"]), "full_output_check": {"equal": True}}
    full["cpu_baseline"]["sample"] = "s" * 4000
    txt = bench.compact_line(full)
    assert len(txt) <= bench.LINE_LIMIT
    line = json.loads(txt)
    assert "error" in line["e2e_c5"] and line["roofline"]["frac"] and line["cpu_baseline"]["value"]
    assert line["e2e_all_devices_text"]["sha256_equal"] is True


def test_emit_prints_the_compact_line_last_and_writes_the_full_record(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _stub_full()
    print("some earlier noise")
    bench.emit(full)
    out = capsys.readouterr().out.splitlines()
    assert json.loads(out[-1])["value"] > 0 and len(out[-1]) < 3072
    assert json.load(open(tmp_path / "bench_full.json"))["e2e"]["runs"]


def _run_bench(argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=timeout, cwd=str(ROOT))
    return p.returncode, p.stdout.decode().splitlines(), p.stderr.decode()


def test_gpus_2_launches_itself_cpu_dry():
    """`python bench.py --gpus 2 --cpu-dry` with no launcher and no WORLD_SIZE: the parent starts two ranks as a child
    torch.distributed.run, relays rank 0's line, exits with its code"""
    rc, out, err = _run_bench(["--gpus", "2", "--cpu-dry", "--steps", "2", "--warmup", "1", "--profile", "c4", "--samples", "40"])
    assert rc == 0, err[-2000:]
    line = json.loads(out[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["per_rank_variants_per_s"]) == 2
    assert all(v > 0 for v in line["per_rank_variants_per_s"])
    assert line["metric"].startswith("cpu-dry") and line["scaling"] == "weak"
    # two ranks x two blocks x 400 rows x 2 steps, every row one variant
    assert line["config"]["rows_per_step_per_gpu"] == 800
    assert len(out[-1]) < 3072


def test_gpus_1_cpu_dry_runs_in_process_and_child_failure_is_relayed():
    rc, out, err = _run_bench(["--gpus", "1", "--cpu-dry", "--steps", "1", "--warmup", "0", "--profile", "c2"])
    assert rc == 0, err[-2000:]
    assert json.loads(out[-1])["n_gpus"] == 1
    # a bad flag inside the children: the parent's exit code is not 0 and no line is invented
    rc, out, err = _run_bench(["--gpus", "2", "--cpu-dry", "--profile", "nope"])
    assert rc != 0 and not any(ln.startswith('{"metric"') for ln in out)
