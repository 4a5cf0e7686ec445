"""More than one ctx in one process: bvcf_run_fd deals the blocks of a stream round-robin to one ctx per
entry of the device list and merges by block number -- the output must be the same bytes in the same order for any
list.  The GPU box has one device, so the lists repeat ordinal 0: two or three ctxs alive on it at once."""
import json
import os
import subprocess
import threading

import numpy as np
import pytest

import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def _cli(args, data, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([EXE] + args, input=data, capture_output=True, timeout=600, env=e)


def _timing(stderr):
    for ln in stderr.decode().splitlines():
        if ln.startswith("[bvcf timing-json] "):
            return json.loads(ln[len("[bvcf timing-json] "):])
    return None


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
def test_cli_device_lists_give_identical_output(bv, golden_1kg, devices):
    """200 MB of real 1000-Genomes lines in 8 MiB blocks (25 of them) dealt to 1, 2 and 3 ctxs: rows == oracle rows
    in input order, and the run summary accounts for every block and line"""
    vcf, want_sorted, hdr = golden_1kg
    p = _cli(["--batchMB", "8", "--devices", devices], vcf, {"BVCF_TIMING": "json"})
    assert p.returncode == 0, p.stderr[-400:]
    rows = p.stdout.split(b"\n")
    assert rows[0] == hdr and rows[-1] == b""
    rc_o, out_o, log_o, n_o = orc.run(vcf)
    assert rows[1:-1] == out_o.split(b"\n")[:-1]          # same bytes, same order
    assert sorted(rows[1:-1]) == want_sorted               # and the reference's own golden output
    t = _timing(p.stderr)
    assert t is not None
    log_lines = [ln for ln in p.stderr.decode().splitlines() if not ln.startswith("[bvcf timing")]
    assert "\n".join(log_lines) + "\n" == log_o
    n_dev = len(devices.split(","))
    assert len(t["devices"]) == n_dev and t["devices_used"] == n_dev
    blocks = [d["blocks"] for d in t["devices"]]
    assert sum(blocks) >= 24 and max(blocks) - min(blocks) <= 1   # round-robin
    assert t["lines_in"] == n_o == t["counters"][0]
    assert t["counters"][5] == sum(d["bytes"] for d in t["devices"])
    assert t["count_gather"] == "host"  # ctxs that share a device: RCCL has one rank per device


def test_eight_workers_on_one_gpu_stay_within_the_cpu_share(bv, golden_1kg, tmp_path):
    """`--devices 0,0,0,0,0,0,0,0` over a FILE (per-device range readers): the threads that copy and format are budgeted
    over ALL workers from the CPU quota (bvcf_plan_threads), so eight workers start no more busy threads than one does;
    the OS thread count of the running process is sampled from /proc as the evidence.  Output == the one-worker run's."""
    import time
    vcf, want_sorted, hdr = golden_1kg
    path = tmp_path / "in.vcf"
    path.write_bytes(vcf)
    env = dict(os.environ, BVCF_TIMING="json")
    runs = {}
    for devices in ("0", "0,0,0,0,0,0,0,0"):
        with open(tmp_path / "out.tsv", "wb") as out, open(tmp_path / "err.txt", "wb") as err:
            p = subprocess.Popen([EXE, "--in", str(path), "--batchMB", "8", "--devices", devices], stdout=out, stderr=err, env=env)
            peak = 0
            while p.poll() is None:
                try:
                    for ln in open("/proc/%d/status" % p.pid):
                        if ln.startswith("Threads:"):
                            peak = max(peak, int(ln.split()[1]))
                except OSError:
                    pass
                time.sleep(0.002)
        assert p.returncode == 0, (tmp_path / "err.txt").read_bytes()[-400:]
        t = _timing((tmp_path / "err.txt").read_bytes())
        runs[devices] = (t, peak, (tmp_path / "out.tsv").read_bytes())
    t1, peak1, out1 = runs["0"]
    t8, peak8, out8 = runs["0,0,0,0,0,0,0,0"]
    assert out8 == out1 and sorted(out1.split(b"\n")[1:-1]) == want_sorted
    assert t8["devices_used"] == 8 and t8["lines_in"] == t1["lines_in"]
    cpus = t8["threads"]["cpus"]
    for t in (t1, t8):
        assert t["threads"]["busy_total"] <= max(cpus, 2 * len(t["devices"])), t["threads"]
    # what is not in busy_total waits: per worker its device thread and its readers' front threads; the writer, the main
    # thread and the HIP runtime's own helpers (measured: about a dozen)
    per_worker_waiting = 1 + max(1, t8["threads"]["readers_per_worker"])
    note = "OS threads at peak: one worker %d, eight workers %d (cpus %d, busy threads planned %d / %d)" % (
        peak1, peak8, cpus, t1["threads"]["busy_total"], t8["threads"]["busy_total"])
    print(note)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):  # (kept as evidence for DESIGN.md section 6)
        with open(os.path.join(ROOT, "gpurun_out", "thread_counts.txt"), "a") as f:
            f.write(note + "\n")
    # (measured: 19 with one worker, 50 with eight; a per-worker budget as before round 4 would start 8 x (2 x 4 + 8) = 128
    # busy threads alone)
    assert peak8 <= t8["threads"]["busy_total"] + 8 * per_worker_waiting + 48, (peak1, peak8, t8["threads"])


def test_cli_count_gather_is_rccl_only_on_request(bv):
    """the end-of-run count gather: the host sums the ctxs' counters unless BVCF_RCCL=1 asks for the all-reduce (its
    communicator bring-up is inside the run's wall time); the totals are the same"""
    vcf = vcfgen.gen_vcf(62, 400, 40, weird=0.02)
    t_host = _timing(_cli(["--devices", "0"], vcf, {"BVCF_TIMING": "json"}).stderr)
    p = _cli(["--devices", "0"], vcf, {"BVCF_TIMING": "json", "BVCF_RCCL": "1"})
    assert p.returncode == 0, p.stderr[-400:]
    t_rccl = _timing(p.stderr)
    assert t_host["count_gather"] == "host" and t_rccl["count_gather"] == "rccl"
    # (the eighth counter is kernel time)
    assert t_host["counters"][:7] == t_rccl["counters"][:7] and t_host["counters"][0] == t_host["lines_in"]


def test_cli_devices_flags_and_small_input(bv):
    vcf = vcfgen.gen_vcf(61, 500, 40, weird=0.02)
    want = (bv.string_header() + "\n").encode() + orc.run(vcf)[1]
    for args in (["--devices", "all"], ["--devices=0"], ["--device", "0"], ["--devices", "0,0,0,0"]):
        p = _cli(args, vcf, {"BVCF_TIMING": "json"})
        assert p.returncode == 0 and p.stdout == want, args
        assert _timing(p.stderr)["devices_used"] == 1  # one block: only the first ctx is ever created
    assert _cli(["--devices", "x"], vcf).returncode == 2
    assert _cli(["--devices", "99"], vcf).returncode == 1  # no such device: fatal, never a CPU path


def test_capacity_growth_with_two_ctxs(bv):
    """short junk lines between the records exceed the first reservation on both ctxs (see
    test_cli_many_batches_with_capacity_growth): each grows its own, the merge keeps input order"""
    import random
    rng = random.Random(78)
    ns = 300
    rows = [vcfgen.header(ns)]
    pos = 1000
    for k in range(16000):
        pos += rng.randint(1, 40)
        if k % 3 == 0:
            gts = ["0|1" if rng.random() < 0.02 else ("1|1" if rng.random() < 0.01 else "0|0") for _ in range(ns)]
            rows.append("\t".join(["chr2", str(pos), ".", "C", "T", ".", "PASS", ".", "GT"] + gts) + "\n")
        else:
            rows.extend(["chr2\t%d\t.\tA\tG\n" % pos] * (200 if k % 50 == 1 else 2))
    vcf = "".join(rows).encode()
    rc_o, out_o, log_o, _ = orc.run(vcf)
    p = _cli(["--batchMB", "1", "--devices", "0,0"], vcf)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == (bv.string_header() + "\n").encode() + out_o
    assert p.stderr.decode() == log_o


def test_two_ctxs_driven_from_two_threads(bv):
    """the ABI's promise (include/bvcf.h): distinct ctxs are independent and may be driven from different host
    threads.  Two ctxs on device 0, each fed its own blocks concurrently, both paths; results == a lone ctx's."""
    blocks = [vcfgen.gen_vcf(70 + i, 400, 2504 if i % 2 else 300, weird=0.02) for i in range(4)]

    def body(vcf):
        return vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:]

    def n_hdr(vcf):
        i = vcf.index(b"#CHROM")
        return vcf[i:vcf.index(b"\n", i)].count(b"\t") + 1

    def key(b):
        from numpy.lib.recfunctions import repack_fields
        L = repack_fields(b.lines[["off", "len", "n_rec", "status", "site_type", "n_fields"]])
        slots = [s for i in range(len(b.lines)) for s in b.record_slots(i)]
        A = repack_fields(b.alleles[slots][["pos", "alt_idx", "alt_len", "ac", "an", "n_het", "n_hom", "n_miss", "ref", "alt_base",
                                            "kind", "trtv"]])
        cls = [b.classes(b.alleles[s]).tobytes() for s in slots if int(b.alleles[s]["cmap_off"]) != bv.NO_CMAP]
        return L.tobytes(), A.tobytes(), cls

    # what a lone ctx returns, per block and device path (the streaming path lists only lines with the right field count)
    want = {}
    for i, vcf in enumerate(blocks):
        for path in (0, 1, 2):
            ctx = bv.Ctx(n_hdr(vcf), allow="", path=path)
            want[(i, path)] = key(ctx.process(body(vcf)))
            ctx.close()

    got, errs = {}, []

    def run(ids, path):
        try:
            for rep in range(3):
                for i in ids:
                    ctx = bv.Ctx(n_hdr(blocks[i]), allow="", path=path)
                    got[(i, path, rep)] = key(ctx.process(body(blocks[i])))
                    ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=([0, 1], 1)), threading.Thread(target=run, args=([2, 3], 2)),
          threading.Thread(target=run, args=([1, 2], 0))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (i, path, rep), k in got.items():
        assert k == want[(i, path)], (i, path, rep)


def test_count_gather_over_rccl(bv, monkeypatch):
    """the path's one collective: ncclAllReduce(sum) of the uint64[8] counters.  One device here, so the RCCL leg is
    forced for a single rank (BVCF_RCCL=1) and must return that ctx's own totals; two ctxs on one device fall to the
    host sum (RCCL has one rank per device) and must add up."""
    vcf = vcfgen.gen_vcf(90, 600, 64, weird=0.02)
    body = vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:]
    a, b = bv.Ctx(9 + 64, allow=""), bv.Ctx(9 + 64, allow="")
    try:
        a.process(body)
        b.process(body)
        b.process(body)
        ca, cb = a.counters(), b.counters()
        n = body.count(b"\n")
        assert ca[0] == n and cb[0] == 2 * n
        tot, used = bv.allreduce_counters([a, b])
        assert not used and tot == [x + y for x, y in zip(ca, cb)]
        monkeypatch.setenv("BVCF_RCCL", "1")
        tot1, used1 = bv.allreduce_counters([b])
        assert used1 and tot1 == cb
    finally:
        a.close()
        b.close()
