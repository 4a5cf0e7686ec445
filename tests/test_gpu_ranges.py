"""bvcf_run_fd over a regular file (`--in`): every device worker reads its own byte ranges (text) or compressed ranges
(BGZF) of the file -- the replacement for the reference's single producer (main.go:349-380) -- and the ordered sink
merges by (range, piece).  The ranges are cut at fixed offsets, inside lines and inside BGZF blocks: the output must be
the oracle's bytes in input order for any device list and any block size.  The GPU box has one device, so the lists
repeat ordinal 0 (two or three workers, each with its own ctx, reader, formatter and buffers)."""
import gzip
import json
import os
import random
import subprocess

import pytest

import bgzf
import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def _cli_file(args, path, env=None):
    e = dict(os.environ, BVCF_TIMING="json")
    e.update(env or {})
    p = subprocess.run([EXE, "--in", path] + args, capture_output=True, timeout=600, env=e)
    t = None
    log = []
    for ln in p.stderr.decode(errors="replace").splitlines(keepends=True):
        if ln.startswith("[bvcf timing-json] "):
            t = json.loads(ln[len("[bvcf timing-json] "):])
        elif not ln.startswith("[bvcf timing"):
            log.append(ln)
    return p, t, "".join(log)


_WANT = {}


def _want(bv, vcf, cfg=None):
    key = (len(vcf), hash(vcf[:4096] + vcf[-4096:]), tuple(sorted((cfg or {}).items())))
    if key not in _WANT:
        rc, out, log, n = orc.run(vcf, cfg or {}, n_threads=8)
        assert rc == 0
        _WANT.clear()  # (one entry: the golden file's rows are 200 MB)
        _WANT[key] = ((bv.string_header(cfg or {}) + "\n").encode() + out, log, n)
    return _WANT[key]


def _write(tmp_path, name, data):
    p = str(tmp_path / name)
    with open(p, "wb") as f:
        f.write(data)
    return p


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
@pytest.mark.parametrize("kind", ["text", "bgzf", "bgzf-small-blocks", "gzip"])
def test_file_input_any_device_list(bv, golden_1kg, tmp_path, devices, kind):
    """200 MB of real 1000-Genomes lines in 8 MiB batches: text and BGZF files go through the per-device range readers,
    a plain gzip file through the single reader; rows == the oracle's in input order == the reference's golden rows"""
    vcf, want_sorted, hdr = golden_1kg
    if kind == "text":
        data = vcf
    elif kind == "gzip":
        data = gzip.compress(vcf, 1)
    else:
        data = bgzf.bgzf_compress(vcf, block=0xFF00 if kind == "bgzf" else 20011, level=1)
    path = _write(tmp_path, "in.vcf" + ("" if kind == "text" else ".gz"), data)
    p, t, log = _cli_file(["--batchMB", "8", "--devices", devices], path)
    assert p.returncode == 0, p.stderr[-400:]
    want, want_log, n = _want(bv, vcf)
    assert p.stdout == want
    rows = p.stdout.split(b"\n")
    assert rows[0] == hdr and sorted(rows[1:-1]) == want_sorted
    assert log == want_log
    n_dev = len(devices.split(","))
    assert t["lines_in"] == n == t["counters"][0]
    if kind == "gzip":
        assert "one reader" in t["readers"]
    else:
        assert "per-device" in t["readers"]
        assert ("device" in t["input"]) == (kind != "text")
    if kind == "text":
        assert t["devices_used"] == n_dev
        blocks = [d["blocks"] for d in t["devices"]]
        assert sum(blocks) >= 24 and max(blocks) - min(blocks) <= 1   # range i -> worker i mod N
        assert t["counters"][5] == sum(d["bytes"] for d in t["devices"])


def _long_info_line(rng, pos, n_info, ns):
    info = "X=" + "".join(rng.choice("ACGT") for _ in range(64)) * (n_info // 64)
    gts = [rng.choice(["0|0"] * 8 + ["0|1", "1|1"]) for _ in range(ns)]
    return "\t".join(["7", str(pos), "rs%d" % pos, "A", "G", "50", "PASS", info, "GT"] + gts) + "\n"


@pytest.mark.parametrize("devices", ["0", "0,0,0"])
@pytest.mark.parametrize("compress", [False, True])
def test_ranges_with_long_lines(bv, tmp_path, devices, compress):
    """1 MiB batches (ranges of 896 KiB + 128 KiB of spare room): lines of 150-980 KB start at arbitrary offsets, so
    there are ranges that lie inside one line (they own nothing), lines that straddle a range end by more than the
    spare room (read separately), and BGZF batches whose look-ahead is many blocks"""
    rng = random.Random(5)
    ns = 60
    parts = [vcfgen.header(ns)]
    pos = 100
    for k in range(40):
        for _ in range(rng.randint(1, 400)):
            pos += 3
            parts.append(_long_info_line(rng, pos, 64, ns))
        pos += 3
        # (a BGZF batch is whole blocks: the text of the block a line starts in and of the one it ends in count against
        # max_batch_bytes too)
        parts.append(_long_info_line(rng, pos, rng.choice([150_000, 300_000, 700_000, 900_000 if compress else 980_000]), ns))
    vcf = "".join(parts).encode()
    cfg = {"keepInfo": True, "keepId": True}
    want, want_log, n = _want(bv, vcf, cfg)
    data = bgzf.bgzf_compress(vcf, block=50000, level=1) if compress else vcf
    path = _write(tmp_path, "long.vcf", data)
    p, t, log = _cli_file(["--batchMB", "1", "--devices", devices, "--keepInfo", "--keepId"], path)
    assert p.returncode == 0, log[-400:]
    assert p.stdout == want and log == want_log
    assert t["lines_in"] == n and "per-device" in t["readers"]
    # a line that does not fit max_batch_bytes is refused, as by the single reader
    too = vcf + _long_info_line(rng, pos + 3, 1_200_000, ns).encode() + _long_info_line(rng, pos + 6, 64, ns).encode()
    path = _write(tmp_path, "too_long.vcf", bgzf.bgzf_compress(too, level=1) if compress else too)
    p, t, log = _cli_file(["--batchMB", "1", "--devices", devices], path)
    assert p.returncode == 1 and b"a line is longer than max_batch_bytes" in p.stderr


def test_range_edges(bv, tmp_path):
    """files around the edges of the rule: no data, one line, an unterminated last line (dropped, main.go:354-358), a
    file shorter than one range, CRLF lines, a header longer than the first look"""
    hdr = vcfgen.header(3)
    line = "1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0"
    big_hdr = "##fileformat=VCFv4.2\n" + "".join("##contig=<ID=c%d,length=%d>\n" % (i, i) for i in range(60000)) + hdr.split("\n", 2)[2]
    cases = [hdr, hdr + line, hdr + line + "\n", hdr + (line + "\n") * 50 + line, (hdr + (line + "\n") * 50).replace("\n", "\r\n"),
             big_hdr + (line + "\n") * 7]
    for k, text in enumerate(cases):
        vcf = text.encode()
        want, want_log, n = _want(bv, vcf)
        for devices in ("0", "0,0"):
            for comp in (False, True):
                path = _write(tmp_path, "edge%d" % k, bgzf.bgzf_compress(vcf, block=4000) if comp else vcf)
                p, t, log = _cli_file(["--devices", devices], path)
                assert p.returncode == 0, (k, p.stderr[-300:])
                assert p.stdout == want and log == want_log, (k, devices, comp)
                assert t["lines_in"] == n
    # fatal paths keep the reference's messages
    for bad, msg in ((b"not a vcf\n", b"Not a VCF file"), (b"##fileformat=VCFv4.2\n##x\n", b"No header found")):
        for comp in (False, True):
            path = _write(tmp_path, "bad", bgzf.bgzf_compress(bad) if comp else bad)
            p, t, log = _cli_file([], path)
            assert p.returncode == 1 and msg in p.stderr
    # damage inside a compressed range, and a truncated file
    vcf = vcfgen.gen_vcf(3, 3000, 100)
    data = bytearray(bgzf.bgzf_compress(vcf, block=4000))
    data[len(data) // 2] ^= 0x10
    p, t, log = _cli_file(["--devices", "0,0"], _write(tmp_path, "dmg", bytes(data)))
    assert p.returncode == 1 and b"bgzf" in p.stderr
    p, t, log = _cli_file(["--devices", "0,0"], _write(tmp_path, "trunc", bgzf.bgzf_compress(vcf, block=4000)[:-100]))
    assert p.returncode == 1 and b"bgzf" in p.stderr


@pytest.mark.parametrize("via", ["file", "pipe"])
def test_bgzf_look_ahead_is_found_not_guessed(bv, tmp_path, via):
    """(a) small BGZF blocks under long lines (2 504 samples, 777-byte blocks: a line spans 13 blocks);
    (b) a late line several times the first one; (c) batches that end on or before an unterminated last line"""
    rng = random.Random(11)
    ns = 2504
    vcf_a = vcfgen.gen_vcf(31, 700, ns, weird=0.01)
    parts = [vcfgen.header(300)]
    for k in range(3000):
        parts.append(_long_info_line(rng, 10 + 3 * k, 64 if k % 500 != 499 else 400_000, 300))
    vcf_b = "".join(parts).encode()
    vcf_c = vcf_a.rstrip(b"\n")  # the last line has no terminator
    for name, vcf, block, mb in (("a", vcf_a, 777, 1), ("b", vcf_b, 0xFF00, 1), ("c", vcf_c, 9000, 1), ("c2", vcf_c, 0xFF00, 2)):
        want, want_log, n = _want(bv, vcf, {"keepInfo": True})
        data = bgzf.bgzf_compress(vcf, block=block, level=1, eof_marker=name != "c2")
        args = ["--batchMB", str(mb), "--keepInfo", "--devices", "0,0"]
        if via == "file":
            p, t, log = _cli_file(args, _write(tmp_path, name + ".vcf.gz", data))
            err = p.stderr
        else:
            p = subprocess.run([EXE] + args, input=data, capture_output=True, timeout=600)
            log, err = p.stderr.decode(), p.stderr
        assert p.returncode == 0, (name, err[-300:])
        assert p.stdout == want and log == want_log, name


def test_stdin_and_file_agree_on_fuzz(bv, tmp_path):
    """seeded fuzz files (junk lines, comments between records, wrong field counts) through the pipe reader and the range
    readers, three batch sizes"""
    for seed, ns, n_lines in ((71, 17, 9000), (72, 0, 30000), (73, 700, 1500)):
        vcf = vcfgen.gen_vcf(seed, n_lines, ns, fmt_extra=seed == 73, weird=0.04)
        want, want_log, n = _want(bv, vcf, {"keepId": True})
        path = _write(tmp_path, "fuzz%d.vcf" % seed, vcf)
        for mb in ("1", "2"):
            p, t, log = _cli_file(["--batchMB", mb, "--keepId", "--devices", "0,0,0"], path)
            assert p.returncode == 0 and p.stdout == want and log == want_log, (seed, mb)
            q = subprocess.run([EXE, "--batchMB", mb, "--keepId"], input=vcf, capture_output=True, timeout=600)
            assert q.returncode == 0 and q.stdout == want and q.stderr.decode() == want_log, (seed, mb)


def test_dosage_rows_keep_input_order_with_range_readers(bv, tmp_path):
    """--dosageOutput with three workers: the Arrow rows are appended by the ordered sink, so the file is the same as
    with one worker"""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.ipc as ipc
    vcf = vcfgen.gen_vcf(91, 6000, 120, weird=0.02)
    path = _write(tmp_path, "d.vcf", vcf)
    tables = []
    for devices in ("0", "0,0,0"):
        out = str(tmp_path / ("dosage_%d.arrow" % len(devices)))
        p, t, log = _cli_file(["--batchMB", "1", "--devices", devices, "--dosageOutput", out], path)
        assert p.returncode == 0, p.stderr[-300:]
        with open(out, "rb") as f:
            tables.append(ipc.open_file(f).read_all())
    assert tables[0].num_rows > 1000 and tables[0].equals(tables[1])


def test_bgzf_file_with_fields_beyond_gt(bv, tmp_path):
    """a GATK-style cohort file (GT:DP:GQ, 700 samples) as BGZF: the first batch is launched before anyone has seen its
    text, so the driver tells the ctx from the header blocks it inflated itself which streaming kernel to start with"""
    vcf = vcfgen.gen_vcf(74, 1500, 700, fmt_extra=True, weird=0.01)
    want, want_log, n = _want(bv, vcf, {"keepId": True})
    path = _write(tmp_path, "gatk.vcf.gz", bgzf.bgzf_compress(vcf, level=6))
    for devices in ("0", "0,0"):
        p, t, log = _cli_file(["--batchMB", "2", "--keepId", "--devices", devices], path)
        assert p.returncode == 0 and p.stdout == want and log == want_log, devices
        assert t["lines_in"] == n and "device" in t["input"]


@pytest.mark.parametrize("kind", ["text", "bgzf"])
def test_numa_binding_path(bv, tmp_path, kind):
    """BVCF_NUMA=1 takes the path a run over several devices takes: every host thread of a worker binds itself to the
    CPUs of its device's NUMA node (best effort: a box without the sysfs entries binds nothing).  Same bytes either way."""
    vcf = vcfgen.gen_vcf(314, 3000, 300, weird=0.02)
    data = vcf if kind == "text" else bgzf.bgzf_compress(vcf, block=0xFF00, level=1)
    path = _write(tmp_path, "numa.vcf" + ("" if kind == "text" else ".gz"), data)
    want, want_log, n = _want(bv, vcf)
    for numa in ("1", "0"):
        p, t, log = _cli_file(["--batchMB", "1", "--devices", "0,0"], path, {"BVCF_NUMA": numa})
        assert p.returncode == 0, p.stderr[-400:]
        assert p.stdout == want and log == want_log and t["lines_in"] == n
        bound = [d["cpus_bound"] for d in t["devices"]]
        assert all(b >= 0 for b in bound) and (numa == "1" or not any(bound))
