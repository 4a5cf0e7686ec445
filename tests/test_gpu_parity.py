"""Parity of the HIP path (through the C-ABI) with the CPU oracle and with the reference's own
known answers / golden output.  Integer, byte and text results must be identical."""
import gzip
import os

import numpy as np
import pytest

import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["census", "streaming", "census-wide", "streaming-general"])
def bvcf_path(request, monkeypatch):
    """every parity test runs on both device paths (bvcf_params.path; BVCF_PATH overrides `choose`), on the census
    path with the regular scan split over waves as it is for cohorts of >= 32 768 samples (k_gt_wide), and on the
    streaming path with k_stream_gen -- the kernel for files whose sample fields carry more than GT -- pinned from the
    first batch (left alone the library switches to it after a batch of such lines)"""
    monkeypatch.setenv("BVCF_PATH", "2" if request.param.startswith("streaming") else "1")
    monkeypatch.setenv("BVCF_GEN_STREAM", "1" if request.param == "streaming-general" else "0")
    if request.param == "census-wide":
        monkeypatch.setenv("BVCF_WIDE", "1")
        monkeypatch.setenv("BVCF_WIDE_WIN", "1000")  # the general scan of one line in 1000-byte shares
    return request.param

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H8 = ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO"]


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def both(bv, vcf, cfg=None, **kw):
    """run the oracle and the HIP path on the same bytes; assert identical TSV and log"""
    rc_o, out_o, log_o, n_o = orc.run(vcf, cfg)
    rc_g, out_g, log_g, n_g = bv.run_buffer(vcf, cfg, **kw)
    assert (rc_g != 0) == (rc_o != 0), (rc_g, rc_o, log_g)
    assert n_g == n_o
    if out_g != out_o:
        a, b = out_o.split(b"\n"), out_g.split(b"\n")
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, "row %d differs:\noracle: %r\nhip:    %r" % (i, x[:300], y[:300])
        assert len(a) == len(b)
    assert log_g == log_o
    return out_g, log_g


def test_end_to_end_known_answers(bv, known_answers):
    from test_oracle import check_rows
    for case in known_answers["end_to_end"]:
        out, _ = both(bv, case["vcf"].encode(), case["config"])
        check_rows(case, out)


def test_get_alleles_known_answers(bv, known_answers):
    hdr = "##fileformat=VCFv4.x\n" + "\t".join(H8) + "\n"
    for case in known_answers["get_alleles"]:
        rec = "\t".join([case["chrom"], case["pos"], ".", case["ref"], case["alt"], ".", "PASS", "."]) + "\n"
        rc, out, _, _ = bv.run_buffer((hdr + rec).encode(), {"allow": ""})
        assert rc == 0
        rows = [r.split("\t") for r in out.decode().split("\n") if r]
        got = [[r[1], r[3], r[4]] for r in rows]
        assert got == [a[:3] for a in case["alleles"]], case["cite"]
        if rows:
            assert {r[2] for r in rows} == {case["type"]}, case["cite"]


def test_make_het_hom_known_answers(bv, known_answers):
    """per-allele ac/an/het/hom/missing straight from bvcf_collect"""
    for case in known_answers["make_het_hom"]:
        fields = case["line"].split("\t")[:case["n_header"]]  # drop the tests' stray 14th field
        a = int(case["allele"])
        fields[4] = ",".join("TGAC"[:max(a, 1)])
        line = ("\t".join(fields) + "\n").encode()
        ctx = bv.Ctx(case["n_header"], allow="")
        b = ctx.process(line)
        ctx.close()
        assert b.lines["status"][0] == bv.LINE_OK, case["cite"]
        rec = [r for r in b.records(0) if r["alt_idx"] == a - 1][0]
        got = (int(rec["n_hom"]), int(rec["n_het"]), int(rec["n_miss"]), int(rec["ac"]), int(rec["an"]))
        want = (case["n_hom"], case["n_het"], case["n_missing"], case["ac"], case["an"])
        assert got == want, case["cite"]
        cls = b.classes(rec)
        ocls, _, _, _ = orc.make_het_hom(case["line"], case["n_header"], case["allele"])
        assert cls.tolist() == ocls, case["cite"]


def test_golden_1kg(bv, golden_1kg):
    """the reference's regression pair: 19 747 rows x 2 504 samples -> 19 821 rows"""
    vcf, want_sorted, _ = golden_1kg
    out, log = both(bv, vcf)
    rows = out.split(b"\n")
    assert rows[-1] == b"" and len(rows) - 1 == 19821
    assert sorted(rows[:-1]) == want_sorted
    assert log.count("ALT not ACTG") == 17


def test_golden_1kg_small_batches_and_flags(bv, golden_1kg):
    vcf = golden_1kg[0]
    cut = vcf[: vcf.index(b"\n", 40_000_000) + 1]
    cfg = {"keepId": True, "keepInfo": True, "keepPos": True, "allow": "PASS", "emptyField": "NA", "fieldDelimiter": "|"}
    both(bv, cut, cfg, max_batch_bytes=3 << 20)


def test_example_query_vcf(bv):
    """BASELINE config 0: examples/test.query.vcf (60 samples, GT:AD:DP:GQ:PL, FILTER '.')"""
    with gzip.open(os.path.join(ROOT, "tests", "golden", "test.query.vcf.gz"), "rb") as f:
        vcf = f.read()
    out, _ = both(bv, vcf)
    assert len(out.split(b"\n")) - 1 == 879
    both(bv, vcf, {"keepId": True, "keepInfo": True, "allow": "*"})


@pytest.mark.parametrize("seed,n_lines,n_samples,fmt_extra,weird", [
    (1, 400, 0, False, 0.05), (2, 400, 1, False, 0.1), (3, 300, 7, False, 0.1), (4, 300, 64, False, 0.02),
    (5, 200, 255, False, 0.02), (6, 200, 256, False, 0.0), (7, 200, 257, False, 0.01), (8, 150, 1000, False, 0.002),
    (9, 200, 33, True, 0.05), (10, 100, 300, True, 0.01), (11, 60, 2504, False, 0.001), (12, 300, 5, False, 0.5),
    # > 2 560 samples: lines no longer fit the streaming kernel's chunk registers (one line at a time);
    # > 16 384 samples: the class map is staged and flushed in several 4 KiB windows
    (13, 40, 3000, False, 0.001), (14, 14, 17000, False, 0.0003), (15, 10, 33000, False, 0.0),
    # sample counts that are multiples of 256 end on a full last chunk (the dword-aligned loads fall back to
    # unaligned ones there), one less/more does not
    (16, 60, 512, False, 0.002), (17, 30, 2560, False, 0.001), (18, 30, 2559, False, 0.001), (19, 40, 1024, False, 0.01),
    (20, 30, 2304, False, 0.001), (21, 30, 2305, False, 0.001),
    # one case per remaining instance of the streaming kernel's pipeline (it is specialised per chunk count 1..10)
    (22, 60, 700, False, 0.003), (23, 50, 1200, False, 0.002), (24, 40, 1500, False, 0.002), (25, 40, 1700, False, 0.002),
    (26, 40, 2000, False, 0.001),
])
def test_fuzz_parity(bv, seed, n_lines, n_samples, fmt_extra, weird):
    vcf = vcfgen.gen_vcf(seed, n_lines, n_samples, fmt_extra, weird)
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"keepId": True, "keepInfo": True, "keepPos": True, "exclude": "q10"})


@pytest.mark.parametrize("name_len,delim", [(3, ";"), (15, ";"), (16, "|"), (30, ";;"), (33, ","), (70, ";")])
def test_sample_name_widths(bv, name_len, delim):
    """the list writer moves names with fixed 16 / 32 byte copies when name + delimiter fit: each width class,
    names of mixed length, one- and two-byte delimiters; dense maps and sparse lists both"""
    import random
    ns = 600
    names = [("N%d_" % i).ljust(max(len("N%d_" % i), name_len - (i % 4)), "x") for i in range(ns)]
    rng = random.Random(name_len)
    rows = [vcfgen.header(ns, names=names)]
    pos = 5000
    for k in range(120):
        pos += rng.randint(1, 50)
        # common alleles (dense map) alternate with singletons/doubletons (sparse list); a few missing genotypes
        p_alt = 0.4 if k % 2 else 1.5 / ns
        gts = []
        for _ in range(ns):
            r = rng.random()
            gts.append(".|." if r < 0.003 else ("1|1" if r < 0.003 + p_alt / 3 else ("0|1" if r < 0.003 + p_alt else "0|0")))
        rows.append("\t".join(["chr1", str(pos), "rs%d" % k, "A", "G", ".", "PASS", "DP=1", "GT"] + gts) + "\n")
    vcf = "".join(rows).encode()
    both(bv, vcf, {"fieldDelimiter": delim})
    both(bv, vcf, {"fieldDelimiter": delim, "keepId": True}, max_batch_bytes=1 << 16)


@pytest.mark.parametrize("n_samples", [32768, 40001, 16384 * 3 + 255])
def test_many_samples_split_scan(bv, n_samples, monkeypatch):
    """cohorts of >= 32 768 samples: the regular scan of one line is split into 16 384-sample windows over several waves
    (k_gt_wide, chosen by the library itself here).  Non-reference genotypes on both sides of every window boundary, a
    line with one irregular field in the middle window (the whole task falls back to the general scan), multiallelic
    lines, missing genotypes, mixed separators between windows"""
    import random
    monkeypatch.delenv("BVCF_WIDE", raising=False)
    rng = random.Random(n_samples)
    ns = n_samples
    rows = [vcfgen.header(ns)]
    edges = [e for w in range(1, ns // 16384 + 1) for e in (w * 16384 - 1, w * 16384) if e < ns]
    for k in range(14):
        alt = "G,T" if k % 4 == 1 else "G"
        sep = "|" if k % 3 else "/"
        gts = ["0%s0" % sep] * ns
        for e in edges + [0, ns - 1] + [rng.randrange(ns) for _ in range(40)]:
            gts[e] = rng.choice(["0%s1", "1%s1", ".%s.", "1%s0", "0%s2" if "," in alt else "0%s1"]) % sep
        if k == 5:
            gts[ns // 2] = "0|1:9"      # not a 4-byte field: the line's length is no longer 4*ns
        if k == 6:
            a, b = ns // 2, ns // 2 + 1  # same length, but two fields are not regular ("0|10", "|1" ...)
            gts[a], gts[b] = "0|10", "|1"
        if k == 7:
            for i in range(20000, min(ns, 36000)):
                gts[i] = gts[i].replace("|", "/")   # another separator from the second window on
        rows.append("\t".join(["chr3", str(1000 + 7 * k), ".", "A", alt, ".", "PASS", ".", "GT"] + gts) + "\n")
    vcf = "".join(rows).encode()
    both(bv, vcf)
    both(bv, vcf, {"keepId": True, "keepInfo": True}, max_batch_bytes=1 << 20)
    # left to itself the library takes the census path for such files (the host driver asks for "choose")
    monkeypatch.delenv("BVCF_PATH", raising=False)
    ctx = bv.Ctx(9 + ns)
    assert ctx.path() == 1
    ctx.close()
    both(bv, vcf, {"allow": ""})


def test_many_samples_split_general_scan(bv, monkeypatch):
    """>= 32 768 samples with sub-fields beyond GT: the general scan of one line runs as 64 KiB shares over several
    waves, each starting from the TAB count of the shares before it"""
    import random
    monkeypatch.delenv("BVCF_WIDE", raising=False)
    monkeypatch.delenv("BVCF_WIDE_WIN", raising=False)
    ns = 33000
    rng = random.Random(5)
    rows = [vcfgen.header(ns)]
    for k in range(10):
        rows.append(vcfgen.gen_line(rng, ns, 2000 + 13 * k, fmt_extra=True, weird=0.02 if k % 2 else 0.0,
                                    filters=("PASS", ".")))
    # a line whose fields are long enough for a field to span a whole share, and one with empty trailing fields
    gts = ["0/1:" + "7" * rng.randint(1, 90000) if i == 17 else rng.choice(["0/0:1", "0/1:22", "./.:0", "1/1:3"]) for i in range(ns)]
    rows.append("\t".join(["chr9", "777", ".", "C", "A", ".", "PASS", ".", "GT:X"] + gts) + "\n")
    gts = [rng.choice(["0|0:5", "1|0:6", "1:1", "."]) for _ in range(ns - 2)] + ["", ""]
    rows.append("\t".join(["chr9", "778", ".", "C", "A,T", ".", "PASS", ".", "GT:X"] + gts) + "\n")
    vcf = "".join(rows).encode()
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"allow": "", "keepInfo": True}, max_batch_bytes=4 << 20)


@pytest.mark.parametrize("n_samples", [300, 512, 1030])
def test_chunk_boundary_alignment(bv, n_samples):
    """a single non-reference genotype next to every 256-sample chunk boundary, at each of the four byte
    alignments of the sample region (the ID column's length shifts it): the realigned window of lane 63
    borrows its last dword from the next chunk"""
    spots = sorted({k for b in range(256, n_samples + 1, 256) for k in (b - 4, b - 2, b - 1, b, b + 1, b + 3)
                    if 0 <= k < n_samples} | {0, 1, n_samples - 2, n_samples - 1})
    lines, pos = [], 1000
    for shift in range(4):
        for k in spots:
            for g in ("0|1", "1|1", ".|.", "1|.", "2|1"):
                gts = ["0|0"] * n_samples
                gts[k] = g
                pos += 7
                lines.append("\t".join(["1", str(pos), "r" + "x" * shift, "A", "C,G", "50", "PASS", "AC=1", "GT"] + gts))
    vcf = (vcfgen.header(n_samples) + "\n".join(lines) + "\n").encode()
    both(bv, vcf)


@pytest.mark.parametrize("n_samples", [260, 2504])
def test_sparse_class_list_boundary(bv, n_samples):
    """alleles whose class map has 0..20 non-zero bytes: the streaming path keeps up to 15 as a list
    (BVCF_ALLELE_CMAP_SPARSE) and turns the line into a map at the 16th, possibly in the middle of a chunk"""
    import random
    rng = random.Random(n_samples)
    lines, pos = [], 5000
    for k in list(range(0, 21)) + [40, 64, 65]:
        for rep in range(3):
            gts = ["0|0"] * n_samples
            n_bytes = (n_samples + 3) // 4
            picks = rng.sample(range(n_bytes), min(k, n_bytes))
            for b in picks:
                for q in rng.sample(range(4), rng.randint(1, 4)):
                    if b * 4 + q < n_samples:
                        gts[b * 4 + q] = rng.choice(["0|1", "1|0", "1|1", ".|.", "1|.", "2|1"])
            if rep == 2 and picks:  # everything in the last chunk(s)
                gts = ["0|0"] * n_samples
                for j in range(min(k, n_samples)):
                    gts[n_samples - 1 - 4 * j if n_samples - 1 - 4 * j >= 0 else 0] = "0|1"
            pos += 3
            lines.append("\t".join(["7", str(pos), ".", "G", "A,T", "9", "PASS", ".", "GT"] + gts))
    vcf = (vcfgen.header(n_samples) + "\n".join(lines) + "\n").encode()
    both(bv, vcf)


def test_crlf_and_lone_cr(bv):
    v = vcfgen.gen_vcf(21, 120, 9, weird=0.05, eol="\r\n")
    both(bv, v, {"allow": ""})


def test_unterminated_tail_and_empty_body(bv):
    h = vcfgen.header(3).encode()
    both(bv, h)
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\n1\t6\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0")
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t")          # empty last sample field
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t\n")
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t\t\t\n", {"allow": ""})
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\n", {"allow": ""})      # region missing entirely
    both(bv, h + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t\n", {"allow": ""})


def test_long_info_pushes_samples_past_first_window(bv):
    h = vcfgen.header(300).encode()
    gts = b"\t".join([b"0|1", b"1|1", b"0|0"] * 100)
    for n in (10, 200, 240, 1000, 1024, 5000):
        line = b"7\t100\t.\tA\tG\t.\tPASS\t" + b"X" * n + b"\tGT\t" + gts + b"\n"
        both(bv, h + line * 3)


def test_result_capacity_growth(bv):
    """a tiny reservation must surface as BVCF_E_CAPACITY, then succeed after bvcf_reserve"""
    vcf = vcfgen.gen_vcf(31, 500, 40, weird=0.0)
    body = vcf[vcf.index(b"#CHROM"):]
    body = body[body.index(b"\n") + 1:]
    ctx = bv.Ctx(49, allow="", max_lines=16, max_alleles=16, cmap_bytes=64)
    ctx.submit(body)
    with pytest.raises(bv.BvcfError) as ei:
        ctx.collect()
    assert ei.value.rc == bv.E_CAPACITY
    ctx.reserve(1000, 4000, 1 << 20)
    b = ctx.process(body)
    assert len(b.lines) == body.count(b"\n")
    ctx.close()
    both(bv, vcf, {"allow": ""})


def test_device_resident_submit_matches_host_submit(bv):
    import torch
    vcf = vcfgen.gen_vcf(41, 200, 100, weird=0.01)
    body = vcf[vcf.index(b"#CHROM"):]
    body = body[body.index(b"\n") + 1:]
    ctx = bv.Ctx(109, allow="")
    a = ctx.process(body)
    t = torch.zeros(len(body) + bv.DEVICE_PAD, dtype=torch.uint8, device="cuda")
    t[: len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    ctx.submit_device(t.data_ptr(), len(body))
    b = ctx.collect()
    ctx.close()
    for f in ("off", "len", "n_rec", "status", "n_fields"):
        assert (a.lines[f] == b.lines[f]).all()
    assert len(a.alleles) == len(b.alleles)
    for i in range(len(a.lines)):
        ra, rb = a.records(i), b.records(i)
        for f in ("pos", "alt_idx", "ac", "an", "n_het", "n_hom", "n_miss", "ref", "kind", "trtv"):
            assert (ra[f] == rb[f]).all()


def _device_dosage_rows(bv, vcf, allow=""):
    """the int8 rows bvcf_collect returns for the output alleles of `vcf`, in input order"""
    hdr_at = vcf.index(b"#CHROM")
    hdr_end = vcf.index(b"\n", hdr_at)
    n_header = vcf[hdr_at:hdr_end].count(b"\t") + 1
    body = vcf[hdr_end + 1:]
    ctx = bv.Ctx(n_header, allow=allow, want_dosage=True)
    b = ctx.process(body)
    ctx.close()
    ns = n_header - 9
    rows = []
    for i in range(len(b.lines)):
        if b.lines[i]["status"] != 0:
            continue
        for k in b.record_slots(i):
            if b.alleles[k]["ac"] == 0:
                continue  # main.go:558-560
            rows.append([int(x) for x in b.dosage[k][:ns]])
    return rows


@pytest.mark.parametrize("seed,n_lines,n_samples,fmt_extra,weird", [
    (31, 200, 3, False, 0.1), (32, 150, 70, False, 0.05), (33, 100, 300, True, 0.03), (34, 40, 2504, False, 0.002),
    (35, 60, 257, True, 0.2), (36, 100, 1, False, 0.3),
    # all-regular files: every row comes from the class map (sparse list or 2-bit map), not from a second scan
    (37, 60, 2504, False, 0.0), (38, 80, 300, False, 0.0),
])
def test_dosage_rows_match_oracle(bv, seed, n_lines, n_samples, fmt_extra, weird):
    """bvcf_params.want_dosage: altCount per sample, -1 when missing (main.go:1069-1178), any ploidy"""
    vcf = vcfgen.gen_vcf(seed, n_lines, n_samples, fmt_extra, weird)
    want = [d for _, d in orc.run_dosage(vcf, {"allow": ""})]
    got = _device_dosage_rows(bv, vcf)
    assert len(got) == len(want)
    for r, (g, w) in enumerate(zip(got, want)):
        assert g == w, "row %d: first difference at sample %d" % (r, next(i for i in range(len(w)) if g[i] != w[i]))


def test_dosage_reference_known_answer(bv):
    """TestGenotypeMatrix, main_test.go:2911-2977"""
    hdr = "##fileformat=VCFv4.x\n" + "\t".join(H8 + ["FORMAT", "S1", "S2", "S3"]) + "\n"
    rows = [["1", "1000", "rs1", "A", "T", ".", "PASS", "DP=100", "GT", "1|1", "0|1", "0|0"],
            ["2", "200", "rs2", "C", "G", ".", "PASS", "DP=100", "GT", "0|1", "0|0", "1|1"],
            ["22", "300", "rs2", "G", "T", ".", "PASS", "DP=100", "GT", "0|.", "0|.", "1|1"]]
    vcf = (hdr + "".join("\t".join(r) + "\n" for r in rows)).encode()
    assert _device_dosage_rows(bv, vcf, allow="PASS,.") == [[2, 1, 0], [1, 0, 2], [-1, -1, 2]]
    assert orc.run_dosage(vcf) == [("chr1:1000:A:T", [2, 1, 0]), ("chr2:200:C:G", [1, 0, 2]), ("chr22:300:G:T", [-1, -1, 2])]


def _read_matrix(path):
    import pyarrow.ipc as ipc
    t = ipc.open_file(str(path)).read_all()
    t.validate(full=True)
    cols = [t.column(i).to_pylist() for i in range(1, t.num_columns)]
    return t.schema.names, [(locus, [c[r] for c in cols]) for r, locus in enumerate(t.column(0).to_pylist())]


@pytest.mark.parametrize("seed,n_lines,n_samples,fmt_extra", [(51, 300, 12, False), (52, 120, 2504, False),
                                                               (53, 150, 40, True)])
def test_dosage_output_file(bv, tmp_path, seed, n_lines, n_samples, fmt_extra):
    """--dosageOutput end to end (main.go:306-342,576-584): the Arrow file read back with pyarrow holds the rows
    the oracle computes, in input order; the TSV is unchanged; --noOut drops the TSV only"""
    pytest.importorskip("pyarrow")
    vcf = vcfgen.gen_vcf(seed, n_lines, n_samples, fmt_extra, 0.03)
    want = orc.run_dosage(vcf, {"allow": ""})
    p = tmp_path / "dosage.feather"
    out, _ = both(bv, vcf, {"allow": "", "dosageOutput": p})
    names, rows = _read_matrix(p)
    assert names == ["locus"] + ["S%05d" % i for i in range(n_samples)]
    assert rows == want
    # small batches cut the same file into many record batches
    rc, out2, _, _ = bv.run_buffer(vcf, {"allow": "", "dosageOutput": p, "noOut": True}, max_batch_bytes=max(1 << 16, 8 * n_samples * 8))
    assert rc == 0 and out2 == b""
    assert _read_matrix(p)[1] == want


def test_dosage_output_no_samples_writes_empty_file(bv, tmp_path):
    """main.go:308-318"""
    vcf = vcfgen.gen_vcf(54, 50, 0)
    p = tmp_path / "empty.feather"
    both(bv, vcf, {"allow": "", "dosageOutput": p})
    assert p.exists() and p.stat().st_size == 0


def test_cli_dosage_and_no_out(bv, tmp_path):
    """TestGenotypeMatrix / TestNoOut through the binary (main_test.go:2911-3029), plus the flag checks of main.go:160-166"""
    pytest.importorskip("pyarrow")
    hdr = "##fileformat=VCFv4.x\n" + "\t".join(H8 + ["FORMAT", "S1", "S2", "S3"]) + "\n"
    rows = [["1", "1000", "rs1", "A", "T", ".", "PASS", "DP=100", "GT", "1|1", "0|1", "0|0"],
            ["2", "200", "rs2", "C", "G", ".", "PASS", "DP=100", "GT", "0|1", "0|0", "1|1"],
            ["22", "300", "rs2", "G", "T", ".", "PASS", "DP=100", "GT", "0|.", "0|.", "1|1"]]
    vcf = (hdr + "".join("\t".join(r) + "\n" for r in rows)).encode()
    want = [("chr1:1000:A:T", [2, 1, 0]), ("chr2:200:C:G", [1, 0, 2]), ("chr22:300:G:T", [-1, -1, 2])]
    p = tmp_path / "m.feather"
    r = _run_cli(["--dosageOutput", str(p)], vcf)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count(b"\n") == 4  # header + 3 rows
    assert _read_matrix(p) == (["locus", "S1", "S2", "S3"], want)
    p2 = tmp_path / "m2.feather"
    r = _run_cli(["--dosageOutput", str(p2), "--noOut"], vcf)
    assert r.returncode == 0 and r.stdout == b""
    assert _read_matrix(p2)[1] == want
    assert _run_cli(["--noOut"], vcf).returncode == 1
    assert _run_cli(["--noOut", "--out", str(tmp_path / "x.tsv"), "--dosageOutput", str(p2)], vcf).returncode == 1


def _run_cli(args, data):
    import subprocess
    exe = os.path.join(ROOT, "bystro-vcf_amd", "bystro-vcf")
    return subprocess.run([exe] + args, input=data, capture_output=True, timeout=300)


def test_cli_matches_oracle(bv, tmp_path):
    """the process surface: stdin -> stdout with header line, log lines on stderr, exit status"""
    vcf = vcfgen.gen_vcf(51, 3000, 120, weird=0.02)
    cfg = {"keepId": True, "keepInfo": True, "emptyField": "NA"}
    rc_o, out_o, log_o, _ = orc.run(vcf, cfg)
    p = _run_cli(["--keepId", "--keepInfo", "--emptyField", "NA", "--batchMB", "1"], vcf)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == (bv.string_header(cfg) + "\n").encode() + out_o
    assert p.stderr.decode() == log_o
    # --in / --out / -flag=value forms, default flags
    src, dst = tmp_path / "in.vcf", tmp_path / "out.tsv"
    src.write_bytes(vcf)
    p = _run_cli(["-in=%s" % src, "--out", str(dst)], b"")
    assert p.returncode == 0
    assert dst.read_bytes() == (bv.string_header() + "\n").encode() + orc.run(vcf)[1]
    # fatal paths: message on stderr, exit status 1 (log.Fatal)
    p = _run_cli([], b"not a vcf\n")
    assert p.returncode == 1 and b"Not a VCF file" in p.stderr
    p = _run_cli([], b"##fileformat=VCFv4.2\n")
    assert p.returncode == 1 and b"No header found" in p.stderr
    p = _run_cli([], b"")
    assert p.returncode == 1


def test_cli_many_batches_with_capacity_growth(bv):
    """the pipelined driver over a dozen one-megabyte batches whose line count exceeds the first reservation (short junk
    lines between the records): the formatter runs behind the device, the reservation grows mid-stream, and the
    rows still come out complete and in input order"""
    import random
    rng = random.Random(77)
    ns = 300
    rows = [vcfgen.header(ns)]
    pos = 1000
    for k in range(24000):
        pos += rng.randint(1, 40)
        if k % 3 == 0:
            gts = ["0|1" if rng.random() < 0.02 else ("1|1" if rng.random() < 0.01 else "0|0") for _ in range(ns)]
            rows.append("\t".join(["chr2", str(pos), ".", "C", "T", ".", "PASS", ".", "GT"] + gts) + "\n")
        else:
            # far more lines per megabyte than the reservation assumes (it divides by the header width)
            rows.extend(["chr2\t%d\t.\tA\tG\n" % pos] * (200 if k % 50 == 1 else 2))
    vcf = "".join(rows).encode()
    rc_o, out_o, log_o, n_o = orc.run(vcf)
    assert rc_o == 0 and out_o.count(b"\n") > 1000
    p = _run_cli(["--batchMB", "1"], vcf)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == (bv.string_header() + "\n").encode() + out_o
    assert p.stderr.decode() == log_o


def test_cli_large_stream(bv, golden_1kg):
    """the 200 MB 1KG regression input through the pipelined driver with small blocks"""
    vcf, want_sorted, hdr = golden_1kg
    p = _run_cli(["--batchMB", "16"], vcf)
    assert p.returncode == 0
    rows = p.stdout.split(b"\n")
    assert rows[0] == hdr and rows[-1] == b""
    assert sorted(rows[1:-1]) == want_sorted
    assert rows[1:-1] == orc.run(vcf)[1].split(b"\n")[:-1]


@pytest.mark.parametrize("tile_kb", [4, 8])
def test_streaming_tile_boundaries(bv, monkeypatch, tile_kb, bvcf_path, golden_1kg):
    """tiny tiles: almost every line straddles a tile (and wave-run) boundary, many tiles hold no
    line start at all, and 10 KB lines span several tiles"""
    if not bvcf_path.startswith("streaming"):
        pytest.skip("tile logic only exists on the streaming path")
    monkeypatch.setenv("BVCF_TILE_KB", str(tile_kb))
    for seed, n_lines, n_samples, fmt_extra, weird in [(61, 500, 3, False, 0.05), (62, 400, 40, True, 0.05),
                                                       (63, 300, 300, False, 0.01), (64, 120, 2504, False, 0.002),
                                                       (65, 300, 1100, False, 0.0)]:
        vcf = vcfgen.gen_vcf(seed, n_lines, n_samples, fmt_extra, weird)
        both(bv, vcf, {"allow": ""})
        both(bv, vcf, {"allow": ""}, max_batch_bytes=1 << 20)
    vcf = golden_1kg[0]
    both(bv, vcf[: vcf.index(b"\n", 30_000_000) + 1])
    # line starts exactly on tile boundaries: rows padded to a power of two
    h = vcfgen.header(2).encode()
    row = b"1\t5\t.\tA\tG\t.\tPASS\t" + b"X" * (tile_kb * 1024 - 29) + b"\tGT\t0|1\t1|1\n"
    assert len(row) == tile_kb * 1024
    both(bv, h + row * 9, {"allow": ""})
    both(bv, h + (row[:-1] + b"\t\n") * 3 + row * 2, {"allow": ""})


def test_sample_list(bv, tmp_path, known_answers):
    """--sample: main_test.go:171-270 (names one per line; nothing when the header has no samples)"""
    path = tmp_path / "samples.txt"
    vcf = vcfgen.gen_vcf(71, 20, 4, weird=0.0)
    rc, _, log, _ = bv.run_buffer(vcf, {"sample": str(path)})
    assert rc == 0, log
    assert path.read_text().split("\n") == ["S00000", "S00001", "S00002", "S00003", ""]
    path2 = tmp_path / "samples2.txt"
    rc, _, _, _ = bv.run_buffer(vcfgen.gen_vcf(72, 5, 0), {"sample": str(path2)})
    assert rc == 0 and path2.read_text() == ""
    p = _run_cli(["--sample", str(tmp_path / "s3.txt")], vcf)
    assert p.returncode == 0 and (tmp_path / "s3.txt").read_text().count("\n") == 4


def test_compressed_input(bv, golden_1kg, tmp_path):
    """the CLI reads gzip and BGZF itself (the reference needs `pigz -d -c` in front)"""
    import gzip
    import bgzf
    vcf = vcfgen.gen_vcf(81, 4000, 150, weird=0.02)
    want = _run_cli(["--keepId"], vcf)
    assert want.returncode == 0
    for name, data in (("gz", gzip.compress(vcf, 1)), ("bgzf", bgzf.bgzf_compress(vcf)),
                       ("bgzf-small", bgzf.bgzf_compress(vcf, block=777, eof_marker=False)),
                       ("gz-members", gzip.compress(vcf[:100000]) + gzip.compress(vcf[100000:]))):
        p = _run_cli(["--keepId", "--batchMB", "2"], data)
        assert p.returncode == 0, (name, p.stderr[-300:])
        assert p.stdout == want.stdout and p.stderr == want.stderr, name
        f = tmp_path / ("in." + name)
        f.write_bytes(data)
        p = _run_cli(["--keepId", "--in", str(f)], b"")
        assert p.returncode == 0 and p.stdout == want.stdout, name
    # the reference's own regression input, exactly as shipped (.vcf.gz, single-stream gzip)
    raw = open(os.path.join(ROOT, "tests", "golden", "1kg_chr1_20klines.vcf.gz"), "rb").read()
    p = _run_cli([], raw)
    assert p.returncode == 0
    rows = p.stdout.split(b"\n")
    assert sorted(rows[1:-1]) == golden_1kg[1]
    # damaged input: message on stderr, exit status 1
    p = _run_cli([], bgzf.bgzf_compress(vcf)[:-40])
    assert p.returncode == 1 and b"bgzf" in p.stderr
    bad = bytearray(gzip.compress(vcf, 1))
    bad[len(bad) // 2] ^= 0xFF
    p = _run_cli([], bytes(bad))
    assert p.returncode == 1 and b"gzip" in p.stderr


def _garbage_vcf(seed, n_lines, n_samples):
    """structurally hostile input: random bytes biased towards the characters the parsers care about"""
    import random
    rng = random.Random(seed)
    alphabet = b"\t\t\t\t\t\t,,..||//::00112ACGTN<>*\r\x00\xff-+ 9PASSq;="
    out = [vcfgen.header(n_samples).encode()]
    for i in range(n_lines):
        r = rng.random()
        if r < 0.5:  # mutate a valid line
            line = bytearray(vcfgen.gen_line(rng, n_samples, 1000 + i, fmt_extra=rng.random() < 0.3, weird=0.2).encode())
            for _ in range(rng.randint(0, 6)):
                k = rng.randrange(len(line) - 1)
                line[k] = rng.choice(alphabet)
            line = bytes(line).replace(b"\n", b" ") + b"\n"
        elif r < 0.8:  # random fields, right count
            nf = 9 + n_samples if n_samples else 8
            line = b"\t".join(bytes(rng.choice(alphabet.replace(b"\t", b"")) for _ in range(rng.randint(0, 6)))
                              for _ in range(nf)) + b"\n"
        else:  # pure junk of random length
            line = bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 400))).replace(b"\n", b"") + b"\n"
        out.append(line)
    return b"".join(out)


@pytest.mark.parametrize("seed,n_lines,n_samples", [(201, 600, 0), (202, 600, 1), (203, 500, 5), (204, 300, 70),
                                                     (205, 200, 300)])
def test_garbage_parity(bv, seed, n_lines, n_samples):
    vcf = _garbage_vcf(seed, n_lines, n_samples)
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"allow": "PASS,q", "exclude": ".", "keepInfo": True, "keepId": True, "keepPos": True})
