"""k_stream_gen (bvcf_streamgen.hip.h), the streaming kernel for files whose sample fields carry more than GT
(/root/reference main.go:1042-1194, the general branch of makeHetHomozygotes): crafted inputs for what its packed-flag
fast path, its byte-parallel medium path and its exact handler each have to get right, compared with the oracle through
the C-ABI.  Every case also runs on k_stream (+ k_gt for the lines it defers) and on the census path."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["streaming-general", "streaming", "census"])
def bvcf_path(request, monkeypatch):
    monkeypatch.setenv("BVCF_PATH", "2" if request.param.startswith("streaming") else "1")
    monkeypatch.setenv("BVCF_GEN_STREAM", "1" if request.param == "streaming-general" else "0")
    return request.param


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def both(bv, vcf, cfg=None, **kw):
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
    return out_g


def _line(pos, fields, alt="G", info="AC=1", fmt="GT:DP:GQ", chrom="1", filt="PASS"):
    return "\t".join([chrom, str(pos), ".", "A", alt, "50", filt, info, fmt] + fields) + "\n"


def _ref_fields(rng, n, sep="/"):
    return ["0%s0:%d:%d" % (sep, rng.randint(0, 99), rng.randint(0, 99)) for _ in range(n)]


@pytest.mark.parametrize("seed,n_lines,n_samples,weird,eol", [
    (101, 60, 2504, 0.001, "\n"), (102, 60, 700, 0.02, "\n"), (103, 200, 64, 0.2, "\n"), (104, 10, 16384, 0.0005, "\n"),
    (105, 10, 16385, 0.0005, "\n"),  # one sample past the LDS stage: left to k_stream + k_gt
    (106, 80, 300, 0.01, "\r\n"), (107, 50, 1000, 0.0, "\n"), (108, 300, 256, 0.05, "\n"),
])
def test_fuzz_fields_beyond_gt(bv, seed, n_lines, n_samples, weird, eol):
    vcf = vcfgen.gen_vcf(seed, n_lines, n_samples, True, weird, eol=eol)
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"keepId": True, "keepInfo": True, "exclude": "q10"})


@pytest.mark.parametrize("ns", [300, 1100])
def test_reference_word_variants_and_carrier_density(bv, ns):
    """lines whose reference genotype is written "0/0:", "0|0:", "0/0<TAB>" (fields without sub-fields) or a mix; no
    carriers, one, a list's worth (15 map bytes), one more, and most samples"""
    rng = random.Random(ns)
    rows = []
    pos = 100
    for sep in ["/", "|", "mix"]:
        for n_car in [0, 1, 2, 15, 16, 40, ns // 2, ns]:
            f = []
            for i in range(ns):
                s = sep if sep != "mix" else rng.choice("/|")
                f.append("0%s0:%d:%d" % (s, rng.randint(0, 99), rng.randint(0, 99)))
            for i in rng.sample(range(ns), n_car):
                s = sep if sep != "mix" else rng.choice("/|")
                f[i] = "%s%s%s:%d:%d" % (rng.choice("01."), s, rng.choice("01."), rng.randint(0, 9), rng.randint(0, 99))
            pos += 7
            rows.append(_line(pos, f))
    # sub-fields dropped in some samples (VCF allows trailing ones to be left out): "0/0<TAB>", and at the line's end
    for k in range(6):
        f = _ref_fields(rng, ns)
        for i in rng.sample(range(ns), ns // 3):
            f[i] = rng.choice(["0/0", "0/1", "1/1", "./."])
        if k % 2:
            f[-1] = rng.choice(["0/0", "0/1", "1|1", "./."])
        pos += 7
        rows.append(_line(pos, f))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})


@pytest.mark.parametrize("ns", [260, 900, 2600])  # class-map slots with room for 1, 3 and 8 lists
def test_multiallelic_lines_carry_every_alt_index_in_lists(bv, ns):
    """1-9 ALTs (and 11: two-digit indices are outside the fast gate), carriers of any index, few (class lists for every
    index, k_head takes the counts from them) and many (a dense map for ALT #1, k_gt for the rest), missing samples
    (listed for every index), several carriers inside one map byte"""
    rng = random.Random(77 + ns)
    alts_all = ["G", "T", "C", "GA", "AT", "ACC", "TT", "GG", "CC", "AG", "TG"]
    rows = []
    pos = 3000
    for n_alt in [1, 2, 3, 4, 5, 8, 9, 11]:
        for n_car in [0, 1, 3, 10, 15, 16, 30, ns // 3]:
            for sep in "/|":
                f = ["0%s0:%d:%d" % (sep, rng.randint(0, 99), rng.randint(0, 99)) for _ in range(ns)]
                who = rng.sample(range(ns), n_car)
                if n_car >= 3:  # neighbours: the same map byte
                    who[1] = (who[0] // 4) * 4 + (who[0] + 1) % 4
                    who[2] = (who[0] // 4) * 4 + (who[0] + 2) % 4
                for i in who:
                    a = rng.choice([str(rng.randint(0, n_alt)), "."]) if rng.random() < 0.9 else "."
                    b = str(rng.randint(0, n_alt))
                    f[i] = "%s%s%s:%d:%d" % (a, sep, b, rng.randint(0, 9), rng.randint(0, 99))
                pos += 5
                rows.append(_line(pos, f, alt=",".join(alts_all[:n_alt])))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})


def test_fields_the_fast_gate_does_not_take(bv):
    """haploid, polyploid, multi-digit and empty fields, further ALT indices: such lines are listed as deferred by
    k_stream_gen (or get a dense map and k_gt tasks for the further indices) -- results as the reference's general branch"""
    ns = 400
    rng = random.Random(5)
    odd = ["1", "0", ".", "", "0/1/1", "1|1|1", "10/1", "1/10", "0|2", "2/2", "3|1", "./1", "1/.", "0/0/0", "00/1", "+1/1",
           "1/ 1", "a/b", "0-0", "0/0;", ":", "0/:", "0"]
    rows = []
    pos = 500
    for o in odd:
        for where in [0, 1, 63, 64, 65, ns - 2, ns - 1]:
            f = _ref_fields(rng, ns)
            f[where] = o + (":5:6" if rng.random() < 0.5 and o != ":" else "")
            pos += 3
            rows.append(_line(pos, f, alt="G,T,C,GA,AT,ACC,TT,GG,CC,AG,TG"))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})


def test_field_counts_terminators_and_heads(bv):
    """too few / too many sample fields, a TAB at the end of the line, lines of fewer than ten fields and comments in
    between, heads longer than a chunk (a 5 KB INFO), bytes >= 0x80 in the head and in a sample field, CRLF"""
    ns = 320
    rng = random.Random(9)
    for eol in ["\n", "\r\n"]:
        rows = []
        pos = 900
        for k in range(40):
            f = _ref_fields(rng, ns)
            for i in rng.sample(range(ns), rng.choice([0, 1, 3, 30])):
                f[i] = "0/1:3:4"
            info = "AC=1"
            kind = k % 10
            if kind == 1:
                f = f[:-1]
            elif kind == 2:
                f = f + ["0/0:1:1"]
            elif kind == 3:
                f[-1] = ""
            elif kind == 4:
                info = "CSQ=" + vcfgen.rand_bases(rng, 5000, "ACGT|,")
            elif kind == 5:
                info = "NOTE=caféü"
            elif kind == 6:
                f[rng.randrange(ns)] = "0/1:é:4"
            elif kind == 7:
                rows.append("1\t5\tshort\n")
                rows.append("#a comment\n")
            elif kind == 8:
                f = f[:ns // 2]
            pos += 11
            rows.append(_line(pos, f, info=info))
        text = vcfgen.header(ns) + "".join(rows)
        if eol != "\n":
            text = text.replace("\n", eol)
        both(bv, text.encode("utf-8"), {"allow": "", "keepInfo": True})


def test_every_alignment_of_a_carrier(bv):
    """one carrier per line, moved sample by sample through two chunks' worth of fields of uneven length: every byte
    offset of a field start inside a dword, a lane and across the chunk edge (the word then comes from the next chunk)"""
    ns = 260
    rng = random.Random(13)
    widths = [rng.choice(["%d", "%d%d", "%d%d%d"]) for _ in range(ns)]
    rows = []
    for who in range(ns):
        f = ["0/0:" + (w.replace("%d", "7")) + ":9" for w in widths]
        f[who] = "1|0:" + f[who][4:] if who % 2 else "1/1:" + f[who][4:]
        rows.append(_line(1000 + who, f))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})


def test_regular_file_through_the_general_kernel_and_back(bv, monkeypatch):
    """left alone (no BVCF_GEN_STREAM) a ctx picks the kernel for the next batch by the shape of the last one: batches of
    GT-only lines, then GT:DP lines, then GT-only again, through one ctx -- every batch equal to the census path's"""
    monkeypatch.delenv("BVCF_GEN_STREAM", raising=False)
    ns = 300
    texts = [vcfgen.gen_vcf(200 + i, 60, ns, fmt_extra=(i // 2) % 2 == 1, weird=0.0) for i in range(8)]
    bodies = [t[t.index(b"#CHROM"):] for t in texts]
    bodies = [b[b.index(b"\n") + 1:] for b in bodies]

    def run(path):
        ctx = bv.Ctx(9 + ns, path=path)
        got = []
        for b in bodies:
            r = ctx.process(b)
            cols = ["pos", "alt_idx", "ac", "an", "n_het", "n_hom", "n_miss", "kind", "site_type", "trtv"]
            recs = []
            for i, L in enumerate(r.lines):
                if int(L["status"]) != 0:
                    continue
                for a in r.records(i):
                    recs.append((i, tuple(int(a[c]) for c in cols), tuple(int(x) for x in r.classes(a))))
            got.append((recs, [(int(L["off"]), int(L["len"]), int(L["status"]), int(L["n_rec"])) for L in r.lines]))
        ctx.close()
        return got

    monkeypatch.setenv("BVCF_PATH", "2")
    a = run(2)
    monkeypatch.setenv("BVCF_PATH", "1")
    b = run(1)
    assert len(a) == len(b) == len(bodies)
    for x, y in zip(a, b):
        assert x == y


def _dosage_rows_match_oracle(bv, vcf, ns):
    """bvcf_params.want_dosage on the streaming path: the int8 rows of the output alleles equal the oracle's (any ploidy)"""
    hdr_at = vcf.index(b"#CHROM")
    body = vcf[vcf.index(b"\n", hdr_at) + 1:]
    ctx = bv.Ctx(9 + ns, allow="", want_dosage=True, max_batch_bytes=len(body))
    b = ctx.process(body)
    ctx.close()
    got = []
    for i in range(len(b.lines)):
        if b.lines[i]["status"] != 0:
            continue
        for k in b.record_slots(i):
            if b.alleles[k]["ac"] != 0:
                got.append([int(x) for x in b.dosage[k][:ns]])
    want = [d for _, d in orc.run_dosage(vcf, {"allow": ""})]
    assert got == want


def _hap_rows(rng, ns, n_lines, p_hap, alts=("G",), dots=True):
    """chrX-style lines: a share of the samples (the same ones on every line: the males) has haploid calls"""
    males = [rng.random() < p_hap for _ in range(ns)]
    rows, pos = [], 100
    for k in range(n_lines):
        n_alt = len(alts[k % len(alts)].split(","))
        f = []
        for i in range(ns):
            r = rng.random()
            if males[i]:
                g = "0" if r < 0.9 else rng.choice([str(x) for x in range(1, n_alt + 1)] + (["."] if dots else []))
            else:
                g = "0/0" if r < 0.9 else "%s/%s" % (rng.choice("01." if dots else "01"), rng.choice([str(x) for x in range(0, n_alt + 1)]))
            f.append("%s:%d:%d" % (g, rng.randint(0, 99), rng.randint(0, 99)))
        pos += 11
        rows.append(_line(pos, f, alt=alts[k % len(alts)], chrom="X"))
    return rows


@pytest.mark.parametrize("ns,p_hap", [(300, 0.05), (2504, 0.5), (700, 1.0)])
def test_haploid_calls(bv, ns, p_hap):
    """one allele character per call (chrX males, chrY, chrM): the general branch's single token -- hom when it is the
    allele (alt == gt), one allele towards an (main.go:1130-1190).  Biallelic and multiallelic lines, '.' calls, with and
    without a dosage matrix asked for (the line is then marked "not regular": k_dosage scans it itself, a haploid carrier's
    dosage is 1 where its class is 2)"""
    rng = random.Random(ns)
    rows = _hap_rows(rng, ns, 40, p_hap, alts=("G", "G,T", "G,T,C"))
    # haploid calls at every byte alignment, as the last field, as the only odd field
    for sh in range(17):
        f = _ref_fields(rng, ns)
        f[3] = "1:%s:9" % ("7" * (1 + sh))
        f[ns - 1] = "1:5:5" if sh % 2 else "1"
        f[ns // 2] = ".:0:0"
        rows.append(_line(5000 + sh, f, chrom="X"))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"allow": "", "keepInfo": True}, max_batch_bytes=1 << 20)
    _dosage_rows_match_oracle(bv, vcf, ns)


@pytest.mark.parametrize("want_dosage", [False, True])
def test_haploid_lines_are_not_deferred(bv, bvcf_path, want_dosage):
    """biallelic lines with 5 % haploid calls stay inside k_stream_gen: no line gets a k_gt task -- with a dosage matrix
    asked for as well (round 4: such a line is marked "not regular" and k_dosage scans it; it used to go back to k_gt)"""
    if bvcf_path != "streaming-general":
        pytest.skip("k_stream_gen only")
    import torch
    ns = 2504
    rng = random.Random(5)
    body = "".join(_hap_rows(rng, ns, 64, 0.05, dots=True)).encode()
    t = torch.frombuffer(bytearray(body + b"\n" * bv.DEVICE_PAD), dtype=torch.uint8).cuda()
    ctx = bv.Ctx(9 + ns, max_batch_bytes=len(body), allow="", want_dosage=want_dosage)
    try:
        chain, scan, counts = ctx.bench_device([t.data_ptr()], [len(body)], 2, slots=1)
    finally:
        ctx.close()
    assert counts[0] == 64          # every line listed
    assert counts[4] == counts[0]   # ... and no task slot past the lines' own: nothing left to k_gt


def _odd_rows(rng, ns, n_lines, alts=("G",), odd=("0/1/1", "1/1/1", "0/0/0", "10/1", "1|12", "", "./1/0", "01/1", "1/0/0/0", "0|1|1|1|0", "2/2/1", "A/1"), per_line=3):
    """mostly-reference lines with a few fields of other shapes: polyploid, alleles of two digits, empty, odd tokens"""
    rows, pos = [], 50
    for k in range(n_lines):
        f = _ref_fields(rng, ns)
        for _ in range(per_line):
            i = rng.randrange(ns)
            g = rng.choice(odd)
            f[i] = g + (":%d:%d" % (rng.randint(0, 99), rng.randint(0, 99)) if rng.random() < 0.8 else "")
        for _ in range(rng.randint(0, 3)):  # ordinary carriers beside them
            f[rng.randrange(ns)] = "%s/%s:3:4" % (rng.choice("01."), rng.choice("01"))
        pos += 7
        rows.append(_line(pos, f, alt=alts[k % len(alts)]))
    return rows


@pytest.mark.parametrize("ns", [300, 2504])
def test_polyploid_multidigit_and_empty_fields(bv, ns):
    """fields the packed tiers do not classify -- "0/1/1", "10/1", "", "0|1|1|1|0", "A/1" (main.go:1126-1190, the general
    branch) -- on biallelic and multiallelic lines (where they carry further ALT indices), at every byte alignment, as
    the last field, beside ordinary carriers, more of them than a class list holds; rows, log and dosage rows equal the
    oracle's on every device path (k_stream_gen defers such lines to k_gt: keeping them in the kernel was built in
    round 4 and cost the kernel 19 % on every GATK-shaped file)"""
    rng = random.Random(ns + 1)
    rows = _odd_rows(rng, ns, 60, alts=("G", "G,T", "G,T,C", "G,T,C,GA,GC,GG,GT,AA,AC,AG,AT,TA"))
    for sh in range(17):  # an odd field at every byte alignment, first, last, next to a carrier
        f = _ref_fields(rng, ns)
        f[5] = "0/0/%s:%s" % (sh % 2, "7" * (1 + sh))
        f[6] = "0/1:1:1"
        f[0] = "1/1/1:5:5" if sh % 3 == 0 else f[0]
        f[ns - 1] = "0/1/0" if sh % 2 else "1|1|1:9:9"
        rows.append(_line(9000 + sh, f))
    # more odd fields than the list holds (such a line is deferred to k_gt), and a line that is dense with carriers
    f = _ref_fields(rng, ns)
    for i in range(0, 40, 2):
        f[i] = "0/1/1:2:2"
    rows.append(_line(9500, f))
    f = ["0/1:1:1" if i % 3 == 0 else x for i, x in enumerate(_ref_fields(rng, ns))]
    f[7] = "1/1/0:4:4"
    rows.append(_line(9600, f))
    vcf = (vcfgen.header(ns) + "".join(rows)).encode()
    both(bv, vcf, {"allow": ""})
    both(bv, vcf, {"allow": "", "keepInfo": True}, max_batch_bytes=1 << 20)
    _dosage_rows_match_oracle(bv, vcf, ns)
