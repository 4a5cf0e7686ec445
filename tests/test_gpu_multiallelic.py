"""Every ALT index of a multiallelic line from ONE pass (main.go:549-556 rescans per allele): on the streaming path a
line whose non-reference samples fit the raw list gets the class lists of ALT #2..#8 from the same entries
(finish_list), k_head takes the counts from the lists; everything else still goes through k_gt.  All cases against the
oracle, on both device paths."""
import random

import numpy as np
import pytest

import oracle_lib as orc
import vcfgen

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["census", "streaming"])
def bvcf_path(request, monkeypatch):
    monkeypatch.setenv("BVCF_PATH", "2" if request.param == "streaming" else "1")
    return request.param


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def both(bv, vcf, cfg=None, **kw):
    rc_o, out_o, log_o, n_o = orc.run(vcf, cfg)
    rc_g, out_g, log_g, n_g = bv.run_buffer(vcf, cfg, **kw)
    assert rc_g == 0 and rc_o == 0 and n_g == n_o
    if out_g != out_o:
        a, b = out_o.split(b"\n"), out_g.split(b"\n")
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, "row %d differs:\noracle: %r\nhip:    %r" % (i, x[:300], y[:300])
        assert len(a) == len(b)
    assert log_g == log_o
    return out_g


def _rows(seed, ns, n_lines, sep="|"):
    """multiallelic lines with 1-9 ALTs: a few carriers of some alleles (list mode), none of others, sometimes many
    (dense), sometimes missing genotypes, carriers right at the 15 / 16 entry boundary of the list"""
    rng = random.Random(seed)
    bases = "ACGT"
    rows = []
    pos = 100
    for li in range(n_lines):
        pos += rng.randint(1, 50)
        n_alt = rng.choice([1, 2, 2, 3, 3, 4, 6, 8, 9])
        ref = rng.choice(bases)
        alts = []
        for k in range(n_alt):
            alts.append(rng.choice([b for b in bases if b != ref]) + ("" if k < 3 else "A" * (k - 2)))
        gts = [["0", "0"] for _ in range(ns)]
        mode = li % 7
        # which alleles have carriers
        present = [k for k in range(1, n_alt + 1) if rng.random() < 0.7] or [1]
        if mode in (0, 1, 2):      # a few carriers: stays a list
            n_car = rng.randint(1, 12)
        elif mode == 3:            # exactly around the list limit, in distinct 4-sample groups
            n_car = rng.choice([14, 15, 16, 17])
        elif mode == 4:            # many carriers: dense map
            n_car = rng.randint(40, max(41, ns // 3))
        elif mode == 5:            # nobody carries anything (row dropped) or only the last allele
            n_car = rng.choice([0, 1])
            present = [n_alt]
        else:
            n_car = rng.randint(1, 8)
        if mode == 3:
            groups = rng.sample(range(ns // 4), min(n_car, ns // 4))
            who = [4 * g + rng.randint(0, 3) for g in groups]
        else:
            who = rng.sample(range(ns), min(n_car, ns))
        for s in who:
            a = str(rng.choice(present))
            b = str(rng.choice(present + [0, 0]))
            gts[s] = [a, b] if rng.random() < 0.5 else [b, a]
        if mode == 6 or rng.random() < 0.1:   # missing genotypes
            for s in rng.sample(range(ns), rng.randint(1, 5)):
                gts[s] = rng.choice([[".", "."], [".", "1"], ["2", "."]])
        row = ["7", str(pos), "rs%d" % li, ref, ",".join(alts), ".", "PASS", "NA=%d" % n_alt, "GT"]
        row += [sep.join(g) for g in gts]
        rows.append("\t".join(row))
    return rows


@pytest.mark.parametrize("seed,ns", [(1, 2504), (2, 2504), (3, 300), (4, 1030), (5, 513), (6, 260)])
def test_every_alt_index_matches_oracle(bv, seed, ns):
    """ns = 2504: lists for ALT #2..#8 fit the 640-byte slot; 513 (slot 144 B: lists up to ALT #2), 300 / 260 (slot
    80 B: no room for a second list, the line falls back to a dense map + k_gt)"""
    vcf = (vcfgen.header(ns) + "\n".join(_rows(seed, ns, 400, "|" if seed % 2 else "/")) + "\n").encode()
    out = both(bv, vcf, {"allow": "", "keepInfo": True})
    assert out.count(b"MULTIALLELIC") > 200
    both(bv, vcf, {"allow": "", "keepId": True}, max_batch_bytes=1 << 20)


def test_device_results_agree_between_paths(bv, monkeypatch, bvcf_path):
    """record by record: streaming-path counts and class maps of further ALT indices (taken from the lists) == the
    census path's (scanned by k_gt)"""
    ns = 2504
    rows = _rows(11, ns, 300)
    body = ("\n".join(rows) + "\n").encode()
    res = {}
    for path in (1, 2):
        ctx = bv.Ctx(9 + ns, allow="", path=path)
        res[path] = ctx.process(body)
        ctx.close()
    a, b = res[1], res[2]
    assert len(a.lines) == len(b.lines) == 300
    n_cmp = 0
    for i in range(300):
        ra, rb = a.records(i), b.records(i)
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            for f in ("alt_idx", "ac", "an", "n_het", "n_hom", "n_miss", "kind", "alt_base"):
                assert x[f] == y[f], (i, f, int(x[f]), int(y[f]))
            if x["ac"] > 0:
                assert int(y["cmap_off"]) != bv.NO_CMAP
                assert (a.classes(x) == b.classes(y)).all(), (i, int(x["alt_idx"]))
                n_cmp += 1
    assert n_cmp > 300


def test_dosage_rows_of_multiallelic_lines(bv):
    """--dosageOutput rows of alleles whose class list came from finish_list (k_dosage expands the list)"""
    import test_gpu_parity as tp
    ns = 2504
    vcf = (vcfgen.header(ns) + "\n".join(_rows(21, ns, 120)) + "\n").encode()
    got = tp._device_dosage_rows(bv, vcf)
    want = [d for _, d in orc.run_dosage(vcf, {"allow": ""})]
    assert len(got) == len(want) > 100
    for i, (dg, dw) in enumerate(zip(got, want)):
        assert dg == dw, i


def _dense_first_rows(seed, ns, n_lines, sep="|"):
    """ALT #1 common (hundreds of carriers: its class map goes dense, more than 63 non-reference groups), ALT #2..#k
    carried by a few samples each -- before, inside and after the stretch of the line where the raw list overflows --,
    sometimes missing genotypes, sometimes one further allele that is common too (no list: k_gt)"""
    rng = random.Random(seed)
    bases = "ACGT"
    rows = []
    pos = 1000
    for li in range(n_lines):
        pos += rng.randint(1, 50)
        n_alt = rng.choice([2, 2, 3, 4, 5, 7, 8])
        ref = rng.choice(bases)
        alts = [rng.choice([b for b in bases if b != ref]) + ("" if k < 3 else "C" * (k - 2)) for k in range(n_alt)]
        gts = [["0", "0"] for _ in range(ns)]
        lo = rng.choice([0, 0, ns // 3, ns // 2])       # where ALT #1's carriers start: the list overflows there
        for s in rng.sample(range(lo, ns), rng.randint(min(300, (ns - lo) // 2), (ns - lo) * 2 // 3)):
            gts[s] = rng.choice([["0", "1"], ["1", "0"], ["1", "1"]])
        mode = li % 6
        for k in range(2, n_alt + 1):
            if mode == 5 and k == n_alt:
                continue                                 # the last allele has no carriers: its row is dropped
            n_car = rng.randint(1, 14) if not (mode == 4 and k == 2) else rng.randint(40, 90)
            for s in rng.sample(range(ns), n_car):
                o = rng.choice(["0", "1", str(k), str(rng.randint(1, n_alt))])
                gts[s] = [str(k), o] if rng.random() < 0.5 else [o, str(k)]
        if mode in (1, 3):
            for s in rng.sample(range(ns), rng.randint(1, 6 if mode == 1 else 70)):
                gts[s] = rng.choice([[".", "."], [".", "1"], ["2", "."]])
        row = ["7", str(pos), "rs%d" % li, ref, ",".join(alts), ".", "PASS", "NA=%d" % n_alt, "GT"]
        row += [sep.join(g) for g in gts]
        rows.append("\t".join(row))
    return rows


@pytest.mark.parametrize("seed,ns", [(31, 2504), (32, 2504), (33, 1030), (34, 513), (35, 2560), (36, 300)])
def test_further_alleles_of_dense_lines(bv, seed, ns):
    """a line whose ALT #1 outgrows the raw list keeps listing the lanes that hold anything but 0 and 1; the further
    alleles are settled by k_gt from those (finish_dense, raw_save) instead of a rescan per allele (main.go:549-556)"""
    vcf = (vcfgen.header(ns) + "\n".join(_dense_first_rows(seed, ns, 240, "|" if seed % 2 else "/")) + "\n").encode()
    out = both(bv, vcf, {"allow": "", "keepInfo": True})
    assert out.count(b"MULTIALLELIC") > 400
    both(bv, vcf, {"allow": "", "keepId": True}, max_batch_bytes=1 << 20)


def test_dense_lines_hand_their_entries_to_k_gt(bv, bvcf_path):
    """streaming path, 2 504 samples: ALT #2.. of a line whose ALT #1 is a dense map are settled by k_gt from the entries
    k_stream saved with the line (raw_save) -- the census path's counts and classes"""
    if bvcf_path != "streaming":
        pytest.skip("streaming path only")
    ns = 2504
    # few carriers of the further alleles, nobody missing, at most four ALTs (their carriers' lanes fit the raw list);
    # three biallelic lines between any two of them, as in a real file: a wave has a quarter more class-map slots than
    # lines, and every such line takes a second one
    dense = [r for i, r in enumerate(_dense_first_rows(41, ns, 240)) if i % 6 in (0, 2) and int(r.split("NA=")[1].split("\t")[0]) <= 4]
    plain = [r for r in _rows(42, ns, 1300) if "NA=1\t" in r][:3 * len(dense)]
    assert len(plain) == 3 * len(dense)
    rows, is_dense = [], []
    for i, r in enumerate(dense):
        rows += plain[3 * i:3 * i + 3] + [r]
        is_dense += [False, False, False, True]
    body = ("\n".join(rows) + "\n").encode()
    res = {}
    for path in (1, 2):
        ctx = bv.Ctx(9 + ns, allow="", path=path)
        res[path] = ctx.process(body)
        ctx.close()
    a, b = res[1], res[2]
    n_lists = n_further = 0
    for i in range(len(rows)):
        ra, rb = a.records(i), b.records(i)
        assert len(ra) == len(rb)
        if not is_dense[i]:
            continue
        assert len(rb) >= 2
        assert not int(rb[0]["flags"]) & 2           # ALT #1: a dense map
        for x, y in zip(ra, rb):
            for f in ("alt_idx", "ac", "an", "n_het", "n_hom", "n_miss"):
                assert x[f] == y[f], (i, f, int(x[f]), int(y[f]))
            if x["ac"] > 0:
                assert (a.classes(x) == b.classes(y)).all(), (i, int(x["alt_idx"]))
                if int(y["alt_idx"]) > 0:
                    n_further += 1
                    n_lists += (int(y["flags"]) & 2) != 0
    assert n_further > 40 and n_lists == 0, (n_lists, n_further)   # (maps, not lists: k_gt writes them)


def test_dense_lines_agree_between_paths(bv, bvcf_path):
    """every shape of _dense_first_rows (further alleles with a few carriers, with more than a class list holds, with
    more lanes than the raw list holds, with missing genotypes), record by record against the census path, where k_gt
    scans every allele from the text"""
    if bvcf_path != "streaming":
        pytest.skip("compares the two paths itself")
    ns = 2504
    dense = _dense_first_rows(43, ns, 96)
    plain = [r for r in _rows(44, ns, 700) if "NA=1\t" in r][:2 * len(dense)]
    rows = []
    for i, r in enumerate(dense):
        rows += plain[2 * i:2 * i + 2] + [r]
    body = ("\n".join(rows) + "\n").encode()
    res = {}
    for path in (1, 2):
        ctx = bv.Ctx(9 + ns, allow="", path=path)
        res[path] = ctx.process(body)
        ctx.close()
    a, b = res[1], res[2]
    n_cmp = n_dense_further = 0
    for i in range(len(rows)):
        ra, rb = a.records(i), b.records(i)
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            for f in ("alt_idx", "ac", "an", "n_het", "n_hom", "n_miss"):
                assert x[f] == y[f], (i, f, int(x[f]), int(y[f]))
            if x["ac"] > 0:
                assert (a.classes(x) == b.classes(y)).all(), (i, int(x["alt_idx"]))
                n_cmp += 1
                n_dense_further += int(y["alt_idx"]) > 0 and not int(y["flags"]) & 2
    assert n_cmp > 250 and n_dense_further > 10


def test_ref_alt_lengths_around_the_word_sizes(bv):
    """k_head reads REF / ALT / POS as 8- or 16-byte words from the line's staged head (eval_token_row) and falls back to the
    byte walk (eval_token, main.go:774-999) past them: lengths around 8 and 16, fields that straddle the staged 64 bytes
    (long ID), shared suffixes and prefixes of every length, mismatches in the padding (MIXED), POS of 9 and 10 digits"""
    rng = random.Random(77)
    ns = 5
    bases = "ACGT"

    def rb(n):
        return "".join(rng.choice(bases) for _ in range(n))

    rows = []
    pos = 1000
    lens = [1, 2, 3, 7, 8, 9, 10, 15, 16, 17, 20]
    for li in range(900):
        pos += rng.randint(1, 40)
        lr = rng.choice(lens)
        ref = rb(lr)
        alts = []
        for _ in range(rng.choice([1, 1, 1, 2, 3])):
            shape = rng.randint(0, 7)
            lt = rng.choice(lens)
            if shape == 0:      # unrelated token of some length
                tok = rb(lt)
            elif shape == 1:    # insertion after a shared prefix, sharing a suffix of some length
                p = rng.randint(1, lr)
                tok = ref[:p] + rb(rng.randint(1, 9)) + ref[p:]
            elif shape == 2:    # deletion, sharing prefix and suffix
                if lr < 2:
                    tok = ref + rb(2)
                else:
                    p = rng.randint(1, lr - 1)
                    q = rng.randint(p, lr)
                    tok = ref[:p] + ref[q:]
            elif shape == 3:    # equal length: a few differing bases (or none)
                tok = "".join(c if rng.random() < 0.7 else rng.choice(bases) for c in ref)
            elif shape == 4:    # shares only a suffix (the prefix test fails: MIXED) or only the first base
                s = rng.randint(0, lr)
                tok = rb(rng.randint(1, 6)) + ref[lr - s:]
            elif shape == 5:    # single base
                tok = rng.choice([ref[0], rng.choice(bases)])
            elif shape == 6:    # junk
                tok = rng.choice(["N", "", "<DEL>", "a", ref[:1] + "N" + ref[1:], "*"])
            else:               # the whole REF repeated / truncated
                tok = (ref * 3)[: rng.choice(lens)]
            alts.append(tok)
        ident = rng.choice(["rs%d" % li, ".", "rs" + "9" * rng.randint(20, 45)])   # (a long ID pushes REF / ALT past byte 48 / 64)
        pos_s = rng.choice([str(pos), str(pos), str(10 ** 8 + pos), str(10 ** 9 + pos), "0%d" % pos, str(2 ** 31 + pos)])
        gts = [rng.choice(["0|0", "0|1", "1|1", "1|2", "2|0", ".|.", "0|3"]) for _ in range(ns)]
        rows.append("\t".join([rng.choice(["1", "chr7"]), pos_s, ident, ref, ",".join(alts), "50", "PASS", "NS=%d" % ns, "GT"] + gts))
    vcf = (vcfgen.header(ns) + "\n".join(rows) + "\n").encode()
    out = both(bv, vcf, {"allow": ""})
    assert out.count(b"\n") > 600 and out.count(b"DEL") > 50 and out.count(b"INS") > 50
    both(bv, vcf, {"allow": "", "keepId": True, "keepInfo": True, "keepPos": True}, max_batch_bytes=1 << 16)
