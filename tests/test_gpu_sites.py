"""The kernels for sites-only input (no sample columns; BASELINE configs[1]) -- k_sites2, tiles behind the newline
census with the common lines on fast lanes, with the packed form of the results (32-byte site records, ABI 6: the
default of the host driver) and with the full one; in builds with -DBVCF_EXPERIMENTS also k_sites1, the same without the
census, the line numbers from a look-back over tile counts (BVCF_SITES=3), and k_sites, round 2's kernel (BVCF_SITES=1) --:
lines of every length around their 8 KiB windows, tiles and runs that start in the middle of a line, rounds of more
than 64 lines, and agreement with the census chain they replace (BVCF_SITES=0: k_scatter_eol + k_head)."""
import random

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

H8 = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
H9 = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\n"


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


@pytest.fixture(autouse=True, params=["k_sites2-packed-rendered", "k_sites2-packed", "k_sites2-packed-chunk-census", "k_sites2", "k_sites2-chunk-census",
                                      "k_sites1", "k_sites"])
def kernel(request, monkeypatch, bv):
    """k_sites2 behind its census per tile (k_census_tiles, the default) and behind the per-chunk census of the other chains,
    each with the packed and with the full form of the results; the packed form also with the rows of the lines it settles
    rendered on the device (bvcf_params.render_sites: the host driver's default for a file without samples)"""
    name = request.param
    monkeypatch.setenv("BVCF_RENDER_SITES", "1" if name.endswith("rendered") else "0")
    if name in ("k_sites1", "k_sites") and b"experiments" not in bv.lib.bvcf_version():
        pytest.skip("%s is only in builds with -DBVCF_EXPERIMENTS" % name)
    monkeypatch.setenv("BVCF_SITES", {"k_sites1": "3", "k_sites": "1"}.get(name, "2"))
    monkeypatch.setenv("BVCF_S2_CENSUS", "chunk" if name.endswith("chunk-census") else "tile")
    monkeypatch.setenv("BVCF_PACKED_SITES", "1" if "packed" in name else "0")
    return name


def both(bv, vcf, cfg=None, **kw):
    rc_o, out_o, log_o, n_o = orc.run(vcf, cfg)
    rc_g, out_g, log_g, n_g = bv.run_buffer(vcf, cfg, **kw)
    assert (rc_g != 0) == (rc_o != 0), (rc_g, rc_o, log_g)
    assert n_g == n_o
    if out_g != out_o:
        a, b = out_o.split(b"\n"), out_g.split(b"\n")
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, "row %d differs:\noracle: %r\nhip:    %r" % (i, x[:200], y[:200])
        assert len(a) == len(b)
    assert log_g == log_o
    return out_g


def _line(rng, pos, info_len=None, kind=None):
    ref = rng.choice("ACGT")
    k = kind if kind is not None else rng.random()
    if k < 0.6:
        alt = rng.choice([b for b in "ACGT" if b != ref])
    elif k < 0.7:
        alt = ref + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 12)))
    elif k < 0.8:
        ref = ref + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 12)))
        alt = ref[0]
    elif k < 0.9:
        alt = ",".join(rng.choice("ACGT") + ("" if rng.random() < 0.5 else "AG") for _ in range(rng.randint(2, 4)))
    elif k < 0.95:
        alt = rng.choice(["<CN0>", ".", "N", ref, "A,<DEL>", ""])
    else:
        ref = "".join(rng.choice("ACGT") for _ in range(5))
        alt = "".join(rng.choice("ACGT") for _ in range(5))
    flt = rng.choice(["PASS", "PASS", "PASS", ".", "q10", "LowQual;s50"])
    n = info_len if info_len is not None else rng.randint(1, 120)
    info = "AC=%d;" % rng.randint(1, 5000) + "X" * n
    return "\t".join(["%s" % rng.choice(["1", "chr2", "X", "contig77"]), str(pos), "rs%d" % pos, ref, alt, "100", flt, info])


@pytest.mark.parametrize("seed,n_lines", [(1, 3000), (2, 40000)])
def test_sites_parity_random(bv, seed, n_lines):
    rng = random.Random(seed)
    rows, pos = [], 1000
    for _ in range(n_lines):
        pos += rng.randint(1, 300)
        rows.append(_line(rng, pos))
    vcf = (H8 + "\n".join(rows) + "\n").encode()
    both(bv, vcf)
    both(bv, vcf, {"allow": "", "keepId": True, "keepInfo": True, "keepPos": True}, max_batch_bytes=1 << 20)


@pytest.mark.parametrize("info_len", [8000, 8100, 8192 - 40, 8192, 16300, 16384, 16500, 40000, 200000])
def test_sites_long_lines(bv, info_len):
    """lines about as long as a window (8 KiB) and as the ring (16 KiB), and far longer, between ordinary ones: the
    start of such a line has left the ring when its terminator arrives (wave-cooperative TAB search, head bytes from
    memory)"""
    rng = random.Random(info_len)
    rows, pos = [], 5
    for i in range(400):
        pos += 7
        rows.append(_line(rng, pos, info_len if i % 23 == 5 else None))
    vcf = (H8 + "\n".join(rows) + "\n").encode()
    both(bv, vcf, {"allow": "", "keepInfo": True})
    both(bv, vcf.replace(b"\n", b"\r\n"), {"allow": ""})


@pytest.mark.parametrize("shift", list(range(0, 48, 5)) + [8191, 8192, 8193])
def test_sites_every_alignment_of_window_boundaries(bv, shift):
    """a first junk line of `shift` bytes moves every later line against the 8 KiB windows and 1 KiB chunks"""
    rng = random.Random(99)
    rows, pos = ["#" + "j" * max(shift - 2, 0)] if shift else [], 10
    for _ in range(1500):
        pos += 3
        rows.append(_line(rng, pos))
    both(bv, (H8 + "\n".join(rows) + "\n").encode(), {"allow": ""})


def test_sites_many_short_lines(bv):
    """hundreds of terminators per chunk (empty lines, two-byte junk): rounds of 64 lines in the middle of a chunk,
    FIFO compaction; every one is a listed line that fails the field count"""
    rng = random.Random(5)
    rows, pos = [], 10
    for i in range(6000):
        pos += 3
        rows.append(_line(rng, pos))
        if i % 40 == 7:
            rows.extend([""] * rng.randint(1, 700))
        if i % 55 == 9:
            rows.extend(["x"] * rng.randint(1, 1500))
    vcf = (H8 + "\n".join(rows) + "\n").encode()
    both(bv, vcf, {"allow": ""})
    # unterminated tail (dropped, main.go:354-358) and a body of terminators only
    both(bv, vcf + b"1\t5\t.\tA\tG\t.\tPASS\t.", {"allow": ""})
    both(bv, (H8 + "\n" * 5000).encode())
    both(bv, H8.encode())


def test_sites_nine_header_fields(bv):
    """FORMAT column without samples: 9 header fields, numSamples 0 (main.go:505-509 logs a warning)"""
    rng = random.Random(11)
    rows, pos = [], 10
    for i in range(3000):
        pos += 3
        rows.append(_line(rng, pos) + ("\tGT" if i % 7 else ""))
    out = both(bv, (H9 + "\n".join(rows) + "\n").encode(), {"allow": ""})
    assert out.count(b"\n") > 1500


def test_sites_bench_shape_runs_of_many_windows(bv, kernel):
    """BASELINE configs[1] rows from the bench generator, 300 000 of them (45 MB: several windows per wave, most runs
    start in the middle of a line), against the oracle; and the census chain it replaces gives the same records"""
    import benchgen as bg
    cfg = bg.make_cfg("c2")
    body = bg.rows_host(cfg, 2_000_000, 300_000)
    vcf = bg.header(cfg) + body
    rc_o, out_o, log_o, n_o = orc.run(vcf, None, n_threads=8)
    rc_g, out_g, log_g, n_g = bv.run_buffer(vcf)
    assert rc_g == 0 and n_g == n_o == 300_000 and out_g == out_o and log_g == log_o
    n_hdr = bg.n_header_fields(cfg)
    ctx = bv.Ctx(n_hdr, max_batch_bytes=len(body), packed_sites="packed" in kernel)
    new = ctx.process(body)
    ctx.close()
    assert (new.sites is not None) == ("packed" in kernel)
    if new.sites is not None:
        assert len(new.full_lines) == 0, "bench rows are all plain SNPs: no line needs full records"
    import os
    mine = os.environ["BVCF_SITES"]
    os.environ["BVCF_SITES"] = "0"
    try:
        ctx = bv.Ctx(n_hdr, max_batch_bytes=len(body))
        old = ctx.process(body)
        ctx.close()
    finally:
        os.environ["BVCF_SITES"] = mine
    assert len(new.lines) == len(old.lines) == 300_000
    for f in ("off", "len", "fend", "n_rec", "n_fields", "status", "site_type"):
        assert (new.lines[f] == old.lines[f]).all(), f
    for f in ("pos", "line", "alt_idx", "alt_len", "ref", "alt_base", "kind", "site_type", "trtv", "flags"):
        assert (new.alleles[:300_000][f] == old.alleles[:300_000][f]).all(), f


def test_batches_of_different_sizes_in_turn_on_the_same_slots(bv, kernel):
    """k_census_tiles adds a batch's line ends to one of the slot's two sets of group totals and clears the other for the
    slot's next batch: batches of very different sizes one after the other, on one slot and on three, each against a ctx
    of its own"""
    if not kernel.startswith("k_sites2"):
        pytest.skip("k_sites2 behind its two censuses")
    import benchgen as bg
    cfg = bg.make_cfg("c2")
    n_hdr = bg.n_header_fields(cfg)
    sizes = [200_000, 7, 60_000, 1, 350_000, 3_000, 350_000, 40]
    bodies = [bg.rows_host(cfg, 1_000 + 400_000 * i, n) for i, n in enumerate(sizes)]
    cap = max(len(b) for b in bodies)
    want = []
    for b, n in zip(bodies, sizes):
        ctx = bv.Ctx(n_hdr, max_batch_bytes=cap, max_lines=max(sizes) + 16, n_slots=1, packed_sites="packed" in kernel)
        r = ctx.process(b)
        want.append((r.lines.tobytes(), r.alleles[:n].tobytes()))
        assert len(r.lines) == n
        ctx.close()
    for n_slots in (1, 3):
        ctx = bv.Ctx(n_hdr, max_batch_bytes=cap, max_lines=max(sizes) + 16, n_slots=n_slots, packed_sites="packed" in kernel)
        for rnd in range(2):
            for i, b in enumerate(bodies):
                r = ctx.process(b)
                assert len(r.lines) == sizes[i], (n_slots, rnd, i)
                assert r.lines.tobytes() == want[i][0], (n_slots, rnd, i)
                assert r.alleles[:sizes[i]].tobytes() == want[i][1], (n_slots, rnd, i)
        ctx.close()


def test_sites_block_of_many_scan_steps(bv, kernel):
    """a sites-only block of 1.9 M rows = 270 MB: 37 k tiles in 74 groups of k_census_tiles (more than the 64 a
    wave sums with one load), line numbers past 2^20 -- against the census chain with k_head on the same block"""
    if not kernel.startswith("k_sites2"):
        pytest.skip("k_sites2 behind its two censuses")
    import os
    import benchgen as bg
    cfg = bg.make_cfg("c2")
    n = 1_900_000
    body = bg.rows_host(cfg, 5_000_000, n)
    assert len(body) > 32_768 * 7168
    n_hdr = bg.n_header_fields(cfg)
    ctx = bv.Ctx(n_hdr, max_batch_bytes=len(body), max_lines=n + 16, packed_sites="packed" in kernel)
    new = ctx.process(body)
    ctx.close()
    mine = os.environ["BVCF_SITES"]
    os.environ["BVCF_SITES"] = "0"
    try:
        ctx = bv.Ctx(n_hdr, max_batch_bytes=len(body), max_lines=n + 16)
        old = ctx.process(body)
        ctx.close()
    finally:
        os.environ["BVCF_SITES"] = mine
    assert len(new.lines) == len(old.lines) == n
    for f in ("off", "len", "fend", "n_rec", "n_fields", "status", "site_type"):
        assert (new.lines[f] == old.lines[f]).all(), f
    for f in ("pos", "line", "alt_idx", "alt_len", "ref", "alt_base", "kind", "site_type", "trtv", "flags"):
        assert (new.alleles[:n][f] == old.alleles[:n][f]).all(), f


def test_packed_form_record_by_record(bv, kernel):
    """ABI 6: the packed form (a 32-byte bvcf_site per line, full records only for the lines that need them) describes
    every line exactly as the full form of the census chain does -- lines of all kinds, in rounds where packed and full
    lines mix, with messages to log"""
    if "packed" not in kernel:
        pytest.skip("the packed form")
    import os
    rng = random.Random(23)
    rows, pos = [], 500
    for i in range(20000):
        pos += rng.randint(1, 50)
        rows.append(_line(rng, pos, kind=None if i % 3 else 0.1))
    body = ("\n".join(rows) + "\n").encode()
    ctx = bv.Ctx(8, allow="PASS,.", max_batch_bytes=len(body), packed_sites=True)
    new = ctx.process(body)
    ctx.close()
    mine = os.environ["BVCF_SITES"]
    os.environ["BVCF_SITES"] = "0"
    try:
        ctx = bv.Ctx(8, allow="PASS,.", max_batch_bytes=len(body))
        old = ctx.process(body)
        ctx.close()
    finally:
        os.environ["BVCF_SITES"] = mine
    assert new.sites is not None and old.sites is None
    n = len(old.lines)
    assert len(new.lines) == n == 20000
    full = (new.sites["status"] & bv.SITE_FULL) != 0
    assert 0 < full.sum() < n and len(new.full_lines) == full.sum()
    # a line stays packed exactly when it is settled without getAlleles' general code: failed the gate, or a plain SNP
    for i in range(n):
        L = old.lines[i]
        if not full[i]:
            assert L["status"] in (bv.LINE_OK, bv.LINE_FIELDS, bv.LINE_FILTER)
            if L["status"] == bv.LINE_OK:
                r = old.records(i)
                assert len(r) == 1 and r[0]["kind"] == bv.ALT_BASE and r[0]["site_type"] == 0 and r[0]["flags"] & 1
    for f in ("off", "len", "n_rec", "status"):
        assert (new.lines[f] == old.lines[f]).all(), f
    ok = old.lines["status"] == bv.LINE_OK
    assert (new.lines["fend"][ok] == old.lines["fend"][ok]).all()
    assert (new.lines["site_type"][ok] == old.lines["site_type"][ok]).all()
    for i in np.nonzero(ok)[0]:
        a, b = new.records(i), old.records(i)
        assert len(a) == len(b)
        for f in ("pos", "alt_idx", "alt_off", "alt_len", "ref", "alt_base", "kind", "site_type", "trtv", "flags"):
            assert (a[f] == b[f]).all(), (i, f)
    key = lambda e: (int(e["line"]), int(e["alt_no"]), int(e["code"]))
    assert sorted(map(key, new.errs)) == sorted(map(key, old.errs)) and len(old.errs) > 0


def test_packed_ctx_grows_lines_and_extra_alt_records_together(bv, kernel):
    """short sites-only lines (about 21 bytes against the 48-byte floor of the first reservation) with three ALTs each:
    the batch overflows on lines AND carries far more than 64 extra ALT records, which a packed ctx keeps behind slot
    cap_lines -- growing the lines moves where they start, so the reservation must grow with the NEW line capacity
    (round 4's advisor finding: the second collect failed with "batch exceeds reserved result capacity").  Library level
    with the driver's own growth rule, then the CLI."""
    import os
    import subprocess
    n = 30_000
    body = "".join("1\t%d\t.\tA\tC,G,T\t.\t.\t.\n" % (100 + i) for i in range(n)).encode()
    vcf = H8.encode() + body
    ctx = bv.Ctx(8, allow="", max_batch_bytes=len(body), max_lines=256, max_alleles=320, packed_sites="packed" in kernel)
    try:
        need = None
        for _ in range(4):
            ctx.submit(body)
            r = bv.Result()
            rc = bv.lib.bvcf_collect(ctx.h, bv.C.byref(r))
            if rc != bv.E_CAPACITY:
                break
            need = (r.need_lines, r.need_alleles)
            ctx.reserve(r.need_lines + r.need_lines // 4 + 64, r.need_alleles + r.need_alleles // 4 + 64, r.need_cmap_bytes + 4096)
        assert need is not None and rc == 0, (rc, need)
        assert r.n_lines == n
    finally:
        ctx.close()
    out = both(bv, vcf, {"allow": ""})
    assert out.count(b"\n") == 3 * n
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bystro-vcf_amd", "bystro-vcf")
    p = subprocess.run([exe, "--allowFilter", "", "--batchMB", "1"], input=vcf, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == (bv.string_header() + "\n").encode() + out


@pytest.mark.parametrize("flags", [{}, {"keepPos": True}, {"keepId": True, "keepInfo": True}, {"keepPos": True, "keepId": True, "keepInfo": True, "emptyField": "NA"},
                                   {"emptyField": "sixteen_bytes_xx"}, {"emptyField": "seventeen_bytes_x"}])
def test_rendered_rows_are_the_oracles_rows(bv, kernel, flags):
    """bvcf_params.render_sites through the C-ABI: a file of plain SNP lines that all pass is settled on the fast lanes, so
    the stream the device renders IS the oracle's output, byte for byte, for every combination of the optional columns and
    --emptyField (17 bytes of it: the library refuses, the host driver then formats itself); lines that do not pass leave
    nothing; a mixed file comes back as the stream plus the cuts of the lines left to the host, and through the host driver
    as the oracle's rows."""
    if not kernel.endswith("rendered"):
        pytest.skip("the rendered form")
    rng = random.Random(5)
    rows, pos = [], 100
    for i in range(20_000):
        pos += rng.randint(1, 500)
        chrom = rng.choice(["1", "chr7", "X", "c", "chrUn_KI270742v1", "22"])
        ref = rng.choice("ACGT")
        alt = rng.choice([b for b in "ACGT" if b != ref])
        flt = "PASS" if i % 11 else "q10"
        rows.append("\t".join([chrom, str(pos), "rs%d" % i if i % 3 else ".", ref, alt, "%d" % rng.randint(1, 999), flt,
                               "AC=%d;AF=0.%04d" % (rng.randint(1, 5000), rng.randint(1, 9999))]))
    vcf = (H8 + "\n".join(rows) + "\n").encode()
    rc_o, out_o, log_o, n_o = orc.run(vcf, flags)
    assert rc_o == 0 and out_o.count(b"\n") == sum(1 for i in range(20_000) if i % 11)
    both(bv, vcf, flags)
    body = vcf[len(H8):]
    kw = dict(empty_field=flags.get("emptyField", "!"), keep_pos=flags.get("keepPos", False), keep_id=flags.get("keepId", False),
              keep_info=flags.get("keepInfo", False))
    if len(kw["empty_field"]) > 16:
        with pytest.raises(bv.BvcfError):
            bv.Ctx(8, max_batch_bytes=len(body), render_sites=True, **kw)
        return
    ctx = bv.Ctx(8, max_batch_bytes=len(body), render_sites=True, **kw)
    try:
        b = ctx.process(body)
        assert b.sites is None and len(b.row_cuts) == 0 and b.n_lines == 20_000
        assert b.n_ok_sites == out_o.count(b"\n")
        assert b.rows == out_o
        # a mixed block: every fifth line an insertion (left to the host: a cut at its place in the stream)
        mixed = []
        for i, ln in enumerate(rows[:5000]):
            f = ln.split("\t")
            if i % 5 == 0:
                f[4] = f[3] + "TT"
            mixed.append("\t".join(f))
        mbody = ("\n".join(mixed) + "\n").encode()
        b = ctx.process(mbody)
        # (an insertion that also fails the FILTER gate may be settled on the fast lanes as FILTER: no cut then)
        cut_lines = b.row_cuts["line"].tolist()
        assert cut_lines == sorted(cut_lines) and {i for i in range(0, 5000, 5) if i % 11} <= set(cut_lines) <= set(range(0, 5000, 5))
        assert (np.diff(b.row_cuts["off"].astype(np.int64)) >= 0).all() and b.row_cuts["off"][-1] <= len(b.rows)
        rc_m, out_m, _, _ = orc.run(H8.encode() + mbody, flags)
        want = [r for r in out_m.split(b"\n")[:-1] if b"\tSNP\t" in r]
        assert b.rows == b"".join(r + b"\n" for r in want)
        both(bv, H8.encode() + mbody, flags, max_batch_bytes=1 << 18)
    finally:
        ctx.close()
    if not flags:
        # rows longer than their lines (16-byte lines, 38-byte rows): the stream, sized for a typical file, grows
        short = "".join("1\t%d\t.\tA\tC\t.\t.\t.\n" % (i % 9 + 1) for i in range(400_000)).encode()
        rc_s, out_s, _, _ = orc.run(H8.encode() + short, {"allow": ""})
        ctx = bv.Ctx(8, allow="", max_batch_bytes=len(short), max_lines=400_100, render_sites=True)
        try:
            for _ in range(2):
                b = ctx.process(short)
                assert len(b.rows) > len(short) // 2 + (1 << 20) and b.rows == out_s
        finally:
            ctx.close()
