"""k_inflate / k_crc32: BGZF blocks inflated on the device must give zlib's bytes, for every DEFLATE block type and
code shape, and corrupt data must be refused (never a hang, never wrong text)."""
import os
import random
import struct
import zlib

import pytest

import bgzf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


@pytest.fixture(autouse=True, params=["w32", "w16", "w4"])
def window(request, monkeypatch):
    """the three inflate kernels: the 32 KiB window, and the 16 and 4 KiB ones that read older bytes back from memory"""
    monkeypatch.setenv("BVCF_INFLATE_W16", {"w32": "0", "w16": "1", "w4": "2"}[request.param])
    return request.param


def _block_raw(payload, data):
    bsize = len(payload) + 25  # header 18 + payload + crc 4 + isize 4 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + payload +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _texts():
    rng = random.Random(7)
    vcf_line = ("1\t%d\trs%d\tA\tG\t100\tPASS\tAC=%d;AF=0.0%d\tGT\t" + "\t".join(rng.choice(["0|0"] * 30 + ["0|1", "1|1"]) for _ in range(2504)) + "\n")
    vcf = "".join(vcf_line % (1000 + i, i, i % 50, i % 9) for i in range(12)).encode()
    return {
        "empty": b"",
        "one": b"x",
        "three": b"abc",
        "short": b"hello, hello, hello world\n",
        "zeros": bytes(65000),
        "run1": b"a" * 50000,
        "period3": b"abc" * 20000,
        "period4_gt": b"0|0\t" * 16000,
        "vcf": vcf[:65000],
        "random": bytes(rng.getrandbits(8) for _ in range(60000)),
        "random_small_alphabet": bytes(rng.choice(b"ACGT\n") for _ in range(65000)),
        "text": (b"The quick brown fox jumps over the lazy dog. " * 2000)[:64000],
        "bytes_all": bytes(range(256)) * 200,
        "max": bytes(rng.choice(b"01|\t") for _ in range(65280)),
        # 20 KB of noise, then copies of it: matches that reach 20 KB back (past the 16 KiB window)
        "far_refs": (lambda n: n + n[:15000] + b"xyz" + n[3000:19000] + n[:9000])(bytes(rng.getrandbits(8) for _ in range(20000))),
        # lines of 30 KB that repeat the line before them
        "long_lines": b"".join(bytes(rng.choice(b"01|\t") for _ in range(40)) + (b"0|0\t" * 7400) + b"\n" for _ in range(2)),
    }


@pytest.mark.parametrize("name", list(_texts().keys()))
def test_inflate_matches_zlib(bv, name):
    data = _texts()[name]
    variants = []
    for level in (0, 1, 6, 9):                       # 0 = stored blocks
        variants.append(_deflate(data, level))
    variants.append(_deflate(data, 6, zlib.Z_FIXED))  # fixed Huffman codes
    variants.append(_deflate(data, 6, zlib.Z_HUFFMAN_ONLY))  # literals only, long codes
    variants.append(_deflate(data, 9, zlib.Z_RLE))
    variants.append(_deflate(data, 6, memlevel=1))    # many small deflate blocks in one member
    comp = b""
    want = b""
    for p in variants:
        if len(p) + 26 > 65536:
            continue
        comp += _block_raw(p, data)
        want += data
    comp += bgzf.bgzf_block(b"")
    rc, text, n = bv.bgzf_inflate_device(comp, cap=len(want) + 64)
    assert rc == 0, (name, rc)
    assert n == len(want) and text == want, name


def test_many_blocks_and_multi_flush_members(bv):
    rng = random.Random(3)
    data = _texts()["vcf"] * 40
    comp = bgzf.bgzf_compress(data, block=0xFF00, level=6)
    rc, text, n = bv.bgzf_inflate_device(comp, cap=len(data) + 64)
    assert rc == 0 and text == data
    # members made of several deflate blocks (Z_FULL_FLUSH / Z_SYNC_FLUSH inside): stored empty blocks between them
    parts, want = b"", b""
    for i in range(30):
        chunk = data[i * 50000:(i + 1) * 50000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = b""
        for j in range(0, len(chunk), 7000):
            payload += c.compress(chunk[j:j + 7000]) + c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        payload += c.flush()
        parts += _block_raw(payload, chunk)
        want += chunk
    rc, text, n = bv.bgzf_inflate_device(parts, cap=len(want) + 64)
    assert rc == 0 and text == want


def test_corrupt_blocks_are_refused(bv):
    rng = random.Random(11)
    data = _texts()["vcf"]
    good = bgzf.bgzf_block(data)
    # CRC / ISIZE damage, payload bit flips, truncation, garbage
    bad = bytearray(good)
    bad[-8] ^= 0x01
    assert bv.bgzf_inflate_device(bytes(bad), cap=70000)[0] == bv.E_FATAL
    for _ in range(40):
        bad = bytearray(good)
        k = rng.randrange(18, len(good) - 8)
        bad[k] ^= 1 << rng.randrange(8)
        rc, text, _ = bv.bgzf_inflate_device(bytes(bad), cap=70000)
        assert rc == bv.E_FATAL or (rc == 0 and text == data)  # (a flip in unused padding bits may leave the text intact)
    assert bv.bgzf_inflate_device(good[:-5], cap=70000)[0] == bv.E_FATAL
    assert bv.bgzf_inflate_device(b"not bgzf at all", cap=100)[0] == bv.E_FATAL
    junk = _block_raw(bytes(rng.getrandbits(8) for _ in range(3000)), data)
    assert bv.bgzf_inflate_device(junk, cap=70000)[0] == bv.E_FATAL
    # output buffer too small: the size needed comes back
    rc, _, need = bv.bgzf_inflate_device(good, cap=100)
    assert rc == bv.E_TOO_BIG and need == len(data)


def test_reference_regression_file_is_bgzf_compatible(bv, golden_1kg):
    """the 200 MB 1000-Genomes input of the reference's own check, BGZF-compressed here, through the device"""
    vcf, _, _ = golden_1kg
    part = vcf[: 40 << 20]
    comp = bgzf.bgzf_compress(part, level=1)
    rc, text, n = bv.bgzf_inflate_device(comp, cap=len(part) + 64)
    assert rc == 0 and text == part


def _bgzf_blocks(data, block, level=6):
    return [bgzf.bgzf_block(data[i:i + block], level) for i in range(0, len(data), block)]


@pytest.mark.parametrize("ns,block,per_batch,look", [(2504, 0xFF00, 5, 1), (300, 3000, 7, 2), (40, 700, 3, 6), (2504, 20000, 1, 2),
                                                     (0, 900, 4, 2)])
def test_submit_bgzf_batches_cover_every_line_once(bv, ns, block, per_batch, look):
    """a VCF body cut into batches of BGZF blocks at arbitrary byte positions: own blocks + look-ahead, skip-first-line
    on all but the first batch.  Every line must come out exactly once, in order, with the records of a text submit."""
    import vcfgen
    vcf = vcfgen.gen_vcf(ns, 300 if ns > 1000 else 2000, ns, weird=0.02)
    body = vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:]
    # a junk prefix stands for the header lines that share the first block with the data
    prefix = b"##junk header bytes\n#CHROM\tPOS\n"
    blocks = _bgzf_blocks(prefix + body, block)
    ctx_t = bv.Ctx(9 + ns, allow="", max_batch_bytes=len(body) + 4096)
    want = ctx_t.process(body)
    ctx_t.close()
    def head(L):  # what comes back per line: all of it without samples, else CHROM..INFO of the lines that passed
        full = body[int(L["off"]):int(L["off"]) + int(L["len"])]
        if ns == 0:
            return full
        return full[:min(int(L["fend"][7]), int(L["len"]))] if int(L["status"]) in (0, 3) else None
    want_lines = [head(L) for L in want.lines]
    ctx = bv.Ctx(9 + ns, allow="", max_batch_bytes=max(1 << 20, (per_batch + look + 1) * (block + 64)), n_slots=3)
    got_lines, got_recs, pending = [], [], []

    def collect():
        b = ctx.collect()
        if len(b.lines):  # (a batch may own no text at all)
            assert (b.head_off is None) == (ns == 0)
        for i, L in enumerate(b.lines):
            got_lines.append(b.line_head(i) if ns == 0 or int(L["status"]) in (0, 3) else None)
            got_recs.append([(int(r["alt_idx"]), int(r["ac"]), int(r["an"]), int(r["n_het"]), int(r["n_hom"]), int(r["n_miss"]))
                             for r in b.records(i)] if L["status"] == 0 else None)

    for b0 in range(0, len(blocks), per_batch):
        own = blocks[b0:b0 + per_batch]
        ahead = blocks[b0 + per_batch:b0 + per_batch + look]
        comp = b"".join(own + ahead)
        if len(pending) == 2:
            collect()
            pending.pop(0)
        ctx.submit_bgzf(comp, sum(len(x) for x in own), skip_first_line=b0 > 0, first_off=len(prefix) if b0 == 0 else 0, seq=b0)
        pending.append(b0)
    while pending:
        collect()
        pending.pop(0)
    ctx.close()
    assert got_lines == want_lines
    want_recs = [[(int(r["alt_idx"]), int(r["ac"]), int(r["an"]), int(r["n_het"]), int(r["n_hom"]), int(r["n_miss"]))
                  for r in want.records(i)] if want.lines[i]["status"] == 0 else None for i in range(len(want.lines))]
    assert got_recs == want_recs


def test_submit_bgzf_errors(bv):
    import vcfgen
    ns = 50
    vcf = vcfgen.gen_vcf(5, 400, ns)
    body = vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:]
    blocks = _bgzf_blocks(body, 2000)
    ctx = bv.Ctx(9 + ns, allow="", max_batch_bytes=1 << 20)
    # a corrupt own block is refused at collect, and the ctx stays usable
    bad = bytearray(blocks[1])
    bad[30] ^= 0x55
    ctx.submit_bgzf(blocks[0] + bytes(bad) + blocks[2], len(blocks[0]) + len(bad), False)
    with pytest.raises(bv.BvcfError) as ei:
        ctx.collect()
    assert ei.value.rc == bv.E_FATAL and "bgzf" in str(ei.value)
    # a line that does not end within the look-ahead
    long_line = b"1\t5\t.\tA\tG\t.\tPASS\t" + b"X" * 9000 + b"\n"
    lb = _bgzf_blocks(body[:3000] + long_line + body[3000:6000], 1500)
    ctx.submit_bgzf(b"".join(lb[:3]), len(lb[0]) + len(lb[1]), False)
    with pytest.raises(bv.BvcfError) as ei:
        ctx.collect()
    assert ei.value.rc == bv.E_FATAL and "look-ahead" in str(ei.value)
    # n_own off a block boundary, not BGZF
    with pytest.raises(bv.BvcfError):
        ctx.submit_bgzf(blocks[0] + blocks[1], len(blocks[0]) + 3, False)
    with pytest.raises(bv.BvcfError):
        ctx.submit_bgzf(b"plain text\n", 5, False)
    ctx.submit_bgzf(b"".join(blocks), sum(len(x) for x in blocks), False)
    b = ctx.collect()
    assert len(b.lines) == body.count(b"\n")
    ctx.close()


def _cli(args, data, env=None):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bystro-vcf_amd", "bystro-vcf")
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([exe] + args, input=data, capture_output=True, timeout=600, env=e)


@pytest.mark.parametrize("ns,n_lines,block,batch_mb,devices", [
    (150, 4000, 0xFF00, 1, "0"), (150, 4000, 777, 1, "0,0"), (2504, 600, 30000, 2, "0"), (0, 30000, 5000, 1, "0,0,0"),
    (40000, 24, 0xFF00, 8, "0"),
])
def test_cli_bgzf_on_device_equals_text_input(bv, ns, n_lines, block, batch_mb, devices):
    """the CLI over BGZF input: blocks inflated on the device (default) and by the host's zlib workers
    (BVCF_DEVICE_INFLATE=0) must both give the text input's bytes; small blocks and batches make every batch start and
    end inside a line; 40 000 samples: lines of 160 KB span three BGZF blocks"""
    import json
    import vcfgen
    vcf = vcfgen.gen_vcf(900 + ns % 97, n_lines, ns, weird=0.02)
    want = _cli(["--keepId", "--batchMB", str(batch_mb)], vcf)
    assert want.returncode == 0
    data = bgzf.bgzf_compress(vcf, block=block, level=1, eof_marker=block != 777)
    for dev_inf in ("1", "0"):
        p = _cli(["--keepId", "--batchMB", str(batch_mb), "--devices", devices], data,
                 {"BVCF_DEVICE_INFLATE": dev_inf, "BVCF_TIMING": "json"})
        assert p.returncode == 0, (dev_inf, p.stderr[-300:])
        assert p.stdout == want.stdout, dev_inf
        log = [ln for ln in p.stderr.decode().splitlines() if not ln.startswith("[bvcf timing")]
        assert "\n".join(log) == want.stderr.decode().rstrip("\n"), dev_inf
        t = [json.loads(ln.split("] ", 1)[1]) for ln in p.stderr.decode().splitlines() if ln.startswith("[bvcf timing-json]")][0]
        assert ("device" in t["input"]) == (dev_inf == "1")
        assert t["lines_in"] == vcf[vcf.index(b"\n", vcf.index(b"#CHROM")) + 1:].count(b"\n")


def test_cli_bgzf_edge_files(bv, golden_1kg):
    import vcfgen
    hdr = vcfgen.header(3).encode()
    # header only; header + unterminated line; data starting exactly at a block boundary; header spread over blocks
    for vcf, blk in ((hdr, 0xFF00), (hdr + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0", 0xFF00), (hdr + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\n" * 50, len(hdr)),
                     (hdr + b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\t0|0\n" * 50, 13)):
        want = _cli([], vcf)
        p = _cli([], bgzf.bgzf_compress(vcf, block=blk))
        assert p.returncode == want.returncode == 0 and p.stdout == want.stdout and p.stderr == want.stderr
    # fatal paths keep their messages
    p = _cli([], bgzf.bgzf_compress(b"not a vcf\n"))
    assert p.returncode == 1 and b"Not a VCF file" in p.stderr
    p = _cli([], bgzf.bgzf_compress(b"##fileformat=VCFv4.2\n##x\n"))
    assert p.returncode == 1 and b"No header found" in p.stderr
    # damage past the header: refused with a message, exit status 1
    vcf = vcfgen.gen_vcf(3, 3000, 100)
    data = bytearray(bgzf.bgzf_compress(vcf, block=4000))
    data[len(data) // 2] ^= 0x10
    p = _cli([], bytes(data))
    assert p.returncode == 1 and b"bgzf" in p.stderr
    p = _cli([], bgzf.bgzf_compress(vcf, block=4000)[:-100])
    assert p.returncode == 1 and b"bgzf" in p.stderr
    # the reference's regression input, BGZF-compressed: the golden rows
    vcf, want_sorted, hdr_line = golden_1kg
    p = _cli(["--batchMB", "16"], bgzf.bgzf_compress(vcf, level=1))
    assert p.returncode == 0
    rows = p.stdout.split(b"\n")
    assert rows[0] == hdr_line and sorted(rows[1:-1]) == want_sorted


def test_cli_bgzf_sites_only_rendered_rows_take_only_the_cut_lines_text_back(bv):
    """sites-only BGZF input with the rows rendered on the device (the default): the text stays on the device and only the
    lines left to the host -- here every tenth line, an insertion -- come back, packed (bvcf_row_cut.text_off); a file
    made of nothing but such lines outgrows that buffer and falls back to the whole text.  Rows and log against the
    oracle, for both forms and with the host's rows (BVCF_RENDER_SITES=0)."""
    import oracle_lib as orc
    hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    for every, n in ((10, 120_000), (1, 250_000)):
        rows = []
        for i in range(n):
            alt = "ATT" if i % every == 0 else "G"
            if i % 997 == 5:
                alt = "<DEL>"  # (a message in the log: its line is read from the cut's text too)
            rows.append("7\t%d\trs%d\tA\t%s\t50\tPASS\tAC=%d;AN=5008" % (100 + 3 * i, i, alt, i % 5000))
        vcf = (hdr + "\n".join(rows) + "\n").encode()
        rc_o, out_o, log_o, _ = orc.run(vcf, {"keepInfo": True})
        assert rc_o == 0
        data = bgzf.bgzf_compress(vcf, level=1)
        for render in ("1", "0"):
            p = _cli(["--keepInfo", "--batchMB", "16"], data, {"BVCF_RENDER_SITES": render})
            assert p.returncode == 0, p.stderr[-300:]
            assert p.stdout == (bv.string_header({"keepInfo": True}) + "\n").encode() + out_o, (every, render)
            assert p.stderr.decode() == log_o, (every, render)
