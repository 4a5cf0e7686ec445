"""Host-only tests of bvcf_run_fd's partition logic (include/bvcf_plan.h): which byte ranges there are, which lines a range
owns, where BGZF blocks start, what the per-device readers hand to their workers -- every line exactly once, in input
order, for any number of workers.  No device is involved: bvcf_plan_fd runs the product's own reader threads with heap
buffers and reports the blocks instead of submitting them.  Counterpart of the reference's single producer,
/root/reference/main.go:345-380 (whose unterminated last line is dropped, main.go:354-358)."""
import ctypes as C
import os
import random
import zlib

import pytest

import bgzf

HDR = b"##fileformat=VCFv4.2\n##source=test\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def make_body(rng, n_lines, lens, final_eol=True, eol=b"\n"):
    """data lines of the given length choices (bytes before the terminator), content that compresses badly"""
    out = []
    for i in range(n_lines):
        n = rng.choice(lens)
        head = b"1\t%d\t.\tA\tG\t.\tPASS\t.\tGT\t" % (i + 1)
        fill = bytes(rng.choices(b"ACGT0123456789|/.:;=", k=max(0, n - len(head))))
        out.append(head[:n] + fill + eol)
    body = b"".join(out)
    return body if final_eol else body[:-len(eol)]


def terminated(body, eol=b"\n"):
    return body[:body.rfind(eol) + 1]


# ---------------------------------------------------------------- pure functions

def ref_cut(window, own_len, first, last, eol=10):
    """the ownership rule, restated: range [a, b) owns (T(a), T(b)]"""
    n = len(window)
    s = 0
    if not first:
        t = window.find(bytes([eol]), 0, own_len)
        if t < 0:
            return ("none", 0, 0, 0)
        s = t + 1
    if last:
        t = window.rfind(bytes([eol]), s)
        return ("lines", s, t + 1 if t >= 0 else s, 0)
    t = window.find(bytes([eol]), own_len)
    if t >= 0:
        return ("lines", s, t + 1, 0)
    t = window.rfind(bytes([eol]), s, own_len)
    ls = t + 1 if t >= 0 else s
    return ("long", s, ls, ls)


def test_cut_text_range_matches_the_rule(bv):
    rng = random.Random(7)
    kinds = {bv.CUT_LINES: "lines", bv.CUT_NONE: "none", bv.CUT_LONG: "long"}
    for case in range(3000):
        n = rng.randrange(1, 200)
        p_eol = rng.choice([0.0, 0.02, 0.1, 0.5])
        w = bytes(10 if rng.random() < p_eol else 65 for _ in range(n))
        own = rng.randrange(0, n + 1)
        first, last = rng.random() < 0.3, rng.random() < 0.3
        cut = bv.TextCut()
        assert bv.lib.bvcf_cut_text_range(w, n, own, first, last, 10, C.byref(cut)) == 0
        want = ref_cut(w, own, first, last)
        got = (kinds[cut.kind], cut.start, cut.end, cut.long_start if cut.kind == bv.CUT_LONG else 0)
        if want[0] == "none":
            assert got[0] == "none", (case, w, own, first, last)
        else:
            assert got == want, (case, w, own, first, last)


def test_text_ranges_cover_the_body(bv):
    for size, off, cap, first_line in ((10**9, 12345, 0, 10168), (5000, 100, 4096, 80), (100, 100, 4096, 0), (70 << 20, 0, 64 << 20, 150),
                                       (63_298_516_895, 20078, 0, 10168)):
        p = bv.RangePlan()
        assert bv.lib.bvcf_plan_text_ranges(size, off, cap, first_line, C.byref(p)) == 0
        cap_eff = cap or (64 << 20)
        assert p.data_off == off and p.range_bytes + p.spare_bytes <= cap_eff and p.range_bytes > 0
        assert p.n_ranges * p.range_bytes >= size - off > (p.n_ranges - 1) * p.range_bytes if size > off else p.n_ranges == 0
        assert p.spare_bytes >= min(cap_eff - p.range_bytes, max(8 * first_line, 64 << 10))


def test_thread_budget_fits_the_quota(bv):
    for cpus in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        for n in (1, 2, 3, 4, 8):
            for mode in (bv.MODE_STREAM, bv.MODE_TEXT_RANGES, bv.MODE_BGZF_RANGES):
                b = bv.ThreadBudget()
                assert bv.lib.bvcf_plan_threads(cpus, n, mode, C.byref(b)) == 0
                assert b.copy_threads >= 1 and b.format_threads >= 1
                per_worker = (b.readers or 0) * b.copy_threads + b.format_threads
                busy = n * per_worker + (1 if mode == bv.MODE_STREAM else 0)
                if mode == bv.MODE_STREAM and b.copy_threads > 1:
                    busy += b.copy_threads  # (the copiers of a text pipe's pages, beside the one reader)
                assert busy == b.busy_total
                # everything that burns CPU fits the CPUs the process may use; with fewer than two per worker each
                # worker still gets one reader and one formatter
                assert busy <= max(cpus, 2 * n) + (1 if mode == bv.MODE_STREAM else 0), (cpus, n, mode, busy)
                if mode == bv.MODE_TEXT_RANGES and cpus >= 4 * n:
                    assert b.readers == 2
    # eight workers on the 16-core share of a one-GPU box (--devices 0,0,0,0,0,0,0,0): 16 busy threads, + 8 device threads
    b = bv.ThreadBudget()
    bv.lib.bvcf_plan_threads(16, 8, bv.MODE_TEXT_RANGES, C.byref(b))
    assert (b.readers, b.copy_threads, b.format_threads, b.busy_total) == (1, 1, 1, 16)
    bv.lib.bvcf_plan_threads(16, 1, bv.MODE_TEXT_RANGES, C.byref(b))
    assert (b.readers, b.copy_threads, b.format_threads, b.busy_total) == (2, 4, 8, 16)


def test_find_bgzf_chain(bv):
    rng = random.Random(3)
    blocks = [bgzf.bgzf_block(bytes(rng.choices(b"ACGT\t\n01|", k=rng.randrange(1, 3000))), 1) for _ in range(40)]
    data = b"".join(blocks)
    starts, p = [], 0
    for b in blocks:
        starts.append(p)
        p += len(b)
    for frm in list(range(0, 200)) + [rng.randrange(len(data)) for _ in range(300)]:
        want = next((s for s in starts if s >= frm and s + 18 <= len(data)), -1)
        got = bv.lib.bvcf_find_bgzf_chain(data, len(data), frm)
        assert got == want, (frm, got, want)
    # a payload that holds the magic bytes of a block header does not start a chain
    fake = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x10\x00" + b"x" * 40
    trap = bgzf.bgzf_block(b"", 1)[:0] + b"\x00" * 7 + fake
    assert bv.lib.bvcf_find_bgzf_chain(trap + data, len(trap) + len(data), 0) == len(trap)
    assert bv.lib.bvcf_find_bgzf_chain(b"no blocks here at all, only text\n" * 4, 132, 0) == -1


# ---------------------------------------------------------------- the readers themselves, without a device

def plan_file(bv, tmp_path, data, n_workers, max_batch, device_inflate=1, name="in.vcf"):
    path = tmp_path / name
    path.write_bytes(data)
    fd = os.open(str(path), os.O_RDONLY)
    err = os.open(str(tmp_path / "err.txt"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        rc, mode, plan, blocks = bv.plan_fd(fd, n_workers, max_batch, device_inflate, fd_err=err)
    finally:
        os.close(fd)
        os.close(err)
    return rc, mode, plan, blocks, (tmp_path / "err.txt").read_text()


def check_text_blocks(bv, data, blocks, n_workers, body_off, eol=b"\n", ranges=True):
    got = []
    last_key = (-1, -1)
    open_range = None
    for b in blocks:
        assert not b.bgzf
        key = (b.range, b.piece)
        assert key > last_key, "blocks in (range, piece) order, none twice"
        last_key = key
        if ranges:
            assert b.worker == b.range % n_workers
        # a range's pieces are consecutive from 0 and exactly one closes it
        if open_range is None:
            assert b.piece == 0
        else:
            assert b.range == open_range[0] and b.piece == open_range[1] + 1
        open_range = None if b.last_piece else (b.range, b.piece)
        seg = data[b.file_off:b.file_off + b.nbytes]
        assert len(seg) == b.nbytes
        if b.nbytes:
            assert seg.endswith(eol), "a block is whole lines"
            assert b.file_off == body_off or data[b.file_off - len(eol):b.file_off] == eol, "a block starts at a line start"
        got.append(seg)
    assert open_range is None
    want = terminated(data[body_off:], eol)
    assert b"".join(got) == want, "every terminated line exactly once, in input order"
    if ranges and blocks:
        assert sorted(set(b.range for b in blocks)) == list(range(blocks[-1].range + 1)), "every range takes its turn"


@pytest.mark.parametrize("n_workers", [1, 2, 3, 4, 5, 8])
def test_text_ranges_every_line_once(bv, tmp_path, n_workers):
    rng = random.Random(100 + n_workers)
    for case in range(10):
        max_batch = rng.choice([4096, 8192, 20000, 65536])
        lens = rng.choice([[30, 60, 120], [10, 11, 12, 400], [200, 2000], [50, 5000, 9000], [1, 2, 3]])
        body = make_body(rng, rng.randrange(1, 1500), lens, final_eol=rng.random() < 0.6)
        data = HDR + body
        rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, n_workers, max_batch)
        if any(ln > max_batch for ln in map(len, body.split(b"\n"))):
            assert rc == bv.E_TOO_BIG and "longer than max_batch_bytes" in err
            continue
        assert rc == 0, err
        assert mode == bv.MODE_TEXT_RANGES and plan.data_off == len(HDR)
        assert plan.n_ranges == -(-len(body) // plan.range_bytes)
        check_text_blocks(bv, data, blocks, n_workers, len(HDR))


def test_text_lines_longer_than_a_range_and_than_the_spare(bv, tmp_path):
    rng = random.Random(5)
    # range 2048 + spare 2048 at max_batch 4096: lines of 2.5 k outrun the spare, lines of 5 k+ cover whole ranges
    for lens, ok in (([2500, 40], True), ([3500, 3900, 10], True), ([4000], True), ([4097, 100], False)):
        body = make_body(rng, 120, lens)
        data = HDR + body
        for n_workers in (1, 2, 3, 8):
            rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, n_workers, 4096)
            if not ok:
                assert rc == bv.E_TOO_BIG and "a line is longer than max_batch_bytes" in err
                continue
            assert rc == 0, err
            assert plan.range_bytes == 2048 and plan.spare_bytes == 2048
            check_text_blocks(bv, data, blocks, n_workers, len(HDR))
            assert any(b.piece > 0 for b in blocks), "some straddling line went the slow way (a second piece)"
    # ranges inside one long line own nothing but still take their turn
    body = make_body(rng, 3, [60]) + make_body(rng, 1, [7000]) + make_body(rng, 3, [60])
    data = HDR + body
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 3, 16384)
    assert rc == 0, err
    check_text_blocks(bv, data, blocks, 3, len(HDR))


def test_text_edges(bv, tmp_path):
    rng = random.Random(9)
    # no data lines at all; one line; one line without terminator (dropped, main.go:354-358); CRLF
    for body, eol in ((b"", b"\n"), (b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\n", b"\n"), (b"1\t5\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1", b"\n")):
        data = HDR + body
        rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 2, 4096)
        if not body:
            # a file that ends with its header: no ranges, nothing to hand over (bvcf_run_fd prints the header line only)
            assert rc == 0 and not blocks, err
            continue
        assert rc == 0, err
        check_text_blocks(bv, data, blocks, 2, len(HDR), eol)
    hdr_crlf = HDR.replace(b"\n", b"\r\n")
    body = make_body(rng, 400, [40, 90], eol=b"\r\n")
    data = hdr_crlf + body
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 3, 4096)
    assert rc == 0, err
    check_text_blocks(bv, data, blocks, 3, len(hdr_crlf), b"\n")  # (the terminator BYTE is \n; \r stays with its line)
    # a range boundary exactly on / right after / right before a terminator
    p = bv.RangePlan()
    bv.lib.bvcf_plan_text_ranges(10**6, len(HDR), 4096, 64, C.byref(p))
    for shift in (-1, 0, 1):
        first = b"A" * (p.range_bytes - 1 + shift) + b"\n"
        data = HDR + first + make_body(rng, 200, [64])
        rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 2, 4096)
        assert rc == 0, err
        check_text_blocks(bv, data, blocks, 2, len(HDR))
    # fatal paths come out of the planner as out of the run
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, b"#CHROM\tPOS\n1\t2\n", 2, 4096)
    assert rc == bv.E_FATAL and "Not a VCF file" in err
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, b"##fileformat=VCFv4.2\n##x\n", 2, 4096)
    assert rc == bv.E_FATAL and "No header found" in err


def inflate_members(comp):
    out, pos = [], 0
    while pos < len(comp):
        d = zlib.decompressobj(31)
        out.append(d.decompress(comp[pos:]))
        assert d.eof
        pos = len(comp) - len(d.unused_data)
    return b"".join(out)


def batch_text(bv, data, b, eol=b"\n"):
    """what bvcf_submit_bgzf's rule (include/bvcf.h; k_cuts on the device) makes of a batch"""
    own_text = inflate_members(data[b.file_off:b.file_off + b.own])
    la_text = inflate_members(data[b.file_off + b.own:b.file_off + b.nbytes])
    text = own_text + la_text
    start = b.first_off
    if b.bgzf_flags & bv.BGZF_SKIP_FIRST_LINE:
        e = text.find(eol)
        start = len(text) if e < 0 else e + 1
    end = len(text)
    if la_text:
        e = text.find(eol, len(own_text))
        if e < 0:
            assert b.bgzf_flags & bv.BGZF_END_OF_STREAM, "a batch whose last line does not end in its look-ahead"
        else:
            end = e + 1
    start = min(start, end)
    return terminated(text[start:end], eol) if text[start:end].find(eol) >= 0 else b""


def check_bgzf_blocks(bv, data, text, blocks, n_workers, ranges=True):
    got = []
    last_key = (-1, -1)
    for b in blocks:
        key = (b.range, b.piece)
        assert key > last_key
        last_key = key
        if ranges:
            assert b.worker == b.range % n_workers
        if not b.nbytes:
            continue
        assert b.bgzf and 0 < b.own <= b.nbytes
        assert data[b.file_off:b.file_off + 4] == b"\x1f\x8b\x08\x04", "a batch starts at a block"
        got.append(batch_text(bv, data, b))
    body = text[text.index(b"#CHROM"):]
    body = body[body.index(b"\n") + 1:]
    assert b"".join(got) == terminated(body), "every terminated line in exactly one batch, in input order"


@pytest.mark.parametrize("n_workers", [1, 2, 3, 8])
def test_bgzf_ranges_every_line_once(bv, tmp_path, n_workers):
    rng = random.Random(40 + n_workers)
    # > 1 MiB of compressed bytes per range: random-ish content, level 1
    body = make_body(rng, 9000, [300, 700, 1500], final_eol=n_workers != 2)
    text = HDR + body
    for block in (0xFF00, 3000, 700):
        data = bgzf.bgzf_compress(text, block=block, level=1, eof_marker=block != 3000)
        for max_batch in (4 << 20, 1 << 20):
            rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, n_workers, max_batch, name="in.vcf.gz")
            assert rc == 0, err
            assert mode == bv.MODE_BGZF_RANGES and plan.range_bytes % 65536 == 0
            if block == 700:
                assert plan.n_ranges >= 3
            check_bgzf_blocks(bv, data, text, blocks, n_workers)
            assert blocks[-1].bgzf_flags & bv.BGZF_END_OF_STREAM


def test_bgzf_line_longer_than_a_batch_and_corrupt_input(bv, tmp_path):
    rng = random.Random(77)
    text = HDR + make_body(rng, 50, [100]) + make_body(rng, 1, [3 << 20]) + make_body(rng, 50, [100])
    data = bgzf.bgzf_compress(text, level=1)
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 2, 1 << 20, name="in.vcf.gz")
    assert rc == bv.E_TOO_BIG and "a line is longer than max_batch_bytes" in err
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data, 2, 8 << 20, name="in.vcf.gz")
    assert rc == 0, err
    check_bgzf_blocks(bv, data, text, blocks, 2)
    # a file cut inside a block
    text = HDR + make_body(rng, 6000, [700])
    data = bgzf.bgzf_compress(text, level=1)
    rc, mode, plan, blocks, err = plan_file(bv, tmp_path, data[:len(data) - 5000], 2, 4 << 20, name="in.vcf.gz")
    assert rc == bv.E_FATAL and "truncated" in err


def test_stream_mode_deals_blocks_round_robin(bv, tmp_path):
    """a pipe has no offsets to seek to: one reader cuts blocks of whole lines, block k goes to worker k % N"""
    rng = random.Random(11)
    body = make_body(rng, 3000, [40, 300], final_eol=False)
    data = HDR + body
    for n_workers in (1, 3):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            try:
                os.write(w, data) if len(data) < 60000 else [os.write(w, data[i:i + 30000]) for i in range(0, len(data), 30000)]
            finally:
                os._exit(0)
        os.close(w)
        try:
            rc, mode, plan, blocks = bv.plan_fd(r, n_workers, 16384, 1)
        finally:
            os.close(r)
            os.waitpid(pid, 0)
        assert rc == 0 and mode == bv.MODE_STREAM
        for k, b in enumerate(blocks):
            assert b.range == k and b.worker == k % n_workers and b.piece == 0 and b.last_piece
        check_text_blocks(bv, data, blocks, n_workers, len(HDR), ranges=False)
    # batches of 8 MiB: a text pipe's pages are then handed on to private pipes and copied out by several threads
    # (bvcf_input.cpp: read_fifo_fanout) -- the blocks must come out the same, and the same as with the plain read()
    big = HDR + make_body(rng, 60000, [300, 900])
    seen = {}
    for fanout in ("1", "0"):
        os.environ["BVCF_PIPE_FANOUT"] = fanout
        try:
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                try:
                    mv, off = memoryview(big), 0
                    while off < len(mv):
                        off += os.write(w, mv[off:off + 777_777])
                finally:
                    os._exit(0)
            os.close(w)
            try:
                rc, mode, plan, blocks = bv.plan_fd(r, 2, 8 << 20, 1)
            finally:
                os.close(r)
                os.waitpid(pid, 0)
        finally:
            os.environ.pop("BVCF_PIPE_FANOUT", None)
        assert rc == 0 and mode == bv.MODE_STREAM and len(blocks) >= 3
        check_text_blocks(bv, big, blocks, 2, len(HDR), ranges=False)
        seen[fanout] = [(b.range, b.worker, b.file_off, b.nbytes) for b in blocks]
    assert seen["1"] == seen["0"]
    # BGZF through a pipe: compressed batches, the same ownership rule
    text = HDR + make_body(rng, 4000, [300, 900])
    comp = bgzf.bgzf_compress(text, block=5000, level=1)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        try:
            for i in range(0, len(comp), 50000):
                os.write(w, comp[i:i + 50000])
        finally:
            os._exit(0)
    os.close(w)
    try:
        rc, mode, plan, blocks = bv.plan_fd(r, 2, 1 << 20, 1)
    finally:
        os.close(r)
        os.waitpid(pid, 0)
    assert rc == 0 and mode == bv.MODE_STREAM and len(blocks) > 2
    check_bgzf_blocks(bv, comp, text, blocks, 2, ranges=False)
