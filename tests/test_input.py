"""The driver's byte source (bvcf_input.cpp) on the CPU: text, single-stream gzip, BGZF."""
import gzip
import os
import random

import pytest

import bgzf


@pytest.fixture(scope="module")
def bv():
    import bystro_vcf_amd as b
    return b


def payload(n, seed=1):
    rng = random.Random(seed)
    words = [b"0|0\t", b"0|1\t", b"1|1\t", b"chr1\t", b"PASS\t", b"AC=5;AF=0.1\t", b"\n"]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words) if rng.random() < 0.97 else bytes([rng.randrange(256)])
    return bytes(out[:n])


@pytest.mark.parametrize("n", [0, 1, 17, 65280, 65281, 1_000_003, 20_000_000])
def test_roundtrips(bv, n):
    data = payload(n)
    for name, comp in (("text", data), ("gzip", gzip.compress(data, 1)), ("bgzf", bgzf.bgzf_compress(data, level=1))):
        if n == 0 and name == "text":
            continue
        rc, out, kind = bv.decompress(comp)
        assert rc == 0 and out == data, (name, n)
        assert kind == (name if n or name != "text" else "text")


@pytest.mark.parametrize("n_threads", [2, 8, 16])
def test_text_through_a_pipe(bv, n_threads, monkeypatch):
    """stdin's shape: text from a pipe is handed on, page by page, to private pipes whose readers copy it out side by side
    (bvcf_input.cpp: read_fifo_fanout; 2 threads: the plain read(), 8: two copiers, 16: four) -- every byte once and in order
    whatever the sizes the producer writes, gzip and BGZF through the same pipe untouched by it"""
    for n, seed in ((0, 1), (1, 2), (5_000_000, 3), (40_000_003, 4)):
        data = payload(n, seed)
        if n:
            rc, out, kind = bv.decompress_pipe(data, n_threads, 0, seed)
            assert rc == 0 and out == data and kind == "text", (n, n_threads)
        for name, comp in (("gzip", gzip.compress(data, 1)), ("bgzf", bgzf.bgzf_compress(data, level=1))):
            rc, out, kind = bv.decompress_pipe(comp, n_threads, 0, seed)
            assert rc == 0 and out == data and kind == name, (name, n, n_threads)
    data = payload(30_000_000, 9)
    for piece in (1 << 20, 65536, 999_983):
        rc, out, kind = bv.decompress_pipe(data, n_threads, piece)
        assert rc == 0 and out == data
    monkeypatch.setenv("BVCF_PIPE_FANOUT", "0")
    rc, out, kind = bv.decompress_pipe(data, n_threads, 0, 5)
    assert rc == 0 and out == data


def test_bgzf_shapes(bv):
    data = payload(300_000, 2)
    # tiny blocks, no EOF marker, empty blocks in the middle, one thread and many
    blocks = [bgzf.bgzf_block(data[i:i + 1000]) for i in range(0, len(data), 1000)]
    blocks.insert(7, bgzf.bgzf_block(b""))
    blocks.insert(7, bgzf.bgzf_block(b""))
    for nt in (1, 3, 64):
        rc, out, kind = bv.decompress(b"".join(blocks), nt)
        assert rc == 0 and out == data and kind == "bgzf"
    # stored (incompressible) blocks
    noise = os.urandom(200_000)
    rc, out, _ = bv.decompress(bgzf.bgzf_compress(noise, block=60000))
    assert rc == 0 and out == noise


def test_gzip_members_and_padding(bv):
    a, b = payload(70_000, 3), payload(90_000, 4)
    rc, out, kind = bv.decompress(gzip.compress(a) + gzip.compress(b) + b"\0" * 512)
    assert rc == 0 and out == a + b and kind == "gzip"


def test_damage_is_reported(bv):
    data = payload(500_000, 5)
    good = bgzf.bgzf_compress(data)
    assert bv.decompress(good[:-30])[0] != 0                       # truncated block
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x55
    assert bv.decompress(bytes(bad))[0] != 0                       # payload or CRC damage
    assert bv.decompress(good[:5000] + gzip.compress(b"x") )[0] != 0  # foreign member inside BGZF
    g = bytearray(gzip.compress(data, 1))
    g[len(g) // 3] ^= 0xFF
    assert bv.decompress(bytes(g))[0] != 0
    assert bv.decompress(gzip.compress(data)[:-9])[0] != 0         # gzip cut short
