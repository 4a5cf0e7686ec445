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
