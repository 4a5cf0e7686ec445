"""BGZF writer for tests (SAM spec §4.1): independent gzip members of <= 64 KiB with a 'BC' extra field."""
import struct
import zlib


def bgzf_block(data, level=6):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = c.compress(data) + c.flush()
    bsize = len(payload) + 25  # header 18 + payload + crc 4 + isize 4 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + payload +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def bgzf_compress(data, block=0xFF00, level=6, eof_marker=True):
    out = []
    for i in range(0, len(data), block):
        out.append(bgzf_block(data[i:i + block], level))
    if eof_marker:
        out.append(bgzf_block(b""))
    return b"".join(out)
