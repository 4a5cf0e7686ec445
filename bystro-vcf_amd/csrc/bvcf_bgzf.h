// bvcf_bgzf.h — BGZF framing on the host (SAM spec 4.1): finds the blocks of a buffer, for the device inflate.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace bvcf_bgzf {

struct Block {
  uint32_t in_off, in_len;  // deflate payload inside the buffer
  uint32_t isize, crc;      // trailer: bytes of text, their CRC-32
  uint32_t total;           // bytes of the whole member (BSIZE + 1)
};

// a complete BGZF block at p[0..n)?  Returns its total size (BSIZE + 1), 0 if more bytes are needed, -1 if p does not
// start a BGZF block (gzip member with FEXTRA subfield 'B','C')
inline long block_size(const uint8_t *p, size_t n, uint32_t *xlen_out) {
  if (n < 18) return 0;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
  const uint32_t xlen = p[10] | (p[11] << 8);
  if (n < 12 + (size_t)xlen) return 0;
  size_t q = 12;
  const size_t xend = 12 + xlen;
  while (q + 4 <= xend) {
    const uint32_t slen = p[q + 2] | (p[q + 3] << 8);
    if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= xend) {
      *xlen_out = xlen;
      return (long)(p[q + 4] | (p[q + 5] << 8)) + 1;
    }
    q += 4 + slen;
  }
  return -1;
}

// the whole blocks at the start of buf[0..n): appended to `out`; returns the bytes they occupy, or -1 on a malformed
// block.  Stops at the first incomplete block.
inline long scan(const uint8_t *buf, size_t n, std::vector<Block> *out) {
  size_t pos = 0;
  while (pos < n) {
    uint32_t xlen = 0;
    const long bs = block_size(buf + pos, n - pos, &xlen);
    if (bs < 0) return -1;
    if (bs == 0 || pos + (size_t)bs > n) break;
    if ((size_t)bs < 12 + (size_t)xlen + 8) return -1;
    const uint8_t *tail = buf + pos + bs - 8;
    Block b;
    b.in_off = (uint32_t)(pos + 12 + xlen);
    b.in_len = (uint32_t)(bs - 12 - xlen - 8);
    b.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    b.total = (uint32_t)bs;
    if (b.isize > (1u << 16)) return -1;
    out->push_back(b);
    pos += (size_t)bs;
  }
  return (long)pos;
}

}  // namespace bvcf_bgzf
