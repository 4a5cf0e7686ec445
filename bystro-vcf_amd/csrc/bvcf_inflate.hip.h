// bvcf_inflate.hip.h — BGZF blocks inflated on the device (device half)
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// The reference leaves decompression to `pigz -d -c` in front of the pipe, and names that single-threaded inflate as
// the limit of its headline run (README.md:5,46).  A .vcf.gz written by bgzip / htslib is BGZF: independent gzip
// members of at most 64 KiB of text each.  k_inflate takes ONE WAVE PER BGZF BLOCK: the compressed bytes cross PCIe
// (a sixth of the text for genotype VCFs), and the text is produced where the scan kernels read it.
//
// DEFLATE (RFC 1951) is a serial bit stream, so the wave decodes it with wave-uniform state -- every lane holds the
// same bit buffer and positions, which costs what one lane would -- and uses its 64 lanes where the format allows:
// building the Huffman tables, match copies (up to 258 bytes each; genotype text is mostly matches), stored blocks
// and the write-out.  Per wave in LDS: a 32 KiB window of the output (the most a match may reach back), written to
// memory in 16 KiB segments as it fills; a 2 KiB ring of compressed input; a 10-bit literal/length table and a 9-bit
// distance table with canonical-code fallbacks for longer codes.
#pragma once

#include "bvcf_common.hip.h"

namespace bvcf_dev {

struct BgzfDesc {
  uint32_t in_off;    // deflate payload in the compressed buffer
  uint32_t in_len;
  uint32_t out_off;   // where the block's text goes
  uint32_t isize;     // bytes it must inflate to (BGZF trailer)
};

enum {
  kInfOk = 0,
  kInfBadBlockType = 1,
  kInfBadStored = 2,
  kInfBadCodeLengths = 3,
  kInfBadSymbol = 4,
  kInfBadDistance = 5,
  kInfOutputOverrun = 6,
  kInfInputOverrun = 7,
  kInfSizeMismatch = 8,
};

constexpr uint32_t kInfWindowFull = 32768;  // the most a DEFLATE match may reach back
// LDS per wave decides how many blocks a CU decodes side by side, and the decoder is bound by the latency of one wave's
// dependent steps: a 1 KiB input ring filled 512 bytes at a time, a 9-bit literal/length table and an 8-bit distance
// table (longer codes -- under 1 % of the symbols of genotype text at zlib's levels 1..9 -- take the canonical walk).
#ifndef BVCF_INF_RING
#define BVCF_INF_RING 1024
#endif
#ifndef BVCF_INF_LIT_BITS
#define BVCF_INF_LIT_BITS 9
#endif
#ifndef BVCF_INF_DIST_BITS
#define BVCF_INF_DIST_BITS 8
#endif
constexpr uint32_t kInfInRing = BVCF_INF_RING;
constexpr uint32_t kInfFetch = kInfInRing / 2;   // compressed bytes per fetch: 8 per lane
constexpr uint32_t kInfLitBits = BVCF_INF_LIT_BITS, kInfDistBits = BVCF_INF_DIST_BITS;
static_assert(kInfFetch == 8u * kWave || kInfFetch == 16u * kWave, "a fetch is one 8- or 16-byte load per lane");
static_assert(kInfDistBits >= 7, "the code-length code's 7-bit table lives in the distance table's storage");
constexpr int kInfThreads = kWave;  // one wave per workgroup

template <uint32_t kInfWindow>
struct InfLds {
  uint8_t win[kInfWindow];                 // output byte p lives at win[p & (kInfWindow - 1)]
  uint8_t in[kInfInRing];                  // compressed byte q lives at in[q & (kInfInRing - 1)]
  uint16_t lit[1u << kInfLitBits];         // code (bit-reversed, low bits) -> len << 9 | symbol; 0 = longer than the table
  uint16_t dist[1u << kInfDistBits];       // the same for distance codes: len << 5 | symbol
  uint16_t lit_sorted[288], dist_sorted[32];  // symbols in canonical order, for codes longer than the tables
  uint16_t lit_count[16], dist_count[16];  // codes per length
  uint8_t lens[384];                       // code lengths being read: [0,19) the code-length code, [32,..) literal/length then distance
};

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t n) { return __brev(v) >> (32u - n); }

// Huffman tables from code lengths lens[0 .. n) (RFC 1951 3.2.2).  table: primary lookup of `bits` bits; sorted /
// count: canonical order for longer codes.  sym_shift: where the length goes in a table entry.  Returns false if the
// lengths over-subscribe the code space.
__device__ inline bool inf_build(const uint8_t *lens, uint32_t n, uint16_t *table, uint32_t bits, uint16_t *sorted,
                                 uint16_t *count, uint32_t sym_shift) {
  const int lane = lane_id();
  // codes per length
  for (uint32_t L = lane; L < 16; L += kWave) count[L] = 0;
  for (uint32_t i = lane; i < (1u << bits); i += kWave) table[i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t cnt[16];
#pragma unroll
  for (uint32_t L = 0; L < 16; L++) cnt[L] = 0;
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t s = base + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
#pragma unroll
    for (uint32_t L = 1; L < 16; L++) cnt[L] += (uint32_t)__popcll(__ballot(l == L));
  }
  // first code of each length; over-subscription check
  uint32_t next[16], offs[16];
  uint32_t code = 0, left = 1, off = 0;
  bool ok = true;
#pragma unroll
  for (uint32_t L = 1; L < 16; L++) {
    left <<= 1;
    if (cnt[L] > left) ok = false;
    left -= min(cnt[L], left);
    code = (code + cnt[L - 1]) << 1;
    next[L] = code;
    offs[L] = off;
    off += cnt[L];
  }
  if (lane < 16) count[lane] = (uint16_t)(lane ? cnt[lane] : 0u);
  if (!ok) return false;
  // every symbol: its code = first code of its length + its rank among the symbols of that length
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t s = base + lane;
    const uint32_t l = s < n ? lens[s] : 0u;
    uint32_t my_code = 0, my_rank = 0;
#pragma unroll
    for (uint32_t L = 1; L < 16; L++) {
      const unsigned long long m = __ballot(l == L);
      if (l == L) {
        my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        my_code = next[L] + my_rank;
        sorted[offs[L] + my_rank] = (uint16_t)s;
      }
      const uint32_t c = (uint32_t)__popcll(m);
      next[L] += c;
      offs[L] += c;
    }
    if (l && l <= bits) {
      const uint16_t e = (uint16_t)((l << sym_shift) | s);
      for (uint32_t i = bitrev(my_code, l); i < (1u << bits); i += 1u << l) table[i] = e;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return true;
}

// length / distance bases and extra bits (RFC 1951 3.2.5), closed forms.  (Round 5 tried tables in constant memory -- the
// symbol is wave-uniform, so a lookup is ONE scalar load instead of 8-12 scalar instructions and two branches --:
// k_inflate_w4 1 600 -> 1 710 us per 3 840 blocks, the load's latency costs more than the instructions it saves;
// -DBVCF_INF_TABLES builds that form, profiles/r05_k_inflate_attempts.txt)
__constant__ uint16_t kInfLenTab[32] = {  // base | extra << 12, codes 257..285
    3,           4,           5,           6,           7,           8,           9,           10,
    11 | 1 << 12, 13 | 1 << 12, 15 | 1 << 12, 17 | 1 << 12, 19 | 2 << 12, 23 | 2 << 12, 27 | 2 << 12, 31 | 2 << 12,
    35 | 3 << 12, 43 | 3 << 12, 51 | 3 << 12, 59 | 3 << 12, 67 | 4 << 12, 83 | 4 << 12, 99 | 4 << 12, 115 | 4 << 12,
    131 | 5 << 12, 163 | 5 << 12, 195 | 5 << 12, 227 | 5 << 12, 258,         0,           0,           0};
__constant__ uint32_t kInfDistTab[32] = {  // base | extra << 16, codes 0..29
    1,               2,               3,               4,               5 | 1 << 16,     7 | 1 << 16,     9 | 2 << 16,     13 | 2 << 16,
    17 | 3 << 16,    25 | 3 << 16,    33 | 4 << 16,    49 | 4 << 16,    65 | 5 << 16,    97 | 5 << 16,    129 | 6 << 16,   193 | 6 << 16,
    257 | 7 << 16,   385 | 7 << 16,   513 | 8 << 16,   769 | 8 << 16,   1025 | 9 << 16,  1537 | 9 << 16,  2049 | 10 << 16, 3073 | 10 << 16,
    4097 | 11 << 16, 6145 | 11 << 16, 8193 | 12 << 16, 12289 | 12 << 16, 16385 | 13 << 16, 24577 | 13 << 16, 0,               0};
__device__ __forceinline__ uint32_t inf_len_base(uint32_t i, uint32_t *extra) {
  // codes 257..285 -> i = 0..28
#ifndef BVCF_INF_TABLES
  if (i < 8) {
    *extra = 0;
    return 3 + i;
  }
  if (i == 28) {
    *extra = 0;
    return 258;
  }
  const uint32_t e = (i - 4) >> 2;
  *extra = e;
  return 3 + ((4 + (i & 3)) << e);
#else
  const uint32_t e = kInfLenTab[i & 31u];
  *extra = e >> 12;
  return e & 0xFFFu;
#endif
}
__device__ __forceinline__ uint32_t inf_dist_base(uint32_t i, uint32_t *extra) {
#ifndef BVCF_INF_TABLES
  if (i < 4) {
    *extra = 0;
    return 1 + i;
  }
  const uint32_t e = (i - 2) >> 1;
  *extra = e;
  return 1 + ((2 + (i & 1)) << e);
#else
  const uint32_t e = kInfDistTab[i & 31u];
  *extra = e >> 16;
  return e & 0xFFFFu;
#endif
}

// one wave per BGZF block; status[k] = kInf*
// kInfWindow: bytes of output kept in LDS.  32 KiB holds everything a match may refer to (39 KB of LDS: four waves per
// CU, i.e. one batch of ~1 000 blocks fills the chip).  16 KiB (23 KB: seven waves per CU, so two batches inflate side
// by side) is for files whose lines are well under 16 KB -- a genotype line mostly repeats the line before it --
// and reads the few older bytes a match asks for back from memory, where the finished segments already are.
template <uint32_t kInfWindow>
__device__ __forceinline__ void k_inflate_body(const uint8_t *comp, const BgzfDesc *desc, uint32_t n_blocks, uint8_t *out,
                                               uint32_t *status, InfLds<kInfWindow> &S) {
  constexpr uint32_t kInfSegment = kInfWindow / 2;
  const int lane = lane_id();
  for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const BgzfDesc d = desc[blk];
    const uint8_t *src = comp + d.in_off;
    uint8_t *dst = out + d.out_off;
    // wave-uniform decoder state
    uint32_t in_fetched = 0;   // compressed bytes in the ring so far
    uint32_t in_pos = 0;       // next byte the bit buffer takes
    unsigned long long bb = 0; // bit buffer, LSB first
    uint32_t nb = 0;           // bits in it
    uint32_t pos = 0;          // output bytes produced
    uint32_t flushed = 0;      // output bytes written to memory
    uint32_t err = kInfOk;

    // compressed bytes [in_fetched, in_fetched + kInfFetch) into the ring (all lanes; bytes past the payload are zeros)
    auto fetch = [&]() {
      constexpr uint32_t kPer = kInfFetch / kWave;  // 8 or 16 bytes per lane
      const uint32_t q = in_fetched + kPer * lane;
      if (kPer == 16u) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q + 16u <= d.in_len) v = *reinterpret_cast<const u32x4_u *>(src + q);
        *reinterpret_cast<u32x4 *>(&S.in[q & (kInfInRing - 1u)]) = v;
      } else {
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef u32x2_t u32x2_unaligned __attribute__((aligned(1)));
        u32x2_t v = {0u, 0u};
        if (q + 8u <= d.in_len) v = *reinterpret_cast<const u32x2_unaligned *>(src + q);
        *reinterpret_cast<u32x2_t *>(&S.in[q & (kInfInRing - 1u)]) = v;
      }
      if (q < d.in_len && q + kPer > d.in_len)  // the payload's last, partial piece: byte by byte over the zeros
        for (uint32_t i = 0; i < d.in_len - q; i++) S.in[(q + i) & (kInfInRing - 1u)] = src[q + i];
      in_fetched += kInfFetch;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
#ifdef BVCF_INF_REFILL64
    // (experiment, round 5) at least 56 bits after every refill -- whole bytes up to bit 63, as libdeflate does --, so that a
    // length + distance pair (15 + 5 + 15 + 13 bits) needs ONE refill test, at the symbol's start.  The bits of `w` that
    // land above position nb + 8 k are the stream's next bits: or-ing them in again at the next refill changes nothing.
    auto refill = [&]() {
      if (nb >= 48u) return;
      if (in_pos > d.in_len + 16u) {  // reading far past the payload: a truncated or corrupt stream
        err = kInfInputOverrun;
        return;
      }
      if (in_pos + 12u > in_fetched) fetch();  // (unread bytes stay below half the ring)
      const uint32_t a = in_pos & (kInfInRing - 1u);
      const uint32_t w0 = *reinterpret_cast<const uint32_t *>(&S.in[a & ~3u]);
      const uint32_t w1 = *reinterpret_cast<const uint32_t *>(&S.in[(a + 4u) & (kInfInRing - 1u) & ~3u]);
      const uint32_t w2 = *reinterpret_cast<const uint32_t *>(&S.in[(a + 8u) & (kInfInRing - 1u) & ~3u]);
      const uint32_t lo = bcast0(__builtin_amdgcn_alignbyte(w1, w0, a & 3u));
      const uint32_t hi = bcast0(__builtin_amdgcn_alignbyte(w2, w1, a & 3u));
      const unsigned long long w = (unsigned long long)lo | ((unsigned long long)hi << 32);
      const uint32_t k = (63u - nb) >> 3;  // whole bytes that fit
      bb |= w << nb;
      nb += 8u * k;
      in_pos += k;
    };
#else
    // keep >= 32 bits in the buffer (a symbol needs at most 15 + 13 extra)
    auto refill = [&]() {
      while (nb <= 32u) {
        if (in_pos > d.in_len + 16u) {  // reading far past the payload: a truncated or corrupt stream
          err = kInfInputOverrun;
          break;
        }
        if (in_pos + 8u > in_fetched) fetch();  // (unread bytes stay below half the ring)
        const uint32_t a = in_pos & (kInfInRing - 1u);
        // four bytes at any alignment, across the ring's end
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(&S.in[a & ~3u]);
        const uint32_t w1 = *reinterpret_cast<const uint32_t *>(&S.in[(a + 4u) & (kInfInRing - 1u) & ~3u]);
        // (every lane reads the same word; saying so keeps the whole decoder state -- bit buffer, positions, symbols --
        // in scalar registers and its branches scalar: a value that comes out of LDS is a vector value to the compiler)
        const uint32_t w = bcast0(__builtin_amdgcn_alignbyte(w1, w0, a & 3u));
        bb |= (unsigned long long)w << nb;
        nb += 32u;
        in_pos += 4u;
      }
    };
#endif
    auto take = [&](uint32_t n) -> uint32_t {
      const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
      bb >>= n;
      nb -= n;
      return v;
    };
    // write out the finished 16 KiB segments
    auto flush_segments = [&](bool all) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      while (flushed + kInfSegment <= pos || (all && flushed < pos)) {
        const uint32_t n = min(kInfSegment, pos - flushed);
        for (uint32_t i = 16u * lane; i < n; i += 16u * kWave) {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(&S.win[(flushed + i) & (kInfWindow - 1u)]);
          if (i + 16u <= n) {
            *reinterpret_cast<u32x4_u *>(dst + flushed + i) = v;
          } else {
            for (uint32_t j = 0; j < n - i; j++) dst[flushed + i + j] = S.win[(flushed + i + j) & (kInfWindow - 1u)];
          }
        }
        flushed += n;
      }
      __builtin_amdgcn_wave_barrier();
    };

    fetch();
    bool last = false;
    while (!last && err == kInfOk) {
      refill();
      last = take(1) != 0;
      const uint32_t type = take(2);
      if (type == 0) {
        // ---- stored: skip to the byte boundary, LEN / NLEN, then LEN bytes
        take(nb & 7u);
        refill();
        const uint32_t len = take(16), nlen = take(16);
        if ((len ^ nlen) != 0xFFFFu) {
          err = kInfBadStored;
          break;
        }
        // the bytes still in the bit buffer come first
        uint32_t q = in_pos - nb / 8u;  // payload offset of the next unread byte
        if (q + len > d.in_len) {
          err = kInfInputOverrun;
          break;
        }
        if (pos + len > d.isize) {
          err = kInfOutputOverrun;
          break;
        }
        for (uint32_t done = 0; done < len;) {
          // in pieces that keep a segment's worth of window
          const uint32_t n = min(len - done, kInfSegment - ((pos + 0u) & (kInfSegment - 1u)));
          for (uint32_t i = lane; i < n; i += kWave) S.win[(pos + i) & (kInfWindow - 1u)] = src[q + done + i];
          pos += n;
          done += n;
          flush_segments(false);
        }
        // restart the input after the stored bytes
        in_pos = q + len;
        bb = 0;
        nb = 0;
        in_fetched = in_pos & ~(kInfFetch - 1u);
        fetch();
        continue;
      }
      if (type == 3) {
        err = kInfBadBlockType;
        break;
      }
      uint32_t n_lit = 288, n_dist = 30;
      if (type == 1) {
        // ---- fixed codes (RFC 1951 3.2.6)
        for (uint32_t i = lane; i < 288; i += kWave) S.lens[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
        for (uint32_t i = lane; i < 32; i += kWave) S.lens[288 + i] = 5;
        n_dist = 32;
      } else {
        // ---- dynamic codes: HLIT, HDIST, HCLEN, the code-length code, then the lengths (serial: run-length coded)
        refill();
        n_lit = take(5) + 257u;
        n_dist = take(5) + 1u;
        const uint32_t n_clen = take(4) + 4u;
        if (n_lit > 286u || n_dist > 30u) {
          err = kInfBadCodeLengths;
          break;
        }
        // code lengths of the code-length alphabet, in the order of RFC 1951 3.2.7
        if (lane < 19) S.lens[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = 0; i < n_clen; i++) {
          refill();
          const uint32_t v = take(3);
          // 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
          const uint32_t ord = i < 3 ? 16u + i : (i == 3 ? 0u : ((i & 1u) ? 8u - ((i - 3u) >> 1) : 7u + ((i - 2u) >> 1)));
          if (lane == 0) S.lens[ord] = (uint8_t)v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // a 7-bit table for the 19 code-length symbols (in the distance table's storage)
        if (!inf_build(S.lens, 19, S.dist, 7, S.dist_sorted, S.dist_count, 5)) {
          err = kInfBadCodeLengths;
          break;
        }
        uint32_t got = 0, prev = 0;
        const uint32_t want = n_lit + n_dist;
        while (got < want && err == kInfOk) {
          refill();
          const uint32_t e = bcast0(S.dist[(uint32_t)bb & 127u]);
          const uint32_t l = e >> 5, sym = e & 31u;
          if (l == 0) {
            err = kInfBadCodeLengths;
            break;
          }
          take(l);
          uint32_t rep = 1, val = sym;
          if (sym == 16) {
            if (got == 0) {
              err = kInfBadCodeLengths;
              break;
            }
            rep = 3 + take(2);
            val = prev;
          } else if (sym == 17) {
            rep = 3 + take(3);
            val = 0;
          } else if (sym == 18) {
            rep = 11 + take(7);
            val = 0;
          }
          if (got + rep > want) {
            err = kInfBadCodeLengths;
            break;
          }
          // (lens[] for the real alphabets start at 32: the first 19 entries are the code-length code's)
          for (uint32_t i = lane; i < rep; i += kWave) S.lens[32 + got + i] = (uint8_t)val;
          got += rep;
          prev = val;
        }
        if (err != kInfOk) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      const uint8_t *lit_lens = type == 1 ? S.lens : S.lens + 32;
      const uint8_t *dist_lens = type == 1 ? S.lens + 288 : S.lens + 32 + n_lit;
      if (type == 2 && bcast0(lit_lens[256]) == 0) {
        err = kInfBadCodeLengths;  // no end-of-block code
        break;
      }
      if (!inf_build(lit_lens, n_lit, S.lit, kInfLitBits, S.lit_sorted, S.lit_count, 9) ||
          !inf_build(dist_lens, n_dist, S.dist, kInfDistBits, S.dist_sorted, S.dist_count, 5)) {
        err = kInfBadCodeLengths;
        break;
      }
      // canonical decode of a code longer than the primary table: returns the symbol, consumes the bits
      auto slow_decode = [&](const uint16_t *count, const uint16_t *sorted) -> uint32_t {
        uint32_t code = 0, first = 0, index = 0;
        unsigned long long b = bb;
        for (uint32_t l = 1; l <= 15; l++) {
          code |= (uint32_t)b & 1u;
          b >>= 1;
          const uint32_t c = bcast0(count[l]);
          if (code < first + c) {
            take(l);
            return bcast0(sorted[index + (code - first)]);
          }
          index += c;
          first += c;
          first <<= 1;
          code <<= 1;
        }
        return 0xFFFFu;
      };
      // ---- the symbols of this block
      for (;;) {
        refill();
        if (err != kInfOk) break;
        uint32_t e = bcast0(S.lit[(uint32_t)bb & ((1u << kInfLitBits) - 1u)]);
        uint32_t sym;
        if (e) {
          take(e >> 9);
          sym = e & 511u;
        } else {
          sym = slow_decode(S.lit_count, S.lit_sorted);
          if (sym == 0xFFFFu) {
            err = kInfBadSymbol;
            break;
          }
        }
        if (sym < 256u) {
          if (pos >= d.isize) {
            err = kInfOutputOverrun;
            break;
          }
          S.win[pos & (kInfWindow - 1u)] = (uint8_t)sym;  // (every lane the same byte to the same place: no exec mask to set up)
          pos++;
          if ((pos & (kInfSegment - 1u)) == 0u) flush_segments(false);
          continue;
        }
        if (sym == 256u) break;
        if (sym > 285u) {
          err = kInfBadSymbol;
          break;
        }
        uint32_t extra;
        uint32_t len = inf_len_base(sym - 257u, &extra);
        len += take(extra);
#ifndef BVCF_INF_REFILL64
        refill();
#endif
        e = bcast0(S.dist[(uint32_t)bb & ((1u << kInfDistBits) - 1u)]);
        uint32_t dsym;
        if (e) {
          take(e >> 5);
          dsym = e & 31u;
        } else {
          dsym = slow_decode(S.dist_count, S.dist_sorted);
        }
        if (dsym > 29u) {
          err = kInfBadDistance;
          break;
        }
        uint32_t dist = inf_dist_base(dsym, &extra);
        dist += take(extra);
        if (dist > pos) {
          err = kInfBadDistance;
          break;
        }
        if (pos + len > d.isize) {
          err = kInfOutputOverrun;
          break;
        }
        // ---- the match, by all lanes, four bytes per lane and step: byte j comes from the `dist` bytes before pos,
        // repeated.  (LDS operations of one wave execute in order: the bytes earlier symbols wrote are there.)
        __builtin_amdgcn_wave_barrier();
        {
          const uint32_t from = pos - dist;
          const bool periodic = dist < len;
          const float inv = 1.0f / (float)dist;
          // a source byte older than this is no longer in the ring (this match may overwrite it): it is in memory
          const bool far = kInfWindow < kInfWindowFull && dist + len > kInfWindow;
          if (far) __builtin_amdgcn_s_waitcnt(0);  // the segment stores that carry those bytes have landed
          auto src_byte = [&](uint32_t p) -> uint32_t {
            if (kInfWindow < kInfWindowFull && far && p + kInfWindow < pos + len) {
              // (cache-bypassing: this CU's L1 may hold an older copy of the line from before its segment was written)
              const uint32_t *wp = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(dst + p) & ~(uintptr_t)3);
              const uint32_t w = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              return (w >> (8u * (uint32_t)(reinterpret_cast<uintptr_t>(dst + p) & 3u))) & 0xFFu;
            }
            return S.win[p & (kInfWindow - 1u)];
          };
          // four bytes of the window at any alignment: two aligned words of the ring
          auto load4u = [&](uint32_t p) -> uint32_t {
            const uint32_t a = p & (kInfWindow - 1u);
            const uint32_t w0 = *reinterpret_cast<const uint32_t *>(&S.win[a & ~3u]);
            const uint32_t w1 = *reinterpret_cast<const uint32_t *>(&S.win[(a + 4u) & (kInfWindow - 1u) & ~3u]);
            return __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
          };
          // ... of the match's source: from the ring, or -- small windows, a match that reaches further back than the ring
          // will hold once it is written -- from memory, where the finished segments are (a match that overlaps its own
          // output is never that far back)
          auto src4 = [&](uint32_t p) -> uint32_t {
            if (kInfWindow < kInfWindowFull && far) {
              if (p + 3u + kInfWindow < pos + len) {
                // (cache-bypassing: this CU's L1 may hold an older copy of the line from before its segment was written)
                const uintptr_t g = reinterpret_cast<uintptr_t>(dst + p);
                const uint32_t *wp = reinterpret_cast<const uint32_t *>(g & ~(uintptr_t)3);
                const uint32_t w0 = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t w1 = __hip_atomic_load(wp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(g & 3u));
              }
              if (p + kInfWindow < pos + len)  // the piece that straddles the ring's oldest byte
                return src_byte(p) | (src_byte(p + 1u) << 8) | (src_byte(p + 2u) << 16) | (src_byte(p + 3u) << 24);
            }
            return load4u(p);
          };
          if (!far && len <= (uint32_t)kWave) {
            // a short match (most of them, between the long runs): one byte per lane
            if ((uint32_t)lane < len) {
              const uint32_t k = periodic ? (uint32_t)lane - dist * (uint32_t)(((float)lane + 0.5f) * inv) : (uint32_t)lane;
              S.win[(pos + lane) & (kInfWindow - 1u)] = S.win[(from + k) & (kInfWindow - 1u)];
            }
          } else {
            // A lane writes one ALIGNED dword of the destination per step (whole dwords with one store; only the match's
            // first and last dword byte by byte).  Genotype text is mostly matches that overlap their own output --
            // "0|0<TAB>" 64 times over is distance 4, length 258 -- i.e. a period of `dist` bytes: the dword at match
            // offset j is the period rotated by j mod dist, read with one or two unaligned loads (three bytes or fewer:
            // from a 64-bit repetition of the period).
            const uint32_t a0 = pos & 3u;
            const uint32_t n_dw = (a0 + len + 3u) >> 2;
            unsigned long long rep = 0;
            if (periodic && dist < 4u) {
              const unsigned long long p = bcast0(load4u(from)) & ((1u << (8u * dist)) - 1u);
              rep = dist == 1u ? p * 0x0101010101010101ull : (dist == 2u ? p * 0x0001000100010001ull : (p | (p << 24) | (p << 48)));
            }
            for (uint32_t base = 0; base < n_dw; base += kWave) {
              const uint32_t dwi = base + lane;
              if (dwi < n_dw) {
                const int j0 = (int)(4u * dwi) - (int)a0;  // match offset of the dword's first byte (-3..-1 for the first dword)
                const uint32_t js = j0 < 0 ? 0u : (uint32_t)j0;
                uint32_t w;
                if (!periodic) {
                  w = src4(from + js);
                } else {
                  // js mod dist for js < 262, dist < 258 from a float reciprocal ((j + 0.5) / dist is never within 0.002
                  // of an integer, the product's error stays below 0.0004)
                  const uint32_t k = js - dist * (uint32_t)(((float)js + 0.5f) * inv);
                  if (dist < 4u) {
                    w = (uint32_t)(rep >> (8u * k));
                  } else {
                    const uint32_t n1 = dist - k;  // bytes to the end of the period
                    w = load4u(from + k);
                    if (n1 < 4u) w = (w & ((1u << (8u * n1)) - 1u)) | (load4u(from) << (8u * n1));
                  }
                }
                if (j0 < 0) w <<= 8u * (uint32_t)(-j0);  // (byte q of w belongs to match offset j0 + q)
                const uint32_t o = ((pos & ~3u) + 4u * dwi) & (kInfWindow - 1u);
                if (j0 >= 0 && (uint32_t)j0 + 4u <= len) {
                  *reinterpret_cast<uint32_t *>(&S.win[o]) = w;
                } else {
#pragma unroll
                  for (uint32_t q = 0; q < 4; q++)
                    if (j0 + (int)q >= 0 && (uint32_t)(j0 + (int)q) < len) S.win[o + q] = (uint8_t)(w >> (8u * q));
                }
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t before = pos;
        pos += len;
        if ((before ^ pos) & ~(kInfSegment - 1u)) flush_segments(false);
      }
    }
    if (err == kInfOk) {
      flush_segments(true);
      if (pos != d.isize) err = kInfSizeMismatch;
    }
    if (lane == 0) status[blk] = err;
    __builtin_amdgcn_wave_barrier();
  }
}


__global__ __launch_bounds__(kInfThreads) void k_inflate(const uint8_t *comp, const BgzfDesc *desc, uint32_t n_blocks,
                                                         uint8_t *out, uint32_t *status) {
  __shared__ __attribute__((aligned(16))) InfLds<kInfWindowFull> S;
  k_inflate_body<kInfWindowFull>(comp, desc, n_blocks, out, status, S);
}
// the 16 KiB-window variant (see k_inflate_body)
__global__ __launch_bounds__(kInfThreads) void k_inflate_w16(const uint8_t *comp, const BgzfDesc *desc, uint32_t n_blocks,
                                                             uint8_t *out, uint32_t *status) {
  __shared__ __attribute__((aligned(16))) InfLds<16384> S;
  k_inflate_body<16384>(comp, desc, n_blocks, out, status, S);
}

// the 4 KiB-window variant: fifteen waves per CU
__global__ __launch_bounds__(kInfThreads) void k_inflate_w4(const uint8_t *comp, const BgzfDesc *desc, uint32_t n_blocks,
                                                            uint8_t *out, uint32_t *status) {
#ifndef BVCF_INF_SMALL
#define BVCF_INF_SMALL 4096
#endif
  __shared__ __attribute__((aligned(16))) InfLds<BVCF_INF_SMALL> S;
  k_inflate_body<BVCF_INF_SMALL>(comp, desc, n_blocks, out, status, S);
}

// ------------------------------------------------------------------ CRC-32 of the inflated blocks
// BGZF's trailer carries the CRC-32 (IEEE 802.3, reflected) of each block's text.  One wave per block, the text read
// in ROWS of 1 KiB -- lane L takes bytes [16 L, 16 L + 16) of every row, so a row is one coalesced load (a lane that
// walks its own contiguous KiB instead touches 64 cache lines per load instruction and the kernel runs at a sixth of
// this one's rate).  A CRC without its start value is linear over GF(2): the block's CRC is the XOR of the CRCs of 64
// messages, lane L's being its own bytes with zeros everywhere else.  Within a row a lane takes its four dwords by
// slicing-by-4 (table j: a byte followed by j zero bytes); to get from the end of its chunk to the same place in the
// next row its state has to pass 1 008 zero bytes, which is another table lookup per state byte (the same tables
// multiplied by x^(8 * 1008) mod P) and needs no step of its own: T(adv(S) ^ w) = T(adv(S)) ^ T(w).  The block is seen
// as the TAIL of a whole number of rows (leading zeros do not change a CRC that starts from 0; the usual 0xFFFFFFFF
// start value is the same as inverting the first four bytes), so every lane's last chunk ends 16 (63 - L) bytes before
// the end: one multiplication by x^(8 * 16 * (63 - L)) per lane and six XOR steps across the wave finish it.
// Tables and constants come from the host (CrcTabs, bvcf_core.hip).
struct CrcTabs {
  uint32_t std4[4][256];  // std4[j][b]: CRC register after byte b and j zero bytes
  uint32_t jump[4][256];  // jump[j][b] = std4[j][b] * x^(8 * 1008)
  uint32_t lane_k[64];    // x^(8 * 16 * (63 - L))
};

__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {  // zlib's multmodp, branch-free
  uint32_t p = 0;
#pragma unroll
  for (int i = 31; i >= 0; i--) {
    p ^= b & (0u - ((a >> i) & 1u));
    b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
  }
  return p;
}

__global__ __launch_bounds__(kWave) void k_crc32(const uint8_t *text, const BgzfDesc *desc, uint32_t n_blocks, const CrcTabs *tabs,
                                                 uint32_t *crc_out) {
  __shared__ uint32_t s_t[2048];  // std4 then jump
  const int lane = lane_id();
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(tabs);
    for (uint32_t i = lane; i < 2048; i += kWave) s_t[i] = g[i];
  }
  const uint32_t my_k = tabs->lane_k[lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  auto fold = [&](uint32_t x, uint32_t base) -> uint32_t {
    return s_t[base + 768u + (x & 0xFFu)] ^ s_t[base + 512u + ((x >> 8) & 0xFFu)] ^ s_t[base + 256u + ((x >> 16) & 0xFFu)] ^
           s_t[base + (x >> 24)];
  };
  for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const BgzfDesc d = desc[blk];
    const uint8_t *p = text + d.out_off;
    const uint32_t n = d.isize;  // <= 65536
    if (n < 4u) {
      // shorter than the start value: the plain definition
      if (lane == 0) {
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t j = 0; j < n; j++) c = s_t[(c ^ p[j]) & 0xFFu] ^ (c >> 8);
        crc_out[blk] = c ^ 0xFFFFFFFFu;
      }
      continue;
    }
    const uint32_t n_rows = (n + 1023u) >> 10;
    const int pad = (int)(n_rows * 1024u - n);  // virtual leading zeros
    uint32_t S = 0;
    for (uint32_t r = 0; r < n_rows; r++) {
      const int r0 = (int)(r * 1024u + 16u * (uint32_t)lane) - pad;  // offset in the text of the chunk's first byte
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (r0 >= 0) {
        const u32x4 v = *reinterpret_cast<const u32x4_u *>(p + r0);
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
      } else if (r0 > -16) {
        // the chunk the text starts in
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int i = -r0; i < 16; i++) {
          const uint32_t b = (uint32_t)p[r0 + i] << (8u * ((uint32_t)i & 3u));
          if (i < 4) a0 |= b; else if (i < 8) a1 |= b; else if (i < 12) a2 |= b; else a3 |= b;
        }
        w[0] = a0;
        w[1] = a1;
        w[2] = a2;
        w[3] = a3;
      }
      if (r0 < 4 && r0 > -16) {
        // the 0xFFFFFFFF start value: the text's first four bytes inverted
        uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        for (int t = 0; t < 4; t++) {
          const int i = t - r0;
          if (i >= 0 && i < 16) {
            const uint32_t m = 0xFFu << (8u * ((uint32_t)i & 3u));
            if (i < 4) m0 |= m; else if (i < 8) m1 |= m; else if (i < 12) m2 |= m; else m3 |= m;
          }
        }
        w[0] ^= m0;
        w[1] ^= m1;
        w[2] ^= m2;
        w[3] ^= m3;
      }
      S = fold(S, 1024u) ^ fold(w[0], 0u);
      S = fold(S ^ w[1], 0u);
      S = fold(S ^ w[2], 0u);
      S = fold(S ^ w[3], 0u);
    }
    uint32_t c = gf2_mulmod(my_k, S);
#pragma unroll
    for (int l = 0; l < 6; l++) c ^= __shfl_xor(c, 1 << l);
    if (lane == 0) crc_out[blk] = c ^ 0xFFFFFFFFu;
  }
}

// ------------------------------------------------------------------ where a batch of inflated text starts and ends
// A batch is cut in the compressed domain (whole BGZF blocks), so its text begins and ends inside lines.  The line
// that straddles the boundary to the next batch belongs to THIS batch: the caller appends the next batch's first
// block(s) as look-ahead, and the text ends after the first terminator at or past the end of the batch's own blocks.
// Accordingly the text before the first terminator of a batch that is not the stream's first belongs to the previous
// batch and is skipped.  One wave; also folds the per-block inflate status and CRC comparison.
struct CutArgs {
  const uint8_t *text;
  uint32_t total;        // bytes of text (own + look-ahead blocks)
  uint32_t own;          // bytes the batch's own blocks inflate to
  uint32_t skip_first;   // 1: start after the first terminator; 0: start at first_off
  uint32_t at_eof;       // the look-ahead reaches the end of the stream: a last line without a terminator ends the batch
  uint32_t first_off;
  uint32_t eol_byte;
  uint32_t n_blocks;
  const uint32_t *status, *crc, *want_crc;
  uint32_t *out;         // {start, end, flags, first bad block}
};
enum { kCutInflateError = 1, kCutCrcMismatch = 2, kCutNoTerminator = 4 };

__global__ __launch_bounds__(kWave) void k_cuts(CutArgs c) {
  const int lane = lane_id();
  // first terminator at a position >= from, or kNoPos
  constexpr uint32_t kNoPos = 0xFFFFFFFFu;
  auto first_eol = [&](uint32_t from) -> uint32_t {
    for (uint32_t base = from & ~15u; base < c.total; base += kChunk) {
      const uint32_t off = base + 16u * lane;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (off + 16u <= c.total) {
        v = *reinterpret_cast<const u32x4_u *>(c.text + off);
      } else if (off < c.total) {
        // the last, partial piece
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        for (uint32_t i = 0; i < c.total - off; i++) {
          const uint32_t b = (uint32_t)c.text[off + i] << (8u * (i & 3u));
          if (i < 4) w0 |= b; else if (i < 8) w1 |= b; else if (i < 12) w2 |= b; else w3 |= b;
        }
        v = u32x4{w0, w1, w2, w3};
      }
      uint32_t m = eq_mask16(v, c.eol_byte) & bits_until(c.total, off);
      if (off < from) m &= ~bits_until(from, off);
      const unsigned long long b = __ballot(m != 0);
      if (b) {
        const int src = __ffsll((long long)b) - 1;
        return lane_value(off + __ffs(m) - 1, src);
      }
    }
    return kNoPos;
  };
  uint32_t flags = 0, bad_block = kNoPos;
  for (uint32_t base = 0; base < c.n_blocks; base += kWave) {
    const uint32_t i = base + lane;
    const bool inf_bad = i < c.n_blocks && c.status[i] != kInfOk;
    const bool crc_bad = i < c.n_blocks && !inf_bad && c.crc[i] != c.want_crc[i];
    const unsigned long long bi = __ballot(inf_bad), bc = __ballot(crc_bad);
    if (bi) flags |= kCutInflateError;
    if (bc) flags |= kCutCrcMismatch;
    if ((bi | bc) && bad_block == kNoPos) bad_block = base + (uint32_t)__ffsll((long long)(bi | bc)) - 1u;
  }
  uint32_t start = c.first_off, end = c.total;
  if (c.skip_first) {
    const uint32_t e = first_eol(0);
    start = e == kNoPos ? c.total : e + 1u;
  }
  if (c.own < c.total) {  // there is look-ahead: the batch ends with the line that straddles `own`
    // (the next batch skips everything up to the first terminator of ITS text, i.e. at or after `own`: even when
    // this batch's own text happens to end on a line boundary, the following line is this batch's)
    const uint32_t e = first_eol(c.own);
    if (e == kNoPos) {
      // (the stream's last line has no terminator: the text ends the batch, the scans ignore the unterminated tail,
      // main.go:354-358; otherwise the caller gave too little look-ahead)
      if (!c.at_eof) flags |= kCutNoTerminator;
    } else {
      end = e + 1u;
    }
  }
  if (start > end) start = end;
  if (lane == 0) {
    c.out[0] = start;
    c.out[1] = end;
    c.out[2] = flags;
    c.out[3] = bad_block;
  }
}

// ------------------------------------------------------------------ the line heads of a batch, packed
// The host's TSV assembly reads CHROM..INFO of every line that passed; when the text was inflated on the device only
// those bytes go back (k_heads_len -> k_heads_scan -> k_heads_copy), not the sample columns.
struct HeadArgs {
  uint32_t *off;              // [max_lines + 1] head length per line -> exclusive prefix
  uint8_t *out;               // the packed heads
  unsigned long long cap;
  unsigned long long *total;  // bytes of all heads
};

__global__ __launch_bounds__(kWgThreads) void k_heads_len(KernelArgs a, HeadArgs h) {
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) {
    const uint32_t st = a.lines[i].status;
    // up to the end of the INFO column (fields past the line's end have fend == len)
    h.off[i] = (st == BVCF_LINE_OK || st == BVCF_LINE_NOALLELE) ? min(a.lines[i].fend[7], a.lines[i].len) : 0u;
  }
}

__global__ __launch_bounds__(1024) void k_heads_scan(KernelArgs a, HeadArgs h) {
  __shared__ unsigned long long s_part[1024];
  const uint32_t n = min(a.counters->n_lines, a.max_lines);
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per;
  unsigned long long sum = 0;
  for (uint32_t i = 0; i < per; i++)
    if (lo + i < n) sum += h.off[lo + i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long t = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0ull;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = s_part[threadIdx.x] - sum;
  for (uint32_t i = 0; i < per; i++) {
    if (lo + i < n) {
      const uint32_t v = h.off[lo + i];
      h.off[lo + i] = (uint32_t)run;
      run += v;
    }
  }
  if (threadIdx.x == 1023) {
    *h.total = s_part[1023];
    if (n <= a.max_lines) h.off[n] = (uint32_t)s_part[1023];
  }
}

// 16 lanes per line
__global__ __launch_bounds__(kWgThreads) void k_heads_copy(KernelArgs a, HeadArgs h) {
  if (*h.total > h.cap) return;
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t gl = threadIdx.x & 15u;
  const uint32_t groups = gridDim.x * (blockDim.x / 16u);
  for (uint32_t i = blockIdx.x * (blockDim.x / 16u) + threadIdx.x / 16u; i < n_lines; i += groups) {
    const uint32_t st = a.lines[i].status;
    if (st != BVCF_LINE_OK && st != BVCF_LINE_NOALLELE) continue;
    const uint32_t n = min(a.lines[i].fend[7], a.lines[i].len);
    const uint8_t *src = a.buf + a.lines[i].off;
    uint8_t *dst = h.out + h.off[i];
    for (uint32_t j = gl; j < n; j += 16u) dst[j] = src[j];
  }
}

}  // namespace bvcf_dev
