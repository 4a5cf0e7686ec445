// bvcf_alleles.hip.h — getAlleles / altIsValid / trTv for one lane (main.go:456-474, 723-1038)
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include "bvcf_common.hip.h"

namespace bvcf_dev {

// ------------------------------------------------------------------ getAlleles (one lane)

struct Span {
  uint32_t off, len;
};

// Byte access for the serial per-line work (one lane per line): the head of the line is staged in LDS (k_head: the
// first windows of the line; k_sites: a text ring), everything else falls through to HBM.
// The LDS pointer carries its address space: with two generic pointers hipcc merges the two loads into ONE
// flat_load_ubyte of a selected address -- a flat load waits on vmcnt as well as lgkmcnt, i.e. every byte of the
// serial work then waits for the text prefetches and the record stores in flight.
typedef const __attribute__((address_space(3))) uint8_t *lds_bytes_t;
__device__ __forceinline__ lds_bytes_t as_lds(const void *p) { return (lds_bytes_t)p; }

// kFallback = false: every byte asked for is known to be in the LDS copy (k_sites: a line that lies in the ring)
template <bool kFallback>
struct BytesT {
  const uint8_t *g;          // the block
  lds_bytes_t lds;           // copy of block bytes [lo, lo + n): byte `off` lives at lds[(off - sub) & mask]
  uint32_t lo, n;
  uint32_t sub = 0, mask = 0xFFFFFFFFu;  // k_head: a flat copy (sub = lo); k_sites: a ring (sub = 0, mask = ring size - 1)
  __device__ __forceinline__ uint8_t operator[](uint32_t off) const {
    if (!kFallback) return lds[(off - sub) & mask];
    const uint32_t d = off - lo;
    if (d < n) return lds[(off - sub) & mask];
    return g[off];
  }
};
typedef BytesT<true> Bytes;

// strconv.Atoi on buf[s.off .. +len): optional sign, digits, must fit int64 (main.go:752,824)
template <class B>
__device__ inline bool go_atoi(const B &buf, Span s, long long *out) {
  if (s.len == 0) return false;
  uint32_t i = 0;
  bool neg = false;
  uint8_t c0 = buf[s.off];
  if (c0 == '+' || c0 == '-') {
    neg = c0 == '-';
    i = 1;
    if (s.len == 1) return false;
  }
  unsigned long long v = 0;
  // v * 10 + d must stay within 2^63 - 1 (2^63 for a negative number): lim / 10 and lim % 10 spelled out -- a 64-bit division
  // per digit was a hundred instructions of every wave that holds one deletion
  const unsigned long long q = 922337203685477580ull;
  const uint32_t r = neg ? 8u : 7u;
  #pragma nounroll
  for (; i < s.len; i++) {
    uint32_t d = (uint32_t)buf[s.off + i] - '0';
    if (d > 9u) return false;
    if (v > q || (v == q && d > r)) return false;
    v = v * 10ull + d;
  }
  *out = neg ? (long long)(0ull - v) : (long long)v;
  return true;
}

// (bit tests instead of chains of ==: hipcc turns such a chain into a switch and lowers that to a tree of branches)
__device__ __forceinline__ bool is_actg(uint8_t c) {
  const uint32_t d = (uint32_t)c - 'A';  // A C G T = bits 0, 2, 6, 19
  return (bool)((uint32_t)(d < 20u) & (0x80045u >> (d & 31u)));
}

// parse.GetTrTv restated (oracle/bvcf_oracle.c orc_get_trtv): 1 = transition (A<->G, C<->T), 2 = transversion, 0 = n/a
__device__ __forceinline__ uint8_t trtv_of(uint8_t ref, uint8_t alt) {
  // bits 1-2 of the letters: A 0, C 1, T 2, G 3 -- a transition pairs the codes that differ in both bits
  const uint32_t x = (((uint32_t)ref ^ (uint32_t)alt) >> 1) & 3u;
  const uint32_t ok = (uint32_t)is_actg(ref) & (uint32_t)is_actg(alt);
  return (uint8_t)(ok ? (x == 3u ? 1u : 2u) : 0u);
}

// per-allele GT statistics (makeHetHomozygotes' return values)
struct GtStats {
  uint32_t ac, an, n_het, n_hom, n_miss;
};

// lane-0 state of one line's getAlleles evaluation
template <class B>
struct AlleleCtxT {
  B buf;
  Span chrom, pos, ref, alt;
  long long int_pos;   // intPos, main.go:767
  bool pos_bad;        // Atoi failed: the ALT loop is over (main.go:826-829)
  uint32_t line;
};
typedef AlleleCtxT<Bytes> AlleleCtx;

// what one ALT token yields
struct AlleleEval {
  uint32_t n;          // records this token produces
  uint32_t err;        // BVCF_ERR_* to log, 0 if none
  bool stop;           // "Invalid POS": break out of the ALT loop
  // single-record description (n == 1 and !mnp)
  bool mnp;            // records are the differing bases of an equal-length block
  long long pos;
  bool pos_text;
  uint8_t ref, alt_base, kind;
  uint32_t alt_off, alt_len;
};

// The single-ALT-byte path, main.go:735-765.  t is the whole ALT field (1 byte).
template <class B>
__device__ inline void eval_single(AlleleCtxT<B> &c, AlleleEval &e) {
  const B &b = c.buf;
  e = AlleleEval{};
  const uint8_t a0 = b[c.alt.off];
  if (a0 != 'A' && a0 != 'C' && a0 != 'G' && a0 != 'T') {
    e.err = BVCF_ERR_BAD_ALT1;
    return;
  }
  if (c.ref.len == 1) {
    e.n = 1;
    e.pos_text = true;
    e.ref = b[c.ref.off];
    e.alt_base = a0;
    e.kind = BVCF_ALT_BASE;
    e.alt_len = 1;
    return;
  }
  if (c.ref.len == 0) {
    e.err = BVCF_ERR_EMPTY_REF;
    return;
  }
  if (a0 != b[c.ref.off]) {
    e.err = BVCF_ERR_DEL1_1;
    return;
  }
  long long p;
  if (!go_atoi(b, c.pos, &p)) {
    e.err = BVCF_ERR_POS1;
    return;
  }
  e.n = 1;
  e.pos = p + 1;
  e.ref = b[c.ref.off + 1];
  e.kind = BVCF_ALT_DEL;
  e.alt_len = c.ref.len - 1;
}

// One token of strings.Split(alt, ","), main.go:774-999.  t = token span.
template <class B>
__device__ inline void eval_token(AlleleCtxT<B> &c, Span t, AlleleEval &e) {
  const B &b = c.buf;
  e = AlleleEval{};
  // altIsValid, main.go:456-474 (empty token: Go would panic; invalid here)
  bool valid = t.len > 0;
  #pragma nounroll
  for (uint32_t i = 0; i < t.len && valid; i++) valid = is_actg(b[t.off + i]);
  if (!valid) {
    e.err = BVCF_ERR_BAD_ALT;
    return;
  }
  const uint32_t nref = c.ref.len, nt = t.len;
  if (nref == 1) {  // main.go:786-815
    if (nt == 1) {
      e.n = 1;
      e.pos_text = true;
      e.ref = b[c.ref.off];
      e.alt_base = b[t.off];
      e.kind = BVCF_ALT_BASE;
      e.alt_len = 1;
      return;
    }
    if (b[t.off] != b[c.ref.off]) {
      e.err = BVCF_ERR_INS1;
      return;
    }
    e.n = 1;
    e.pos_text = true;
    e.ref = b[c.ref.off];
    e.kind = BVCF_ALT_INS;
    e.alt_off = t.off + 1;
    e.alt_len = nt - 1;
    return;
  }
  // main.go:822-830
  if (c.int_pos == 0) {
    long long p;
    if (!go_atoi(b, c.pos, &p)) {
      e.err = BVCF_ERR_POS;
      e.stop = true;
      return;
    }
    c.int_pos = p;
  }
  if (nt == 1) {  // main.go:832-847
    if (b[t.off] != b[c.ref.off]) {
      e.err = BVCF_ERR_DEL1;
      return;
    }
    e.n = 1;
    e.pos = c.int_pos + 1;
    e.ref = b[c.ref.off + 1];
    e.kind = BVCF_ALT_DEL;
    e.alt_len = nref - 1;
    return;
  }
  if (nt == nref) {  // main.go:855-873
    uint32_t n = 0;
    #pragma nounroll
    for (uint32_t i = 0; i < nref; i++) n += b[c.ref.off + i] != b[t.off + i];
    e.n = n;
    e.mnp = true;
    return;
  }
  if (nt > nref) {  // main.go:899-958
    int r = 0;
    const int lt = (int)nt, lr = (int)nref;
    #pragma nounroll
    while (lt + r > 0 && lr + r > 1 && b[t.off + lt + r - 1] == b[c.ref.off + lr + r - 1]) r--;
    const int offset = lr + r;
    #pragma nounroll
    for (int i = 0; i < offset; i++)
      if (b[c.ref.off + i] != b[t.off + i]) {
        e.err = BVCF_ERR_MIXED;
        return;
      }
    e.n = 1;
    e.pos = c.int_pos + offset - 1;
    e.ref = b[c.ref.off + offset - 1];
    e.kind = BVCF_ALT_INS;
    e.alt_off = t.off + offset;
    e.alt_len = (uint32_t)(lt + r - offset);
    return;
  }
  {  // main.go:971-998
    int r = 0;
    const int lt = (int)nt, lr = (int)nref;
    #pragma nounroll
    while (lt + r > 1 && lr + r > 0 && b[t.off + lt + r - 1] == b[c.ref.off + lr + r - 1]) r--;
    const int offset = lt + r;
    #pragma nounroll
    for (int i = 0; i < offset; i++)
      if (b[c.ref.off + i] != b[t.off + i]) {
        e.err = BVCF_ERR_MIXED;
        return;
      }
    e.n = 1;
    e.pos = c.int_pos + offset;
    e.ref = b[c.ref.off + offset];
    e.kind = BVCF_ALT_DEL;
    e.alt_len = (uint32_t)(lr + r - offset);
  }
}

// ---- the same from registers, for the shapes nearly every token has
// eval_token / eval_single walk REF, ALT and POS byte by byte: every byte a dependent LDS read, a bounds test and a branch of
// a divergent loop -- 15 000 clock ticks of a k_head step whose wave holds one indel, against 2 000 for SNPs alone.  When the
// three fields lie inside the line's staged head (row: the LDS copy of the line's first kHeadRowBytes bytes, dwords), REF
// is at most 8 bytes, ALT at most 16 with the token at most 8, and POS is 1-9 digits, the fields are read once as whole
// words and the work is byte-parallel arithmetic on them.  Anything else: returns false, and the caller takes the walk.
typedef const __attribute__((address_space(3))) uint32_t *lds_words_t;
__device__ __forceinline__ lds_words_t as_lds_words(const void *p) { return (lds_words_t)p; }
// 16 bytes of the row from byte `rel` (rel + 16 <= row bytes; the row has one more dword behind them)
__device__ __forceinline__ u32x4 row_bytes16(lds_words_t row, uint32_t rel) {
  const uint32_t i = rel >> 2, sh = rel & 3u;
  const uint32_t w0 = row[i], w1 = row[i + 1u], w2 = row[i + 2u], w3 = row[i + 3u], w4 = row[i + 4u];
  return u32x4{__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
               __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh)};
}
// byte-parallel helpers over a word of 8 (unsigned long long) or 16 bytes (unsigned __int128)
typedef unsigned __int128 u128_t;
template <class W> struct WordBytes;
template <> struct WordBytes<unsigned long long> {
  static constexpr uint32_t kBytes = 8;
  static __device__ __forceinline__ unsigned long long of(const u32x4 &v) { return ((unsigned long long)v.y << 32) | v.x; }
  static __device__ __forceinline__ uint32_t clz(unsigned long long x) { return (uint32_t)__clzll((long long)x); }  // x != 0
};
template <> struct WordBytes<u128_t> {
  static constexpr uint32_t kBytes = 16;
  static __device__ __forceinline__ u128_t of(const u32x4 &v) {
    return ((u128_t)(((unsigned long long)v.w << 32) | v.z) << 64) | (((unsigned long long)v.y << 32) | v.x);
  }
  static __device__ __forceinline__ uint32_t clz(u128_t x) {
    const unsigned long long hi = (unsigned long long)(x >> 64), lo = (unsigned long long)x;
    return hi ? (uint32_t)__clzll((long long)hi) : 64u + (uint32_t)__clzll((long long)lo);
  }
};
template <class W> __device__ __forceinline__ W rep_byte(uint32_t b) {
  const unsigned long long r = 0x0101010101010101ull * b;
  return sizeof(W) == 8 ? (W)r : (W)(((u128_t)r << 64) | r);
}
template <class W> __device__ __forceinline__ W low_bytes_w(uint32_t n) { return n >= sizeof(W) ? ~(W)0 : (((W)1 << (8u * n)) - (W)1); }
__device__ __forceinline__ unsigned long long low_bytes(uint32_t n) { return low_bytes_w<unsigned long long>(n); }
// 0x80 in every byte of x that is zero, exact
template <class W> __device__ __forceinline__ W zero_bytes_w(W x) {
  const W m = rep_byte<W>(0x7Fu);
  return ~(((x & m) + m) | x | m);
}
template <class W> __device__ __forceinline__ uint32_t popc_w(W x) {
  return sizeof(W) == 8 ? (uint32_t)__popcll((unsigned long long)x)
                        : (uint32_t)__popcll((unsigned long long)x) + (uint32_t)__popcll((unsigned long long)((u128_t)x >> (sizeof(W) == 8 ? 0 : 64)));
}
// strconv.Atoi for 1-9 plain digits at row byte `rel`; false: not that (the caller walks the field)
__device__ __forceinline__ bool row_atoi9(lds_words_t row, uint32_t rel, uint32_t len, long long *out) {
  if (len - 1u > 8u) return false;
  const u32x4 v = row_bytes16(row, rel);
  const unsigned long long lo = ((unsigned long long)v.y << 32) | v.x;
  const uint32_t last = v.z & 0xFFu;  // (the ninth digit)
  // every byte a digit: b - '0' <= 9, byte-parallel over the first eight (bytes past len masked to '0')
  const unsigned long long keep = low_bytes(len);
  const unsigned long long d = ((lo & keep) | (0x3030303030303030ull & ~keep)) ^ 0x3030303030303030ull;  // digits -> 0..9
  if ((d & 0xF0F0F0F0F0F0F0F0ull) != 0ull) return false;                 // high nibble set: not '0'..'?'
  if ((((d + 0x0606060606060606ull) & 0x1010101010101010ull)) != 0ull) return false;  // 10..15
  if (len == 9u && (last - '0') > 9u) return false;
  unsigned long long val = 0;
#pragma unroll
  for (uint32_t i = 0; i < 8u; i++)
    if (i < len) val = val * 10ull + ((d >> (8u * i)) & 0xFull);
  if (len == 9u) val = val * 10ull + (last - '0');
  *out = (long long)val;
  return true;
}
// the decisions of eval_token / eval_single on a token tk of nt bytes and a REF rf of nref bytes held in words of W
template <class W>
__device__ __forceinline__ bool eval_words(lds_words_t row, uint32_t pos_rel, AlleleCtx &c, bool single, const Span &t, W tk, W rf,
                                           AlleleEval &e) {
  const uint32_t nref = c.ref.len, nt = t.len;
  constexpr uint32_t kB = WordBytes<W>::kBytes;
  e = AlleleEval{};
  // altIsValid, main.go:456-474
  const W flags = rep_byte<W>(0x80u) & low_bytes_w<W>(nt);
  const W actg = zero_bytes_w<W>(tk ^ rep_byte<W>('A')) | zero_bytes_w<W>(tk ^ rep_byte<W>('C')) |
                 zero_bytes_w<W>(tk ^ rep_byte<W>('G')) | zero_bytes_w<W>(tk ^ rep_byte<W>('T'));
  if (nt == 0u || (actg & flags) != flags) {
    e.err = single ? BVCF_ERR_BAD_ALT1 : BVCF_ERR_BAD_ALT;
    return true;
  }
  const uint8_t t0 = (uint8_t)tk, r0 = (uint8_t)rf;
  if (nref == 1u) {  // main.go:743-750, 786-815
    e.n = 1;
    e.pos_text = true;
    e.ref = r0;
    if (nt == 1u) {
      e.alt_base = t0;
      e.kind = BVCF_ALT_BASE;
      e.alt_len = 1;
      return true;
    }
    if (t0 != r0) {
      e = AlleleEval{};
      e.err = BVCF_ERR_INS1;
      return true;
    }
    e.kind = BVCF_ALT_INS;
    e.alt_off = t.off + 1u;
    e.alt_len = nt - 1u;
    return true;
  }
  if (single && t0 != r0) {  // main.go:751 (before the Atoi there)
    e.err = BVCF_ERR_DEL1_1;
    return true;
  }
  long long ip;
  if (!row_atoi9(row, pos_rel, c.pos.len, &ip)) return false;  // (odd POS fields: the walk knows every case)
  c.int_pos = ip;  // (intPos, main.go:822: the records of an equal-length block count from it)
  if (nt == 1u) {  // main.go:752-764, 832-847
    if (t0 != r0) {
      e.err = BVCF_ERR_DEL1;
      return true;
    }
    e.n = 1;
    e.pos = ip + 1;
    e.ref = (uint8_t)(rf >> 8);
    e.kind = BVCF_ALT_DEL;
    e.alt_len = nref - 1u;
    return true;
  }
  const W x = tk ^ rf;
  if (nt == nref) {  // main.go:855-873: one record per differing base
    const W same = zero_bytes_w<W>(x) & (rep_byte<W>(0x80u) & low_bytes_w<W>(nref));
    e.n = nref - popc_w<W>(same);
    e.mnp = true;
    return true;
  }
  // common suffix: both ends moved to the word's last byte (a byte before the shorter one's start is zero there, a letter in the other)
  const W xs = (tk << (8u * (kB - nt))) ^ (rf << (8u * (kB - nref)));
  const uint32_t suffix = xs ? WordBytes<W>::clz(xs) >> 3 : kB;
  if (nt > nref) {  // main.go:899-958
    const uint32_t m = min(suffix, nref - 1u);  // (lr + r > 1)
    const uint32_t offset = nref - m;
    if (x & low_bytes_w<W>(offset)) {
      e.err = BVCF_ERR_MIXED;
      return true;
    }
    e.n = 1;
    e.pos = ip + (long long)offset - 1;
    e.ref = (uint8_t)(rf >> (8u * (offset - 1u)));
    e.kind = BVCF_ALT_INS;
    e.alt_off = t.off + offset;
    e.alt_len = nt - m - offset;
    return true;
  }
  {  // main.go:971-998
    const uint32_t m = min(suffix, nt - 1u);  // (lt + r > 1)
    const uint32_t offset = nt - m;
    if (x & low_bytes_w<W>(offset)) {
      e.err = BVCF_ERR_MIXED;
      return true;
    }
    e.n = 1;
    e.pos = ip + (long long)offset;
    e.ref = (uint8_t)(rf >> (8u * offset));
    e.kind = BVCF_ALT_DEL;
    e.alt_len = nref - m - offset;
    return true;
  }
}
// One ALT token (k-th of strings.Split(alt, ","); `single`: the single-byte ALT path, eval_single) from the staged row.
// ls: the line's start, staged: bytes of it in the row.  *t: the token's span (set whenever true is returned).
template <uint32_t kRowBytes>
__device__ inline bool eval_token_row(lds_words_t row, uint32_t staged, uint32_t ls, AlleleCtx &c, uint32_t k, bool single,
                                      Span *t, AlleleEval &e) {
  const uint32_t nref = c.ref.len, nalt = c.alt.len;
  const uint32_t ref_rel = c.ref.off - ls, alt_rel = c.alt.off - ls, pos_rel = c.pos.off - ls;
  if (nref - 1u > 15u || nalt - 1u > 15u) return false;
  if (ref_rel + 16u > kRowBytes || alt_rel + 16u > kRowBytes || pos_rel + 16u > kRowBytes) return false;  // (the reads stay in the row)
  if (ref_rel + nref > staged || alt_rel + nalt > staged || pos_rel + c.pos.len > staged) return false;
  const u32x4 av = row_bytes16(row, alt_rel);
  const u32x4 rv = row_bytes16(row, ref_rel);
  // ---- the token
  uint32_t start = 0, end = nalt;
  if (!single) {
    uint32_t cm = eq_mask16(av, ',') & ((1u << nalt) - 1u);
    if (k > (uint32_t)__popc(cm)) return false;
    uint32_t m = cm;
    for (uint32_t i = 0; i + 1u < k; i++) m &= m - 1u;
    if (k > 0u) {
      start = (uint32_t)__ffs(m);  // (one past the k-th comma)
      m &= m - 1u;
    }
    end = m ? (uint32_t)__ffs(m) - 1u : nalt;
  }
  t->off = c.alt.off + start;
  t->len = end - start;
  const u128_t a16 = WordBytes<u128_t>::of(av) >> (8u * start);
  if (nref <= 8u && t->len <= 8u)
    return eval_words<unsigned long long>(row, pos_rel, c, single, *t, (unsigned long long)a16 & low_bytes(t->len),
                                          WordBytes<unsigned long long>::of(rv) & low_bytes(nref), e);
  return eval_words<u128_t>(row, pos_rel, c, single, *t, a16 & low_bytes_w<u128_t>(t->len), WordBytes<u128_t>::of(rv) & low_bytes_w<u128_t>(nref), e);
}

// next token of the ALT field starting at *cursor (relative to alt.off); false when exhausted
template <class B>
__device__ inline bool next_token(const AlleleCtxT<B> &c, uint32_t *cursor, Span *t) {
  if (*cursor > c.alt.len) return false;
  uint32_t s = *cursor, i = s;
  #pragma nounroll
  while (i < c.alt.len && c.buf[c.alt.off + i] != ',') i++;
  t->off = c.alt.off + s;
  t->len = i - s;
  *cursor = i + 1;
  return true;
}

// ord: the message's place among its line's (the ALT index it is about), for callers whose lanes log a line's messages
// in no particular order (k_head: one lane per ALT token); 0 where one lane logs them in order
__device__ inline void log_err(const KernelArgs &a, uint32_t line, uint32_t alt_no, uint32_t code, uint32_t ord = 0u) {
  uint32_t i = atomicAdd(&a.counters->n_errs, 1u);
  if (i < a.max_errs) {
    bvcf_err e;
    e.line = line;
    e.alt_no = alt_no;
    e.code = code;
    e.pad = ord;
    a.errs[i] = e;
  }
}


}  // namespace bvcf_dev
