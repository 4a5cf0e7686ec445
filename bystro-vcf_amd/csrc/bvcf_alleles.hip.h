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
  const unsigned long long lim = neg ? 9223372036854775808ull : 9223372036854775807ull;
  #pragma nounroll
  for (; i < s.len; i++) {
    uint32_t d = (uint32_t)buf[s.off + i] - '0';
    if (d > 9u) return false;
    if (v > (lim - d) / 10ull) return false;
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
