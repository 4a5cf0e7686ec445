// bvcf_device.hip.h — gfx950 device code of the per-line variant pipeline.
//
// Kernels, in launch order per batch:
//   k_count_eol    newline census per 1 KiB chunk              (readVcf's ReadBytes, main.go:354)
//   k_scan_groups  exclusive scan of the census, level 1
//   k_scan_top     level 2 + batch totals
//   k_scatter_eol  line start offsets from the census           ("workQueue <- buff", main.go:366)
//   k_head         16 lanes per line: tokenise the fixed columns, FILTER gate, getAlleles;
//                  emits allele records and one genotype-scan task per (line, ALT index)
//                                                                (main.go:535-545, 723-1038)
//   k_gt           one wavefront per task: the per-sample GT byte scan -> ac/an/het/hom/missing
//                  and the 2-bit class map.  THE HBM-bound kernel.   (main.go:1042-1194)
//   k_finish       field-count verdict per line, scan results into the allele records
//
// Streaming variant for files with samples (KernelArgs.fused): the census, its scans, the scatter
// and the ALT #1 genotype scan are replaced by ONE pass over the text,
//   k_stream       one wave walks a 64 KiB tile: finds the lines that start in it, tokenises their
//                  fixed columns and scans ALT #1 straight away (the scan's loads ARE the newline
//                  search: a regular line ends where 4*ns bytes of "x|y<TAB>" end)
//   k_scan_*       exclusive scan of the per-tile line counts
//   k_order        tile-local line entries -> input-ordered line_off / line_len / results
// after which k_head, k_gt (further ALT indices only) and k_finish run as above.
//
// Everything is byte/integer work bounded by the HBM read of the line bytes; no MFMA.
// Loads are 16 B per lane, 1 KiB per wave-instruction, starting exactly at the byte the
// record window starts at (unaligned dwordx4), so that in a regular sample region
// ("x|y\t" per sample) every dword in a lane is exactly one sample field.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvcf.h"

namespace bvcf_dev {

constexpr int kWave = 64;
constexpr int kWavesPerWg = 4;
constexpr int kWgThreads = kWave * kWavesPerWg;
constexpr uint32_t kChunk = 1024;      // bytes per wave-iteration (16 B x 64 lanes)
constexpr uint32_t kScanGroup = 1024;  // census entries per level-1 scan group

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

// FILTER allow / deny sets (config.allowedFilters / excludedFilters, main.go:78-79)
struct FilterTable {
  uint32_t allow_nil, allow_n;
  uint32_t deny_nil, deny_n;
  uint16_t allow_off[32], allow_len[32];
  uint16_t deny_off[32], deny_len[32];
  uint8_t text[2048];
};

// device-resident batch state
struct BatchCounters {
  uint32_t n_lines;      // lines listed in lines[] (may exceed max_lines)
  uint32_t n_alleles;    // bvcf_allele slots requested past the first n_lines
  uint32_t n_errs;
  uint32_t n_tasks;      // genotype-scan tasks requested past the first n_lines
  uint32_t lines_seen;   // terminated lines in the block (== n_lines on the census path)
  uint32_t cmap_maps;    // streaming path: class maps handed out
  uint32_t pad[2];
};

// streaming path: what k_stream knows about a line when it has scanned it
struct StreamEntry {
  uint32_t ls, len;                       // start offset, content length (terminator chomped)
  uint32_t ac, an, n_het, n_hom, n_miss;  // ALT #1
  uint32_t cmap_off;
};

// one genotype scan: all samples of one line against one ALT index
struct GtTask {
  uint32_t line;
  uint32_t allele;       // alleleNum = ALT index + 1 (main.go:552)
  uint32_t s_begin;      // first byte after the FORMAT column's TAB
  uint32_t cend;         // end of the line content (terminator excluded)
  uint32_t cmap_off;     // BVCF_NO_CMAP if no class map is wanted
  uint32_t pad[3];
};

// makeHetHomozygotes' return values for one task, plus the fields it walked
struct GtResult {
  uint32_t ac, an, n_het, n_hom, n_miss;
  uint32_t n_fields;     // sample fields present on the line
  uint32_t pad[2];
};

struct KernelArgs {
  const uint8_t *buf;
  uint32_t nbytes;       // bytes of whole lines
  uint32_t cap;          // bytes that may be read (nbytes + pad)
  uint32_t n_header;     // len(header)
  uint32_t n_samples;    // len(header) - 9, or 0
  uint32_t eol_chars;
  uint32_t eol_byte;
  uint32_t want_cmap;
  uint32_t cmap_stride;
  uint32_t max_lines, max_alleles, max_errs, max_tasks;
  unsigned long long max_cmap;
  const FilterTable *filters;
  uint32_t *census;      // [n_chunks] newline count per chunk -> exclusive prefix within group
  uint32_t *group_base;  // [n_groups]
  uint32_t *line_off;    // [max_lines + 1]
  bvcf_line *lines;
  bvcf_allele *alleles;
  bvcf_err *errs;
  uint8_t *cmap;
  GtTask *tasks;
  GtResult *results;
  BatchCounters *counters;
  // streaming path
  uint32_t fused;        // 1: k_stream found the lines and scanned ALT #1
  uint32_t tile_bytes;   // bytes of text a wave owns (lines belong to the tile they start in)
  uint32_t tile_quota;   // entries reserved per tile: a line that passes the field count is at
                         // least n_header - 1 + eol_chars bytes long
  uint32_t n_tiles;
  StreamEntry *entries;  // [n_tiles * tile_quota]
  uint32_t *line_len;    // [max_lines]
  uint32_t *line_cmap;   // [max_lines] class map of ALT #1
};

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// three per-lane counts -> wave totals; two of them share a register while they fit 16 bits
__device__ __forceinline__ void wave_sum3(uint32_t a, uint32_t b, uint32_t c, uint32_t limit, uint32_t *sa,
                                          uint32_t *sb, uint32_t *sc) {
  if (limit < 65536u) {
    const uint32_t ab = wave_sum(a | (b << 16));
    *sa = ab & 0xFFFFu;
    *sb = ab >> 16;
  } else {
    *sa = wave_sum(a);
    *sb = wave_sum(b);
  }
  *sc = wave_sum(c);
}

// exclusive prefix sum over the 64 lanes; *total receives the wave sum
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t *total) {
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, kWave);
    if (lane_id() >= d) inc += t;
  }
  *total = __shfl(inc, kWave - 1, kWave);
  return inc - v;
}

__device__ __forceinline__ uint32_t bcast0(uint32_t v) { return __shfl(v, 0, kWave); }

// 16 bytes at buf+off for this lane (any alignment); zeros if the window leaves [0, cap)
__device__ __forceinline__ u32x4 load16(const uint8_t *buf, uint32_t off, uint32_t cap) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (off + 16u <= cap) v = *reinterpret_cast<const u32x4_u *>(buf + off);
  return v;
}

// 16 bytes of text that this kernel reads exactly once: non-temporal, so the stream does not evict
// what the caches are asked to keep (measured on k_stream: -3 %)
__device__ __forceinline__ u32x4 ld_stream(const uint8_t *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
}

// 0x80 in every byte of x that is zero, exact (no borrow artefacts)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
  uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}

// 4-bit mask: bit k set iff byte k of d equals c
__device__ __forceinline__ uint32_t eq_mask4(uint32_t d, uint32_t c4) {
  uint32_t z = zero_bytes(d ^ c4) >> 7;  // bits 0,8,16,24
  return ((z * 0x00204081u) >> 21) & 0xFu;
}

// 16-bit mask over the lane's 16 bytes
__device__ __forceinline__ uint32_t eq_mask16(u32x4 v, uint32_t c) {
  uint32_t c4 = c * 0x01010101u;
  return eq_mask4(v.x, c4) | (eq_mask4(v.y, c4) << 4) | (eq_mask4(v.z, c4) << 8) | (eq_mask4(v.w, c4) << 12);
}

// bits [0, n) of a 16-bit mask, n may be <= 0 or >= 16
__device__ __forceinline__ uint32_t low_bits16(int n) {
  return n <= 0 ? 0u : (n >= 16 ? 0xFFFFu : ((1u << n) - 1u));
}

// ------------------------------------------------------------------ line index

// newline census: a wave takes 4 consecutive 1 KiB chunks per step so that 4 KiB are in flight
__global__ __launch_bounds__(kWgThreads) void k_count_eol(KernelArgs a, uint32_t n_chunks) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  const uint32_t stride = gridDim.x * kWavesPerWg * 4u;
  const uint32_t last_off = a.cap - 16u;
  for (uint32_t c0 = wave * 4u; c0 < n_chunks; c0 += stride) {
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t off = min((c0 + q) * kChunk + 16u * lane, last_off);
      v[q] = ld_stream(a.buf + off);
    }
    uint32_t cnt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t off = (c0 + q) * kChunk + 16u * lane;
      cnt[q] = __popc(eq_mask16(v[q], a.eol_byte) & low_bits16((int)a.nbytes - (int)off));
    }
    // two 16-bit sums per register: a chunk holds at most 1024 terminators
    const uint32_t s01 = wave_sum(cnt[0] | (cnt[1] << 16));
    const uint32_t s23 = wave_sum(cnt[2] | (cnt[3] << 16));
    if (lane < 4 && c0 + lane < n_chunks) {
      const uint32_t s = lane < 2 ? s01 : s23;
      a.census[c0 + lane] = (lane & 1) ? (s >> 16) : (s & 0xFFFFu);
    }
  }
}

// level 1: exclusive scan inside groups of kScanGroup census entries; group totals out
__global__ __launch_bounds__(kWgThreads) void k_scan_groups(KernelArgs a, uint32_t n_chunks) {
  __shared__ uint32_t s_wave[kWavesPerWg];
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  const uint32_t g = blockIdx.x;
  const uint32_t base = g * kScanGroup + threadIdx.x * 4u;  // 4 entries per thread
  uint32_t e[4];
#pragma unroll
  for (int i = 0; i < 4; i++) e[i] = (base + i < n_chunks) ? a.census[base + i] : 0u;
  uint32_t mine = e[0] + e[1] + e[2] + e[3];
  uint32_t wtot;
  uint32_t pre = wave_excl_scan(mine, &wtot);
  if (lane == 0) s_wave[w] = wtot;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < w; i++) wbase += s_wave[i];
  uint32_t run = wbase + pre;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (base + i < n_chunks) a.census[base + i] = run;
    run += e[i];
  }
  if (threadIdx.x == kWgThreads - 1) a.group_base[g] = run;  // group total (scanned next)
}

// level 2: exclusive scan of the group totals (single workgroup), batch line count, counters reset
__global__ __launch_bounds__(1024) void k_scan_top(KernelArgs a, uint32_t n_groups) {
  __shared__ uint32_t s_part[1024];
  const uint32_t per = (n_groups + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < per; i++)
    if (lo + i < n_groups) sum += a.group_base[lo + i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  // Hillis-Steele over 1024 partials
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t t = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
  for (uint32_t i = 0; i < per; i++) {
    if (lo + i < n_groups) {
      uint32_t v = a.group_base[lo + i];
      a.group_base[lo + i] = run;
      run += v;
    }
  }
  if (threadIdx.x == 1023) {
    a.counters->n_lines = s_part[1023];
    if (!a.fused) {  // the streaming path zeroes the counters before k_stream uses them
      a.counters->n_alleles = 0;
      a.counters->n_errs = 0;
      a.counters->n_tasks = 0;
      a.counters->lines_seen = s_part[1023];
      a.counters->cmap_maps = 0;
      a.counters->pad[0] = a.counters->pad[1] = 0;
      a.line_off[0] = 0u;
    }
  }
}

// line_off[i + 1] = offset just past line i's terminator.  A wave looks at 64 census entries at
// once (one per lane) and revisits only the chunks that hold a terminator.
__global__ __launch_bounds__(kWgThreads) void k_scatter_eol(KernelArgs a, uint32_t n_chunks) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  const uint32_t stride = gridDim.x * kWavesPerWg * kWave;
  for (uint32_t c0 = wave * kWave; c0 < n_chunks; c0 += stride) {
    const uint32_t c = c0 + lane;
    uint32_t mine = 0, cnt = 0;
    if (c < n_chunks) {
      mine = a.census[c];
      // exclusive prefixes restart at group boundaries; the last chunk of a group (and of the
      // batch) cannot be sized from its successor, so it is always revisited
      const bool has_next = c + 1 < n_chunks && ((c + 1) % kScanGroup) != 0;
      cnt = has_next ? a.census[c + 1] - mine : 1u;
      mine += a.group_base[c / kScanGroup];
    }
    unsigned long long todo = __ballot(cnt != 0);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t cc = c0 + src;
      uint32_t idx = __shfl(mine, src, kWave);
      const uint32_t off = cc * kChunk + 16u * lane;
      u32x4 v = load16(a.buf, off, a.cap);
      uint32_t m = eq_mask16(v, a.eol_byte) & low_bits16((int)a.nbytes - (int)off);
      uint32_t tot;
      idx += wave_excl_scan(__popc(m), &tot);
      while (m) {
        const uint32_t k = __ffs(m) - 1;
        m &= m - 1;
        if (idx < a.max_lines) a.line_off[idx + 1] = off + k + 1;
        idx++;
      }
    }
  }
}

// ------------------------------------------------------------------ getAlleles (one lane)

struct Span {
  uint32_t off, len;
};

// Byte access for the leader lane's serial work: the first windows of the line are staged in LDS
// (k_head), everything else falls through to HBM.
struct Bytes {
  const uint8_t *g;          // the block
  const uint8_t *lds;        // copy of block bytes [lo, lo + n)
  uint32_t lo, n;
  __device__ __forceinline__ uint8_t operator[](uint32_t off) const {
    const uint32_t d = off - lo;
    return d < n ? lds[d] : g[off];
  }
};

// strconv.Atoi on buf[s.off .. +len): optional sign, digits, must fit int64 (main.go:752,824)
__device__ inline bool go_atoi(const Bytes &buf, Span s, long long *out) {
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

__device__ __forceinline__ bool is_actg(uint8_t c) { return c == 'A' || c == 'C' || c == 'T' || c == 'G'; }

// parse.GetTrTv restated (oracle/bvcf_oracle.c orc_get_trtv)
__device__ __forceinline__ uint8_t trtv_of(uint8_t ref, uint8_t alt) {
  if (!is_actg(ref) || !is_actg(alt)) return 0;
  bool tr = (ref == 'A' && alt == 'G') || (ref == 'G' && alt == 'A') || (ref == 'C' && alt == 'T') ||
            (ref == 'T' && alt == 'C');
  return tr ? 1 : 2;
}

// per-allele GT statistics (makeHetHomozygotes' return values)
struct GtStats {
  uint32_t ac, an, n_het, n_hom, n_miss;
};

// lane-0 state of one line's getAlleles evaluation
struct AlleleCtx {
  Bytes buf;
  Span chrom, pos, ref, alt;
  long long int_pos;   // intPos, main.go:767
  bool pos_bad;        // Atoi failed: the ALT loop is over (main.go:826-829)
  uint32_t line;
};

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
__device__ inline void eval_single(AlleleCtx &c, AlleleEval &e) {
  const Bytes &b = c.buf;
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
__device__ inline void eval_token(AlleleCtx &c, Span t, AlleleEval &e) {
  const Bytes &b = c.buf;
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
__device__ inline bool next_token(const AlleleCtx &c, uint32_t *cursor, Span *t) {
  if (*cursor > c.alt.len) return false;
  uint32_t s = *cursor, i = s;
  #pragma nounroll
  while (i < c.alt.len && c.buf[c.alt.off + i] != ',') i++;
  t->off = c.alt.off + s;
  t->len = i - s;
  *cursor = i + 1;
  return true;
}

__device__ inline void log_err(const KernelArgs &a, uint32_t line, uint32_t alt_no, uint32_t code) {
  uint32_t i = atomicAdd(&a.counters->n_errs, 1u);
  if (i < a.max_errs) {
    bvcf_err e;
    e.line = line;
    e.alt_no = alt_no;
    e.code = code;
    e.pad = 0;
    a.errs[i] = e;
  }
}

// ------------------------------------------------------------------ GT scan (whole wave)

// Exact restatement of one sample field of makeHetHomozygotes (main.go:1057-1190) for the
// allele whose decimal text is itoa(a): byte-serial, used for irregular lines.
// p = field start, cend = end of line content; a field ends at '\t' or cend.
__device__ inline void classify_field(const uint8_t *buf, uint32_t p, uint32_t cend, uint32_t a, uint32_t a_ndigits,
                                      uint32_t *cls, uint32_t *altc, uint32_t *gtc) {
  auto getc = [&](uint32_t q) -> uint32_t { return q < cend ? (uint32_t)buf[q] : (uint32_t)'\t'; };
  *altc = 0;
  *gtc = 0;
  *cls = BVCF_CLS_NONE;
  // fast gate, main.go:1063-1064: (len == 3 || g[3] == ':') && g[1] in {'|','/'}
  uint32_t c0 = getc(p), c1 = '\t', c2 = '\t', c3 = '\t';
  if (c0 != '\t') {
    c1 = getc(p + 1);
    if (c1 != '\t') {
      c2 = getc(p + 2);
      if (c2 != '\t') c3 = getc(p + 3);
    }
  }
  const bool have3 = c0 != '\t' && c1 != '\t' && c2 != '\t';
  if (have3 && (c3 == '\t' || c3 == ':') && (c1 == '|' || c1 == '/')) {
    if (c0 == '0' && c2 == '0') {
      *gtc = 2;
      return;
    }
    if (a_ndigits == 1) {
      const uint32_t ac = '0' + a;
      if ((c0 == '0' && c2 == ac) || (c0 == ac && c2 == '0')) {
        *gtc = 2;
        *altc = 1;
        *cls = BVCF_CLS_HET;
        return;
      }
      if (c0 == ac && c2 == ac) {
        *gtc = 2;
        *altc = 2;
        *cls = BVCF_CLS_HOM;
        return;
      }
    }
    if (c0 == '.' || c2 == '.') {
      *cls = BVCF_CLS_MISSING;
      return;
    }
  }
  // general path, main.go:1126-1190.  f = field up to the first ':'
  uint32_t nf = 0;
  bool has_bar = false, has_slash = false;
  #pragma nounroll
  for (;; nf++) {
    uint32_t ch = getc(p + nf);
    if (ch == '\t' || ch == ':') break;
    has_bar |= ch == '|';
    has_slash |= ch == '/';
  }
  const uint32_t sep = has_bar ? '|' : (has_slash ? '/' : 0xFFFFFFFFu);
  uint32_t alt_count = 0, gt_count = 0;
  // token state
  uint32_t tlen = 0;
  unsigned long long val = 0;
  bool digits = true, lead0 = false, dot = false;
  #pragma nounroll
  for (uint32_t k = 0; k <= nf; k++) {
    uint32_t ch = k < nf ? getc(p + k) : sep;
    if (k == nf || ch == sep) {
      if (tlen == 1 && dot) {  // allele == "." => whole sample missing, nothing counted
        *cls = BVCF_CLS_MISSING;
        return;
      }
      if (tlen >= 1 && tlen <= 10 && digits && !lead0 && val == (unsigned long long)a) alt_count++;
      gt_count++;
      tlen = 0;
      val = 0;
      digits = true;
      lead0 = false;
      dot = false;
      continue;
    }
    if (tlen == 0) {
      dot = ch == '.';
      lead0 = ch == '0';
    }
    uint32_t d = ch - '0';
    if (d > 9u)
      digits = false;
    else if (tlen < 11)
      val = val * 10ull + d;
    tlen++;
  }
  *gtc = gt_count;
  *altc = alt_count;
  if (alt_count != 0) *cls = alt_count == gt_count ? BVCF_CLS_HOM : BVCF_CLS_HET;
}

// ---- regular sample region: exactly 4 bytes per sample, "x<sep>y<TAB>" ----
//
// Every dword a lane loads is one sample field.  With t = w ^ "0<sep>0<TAB>":
//   t == 0                      the field is the reference genotype (the common case)
//   t & 0xFFE0FFE0 != 0         separator / TAB bytes differ, or an allele byte is outside
//                               '0'^[0,31]: not a regular field
//   v = allele byte ^ '0'       0..9 for digits, 0x1E for '.'; valid iff bit v of 0x400003FF
// Classes come from a 16-entry x 2-bit table indexed by v & 15 (digit d -> 1 iff d == allele,
// 14 ('.') -> 3): cls = min(code(b0) + code(b2), 3) gives none/het/hom/missing (main.go:1063-1124).
constexpr int kFastGroup = 5;  // chunks per buffer; two buffers => 10 KiB in flight per wave

// v_bfe_u32 and the shifts use only the low 5 bits of their offset operand, so (t << 1) selects entry
// t & 15 of the 2-bit table for byte 0, and (t >> 15) entry (t >> 16) & 15 for byte 2 (bit 0 of that
// offset is bit 7 of the separator xor, zero whenever the frame test passes).
__device__ __forceinline__ uint32_t fast_codes(uint32_t t, uint32_t table) {
  const uint32_t k = __builtin_amdgcn_ubfe(table, t << 1, 2u) + __builtin_amdgcn_ubfe(table, t >> 15, 2u);
  return k < 3u ? k : 3u;
}

// both allele bytes in {0-9, .}: bit (byte ^ '0') of 0x400003FF (the bytes are < 32 when the frame
// test passes; the hardware shift takes the amount mod 32)
__device__ __forceinline__ uint32_t fast_valid(uint32_t t) { return (0x400003FFu >> t) & (0x400003FFu >> (t >> 16)); }

struct FastAcc {
  uint32_t bad, ok, het, hom, miss;
};

constexpr uint32_t kStageChunks = 64;                 // class-map bytes staged in LDS per wave:
constexpr uint32_t kStageBytes = kStageChunks * 64u;  // 64 chunks x 64 B = 4 KiB = 16 384 samples

// all-reference chunks never touch the stage: it is zeroed once per window instead
__device__ __forceinline__ void zero_stage(uint8_t *stage) {
  const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (uint32_t i = 0; i < kStageBytes / (16u * kWave); i++)
    *reinterpret_cast<u32x4 *>(stage + 16u * (lane_id() + i * kWave)) = z;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// write staged class-map bytes [0, n) of the window starting at chunk c_base to the task's map
__device__ __forceinline__ void flush_stage(const uint8_t *stage, uint8_t *cmap, uint32_t c_base, uint32_t n,
                                            uint32_t stride) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t g0 = c_base * 64u;
  if (g0 >= stride) return;
  n = min(n, stride - g0);  // the slot is `stride` bytes (a multiple of 16)
  for (uint32_t i = 16u * lane_id(); i < n; i += 16u * kWave)
    *reinterpret_cast<u32x4 *>(cmap + g0 + i) = *reinterpret_cast<const u32x4 *>(stage + i);
  __builtin_amdgcn_wave_barrier();
}

// one 1 KiB chunk (this lane's 4 fields) of a regular region; class bytes go to the LDS stage
__device__ __forceinline__ void fast_chunk(u32x4 v, uint32_t c, uint32_t n_chunks, uint32_t ns, uint32_t kref,
                                           uint32_t table, uint8_t *cmap, uint8_t *stage, uint32_t stride,
                                           uint32_t term_xor, FastAcc &acc) {
  const int lane = lane_id();
  const uint32_t f0 = c * 256u + 4u * lane;  // sample index of the lane's first dword
  uint32_t t[4] = {v.x ^ kref, v.y ^ kref, v.z ^ kref, v.w ^ kref};
  if (c + 1 == n_chunks) {
    // tail: slots past the last sample count as reference; the last sample's terminator byte
    // (eol or '\r') stands in for its TAB
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (f0 + q >= ns) t[q] = 0;
      if (f0 + q == ns - 1) {
        // term_xor = (expected terminator ^ TAB) << 24; anything above 0xFF000000 = no check
        if (term_xor <= 0xFF000000u) acc.bad |= (t[q] ^ term_xor) & 0xFF000000u;
        t[q] &= 0x00FFFFFFu;
      }
    }
  }
  if (__any((t[0] | t[1] | t[2] | t[3]) != 0)) {
    uint32_t byte = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      acc.bad |= t[q] & 0xFFE0FFE0u;
      acc.ok &= fast_valid(t[q]);
      byte |= fast_codes(t[q], table) << (2 * q);
    }
    const uint32_t lo = byte & 0x55u, hi = (byte >> 1) & 0x55u;
    acc.het += __popc(lo & ~hi);
    acc.hom += __popc(hi & ~lo);
    acc.miss += __popc(lo & hi);
    if (cmap) stage[(c % kStageChunks) * 64u + lane] = (uint8_t)byte;  // the stage starts zeroed
  }
  if (cmap && ((c % kStageChunks) == kStageChunks - 1u || c + 1 == n_chunks)) {
    flush_stage(stage, cmap, c - (c % kStageChunks), ((c % kStageChunks) + 1u) * 64u, stride);
    if (c + 1 != n_chunks) zero_stage(stage);
  }
}

// check_term: also require the byte after the last sample to be the line terminator (the caller
// predicted the end of the line from the region's regular length)
__device__ inline bool gt_scan_fast(const KernelArgs &a, uint32_t s_begin, uint32_t ns, uint32_t allele, uint8_t *cmap,
                                    uint8_t *stage, bool check_term, GtStats *st) {
  const int lane = lane_id();
  const uint32_t table = (allele <= 9 ? (1u << (2u * allele)) : 0u) | (3u << 28);
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;
  const uint32_t last_off = a.cap - 16u;
  const uint8_t *base = a.buf;
  // every chunk of the region ends before the buffer does (the common case): no per-load clamp
  const bool inside = (unsigned long long)s_begin + (unsigned long long)n_chunks * kChunk <= a.cap;
  const uint8_t *lane_base = base + s_begin + 16u * lane;
  auto fetch = [&](uint32_t c) -> u32x4 {
    if (inside) return ld_stream(lane_base + c * kChunk);
    const uint32_t off = min(s_begin + c * kChunk + 16u * lane, last_off);
    return ld_stream(base + off);
  };
  FastAcc acc = {0, 1, 0, 0, 0};
  if (cmap) zero_stage(stage);
  u32x4 va[kFastGroup], vb[kFastGroup];
#pragma unroll
  for (int g = 0; g < kFastGroup; g++)
    if ((uint32_t)g < n_chunks) va[g] = fetch(g);
#pragma unroll
  for (int g = 0; g < kFastGroup; g++)
    if ((uint32_t)(kFastGroup + g) < n_chunks) vb[g] = fetch(kFastGroup + g);
  // the separator of the first field is the line's separator; mixed lines fail the frame test
  const uint32_t sep = (__builtin_amdgcn_readfirstlane(va[0].x) >> 8) & 0xFFu;
  if (sep != '|' && sep != '/') return false;
  const uint32_t kref = 0x09300030u | (sep << 8);
  // the 32-bit compare below cannot be expressed with a 0 sentinel (0 is a valid xor), so "no check"
  // is any value above 0xFF000000
  const uint32_t term_xor = check_term ? ((a.eol_byte ^ 0x09u) << 24) : 0xFFFFFFFFu;

  for (uint32_t c0 = 0; c0 < n_chunks; c0 += 2 * kFastGroup) {
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + g < n_chunks) fast_chunk(va[g], c0 + g, n_chunks, ns, kref, table, cmap, stage, a.cmap_stride, term_xor, acc);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + 2 * kFastGroup + g < n_chunks) va[g] = fetch(c0 + 2 * kFastGroup + g);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + kFastGroup + g < n_chunks) fast_chunk(vb[g], c0 + kFastGroup + g, n_chunks, ns, kref, table, cmap, stage, a.cmap_stride, term_xor, acc);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + 3 * kFastGroup + g < n_chunks) vb[g] = fetch(c0 + 3 * kFastGroup + g);
  }
  if (__any(acc.bad != 0 || !(acc.ok & 1u))) return false;
  wave_sum3(acc.het, acc.hom, acc.miss, ns, &st->n_het, &st->n_hom, &st->n_miss);
  st->ac = st->n_het + 2u * st->n_hom;
  st->an = 2u * (ns - st->n_miss);
  return true;
}

// Any sample region: delimiter masks per lane, wave prefix-sum for the sample index.  A field whose
// first four bytes are "x<sep>y" + (':' | TAB) with x, y in {0-9, .} — the reference's own fast gate,
// main.go:1063-1124, at any stride — is classified from registers (the lane's 16 bytes and the next
// lane's first dword); everything else goes through the byte-serial restatement (classify_field).
// *n_tabs receives the number of TABs in [s_begin, cend).
__device__ inline void gt_scan_general(const KernelArgs &a, uint32_t s_begin, uint32_t cend, uint32_t ns,
                                       uint32_t allele, uint8_t *cmap, GtStats *st, uint32_t *n_tabs) {
  const int lane = lane_id();
  uint32_t a_nd = 1;
  for (uint32_t t = allele; t >= 10; t /= 10) a_nd++;
  const uint32_t table = (allele <= 9 ? (1u << (2u * allele)) : 0u) | (3u << 28);
  if (cmap) {  // zero this allele's map, then OR classes in
    for (uint32_t i = lane * 4u; i < a.cmap_stride; i += kWave * 4u) *reinterpret_cast<uint32_t *>(cmap + i) = 0u;
    __builtin_amdgcn_s_waitcnt(0);  // stores retired before the atomics below touch the same words
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  uint32_t ac = 0, an = 0, het = 0, hom = 0, miss = 0;
  uint32_t tabs_before = 0;    // TABs in earlier chunks
  uint32_t prev_last_tab = 1;  // the byte before the region start behaves like a TAB (field start)
  u32x4 v = {0u, 0u, 0u, 0u};
  if (s_begin < cend) v = load16(a.buf, s_begin + 16u * lane, a.cap);
  for (uint32_t base = s_begin; base < cend; base += kChunk) {
    u32x4 nxt = {0u, 0u, 0u, 0u};
    if (base + kChunk < cend) nxt = load16(a.buf, base + kChunk + 16u * lane, a.cap);  // in flight during this chunk
    const uint32_t off = base + 16u * lane;
    const uint32_t valid = low_bits16((int)cend - (int)off);
    const uint32_t m = eq_mask16(v, '\t') & valid;
    uint32_t tot;
    const uint32_t pre = wave_excl_scan(__popc(m), &tot);
    // field starts: the byte after each TAB, plus the region start
    uint32_t carry = __shfl_up(m >> 15, 1, kWave) & 1u;
    if (lane == 0) carry = prev_last_tab;
    uint32_t starts = ((m << 1) | carry) & valid & 0xFFFFu;
    // bytes 16..19 of this lane's window: the next lane's first dword (next chunk's for lane 63)
    uint32_t d4 = __shfl_down(v.x, 1, kWave);
    const uint32_t nx0 = __shfl(nxt.x, 0, kWave);
    if (lane == kWave - 1) d4 = nx0;
    while (starts) {
      const uint32_t k = __ffs(starts) - 1;
      starts &= starts - 1;
      // sample index = TABs before this byte
      const uint32_t s = tabs_before + pre + __popc(m & ((1u << k) - 1u));
      if (s < ns) {
        uint32_t cls = 0, altc = 0, gtc = 0;
        bool done = false;
        if (off + k + 4u <= cend) {  // four real bytes: c0 c1 c2 c3
          const uint32_t i = k >> 2;
          const uint32_t lo = i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
          const uint32_t hi = i == 0 ? v.y : (i == 1 ? v.z : (i == 2 ? v.w : d4));
          const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, k & 3u);
          const uint32_t c1 = (w >> 8) & 0xFFu, c3 = w >> 24;
          const uint32_t v0 = (w & 0xFFu) ^ '0', v2 = ((w >> 16) & 0xFFu) ^ '0';
          const bool frame = (c1 == '|' || c1 == '/') && (c3 == ':' || c3 == '\t');
          const bool plain = v0 < 32u && v2 < 32u && ((0x400003FFu >> v0) & (0x400003FFu >> v2) & 1u);
          if (frame && plain) {
            const uint32_t code = ((table >> ((v0 & 15u) * 2u)) & 3u) + ((table >> ((v2 & 15u) * 2u)) & 3u);
            cls = code < 3u ? code : 3u;
            gtc = cls == 3u ? 0u : 2u;
            altc = cls == 3u ? 0u : cls;
            done = true;
          }
        }
        if (!done) classify_field(a.buf, off + k, cend, allele, a_nd, &cls, &altc, &gtc);
        ac += altc;
        an += gtc;
        het += cls == BVCF_CLS_HET;
        hom += cls == BVCF_CLS_HOM;
        miss += cls == BVCF_CLS_MISSING;
        if (cmap && cls) atomicOr(reinterpret_cast<uint32_t *>(cmap + (s >> 4) * 4u), cls << (2u * (s & 15u)));
      }
    }
    prev_last_tab = __shfl(m >> 15, kWave - 1, kWave) & 1u;
    tabs_before += tot;
    v = nxt;
  }
  // a field that starts exactly at cend (empty last field) was not visited above
  if (lane == 0) {
    const bool empty_last = (cend == s_begin) || (cend > s_begin && a.buf[cend - 1] == '\t');
    if (empty_last && tabs_before < ns) an += 1;  // "" is one non-matching allele token
  }
  st->ac = wave_sum(ac);
  st->an = wave_sum(an);
  st->n_het = wave_sum(het);
  st->n_hom = wave_sum(hom);
  st->n_miss = wave_sum(miss);
  *n_tabs = tabs_before;
}

// ------------------------------------------------------------------ k_gt: one wave per task

__global__ __launch_bounds__(kWgThreads) void k_gt(KernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerWg][kStageBytes];
  uint8_t *stage = s_stage[threadIdx.x >> 6];
  const int lane = lane_id();
  const uint32_t n_tasks = min(min(a.counters->n_lines, a.max_lines) + a.counters->n_tasks, a.max_tasks);
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t ns = a.n_samples;
  uint32_t ti = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  GtTask nxt = GtTask{};
  if (ti < n_tasks) nxt = a.tasks[ti];
  for (; ti < n_tasks; ti += stride) {
    const GtTask t = nxt;
    if (ti + stride < n_tasks) nxt = a.tasks[ti + stride];  // in flight while this task is scanned
    if (t.allele == 0) continue;  // line rejected before getAlleles: nothing to scan
    uint8_t *cm = t.cmap_off != BVCF_NO_CMAP ? a.cmap + t.cmap_off : nullptr;
    GtStats st = {0, 0, 0, 0, 0};
    uint32_t n_fields;
    // regular region: 4 bytes per sample, every dword of a lane is one "x|y<TAB>" field
    if (t.cend + 1u - t.s_begin == 4u * ns && gt_scan_fast(a, t.s_begin, ns, t.allele, cm, stage, false, &st)) {
      n_fields = ns;
    } else {
      uint32_t tabs;
      gt_scan_general(a, t.s_begin, t.cend, ns, t.allele, cm, &st, &tabs);
      n_fields = tabs + 1u;
    }
    if (lane == 0) {
      GtResult r;
      r.ac = st.ac;
      r.an = st.an;
      r.n_het = st.n_het;
      r.n_hom = st.n_hom;
      r.n_miss = st.n_miss;
      r.n_fields = n_fields;
      r.pad[0] = r.pad[1] = 0;
      a.results[ti] = r;
    }
  }
}

// ------------------------------------------------------------------ k_head: 16 lanes per line

constexpr int kGroup = 16;                       // lanes per line in k_head
constexpr int kGroupsPerWg = kWgThreads / kGroup;
constexpr uint32_t kWindow = kGroup * 16;        // bytes per group step
constexpr uint32_t kHeadStage = 128;             // line-head bytes kept in LDS for the serial phase

__device__ __forceinline__ int glane() { return threadIdx.x & (kGroup - 1); }

// exclusive prefix sum inside a 16-lane group; *total = group sum
__device__ __forceinline__ uint32_t group_excl_scan(uint32_t v, uint32_t *total) {
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < kGroup; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, kGroup);
    if (glane() >= d) inc += t;
  }
  *total = __shfl(inc, kGroup - 1, kGroup);
  return inc - v;
}

__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
  for (int d = kGroup / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kGroup);
  return v;
}

__device__ __forceinline__ uint32_t gbcast0(uint32_t v) { return __shfl(v, 0, kGroup); }

__device__ inline bool filter_in(const Bytes &buf, Span f, const uint16_t *off, const uint16_t *len, uint32_t n,
                                 const uint8_t *text) {
#pragma nounroll
  for (uint32_t i = 0; i < n; i++) {
    if (len[i] != f.len) continue;
    bool eq = true;
#pragma nounroll
    for (uint32_t k = 0; k < f.len && eq; k++) eq = buf[f.off + k] == text[off[i] + k];
    if (eq) return true;
  }
  return false;
}

__device__ inline void write_allele(const KernelArgs &a, uint32_t idx, uint32_t line, uint32_t alt_idx,
                                    const AlleleEval &e, long long pos, uint8_t ref, uint8_t alt_base,
                                    uint8_t site_type, uint32_t task, uint32_t cmap_off) {
  bvcf_allele r;
  r.pos = pos;
  r.line = line;
  r.alt_idx = alt_idx;
  r.alt_off = e.alt_off;
  r.alt_len = e.mnp ? 1u : e.alt_len;
  r.ac = 0;
  r.an = 0;
  r.n_het = 0;
  r.n_hom = 0;
  r.n_miss = 0;
  r.cmap_off = cmap_off;
  r.ref = ref;
  r.alt_base = alt_base;
  r.kind = e.mnp ? (uint8_t)BVCF_ALT_BASE : e.kind;
  r.site_type = site_type;
  r.trtv = (site_type == BVCF_SITE_MULTI || r.kind != BVCF_ALT_BASE) ? 0 : trtv_of(ref, alt_base);
  r.flags = (!e.mnp && e.pos_text) ? BVCF_ALLELE_POS_TEXT : 0;
  r.pad[0] = r.pad[1] = 0;
  r.gt_task = task;
  r.pad2 = 0;
  a.alleles[idx] = r;
}

constexpr uint32_t kNoTask = 0xFFFFFFFFu;

// Write genotype-scan task `ti` (allele == 0 marks a slot without a scan).  Task i < n_lines is
// "line i, ALT #1"; tasks past n_lines are the further ALT indices of multiallelic lines.
__device__ inline void put_task(const KernelArgs &a, uint32_t ti, uint32_t line, uint32_t allele, uint32_t s_begin,
                                uint32_t cend, uint32_t cmap_off) {
  if (ti < a.max_tasks) {
    GtTask t;
    t.line = line;
    t.allele = allele;
    t.s_begin = s_begin;
    t.cend = cend;
    t.cmap_off = cmap_off;
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    a.tasks[ti] = t;
  }
}

// class map of map slot `mi` (census path: slot == task index), or BVCF_NO_CMAP past the arena
__device__ __forceinline__ uint32_t cmap_of(const KernelArgs &a, uint32_t mi, bool want) {
  return (want && ((unsigned long long)mi + 1ull) * a.cmap_stride <= a.max_cmap) ? mi * a.cmap_stride : BVCF_NO_CMAP;
}

// ------------------------------------------------------------------ k_stream: one wave per tile

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kDeferred = 0xFFFFFFFEu;  // StreamEntry.n_miss / GtResult.n_fields: scan left to k_gt

// first terminator byte at a position in [from, limit), or kNone; 4 KiB in flight per step
__device__ inline uint32_t find_eol(const KernelArgs &a, uint32_t from, uint32_t limit) {
  const int lane = lane_id();
  const uint32_t last_off = a.cap - 16u;
  for (uint32_t base = from; base < limit; base += 4u * kChunk) {
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = ld_stream(a.buf + min(base + q * kChunk + 16u * lane, last_off));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t off = base + q * kChunk + 16u * lane;
      const uint32_t m = eq_mask16(v[q], a.eol_byte) & low_bits16((int)limit - (int)off);
      const unsigned long long b = __ballot(m != 0);
      if (b) {
        const int src = __ffsll((long long)b) - 1;
        return __builtin_amdgcn_readfirstlane(__shfl(off + __ffs(m) - 1, src, kWave));
      }
    }
  }
  return kNone;
}

// position of the k-th (0-based) set bit of m; m has more than k bits set
__device__ __forceinline__ uint32_t nth_bit(uint32_t m, uint32_t k) {
  for (uint32_t i = 0; i < k; i++) m &= m - 1;
  return __ffs(m) - 1;
}

// head of a line from one window of bytes: position of the 9th TAB, or kNone with *eolp = first
// terminator seen (kNone if none).  `v` holds 16 B per lane starting at `base`; lanes >= n_lanes hold
// nothing.  found_io carries the TAB count across windows.
__device__ __forceinline__ uint32_t head_window(const KernelArgs &a, u32x4 v, uint32_t base, uint32_t n_lanes,
                                                uint32_t *found_io, uint32_t *eolp) {
  const int lane = lane_id();
  const uint32_t need = 9;
  const uint32_t off = base + 16u * lane;
  uint32_t valid = low_bits16((int)a.nbytes - (int)off);
  if ((uint32_t)lane >= n_lanes) valid = 0;
  const uint32_t me = eq_mask16(v, a.eol_byte) & valid;
  uint32_t mt = eq_mask16(v, '\t') & valid;
  const unsigned long long be = __ballot(me != 0);
  uint32_t eol_here = kNone;
  if (be) {
    const int src = __ffsll((long long)be) - 1;
    eol_here = __builtin_amdgcn_readfirstlane(__shfl(off + __ffs(me) - 1, src, kWave));
    mt &= low_bits16((int)eol_here - (int)off);  // TABs of this line only
  }
  uint32_t tot;
  const uint32_t cnt = __popc(mt);
  const uint32_t prefix = wave_excl_scan(cnt, &tot);
  if (*found_io + tot >= need) {
    const uint32_t target = need - 1 - *found_io;
    const bool mine = prefix <= target && target < prefix + cnt;
    const unsigned long long bm = __ballot(mine);
    const int src = __ffsll((long long)bm) - 1;
    const uint32_t pos = mine ? off + nth_bit(mt, target - prefix) : 0u;
    return __builtin_amdgcn_readfirstlane(__shfl(pos, src, kWave));
  }
  *found_io += tot;
  *eolp = eol_here;
  return kNone;
}

// The same for a 256 B window held by lanes 0..15 (the prefetched head of the next line): a DPP row
// scan replaces the 64-lane shuffle scan, and a terminator anywhere in the window simply declines
// (returns kNone: such a line is shorter than 256 B and goes through the general head scan).
__device__ __forceinline__ uint32_t head_window16(const KernelArgs &a, u32x4 v, uint32_t base) {
  const int lane = lane_id();
  const uint32_t need = 9;
  const uint32_t off = base + 16u * lane;
  uint32_t valid = low_bits16((int)a.nbytes - (int)off);
  if (lane >= 16) valid = 0;
  const uint32_t e4 = a.eol_byte * 0x01010101u;
  const uint32_t eol_any = (zero_bytes(v.x ^ e4) | zero_bytes(v.y ^ e4) | zero_bytes(v.z ^ e4) | zero_bytes(v.w ^ e4));
  if (__ballot(eol_any != 0 && lane < 16)) return kNone;
  const uint32_t mt = eq_mask16(v, '\t') & valid;
  const uint32_t cnt = __popc(mt);
  // inclusive scan inside the row of 16 lanes: row_shr:1,2,4,8 with zero fill
  uint32_t x = cnt;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  const uint32_t prefix = x - cnt;
  const bool mine = lane < 16 && prefix < need && need <= x;  // the 9th TAB is one of this lane's
  const unsigned long long bm = __ballot(mine);
  if (!bm) return kNone;
  const int src = __ffsll((long long)bm) - 1;
  const uint32_t pos = mine ? off + nth_bit(mt, need - 1 - prefix) : 0u;
  return __builtin_amdgcn_readfirstlane(__shfl(pos, src, kWave));
}

constexpr int kPipeChunks = 10;  // chunk registers of the cross-line pipeline: lines of <= 2560 samples

__global__ __launch_bounds__(kWgThreads) void k_stream(KernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerWg][kStageBytes];
  uint8_t *stage = s_stage[threadIdx.x >> 6];
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * kWavesPerWg + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * kWavesPerWg;
  const uint32_t ns = a.n_samples;
  const uint32_t nb = a.nbytes;
  const uint32_t T = a.tile_bytes;
  const bool maps = a.want_cmap != 0;
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;  // of a regular line
  const uint32_t table1 = (1u << 2) | (3u << 28);              // ALT #1
  // cross-line pipelining needs the whole line in the chunk registers and the in-scan terminator check
  const bool pipelined = n_chunks <= (uint32_t)kPipeChunks && a.eol_chars == 1;
  uint32_t cm_next = 0, cm_end = 0;  // this wave's private block of class-map slots
  uint32_t seen = 0;                 // terminated lines this wave walked over

  // A wave owns a contiguous run of tiles and walks it front to back, so only the first tile needs
  // a search for its first line start (those bytes are the previous wave's last line).  Entries
  // stay per tile: the quota argument is about bytes, not about who scans them.
  const uint32_t per_wave = (a.n_tiles + n_waves - 1) / n_waves;
  const uint32_t tile_lo = min(wave * per_wave, a.n_tiles), tile_hi = min(tile_lo + per_wave, a.n_tiles);
  const uint32_t r0 = tile_lo * T;
  const uint32_t r1 = (uint32_t)min((unsigned long long)tile_hi * T, (unsigned long long)nb);
  uint32_t tile = tile_lo, n_local = 0;
  uint32_t p = kNone;
  if (tile_lo < tile_hi) {
    p = 0;
    if (r0 > 0) {
      const uint32_t q = find_eol(a, r0 - 1, r1);  // a terminator at r1 - 1 starts a line of the next run
      p = q == kNone ? kNone : q + 1;
    }
  }

  // class-map slot for the next listed line
  auto map_slot = [&]() -> uint32_t {
    if (!maps) return BVCF_NO_CMAP;
    if (cm_next == cm_end) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&a.counters->cmap_maps, 16u);
      cm_next = __builtin_amdgcn_readfirstlane(b);
      cm_end = cm_next + 16u;
    }
    return cmap_of(a, cm_next, true);
  };
  // list a line (in input order) in the tile it starts in
  auto commit = [&](uint32_t ls, uint32_t cend, const GtStats &st, bool deferred, uint32_t cm_off) {
    while (ls >= (tile + 1) * T) {  // ls moved into a later tile of the run
      if (lane == 0) a.census[tile] = n_local;
      tile++;
      n_local = 0;
    }
    if (n_local >= a.tile_quota) {
      if (lane == 0) a.counters->pad[0] = 1;  // cannot happen: see tile_quota
      return;
    }
    if (lane == 0) {
      StreamEntry en;
      en.ls = ls;
      en.len = cend - ls;
      en.ac = st.ac;
      en.an = st.an;
      en.n_het = st.n_het;
      en.n_hom = st.n_hom;
      en.n_miss = deferred ? kDeferred : st.n_miss;
      en.cmap_off = cm_off;
      a.entries[(size_t)tile * a.tile_quota + n_local] = en;
    }
    n_local++;
    if (maps) cm_next++;
  };
  auto chunk_at = [&](uint32_t s_begin, uint32_t c) -> u32x4 {
    const uint32_t off = min(s_begin + c * kChunk + 16u * lane, a.cap - 16u);
    return ld_stream(a.buf + off);
  };
  auto finish_stats = [&](const FastAcc &acc, GtStats *st) {
    wave_sum3(acc.het, acc.hom, acc.miss, ns, &st->n_het, &st->n_hom, &st->n_miss);
    st->ac = st->n_het + 2u * st->n_hom;
    st->an = 2u * (ns - st->n_miss);
  };

  while (p != kNone && p < r1) {
    // ---- fixed columns: the 9th TAB, or the terminator if it comes first (main.go:535)
    uint32_t found = 0, tab9 = kNone, eolp = kNone;
    for (uint32_t base = p; base < nb; base += kChunk) {
      tab9 = head_window(a, load16(a.buf, base + 16u * lane, a.cap), base, kWave, &found, &eolp);
      if (tab9 != kNone || eolp != kNone) break;
    }
    if (tab9 == kNone) {
      if (eolp == kNone) break;  // unterminated tail of the block: dropped (main.go:354-358)
      seen++;                    // fewer than 10 fields: cannot pass linePasses
      p = eolp + 1;
      continue;
    }
    uint32_t s_begin = tab9 + 1;
    GtStats st = {0, 0, 0, 0, 0};
    const unsigned long long pred = (unsigned long long)s_begin + 4ull * ns - 1ull;  // predicted content end
    uint32_t cend = kNone;

    if (pred + a.eol_chars <= nb && pipelined) {
      // ================= cross-line pipeline over consecutive regular lines =================
      // A = the line being scanned (chunks in va), B = the next one: its head window (hv) is
      // requested before A's chunks, parsed as soon as A starts, and every chunk register is
      // re-issued for B right after A's chunk in it has been processed.
      uint32_t pA = p, sA = s_begin, peA = (uint32_t)pred;
      u32x4 va[kPipeChunks];
      u32x4 hv = {0u, 0u, 0u, 0u};
      bool hv_ok = peA + 1u < r1;  // B starts inside this wave's run
      if (hv_ok && lane < 16) hv = load16(a.buf, peA + 1u + 16u * lane, a.cap);
#pragma unroll
      for (int g = 0; g < kPipeChunks; g++)
        if ((uint32_t)g < n_chunks) va[g] = chunk_at(sA, g);
      for (;;) {
        // ---- B's head from the 256 B window
        uint32_t sB = 0, peB = 0;
        bool b_ok = false;
        if (hv_ok) {
          const uint32_t t9 = head_window16(a, hv, peA + 1u);
          if (t9 != kNone) {
            sB = t9 + 1;
            const unsigned long long pb = (unsigned long long)sB + 4ull * ns - 1ull;
            if (pb + 1ull <= nb) {
              peB = (uint32_t)pb;
              b_ok = true;
            }
          }
        }
        const bool hvc_ok = b_ok && peB + 1u < r1;
        if (hvc_ok && lane < 16) hv = load16(a.buf, peB + 1u + 16u * lane, a.cap);  // C's head, ahead of B's chunks
        // ---- scan A, re-issuing each register for B
        const uint32_t cmA = map_slot();
        uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;
        if (cm) zero_stage(stage);
        FastAcc acc = {0, 1, 0, 0, 0};
        const uint32_t sep = (__builtin_amdgcn_readfirstlane(va[0].x) >> 8) & 0xFFu;
        if (sep != '|' && sep != '/') acc.bad = 1;
        const uint32_t kref = 0x09300030u | (sep << 8);
        const uint32_t term_xor = (a.eol_byte ^ 0x09u) << 24;
#pragma unroll
        for (int g = 0; g < kPipeChunks; g++) {
          if ((uint32_t)g < n_chunks) {
            fast_chunk(va[g], g, n_chunks, ns, kref, table1, cm, stage, a.cmap_stride, term_xor, acc);
            if (b_ok) va[g] = chunk_at(sB, g);
          }
        }
        if (__any(acc.bad != 0 || !(acc.ok & 1u))) {
          // A is not regular after all: B was predicted from a wrong line end.  Leave the
          // pipeline (the loads in flight are simply dropped) and take A the slow way.
          s_begin = sA;
          p = pA;
          break;
        }
        finish_stats(acc, &st);
        seen++;
        commit(pA, peA, st, false, cmA);
        p = peA + 1u;
        if (!b_ok) {
          s_begin = kNone;  // nothing pending: rediscover from p
          break;
        }
        pA = peA + 1u;
        sA = sB;
        peA = peB;
        hv_ok = hvc_ok;
      }
      if (s_begin == kNone) continue;
      // fall through with (p, s_begin) of the line that failed the regular scan
    } else if (pred + a.eol_chars <= nb) {
      // ---- one line at a time (more than kPipeChunks chunks per line, or "\r\n")
      const uint32_t pe = (uint32_t)pred;
      bool term = true;
      if (a.eol_chars == 2) term = a.buf[pe + 1] == a.eol_byte && a.buf[pe] != a.eol_byte;
      if (term) {
        const uint32_t cm_off = map_slot();
        uint8_t *cm = cm_off != BVCF_NO_CMAP ? a.cmap + cm_off : nullptr;
        if (gt_scan_fast(a, s_begin, ns, 1, cm, stage, a.eol_chars == 1, &st)) {
          seen++;
          commit(p, pe, st, false, cm_off);
          p = pe + a.eol_chars;
          continue;
        }
      }
    }

    // ---- not a regular "x|y<TAB>" region: only find where the line ends here; its ALT #1 scan is
    // left to k_gt (k_head turns the entry into a task), which also settles its field count
    {
      const uint32_t e = find_eol(a, s_begin, nb);
      if (e == kNone) break;  // unterminated tail
      seen++;
      if (e + 1 < s_begin + a.eol_chars) {
        // chomping numChars bytes (main.go:535) eats the 9th TAB: at most 9 fields remain
        p = e + 1;
        continue;
      }
      cend = e + 1 - a.eol_chars;
      // a line shorter than n_header - 1 bytes cannot have n_header fields: never listed (this
      // is what bounds the per-tile quota)
      const GtStats none = {0, 0, 0, 0, 0};
      if (cend - p + 1u >= a.n_header) commit(p, cend, none, true, map_slot());
      p = cend + a.eol_chars;
    }
  }
  for (; tile < tile_hi; tile++) {  // the rest of the run has no line starts
    if (lane == 0) a.census[tile] = n_local;
    n_local = 0;
  }
  if (lane == 0 && seen) atomicAdd(&a.counters->lines_seen, seen);
}

// tile-local entries -> input order (the exclusive scan of the tile counts is in census/group_base)
__global__ __launch_bounds__(kWgThreads) void k_order(KernelArgs a) {
  const uint32_t total = a.n_tiles * a.tile_quota;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t tile = i / a.tile_quota, k = i % a.tile_quota;
    const uint32_t first = a.census[tile] + a.group_base[tile / kScanGroup];
    const uint32_t next = (tile + 1 < a.n_tiles)
                              ? a.census[tile + 1] + a.group_base[(tile + 1) / kScanGroup]
                              : a.counters->n_lines;
    if (k >= next - first) continue;
    const uint32_t g = first + k;
    if (g >= a.max_lines) continue;
    const StreamEntry en = a.entries[i];
    a.line_off[g] = en.ls;
    a.line_len[g] = en.len;
    a.line_cmap[g] = en.cmap_off;
    if (g < a.max_tasks) {
      GtResult r;
      r.ac = en.ac;
      r.an = en.an;
      r.n_het = en.n_het;
      r.n_hom = en.n_hom;
      r.n_miss = en.n_miss;
      r.n_fields = en.n_miss == kDeferred ? kDeferred : a.n_header - 9u;
      r.pad[0] = r.pad[1] = 0;
      a.results[g] = r;
    }
  }
}

// k_head handles 256 lines per workgroup step in two phases:
//   T  tokenise: 16 lanes per line find the TABs of the fixed columns (per-lane masks, 16-lane
//      prefix sum) and stage the first kHeadStage bytes of the line in LDS; 16 rounds x 16 lines
//   S  serial:   ONE LANE PER LINE runs the gate + getAlleles on the staged bytes, so a wave
//      instruction serves 64 lines (with 16 lanes per line it served 4 and the kernel was
//      issue-bound on this code)
constexpr uint32_t kLinesPerStep = kWgThreads;
constexpr uint32_t kHeadRow = kHeadStage / 4 + 1;  // dwords per staged line; odd => conflict-free columns
constexpr uint32_t kTabRow = 11;                   // 9 TAB offsets + pad, odd stride

__global__ __launch_bounds__(kWgThreads) void k_head(KernelArgs a) {
  __shared__ uint32_t s_head[kLinesPerStep * kHeadRow];
  __shared__ uint32_t s_tab[kLinesPerStep * kTabRow];
  __shared__ uint32_t s_ls[kLinesPerStep], s_len[kLinesPerStep], s_found[kLinesPerStep], s_staged[kLinesPerStep],
      s_extra[kLinesPerStep];
  __shared__ uint32_t s_wave[kWavesPerWg][2];
  __shared__ uint32_t s_base[3];
  __shared__ FilterTable s_ft;  // FILTER sets
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.filters);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ft);
    for (uint32_t i = threadIdx.x; i < sizeof(FilterTable) / 4; i += kWgThreads) dst[i] = src[i];
  }
  const int gl = glane();
  const int g = threadIdx.x / kGroup;
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t stride = gridDim.x * kLinesPerStep;
  const uint32_t need = min(9u, a.n_header - 1u);  // TABs that bound the fixed columns we read
  const uint32_t ns = a.n_samples;
  const bool maps = a.want_cmap && ns > 0;

  for (uint32_t line0 = blockIdx.x * kLinesPerStep; line0 < n_lines; line0 += stride) {
    __syncthreads();  // LDS of the previous step is free (also covers the s_ft copy)

    // ================= phase T: 16 lanes per line =================
    // lane gl of a group fetches the offsets of the group's round-gl line, so the 16 rounds' offsets
    // are in flight together; the first window of round r + 1 is requested before round r is parsed
    uint32_t my_ls = 0, my_len = 0;
    {
      const uint32_t l = line0 + (uint32_t)gl * kGroupsPerWg + g;
      if (l < n_lines) {
        my_ls = a.line_off[l];
        if (a.fused) {
          my_len = a.line_len[l];
        } else {
          const uint32_t le = a.line_off[l + 1];
          my_len = le - my_ls >= a.eol_chars ? le - my_ls - a.eol_chars : 0u;  // chomp, main.go:535
        }
      }
    }
    u32x4 v_next = load16(a.buf, __shfl(my_ls, 0, kGroup) + 16u * gl, a.cap);
    for (uint32_t r = 0; r < kLinesPerStep / kGroupsPerWg; r++) {
      const uint32_t ll = r * kGroupsPerWg + g;
      const uint32_t line = line0 + ll;
      const uint32_t ls = __shfl(my_ls, r, kGroup);
      const uint32_t len = __shfl(my_len, r, kGroup);
      const u32x4 v_first = v_next;
      if (r + 1 < kLinesPerStep / kGroupsPerWg) v_next = load16(a.buf, __shfl(my_ls, r + 1, kGroup) + 16u * gl, a.cap);
      if (line >= n_lines) continue;
      const uint32_t cend = ls + len;
      uint32_t found = 0, base = ls;
      // strings.Split(row, "\t") for the fixed columns, main.go:535
      for (; base < cend && found < need; base += kWindow) {
        const uint32_t off = base + 16u * gl;
        u32x4 v = base == ls ? v_first : load16(a.buf, off, a.cap);
        const uint32_t rel = off - ls;
        if (rel < kHeadStage) {
          uint32_t *row = &s_head[ll * kHeadRow + rel / 4];
          row[0] = v.x;
          row[1] = v.y;
          row[2] = v.z;
          row[3] = v.w;
        }
        uint32_t m = eq_mask16(v, '\t') & low_bits16((int)cend - (int)off);
        uint32_t tot;
        uint32_t rk = found + group_excl_scan(__popc(m), &tot);
        while (m && rk < need) {
          s_tab[ll * kTabRow + rk] = off + __ffs(m) - 1;
          m &= m - 1;
          rk++;
        }
        found += tot;
      }
      const uint32_t staged = min(base - ls, kHeadStage);
      uint32_t extra = 0;
      if (ns == 0 && found >= need) {
        // no samples: every TAB after the last fixed column is an extra field; `found` already
        // counts the TABs of the windows read so far
        for (; base < cend; base += kWindow) {
          const uint32_t off = base + 16u * gl;
          u32x4 v = load16(a.buf, off, a.cap);
          extra += __popc(eq_mask16(v, '\t') & low_bits16((int)cend - (int)off));
        }
        extra = group_sum(extra);
      }
      if (gl == 0) {
        s_ls[ll] = ls;
        s_len[ll] = len;
        s_found[ll] = found;
        s_staged[ll] = staged;
        s_extra[ll] = extra;
      }
    }
    __syncthreads();

    // ================= phase S: one lane per line =================
    const uint32_t ll = threadIdx.x;
    const uint32_t line = line0 + ll;
    const bool active = line < n_lines;
    const uint32_t ls = active ? s_ls[ll] : 0u, len = active ? s_len[ll] : 0u, found = active ? s_found[ll] : 0u;
    const uint32_t cend = ls + len;
    const uint32_t *tab = &s_tab[ll * kTabRow];
    Bytes hb;
    hb.g = a.buf;
    hb.lds = reinterpret_cast<const uint8_t *>(&s_head[ll * kHeadRow]);
    hb.lo = ls;
    hb.n = active ? s_staged[ll] : 0u;

    uint32_t status = BVCF_LINE_OK;
    uint32_t n_fields = 0;
    if (active && found < need) {
      status = BVCF_LINE_FIELDS;
      n_fields = found + 1;
    } else if (active && ns == 0) {
      n_fields = found + s_extra[ll] + 1;
      if (n_fields != a.n_header) status = BVCF_LINE_FIELDS;
    }

    // field i = [fstart(i), tab[i]) ; fields beyond the line: empty at cend
    auto fspan = [&](uint32_t i) -> Span {
      Span sp;
      sp.off = i == 0 ? ls : tab[i - 1] + 1;
      const uint32_t e = i < need ? tab[i] : cend;
      sp.len = e - sp.off;
      return sp;
    };

    uint32_t rec_first = 0, n_rec = 0, site_type = 0;
    bool task_written = false, primary_written = false;

    // ---- part 1: gate and what the line will need
    AlleleCtx c;
    uint32_t mode = 0, n_commas = 0, bound = 0, s_begin = cend;
    if (active && status == BVCF_LINE_OK && a.n_header > 6) {
      // FILTER gate, main.go:447-454
      const FilterTable *ft = &s_ft;
      Span f = fspan(6);
      if (!ft->allow_nil && !filter_in(hb, f, ft->allow_off, ft->allow_len, ft->allow_n, ft->text))
        status = BVCF_LINE_FILTER;
      else if (!ft->deny_nil && filter_in(hb, f, ft->deny_off, ft->deny_len, ft->deny_n, ft->text))
        status = BVCF_LINE_FILTER;
    }
    const bool eval = active && status == BVCF_LINE_OK;
    if (eval) {
      // getAlleles set-up, main.go:723-735
      c.buf = hb;
      c.chrom = fspan(0);
      c.pos = fspan(1);
      c.ref = fspan(3);
      c.alt = fspan(4);
      c.int_pos = 0;
      c.pos_bad = false;
      c.line = line;
      s_begin = need == 9 ? tab[8] + 1 : cend;
      // mode 0: REF == ALT; 1: single-byte ALT path; 2: ALT token loop; 3: empty REF (Go panics)
      // bound: a token yields one record, or one per differing base when it is as long as a
      // multi-base REF (main.go:855-873)
      bool same = c.alt.len == c.ref.len;
      uint32_t tl = 0, b2 = 0;
#pragma nounroll
      for (uint32_t i = 0; i <= c.alt.len; i++) {
        const uint8_t ch = i < c.alt.len ? hb[c.alt.off + i] : (uint8_t)',';
        if (i < c.alt.len && same) same = ch == hb[c.ref.off + i];
        if (ch == ',') {
          n_commas += i < c.alt.len;
          b2 += (tl == c.ref.len && c.ref.len > 1) ? c.ref.len : 1u;
          tl = 0;
        } else {
          tl++;
        }
      }
      mode = same ? 0u : (c.alt.len == 1 ? 1u : (c.ref.len == 0 ? 3u : 2u));
      bound = mode == 1 ? 1u : (mode == 2 ? b2 : 0u);
    }

    // ---- slot reservation, once per workgroup step: record slot `line` and task slot `line` are
    // the line's own; only further records / ALT indices draw from the batch counters.  Biallelic
    // lines — all of a 1KG-shaped file — never touch an atomic.
    const uint32_t want_rec = bound > 1 ? bound - 1 : 0u;
    const uint32_t want_task = (eval && ns > 0 && mode == 2) ? n_commas : 0u;
    uint32_t wt_rec, wt_task;
    uint32_t extra_base = wave_excl_scan(want_rec, &wt_rec);
    uint32_t task_base = wave_excl_scan(want_task, &wt_task);
    if (lane == 0) {
      s_wave[w][0] = wt_rec;
      s_wave[w][1] = wt_task;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      uint32_t sum = 0;
      for (int k = 0; k < kWavesPerWg; k++) sum += s_wave[k][threadIdx.x];
      uint32_t got = 0;
      if (sum) got = atomicAdd(threadIdx.x == 0 ? &a.counters->n_alleles : &a.counters->n_tasks, sum);
      s_base[threadIdx.x] = got;
      // streaming path: the class maps of the extra tasks come from the same cursor k_stream used
      if (threadIdx.x == 1) s_base[2] = (sum && a.fused && maps) ? atomicAdd(&a.counters->cmap_maps, sum) : 0u;
    }
    __syncthreads();
    uint32_t task_rank = task_base;  // this line's first extra task, counted inside the workgroup
    for (int k = 0; k < w; k++) {
      extra_base += s_wave[k][0];
      task_rank += s_wave[k][1];
    }
    extra_base += n_lines + s_base[0];
    task_base = n_lines + s_base[1] + task_rank;
    const uint32_t map_base = a.fused ? s_base[2] + task_rank : task_base;

    // ---- part 2: evaluate the ALT tokens, write records and scan tasks
    if (eval) {
      if (mode == 0) log_err(a, line, 0, BVCF_ERR_SAME);
      if (mode == 3) log_err(a, line, 0, BVCF_ERR_EMPTY_REF);
      const bool fits = (unsigned long long)extra_base + want_rec <= a.max_alleles;
      // slot of this line's j-th record
      auto slot = [&](uint32_t j) -> uint32_t { return j == 0 ? line : extra_base + j - 1; };

      // With samples, the scan for ALT #1 always runs: it also settles len(record) == len(header).
      // On the streaming path k_stream has already done it (results[line], line_cmap[line]).
      uint32_t cm0 = BVCF_NO_CMAP;
      if (ns > 0 && !a.fused) {
        cm0 = cmap_of(a, line, maps && (mode == 1 || mode == 2));
        put_task(a, line, line, 1, s_begin, cend, cm0);
        task_written = true;
      }
      if (ns > 0 && a.fused) {
        if (maps) cm0 = a.line_cmap[line];
        if (line < a.max_tasks && a.results[line].n_fields == kDeferred) {  // k_stream left the scan to k_gt
          put_task(a, line, line, 1, s_begin, cend, cm0);
          task_written = true;
        }
      }

      uint32_t cur = 0, emitted = 0, tasks_used = 0;
      if (mode == 1 || mode == 2) {
#pragma nounroll
        for (uint32_t k = 0;; k++) {
          AlleleEval e;
          Span t;
          if (mode == 1) {
            if (k > 0) break;
            eval_single(c, e);
            t = c.alt;
          } else {
            if (!next_token(c, &cur, &t)) break;
            eval_token(c, t, e);
          }
          if (e.err) log_err(a, line, (e.err == BVCF_ERR_POS) ? 0u : k + 1u, e.err);
          if (e.stop) break;
          if (!e.n) continue;
          uint32_t task = line, cm_off = cm0;
          if (ns > 0 && k > 0) {
            task = task_base + tasks_used;
            cm_off = cmap_of(a, map_base + tasks_used, maps);
            put_task(a, task, line, k + 1, s_begin, cend, cm_off);
            tasks_used++;
          }
          if (ns == 0) task = kNoTask;
          if (fits) {
            // type call, main.go:1004-1037 (single-ALT path: main.go:743,764)
            uint8_t stype;
            if (n_commas > 0)
              stype = BVCF_SITE_MULTI;
            else if (!e.mnp && e.kind == BVCF_ALT_DEL)
              stype = BVCF_SITE_DEL;
            else if (!e.mnp && e.kind == BVCF_ALT_INS)
              stype = BVCF_SITE_INS;
            else
              stype = e.n > 1 ? BVCF_SITE_MNP : BVCF_SITE_SNP;
            site_type = stype;
            if (e.mnp) {
              uint32_t j = 0;
#pragma nounroll
              for (uint32_t i = 0; i < c.ref.len; i++) {
                const uint8_t rb = hb[c.ref.off + i], ab = hb[t.off + i];
                if (rb == ab) continue;
                write_allele(a, slot(emitted + j), line, k, e, c.int_pos + (long long)i, rb, ab, stype, task, cm_off);
                j++;
              }
            } else {
              write_allele(a, slot(emitted), line, k, e, e.pos, e.ref, e.alt_base, stype, task, cm_off);
            }
          }
          emitted += e.n;
        }
      }
      // reserved but unused slots must not look like records / tasks to the later kernels
      if (fits)
#pragma nounroll
        for (uint32_t j = emitted > 1 ? emitted : 1; j < bound; j++) a.alleles[slot(j)].gt_task = kNoTask;
#pragma nounroll
      for (uint32_t j = tasks_used; j < want_task; j++) put_task(a, task_base + j, line, 0, cend, cend, BVCF_NO_CMAP);
      if (emitted) primary_written = true;
      if (fits) rec_first = extra_base;
      if (emitted == 0)
        status = BVCF_LINE_NOALLELE;  // k_finish may still turn this into FIELDS
      else if (fits)
        n_rec = emitted;
      n_fields = 0;  // settled by k_finish from the scan when there are samples
      if (ns == 0) n_fields = a.n_header;
    }

    // ---- line record
    if (active) {
      bvcf_line L;
      L.off = ls;
      L.len = len;
#pragma unroll
      for (uint32_t i = 0; i < 9; i++) L.fend[i] = (i < need && i < found) ? tab[i] - ls : len;
      L.rec_first = rec_first;
      L.n_rec = n_rec;
      L.n_fields = n_fields;
      L.gt_task = line;
      L.status = (uint8_t)status;
      L.site_type = (uint8_t)site_type;
      L.pad[0] = L.pad[1] = 0;
      a.lines[line] = L;
      // every line owns task slot `line` and record slot `line`: mark the ones it did not fill
      if (ns > 0 && !task_written) put_task(a, line, line, 0, cend, cend, BVCF_NO_CMAP);
      if (!primary_written && line < a.max_alleles) a.alleles[line].gt_task = kNoTask;
    }
  }
}

// ------------------------------------------------------------------ k_finish

// One thread per line and per allele record: the field-count half of linePasses (main.go:449) from
// the scan of ALT #1, and the scan results copied into the records that reference them.
__global__ __launch_bounds__(kWgThreads) void k_finish(KernelArgs a) {
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_alleles = min(n_lines + a.counters->n_alleles, a.max_alleles);
  const uint32_t n_tasks = min(n_lines + a.counters->n_tasks, a.max_tasks);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nthreads = gridDim.x * blockDim.x;
  if (a.n_samples == 0) return;
  for (uint32_t i = tid; i < n_lines; i += nthreads) {
    bvcf_line *L = &a.lines[i];
    const uint32_t st = L->status;
    if (st != BVCF_LINE_OK && st != BVCF_LINE_NOALLELE) continue;
    const uint32_t nf = 9u + a.results[i].n_fields;
    L->n_fields = nf;
    if (nf != a.n_header) {
      L->status = BVCF_LINE_FIELDS;
      L->n_rec = 0;
    }
  }
  for (uint32_t i = tid; i < n_alleles; i += nthreads) {
    bvcf_allele *r = &a.alleles[i];
    const uint32_t t = r->gt_task;
    if (t >= n_tasks) continue;  // kNoTask: slot without a record
    const GtResult g = a.results[t];
    r->ac = g.ac;
    r->an = g.an;
    r->n_het = g.n_het;
    r->n_hom = g.n_hom;
    r->n_miss = g.n_miss;
  }
}

}  // namespace bvcf_dev
