// bvcf_gtscan.hip.h — makeHetHomozygotes: fast (regular) and general genotype scans, k_gt (main.go:1042-1194)
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include "bvcf_common.hip.h"
#include "bvcf_alleles.hip.h"

namespace bvcf_dev {

// ------------------------------------------------------------------ GT scan (whole wave)

// Exact restatement of one sample field of makeHetHomozygotes (main.go:1057-1190) for the
// allele whose decimal text is itoa(a): byte-serial, used for irregular lines.
// getc(i) = byte i of the field, '\t' from the field's end on.  *max_allele (optional): the highest allele number among
// the field's tokens that are plain decimal numbers, 15 at most (what k_stream_gen needs to know before it promises
// k_head that no sample carries a further ALT index).
template <class G>
__device__ inline void classify_field_g(G getc, uint32_t a, uint32_t a_ndigits, uint32_t *cls, uint32_t *altc, uint32_t *gtc,
                                        uint32_t *max_allele = nullptr) {
  *altc = 0;
  *gtc = 0;
  *cls = BVCF_CLS_NONE;
  if (max_allele) *max_allele = 0;
  // fast gate, main.go:1063-1064: (len == 3 || g[3] == ':') && g[1] in {'|','/'}
  uint32_t c0 = getc(0), c1 = '\t', c2 = '\t', c3 = '\t';
  if (c0 != '\t') {
    c1 = getc(1);
    if (c1 != '\t') {
      c2 = getc(2);
      if (c2 != '\t') c3 = getc(3);
    }
  }
  const bool have3 = c0 != '\t' && c1 != '\t' && c2 != '\t';
  if (have3 && (c3 == '\t' || c3 == ':') && (c1 == '|' || c1 == '/')) {
    if (max_allele) {
      const uint32_t d0 = c0 - '0', d2 = c2 - '0';
      *max_allele = max(d0 <= 9u ? d0 : 0u, d2 <= 9u ? d2 : 0u);
    }
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
    uint32_t ch = getc(nf);
    if (ch == '\t' || ch == ':') break;
    has_bar |= ch == '|';
    has_slash |= ch == '/';
  }
  const uint32_t sep = has_bar ? '|' : (has_slash ? '/' : 0xFFFFFFFFu);
  uint32_t alt_count = 0, gt_count = 0, top = 0;
  // token state
  uint32_t tlen = 0;
  unsigned long long val = 0;
  bool digits = true, lead0 = false, dot = false;
  #pragma nounroll
  for (uint32_t k = 0; k <= nf; k++) {
    uint32_t ch = k < nf ? getc(k) : sep;
    if (k == nf || ch == sep) {
      if (tlen == 1 && dot) {  // allele == "." => whole sample missing, nothing counted
        *cls = BVCF_CLS_MISSING;
        if (max_allele) *max_allele = top;
        return;
      }
      if (tlen >= 1 && tlen <= 10 && digits && !lead0) {
        if (val == (unsigned long long)a) alt_count++;
        top = max(top, (uint32_t)(val < 15ull ? val : 15ull));
      }
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
  if (max_allele) *max_allele = top;
  if (alt_count != 0) *cls = alt_count == gt_count ? BVCF_CLS_HOM : BVCF_CLS_HET;
}
// p = field start, cend = end of line content; a field ends at '\t' or cend.
__device__ inline void classify_field(const uint8_t *buf, uint32_t p, uint32_t cend, uint32_t a, uint32_t a_ndigits,
                                      uint32_t *cls, uint32_t *altc, uint32_t *gtc, uint32_t *max_allele = nullptr) {
  classify_field_g([&](uint32_t i) -> uint32_t { return p + i < cend ? (uint32_t)buf[p + i] : (uint32_t)'\t'; }, a, a_ndigits, cls, altc, gtc,
                   max_allele);
}

// ---- regular sample region: exactly 4 bytes per sample, "x<sep>y<TAB>" ----
//
// Every dword a lane loads is one sample field.  With t = w ^ "0<sep>0<TAB>":
//   t == 0                      the field is the reference genotype (the common case)
//   t & 0xFFE0FFE0 != 0         separator / TAB bytes differ, or an allele byte is outside
//                               '0'^[0,31]: not a regular field
//   v = allele byte ^ '0'       0..9 for digits, 0x1E for '.'
// The allele characters of four fields are gathered into two dwords (first alleles, second alleles) with v_perm and
// classified byte-parallel: a byte equal to the counted allele's digit scores one, '.' in either position makes the
// field missing, anything outside {0-9, .} fails the line over to the general scan (main.go:1063-1124).
constexpr int kFastGroup = 5;  // chunks per buffer; two buffers => 10 KiB in flight per wave

struct FastAcc {
  uint32_t bad, het, hom, miss;
  uint32_t n_sp;  // wave-uniform: entries in the raw list of the line; > BVCF_CMAP_SPARSE_MAX once the line went dense
  uint32_t hi = 0;  // dense mode: allele bytes seen that may be a digit >= 2 (or a dot): bits of tor & 0x000E000E
  // wave-uniform, dense mode with a raw list at hand: the list now collects only the lanes that hold a digit >= 2 or a
  // dot -- what the class lists of the further ALT indices are made of (finish_dense) --; kDenseMode once they outgrew it
  uint32_t n_oth = 0;
};

// Most alleles of a cohort file are carried by a handful of samples.  A line therefore starts in LIST MODE: a lane
// whose four fields are not all the reference genotype only appends them, untouched, to a short list in LDS
// (RawList: the xor-ed field words and the map byte index) -- no classification, no class-map staging.  When the
// line ends with at most BVCF_CMAP_SPARSE_MAX such lanes, lanes 0..n-1 classify one entry each (finish_list): the
// counts, the class list of ALT #1 (BVCF_ALLELE_CMAP_SPARSE) -- and, from the same few entries, the class lists of
// EVERY further ALT index the samples carry (main.go:549-556 rescans the line once per allele; here a multiallelic
// line is read once).  A line that outgrows the list is replayed into the LDS stage and continues as a dense map
// (then further ALT indices are left to k_gt, as are all alleles of irregular lines).
constexpr uint32_t kSparseWords = 16;  // words of one class list: count + BVCF_CMAP_SPARSE_MAX entries
constexpr uint32_t kRawMax = 63;       // entries a line may collect in list mode: one lane each in finish_list
constexpr uint32_t kDenseMode = kRawMax + 1u;
constexpr uint32_t kListAlleles = 8;   // ALT indices 1..8 can get a class list from finish_list (3 bits in the entry)
struct RawList {
  u32x4 t[kRawMax + 1];        // entry i: the lane's four field words ^ "0<sep>0<TAB>"
  uint32_t idx[kRawMax + 1];   // ... and its class-map byte index (chunk * 64 + lane)
};

constexpr uint32_t kWideSamples = BVCF_WIDE_SAMPLES;  // from here up the census path splits a line's regular scan over waves (k_gt_wide)
constexpr uint32_t kStageChunks = 64;                 // class-map bytes staged in LDS per wave:
constexpr uint32_t kStageBytes = kStageChunks * 64u;  // 64 chunks x 64 B = 4 KiB = 16 384 samples

// all-reference chunks never touch the stage: it is zeroed once per window instead
// (only the part the next `n_chunks` chunks can touch: 1 KiB of stage per 16 chunks)
__device__ __forceinline__ void zero_stage(uint8_t *stage, uint32_t n_chunks = kStageChunks) {
  const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (uint32_t i = 0; i < kStageBytes / (16u * kWave); i++)
    if (i * 16u < n_chunks) *reinterpret_cast<u32x4 *>(stage + 16u * (lane_id() + i * kWave)) = z;
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
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(stage + i),
                                reinterpret_cast<u32x4 *>(cmap + g0 + i));  // written once, read by the host
  __builtin_amdgcn_wave_barrier();
}

// The lane's four fields at once, one byte lane per field.  t[q] = field word ^ "0<sep>0<TAB>" with the frame bytes
// already checked: byte 0 / byte 2 are the allele characters ^ '0' (0..9 for digits, 0x1E for '.'; below 32 whenever
// the frame test passes, which makes the carry-free zero-byte test ~((x + 0x7F..) | x) & 0x80.. exact).
// A = first alleles, B = second alleles of the four fields.
struct Alleles4 {
  uint32_t A, B, dA, dB;  // dA / dB: 0x80 in every byte that is '.'
};
__device__ __forceinline__ uint32_t zero_b(uint32_t x) { return ~((x + 0x7F7F7F7Fu) | x) & 0x80808080u; }
__device__ __forceinline__ Alleles4 gather4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
  const uint32_t p01 = __builtin_amdgcn_perm(t1, t0, 0x06020400u);  // a0 a1 b0 b1
  const uint32_t p23 = __builtin_amdgcn_perm(t3, t2, 0x06020400u);  // a2 a3 b2 b3
  Alleles4 g;
  g.A = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
  g.B = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
  g.dA = zero_b(g.A ^ 0x1E1E1E1Eu);
  g.dB = zero_b(g.B ^ 0x1E1E1E1Eu);
  return g;
}
// alphabet: every allele byte is a digit (<= 9) or '.'; non-zero = not a regular field
__device__ __forceinline__ uint32_t alphabet_bad(const Alleles4 &g) {
  return (((g.A + 0x76767676u) & ~g.dA) | ((g.B + 0x76767676u) & ~g.dB)) & 0x80808080u;
}
// class bits of the four fields for the allele whose digit is replicated in ka: LO / HI at bits 7, 15, 23, 31
// (none 0, het 1, hom 2, missing 3; main.go:1063-1124 on "x<sep>y")
__device__ __forceinline__ void classes4(const Alleles4 &g, uint32_t ka, uint32_t *LO, uint32_t *HI) {
  const uint32_t eA = zero_b(g.A ^ ka), eB = zero_b(g.B ^ ka);
  const uint32_t dm = g.dA | g.dB;
  *LO = (eA ^ eB) | dm;
  *HI = (eA & eB) | dm;
}
// the four (lo, hi) pairs as one class-map byte: pair i sits at bits 8i+7 and moves to bits 2i, 2i+1
__device__ __forceinline__ uint32_t class_byte(uint32_t LO, uint32_t HI) {
  return ((((LO >> 7) | (HI >> 6)) & 0x03030303u) * 0x01041040u) >> 24;
}

// list mode -> dense: the entries classified for ALT #1 into the zeroed stage, their counts into the lanes that replay
__device__ __forceinline__ void list_to_stage(RawList *sp, uint32_t n, uint32_t ka, uint8_t *stage, uint32_t n_chunks,
                                              FastAcc &acc) {
  zero_stage(stage, n_chunks);
  u32x4 e = {0u, 0u, 0u, 0u};
  uint32_t idx = 0;
  if ((uint32_t)lane_id() < n) {
    e = sp->t[lane_id()];
    idx = sp->idx[lane_id()];
  }
  acc.hi |= e.x | e.y | e.z | e.w;
  {
    // the list goes on as the list of lanes that hold anything but 0 and 1 (every lane has read its entry above; LDS
    // accesses of a wave complete in order)
    const bool oth = ((e.x | e.y | e.z | e.w) & 0x000E000Eu) != 0;
    const unsigned long long nz = __ballot(oth);
    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (oth) {
      sp->t[at] = e;
      sp->idx[at] = idx;
    }
    acc.n_oth = bcast0((uint32_t)__popcll(nz));
  }
  idx %= kStageBytes;
  const Alleles4 g = gather4(e.x, e.y, e.z, e.w);
  uint32_t LO, HI;
  classes4(g, ka, &LO, &HI);
  acc.bad |= alphabet_bad(g);
  acc.het += __popc(LO & ~HI);
  acc.hom += __popc(HI & ~LO);
  acc.miss += __popc(LO & HI);
  if ((uint32_t)lane_id() < n) stage[idx] = (uint8_t)class_byte(LO, HI);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// one 1 KiB chunk (this lane's 4 fields) of a regular region
// sp (optional, lines of <= kStageChunks chunks only): the line is in list mode while acc.n_sp < kDenseMode
__device__ __forceinline__ void fast_chunk(u32x4 v, uint32_t c, uint32_t n_chunks, uint32_t ns, uint32_t kref,
                                           uint32_t table, uint8_t *cmap, uint8_t *stage, uint32_t stride,
                                           uint32_t term_xor, FastAcc &acc, RawList *sp = nullptr) {
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
  const uint32_t tor = t[0] | t[1] | t[2] | t[3];
  const uint32_t low = table & 0x0FFFFFFFu;  // digit d -> 1 at bits 2d..2d+1: the allele digit
  const uint32_t ka = low ? (uint32_t)(__builtin_ctz(low) >> 1) * 0x01010101u : 0xFFFFFFFFu;
  if (__any(tor != 0)) {
    // frame: separator and TAB bytes as expected, allele bytes within '0'^[0,31]
    acc.bad |= tor & 0xFFE0FFE0u;
    bool dense = !(sp && acc.n_sp < kDenseMode);
    if (!dense) {
      const unsigned long long nz = __ballot(tor != 0);
      const uint32_t cnt = (uint32_t)__popcll(nz);
      if (acc.n_sp + cnt <= kRawMax) {
        const uint32_t at = acc.n_sp + __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
        if (tor) {
          sp->t[at] = u32x4{t[0], t[1], t[2], t[3]};
          sp->idx[at] = c * 64u + (uint32_t)lane;
        }
        acc.n_sp = bcast0(acc.n_sp + cnt);  // (kept provably wave-uniform: the tests on it stay scalar branches)
      } else {
        // too many for the list: from here on the line is a map
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        list_to_stage(sp, acc.n_sp, ka, stage, n_chunks, acc);
        acc.n_sp = bcast0(kDenseMode);
        dense = true;
      }
    }
    if (dense) {
      acc.hi |= tor;
      if (sp && acc.n_oth < kDenseMode) {
        // lanes with a digit >= 2 or a dot: the further ALT indices' class lists come from them at the line's end
        const unsigned long long nz = __ballot((tor & 0x000E000Eu) != 0);
        if (nz) {
          const uint32_t cnt = (uint32_t)__popcll(nz);
          if (acc.n_oth + cnt <= kRawMax) {
            const uint32_t at = acc.n_oth + __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
            if (tor & 0x000E000Eu) {
              sp->t[at] = u32x4{t[0], t[1], t[2], t[3]};
              sp->idx[at] = c * 64u + (uint32_t)lane;
            }
            acc.n_oth = bcast0(acc.n_oth + cnt);
          } else {
            acc.n_oth = bcast0(kDenseMode);
          }
        }
      }
      const Alleles4 g = gather4(t[0], t[1], t[2], t[3]);
      uint32_t LO, HI;
      classes4(g, ka, &LO, &HI);
      acc.bad |= alphabet_bad(g);
      acc.het += __popc(LO & ~HI);
      acc.hom += __popc(HI & ~LO);
      acc.miss += __popc(LO & HI);
      if (cmap) stage[(c % kStageChunks) * 64u + lane] = (uint8_t)class_byte(LO, HI);  // the stage starts zeroed
    }
  }
  if (cmap && !(sp && acc.n_sp < kDenseMode) && ((c % kStageChunks) == kStageChunks - 1u || c + 1 == n_chunks)) {
    flush_stage(stage, cmap, c - (c % kStageChunks), ((c % kStageChunks) + 1u) * 64u, stride);
    if (c + 1 != n_chunks) zero_stage(stage, n_chunks - (c + 1u));
  }
}

// the highest allele digit (2..9) some entry carries, 1 when there is none (dots alone, 0x1E, do not count: they are in
// every list anyway).  tor = the entry's four words or-ed
__device__ __forceinline__ uint32_t highest_allele(const Alleles4 &g, uint32_t tor) {
  uint32_t kmax = 1;
  if (__any((tor & 0x000E000Eu) != 0)) {
    // bytes >= 2 that are not dots
    const uint32_t hi2 = (((g.A + 0x7E7E7E7Eu) & ~g.dA) | ((g.B + 0x7E7E7E7Eu) & ~g.dB)) & 0x80808080u;
    if (__any(hi2 != 0)) {
#pragma nounroll
      for (uint32_t k = 9; k >= 2; k--) {
        const uint32_t kk = k * 0x01010101u;
        if (__any((zero_b(g.A ^ kk) | zero_b(g.B ^ kk)) != 0)) {
          kmax = k;
          break;
        }
      }
    }
  }
  return kmax;
}
// A line whose ALT #1 became a dense map hands its further alleles on as a RAW LIST: the (up to kRawMax) entries that
// can tell who carries them -- all non-reference lanes of a line of 16..63 of them, the lanes with a digit >= 2 or a dot
// of a line that went dense during the scan -- saved in the class-map slots behind the line's own.  k_head gives every
// further ALT index of such a line a k_gt task that points here, and k_gt classifies the entries (1.3 KB) instead of
// reading the line (10 KB) again (main.go:549-556 rescans per allele).  k_stream itself only stores the entries: anything
// more at the end of a line costs the scan of every line registers (class lists and maps built here: +1.4 % on biallelic
// files).
//   area: +0 n, +16 the entries' map byte indices (4 B each), +272 their four field words (16 B each)
constexpr uint32_t kRawEnc = 7u << 1;  // low bits of the line's class-map offset: "raw list in the next slots"
constexpr uint32_t kRawAreaBytes = 16u + 4u * (kRawMax + 1u) + 16u * kRawMax;
constexpr uint32_t kRawTask = 1u;      // GtTask.pad[0], low byte: s_begin is the raw area's offset in the class-map arena
constexpr uint32_t kTaskRecShift = 8;  // GtTask.pad[0] >> 8: allele records k_gt fills in (pad[1], pad[2]: where; put_task)
__device__ __forceinline__ void raw_save(uint8_t *area, uint32_t n, const u32x4 &e, uint32_t idx) {
  const uint32_t lane = (uint32_t)lane_id();
  if (lane == 0) __builtin_nontemporal_store(n, reinterpret_cast<uint32_t *>(area));
  if (lane < n) {
    __builtin_nontemporal_store(idx, reinterpret_cast<uint32_t *>(area + 16u) + lane);
    __builtin_nontemporal_store(e, reinterpret_cast<u32x4 *>(area + 16u + 4u * (kRawMax + 1u)) + lane);
  }
}
// the class lists of ALT #2..#kmax from the entries the lanes hold (idx = the entry's map byte): ALT #k's at
// lists + 64 * (k - 1)
__device__ __forceinline__ void write_further_lists(const Alleles4 &g, uint32_t idx, uint32_t kmax, uint8_t *lists) {
#pragma nounroll
  for (uint32_t k = 2; k <= kmax; k++) {
    uint32_t lo_k, hi_k;
    classes4(g, k * 0x01010101u, &lo_k, &hi_k);
    const uint32_t byte_k = class_byte(lo_k, hi_k);
    const unsigned long long nz = __ballot(byte_k != 0);
    uint32_t *list = reinterpret_cast<uint32_t *>(lists + 64u * (k - 1u));
    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0u));
    if (byte_k) __builtin_nontemporal_store((idx << 8) | byte_k, list + 1u + at);
    if (lane_id() == 0) __builtin_nontemporal_store((uint32_t)__popcll(nz), list);
  }
}

// End of a line that went dense with a raw list at hand: the list holds the n (1..kRawMax) lanes that saw anything but
// 0 and 1 (fast_chunk, list_to_stage), in scan order.  Returns 1 << 1 when the entries are dots only (nobody carries a
// further allele), kRawEnc when they were saved for k_gt (raw_area: the slots behind the line's own, null when the
// wave has none to spare), 0 otherwise (k_gt reads the line).
__device__ __forceinline__ uint32_t finish_dense(const RawList *sp, uint32_t n, uint8_t *raw_area) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  u32x4 e = {0u, 0u, 0u, 0u};
  uint32_t idx = 0;
  if ((uint32_t)lane_id() < n) {
    e = sp->t[lane_id()];
    idx = sp->idx[lane_id()];
  }
  const Alleles4 g = gather4(e.x, e.y, e.z, e.w);
  if (highest_allele(g, e.x | e.y | e.z | e.w) == 1u) return 1u << 1;
  if (!raw_area) return 0u;
  raw_save(raw_area, n, e, idx);
  return kRawEnc;
}

// End of a line that stayed in list mode (n = acc.n_sp <= kRawMax entries): lane i classifies entry i.  Returns what
// k_head is told about the line in the low bits of its class-map offset (offsets are multiples of 16):
//   bit 0 = 1, bits 1-3 = kmax - 1   at most BVCF_CMAP_SPARSE_MAX entries: the class list of ALT #1 is at cmap and --
//                                    when the entries carry allele digits 2..kmax -- the lists of ALT #2..#kmax at
//                                    cmap + 64 * (k - 1); no sample carries a higher allele
//   1 << 1                           more entries than a list holds: cmap is the dense map of ALT #1, and no sample carries
//                                    a further allele
//   kRawEnc                          ... some do: the entries were saved behind the line's slot (raw_save) for k_gt
//   0                                a dense map of ALT #1 and nothing known about further alleles (k_gt reads the line):
//                                    no spare slots
// Either way a multiallelic line whose non-reference samples fit the raw list is read once (the reference rescans the
// line once per allele, main.go:549-556).
__device__ __forceinline__ uint32_t finish_list(const RawList *sp, FastAcc &acc, uint8_t *cmap, uint32_t max_k,
                                                uint8_t *stage, uint32_t n_chunks, uint32_t stride, uint8_t *raw_area) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int lane = lane_id();
  const uint32_t n = acc.n_sp;
  u32x4 e = {0u, 0u, 0u, 0u};
  uint32_t idx = 0;
  if ((uint32_t)lane < n) {
    e = sp->t[lane];
    idx = sp->idx[lane];
  }
  const Alleles4 g = gather4(e.x, e.y, e.z, e.w);
  acc.bad |= alphabet_bad(g);
  uint32_t LO, HI;
  classes4(g, 0x01010101u, &LO, &HI);
  acc.het = __popc(LO & ~HI);
  acc.hom = __popc(HI & ~LO);
  acc.miss = __popc(LO & HI);
  const uint32_t byte1 = class_byte(LO, HI);
  const uint32_t kmax = highest_allele(g, e.x | e.y | e.z | e.w);
  const bool sparse1 = n <= BVCF_CMAP_SPARSE_MAX && kmax <= max_k;
  if (!sparse1) {
    // a dense map of ALT #1 after all
    zero_stage(stage, n_chunks);
    if ((uint32_t)lane < n) stage[idx % kStageBytes] = (uint8_t)byte1;
    flush_stage(stage, cmap, 0u, n_chunks * 64u, stride);
    acc.n_sp = bcast0(kDenseMode);
    if (kmax == 1u) return 1u << 1;  // nobody carries a further allele
    // (Tried: dense maps of ALT #2..#4 grown beside ALT #1's in the stage during the scan -- no rescans at all, but
    // k_stream, which is bound by instruction issue, took 23 % longer on configs[3] and 1-5 % on biallelic files, more
    // than k_gt's rescans cost; class lists and maps of the further alleles built here from the entries: +1.4 %.)
    if (!raw_area) return 0u;
    raw_save(raw_area, n, e, idx);
    return kRawEnc;
  }
  write_further_lists(g, idx, kmax, cmap);  // (the list of ALT #k at cmap + 64 * (k - 1))
  // the list of ALT #1: count, then the entries (an entry may carry a zero byte: a lane whose fields only hold other
  // alleles)
  const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)((idx << 8) | byte1), 0x138, 0xF, 0xF, false);  // wave_shr:1
  if ((uint32_t)lane <= n) __builtin_nontemporal_store(lane == 0 ? n : prev, reinterpret_cast<uint32_t *>(cmap) + lane);
  return 1u | ((kmax - 1u) << 1);
}

// check_term: also require the byte after the last sample to be the line terminator (the caller
// predicted the end of the line from the region's regular length)
// [c_lo, c_hi) (optional; c_lo a multiple of kStageChunks): only these chunks of the region -- one wave's share of
// a line that is split across waves (k_gt_wide).  st then holds the share's n_het / n_hom / n_miss only.
__device__ inline bool gt_scan_fast(const KernelArgs &a, uint32_t s_begin, uint32_t ns, uint32_t allele, uint8_t *cmap,
                                    uint8_t *stage, bool check_term, GtStats *st, uint32_t c_lo = 0,
                                    uint32_t c_hi = 0xFFFFFFFFu) {
  const int lane = lane_id();
  const uint32_t table = (allele <= 9 ? (1u << (2u * allele)) : 0u) | (3u << 28);
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;
  c_hi = min(c_hi, n_chunks);
  const uint32_t f_hi = min(c_hi + 1u, n_chunks);  // loads go one chunk further: its first dword ends chunk c_hi - 1
  // dword-aligned loads, shift undone in registers (see realign).  The last field of a full last
  // chunk would need one dword past the chunks: such geometries (ns % 256 == 0) load unaligned.
  const uint32_t r = (ns & 255u) ? (s_begin & 3u) : 0u;
  const uint32_t l_begin = s_begin - r;
  const uint32_t last_off = (a.cap - 16u) & ~3u;
  const uint8_t *base = a.buf;
  // every chunk of the region ends before the buffer does (the common case): no per-load clamp
  const bool inside = (unsigned long long)l_begin + (unsigned long long)n_chunks * kChunk <= a.cap;
  const uint8_t *lane_base = base + l_begin + 16u * lane;
  auto fetch = [&](uint32_t c) -> u32x4 {
    if (inside) return ld_stream(lane_base + c * kChunk);
    const uint32_t off = min(l_begin + c * kChunk + 16u * lane, last_off);
    return ld_stream(base + off);
  };
  // first dword of the chunk after c (held in register `nxt`), or 0 past the region
  auto next0 = [&](uint32_t c, const u32x4 &nxt) -> uint32_t {
    return c + 1 < n_chunks ? (uint32_t)__builtin_amdgcn_readfirstlane(nxt.x) : 0u;
  };
  FastAcc acc = {0, 0, 0, 0, kDenseMode};
  if (cmap) zero_stage(stage, c_hi - c_lo);
  u32x4 va[kFastGroup], vb[kFastGroup];
#pragma unroll
  for (int g = 0; g < kFastGroup; g++) va[g] = c_lo + g < f_hi ? fetch(c_lo + g) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int g = 0; g < kFastGroup; g++) vb[g] = c_lo + kFastGroup + g < f_hi ? fetch(c_lo + kFastGroup + g) : u32x4{0u, 0u, 0u, 0u};
  // the separator of the first field is the line's separator; mixed lines fail the frame test
  const uint32_t w0 = __builtin_amdgcn_alignbyte(__builtin_amdgcn_readfirstlane(va[0].y), __builtin_amdgcn_readfirstlane(va[0].x), r);
  const uint32_t sep = (w0 >> 8) & 0xFFu;
  if (sep != '|' && sep != '/') return false;
  const uint32_t kref = 0x09300030u | (sep << 8);
  // the 32-bit compare below cannot be expressed with a 0 sentinel (0 is a valid xor), so "no check"
  // is any value above 0xFF000000
  const uint32_t term_xor = check_term ? ((a.eol_byte ^ 0x09u) << 24) : 0xFFFFFFFFu;

  for (uint32_t c0 = c_lo; c0 < c_hi; c0 += 2 * kFastGroup) {
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + g < c_hi)
        fast_chunk(realign(va[g], next0(c0 + g, g + 1 < kFastGroup ? va[g + 1 < kFastGroup ? g + 1 : 0] : vb[0]), r), c0 + g,
                   n_chunks, ns, kref, table, cmap, stage, a.cmap_stride, term_xor, acc);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + 2 * kFastGroup + g < f_hi) va[g] = fetch(c0 + 2 * kFastGroup + g);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + kFastGroup + g < c_hi)
        fast_chunk(realign(vb[g], next0(c0 + kFastGroup + g, g + 1 < kFastGroup ? vb[g + 1 < kFastGroup ? g + 1 : 0] : va[0]), r),
                   c0 + kFastGroup + g, n_chunks, ns, kref, table, cmap, stage, a.cmap_stride, term_xor, acc);
#pragma unroll
    for (int g = 0; g < kFastGroup; g++)
      if (c0 + 3 * kFastGroup + g < f_hi) vb[g] = fetch(c0 + 3 * kFastGroup + g);
  }
  if (__any(acc.bad != 0)) return false;
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
// dos (optional): the sample's dosage -- altCount, 127 at most, -1 when missing (main.go:1117-1178) -- goes to dos[s]
// win (optional): only the fields that START in [win->lo, win->hi) -- one wave's share of a line whose scan is split
// over waves (k_gt_wide_general).  win->base = TABs of the region before win->lo, i.e. the sample index of the field
// that holds byte win->lo.  The class map is then zeroed by the caller's predecessor, not here; st and *n_tabs are
// the share's.
struct ScanWindow {
  uint32_t lo, hi, base;
};
__device__ inline void gt_scan_general(const KernelArgs &a, uint32_t s_begin, uint32_t cend, uint32_t ns,
                                       uint32_t allele, uint8_t *cmap, GtStats *st, uint32_t *n_tabs,
                                       int8_t *dos = nullptr, const ScanWindow *win = nullptr) {
  const int lane = lane_id();
  const uint32_t line_begin = s_begin;
  const uint32_t sample_base = win ? win->base : 0u;
  const uint32_t starts_end = win ? win->hi : cend;  // fields starting from here on belong to the next share
  bool first_is_start = true;
  if (win) {
    first_is_start = win->lo == line_begin || a.buf[win->lo - 1u] == '\t';
    s_begin = win->lo;
  }
  uint32_t a_nd = 1;
  for (uint32_t t = allele; t >= 10; t /= 10) a_nd++;
  const uint32_t table = (allele <= 9 ? (1u << (2u * allele)) : 0u) | (3u << 28);
  if (cmap && !win) {  // zero this allele's map, then OR classes in
    for (uint32_t i = lane * 4u; i < a.cmap_stride; i += kWave * 4u) *reinterpret_cast<uint32_t *>(cmap + i) = 0u;
    __builtin_amdgcn_s_waitcnt(0);  // stores retired before the atomics below touch the same words
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  uint32_t ac = 0, an = 0, het = 0, hom = 0, miss = 0;
  uint32_t tabs_before = 0;    // TABs in earlier chunks
  uint32_t prev_last_tab = 0;  // did the previous chunk end in a TAB?
  // Chunks are loaded dword-aligned (see realign: -25 % bandwidth otherwise) kGenDepth ahead, and
  // unconditionally -- past the region the address is clamped -- so that the loads in flight can be
  // counted (see k_stream).  The bytes before s_begin in the first chunk are masked out.
  constexpr int kGenDepth = 4;
  const uint32_t r0 = s_begin & 3u, lb = s_begin - r0;
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  const uint32_t n_chunks = starts_end > s_begin ? (starts_end - lb + kChunk - 1u) / kChunk : 0u;
  auto fetch = [&](uint32_t c) -> u32x4 { return ld_stream(a.buf + min(lb + c * kChunk + 16u * lane, cap_off)); };
  u32x4 vb[kGenDepth];
#pragma unroll
  for (int j = 0; j < kGenDepth; j++) vb[j] = fetch(j);
  for (uint32_t c0 = 0; c0 < n_chunks; c0 += kGenDepth) {
#pragma unroll
    for (int j = 0; j < kGenDepth; j++) {
      const uint32_t c = c0 + j;
      if (c < n_chunks) {
        const u32x4 v = vb[j];
        const uint32_t off = lb + c * kChunk + 16u * lane;
        uint32_t valid = bits_until(starts_end, off);
        if (off < s_begin) valid &= ~bits_until(s_begin, off);  // lane 0 of the first chunk
        const uint32_t m = eq_mask16(v, '\t') & valid;
        uint32_t tot;
        const uint32_t pre = wave_excl_scan(__popc(m), &tot);
        // field starts: the byte after each TAB, plus the region start.  The previous lane's last byte comes
        // over one wave_shr DPP move (lane 0: the previous chunk's lane 63)
        uint32_t starts = (m << 1) | ((uint32_t)__builtin_amdgcn_update_dpp((int)prev_last_tab, (int)(m >> 15), 0x138, 0xF, 0xF, false) & 1u);
        if (c == 0 && lane == 0 && first_is_start) starts |= 1u << r0;
        starts &= valid & 0xFFFFu;
        // bytes 16..19 of this lane's window: the next lane's first dword (next chunk's for lane 63)
        const uint32_t nx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)vb[(j + 1) % kGenDepth].x);
        const uint32_t d4 = (uint32_t)__builtin_amdgcn_update_dpp((int)nx0, (int)v.x, 0x130, 0xF, 0xF, false);
        while (starts) {
          const uint32_t k = __ffs(starts) - 1;
          starts &= starts - 1;
          // sample index = TABs before this byte
          const uint32_t s = sample_base + tabs_before + pre + __popc(m & ((1u << k) - 1u));
          // four real bytes c0 c1 c2 c3 of the field, from registers
          const bool in4 = off + k + 4u <= cend;
          const uint32_t i = k >> 2;
          const uint32_t lo = i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
          const uint32_t hi = i == 0 ? v.y : (i == 1 ? v.z : (i == 2 ? v.w : d4));
          const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, k & 3u);
          const uint32_t c1 = (w >> 8) & 0xFFu, c3 = w >> 24;
          const uint32_t v0 = (w & 0xFFu) ^ '0', v2 = ((w >> 16) & 0xFFu) ^ '0';
          // (bitwise on purpose: a chain of || over compares with constants becomes a switch, which the backend lowers
          // to a tree of divergent branches)
          const bool frame = in4 & (bool)((uint32_t)(c1 == '|') | (uint32_t)(c1 == '/')) &
                             (bool)((uint32_t)(c3 == ':') | (uint32_t)(c3 == '\t'));
          // the reference genotype -- nearly every field of a cohort file -- only counts two called alleles;
          // the rest of the body runs when some lane of the wave holds anything else
          const bool ref = s < ns && frame && (v0 | v2) == 0;
          if (ref) {
            an += 2;
            if (dos) dos[s] = 0;
          }
          if (__any(s < ns && !ref)) {
            if (s < ns && !ref) {
              uint32_t cls = 0, altc = 0, gtc = 0;
              bool done = false;
              const bool plain = v0 < 32u && v2 < 32u && ((0x400003FFu >> v0) & (0x400003FFu >> v2) & 1u);
              if (frame && plain) {
                const uint32_t code = ((table >> ((v0 & 15u) * 2u)) & 3u) + ((table >> ((v2 & 15u) * 2u)) & 3u);
                cls = code < 3u ? code : 3u;
                gtc = cls == 3u ? 0u : 2u;
                altc = cls == 3u ? 0u : cls;
                done = true;
              }
              if (!done) classify_field(a.buf, off + k, cend, allele, a_nd, &cls, &altc, &gtc);
              ac += altc;
              an += gtc;
              het += cls == BVCF_CLS_HET;
              hom += cls == BVCF_CLS_HOM;
              miss += cls == BVCF_CLS_MISSING;
              if (dos) dos[s] = cls == BVCF_CLS_MISSING ? (int8_t)-1 : (int8_t)(altc < 127u ? altc : 127u);
              if (cmap && cls) atomicOr(reinterpret_cast<uint32_t *>(cmap + (s >> 4) * 4u), cls << (2u * (s & 15u)));
            }
          }
        }
        prev_last_tab = lane_value(m >> 15, kWave - 1) & 1u;
        tabs_before += tot;
      }
      vb[j] = fetch(c + kGenDepth);
    }
  }
  // a field that starts exactly at cend (empty last field) was not visited above
  if (lane == 0 && starts_end == cend) {
    const bool empty_last = (cend == line_begin) || (cend > line_begin && a.buf[cend - 1] == '\t');
    if (empty_last && sample_base + tabs_before < ns) {
      an += 1;  // "" is one non-matching allele token
      if (dos) dos[sample_base + tabs_before] = 0;
    }
  }
  st->ac = wave_sum(ac);
  st->an = wave_sum(an);
  st->n_het = wave_sum(het);
  st->n_hom = wave_sum(hom);
  st->n_miss = wave_sum(miss);
  *n_tabs = tabs_before;
}

// ------------------------------------------------------------------ k_gt: one wave per task

#ifdef BVCF_EXP_GT_KINDS  // (one atomic per task on one address: the kernel's time is not to be read in such a build)
__device__ unsigned int g_gt_kinds[4];  // tasks k_gt ran: raw list, regular text, general text, summed windows (bvcf_debug_gt_kinds)
#define GT_KIND(k) if (lane == 0) atomicAdd(&g_gt_kinds[k], 1u)
#else
#define GT_KIND(k)
#endif
__global__ __launch_bounds__(kWgThreads) void k_gt(KernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerWg][kStageBytes];
  uint8_t *stage = s_stage[wave_in_wg()];
  const int lane = lane_id();
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_tasks = min(n_lines + a.counters->n_tasks, a.max_tasks);
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t ns = a.n_samples;
  // Streaming path: ALT #1 of every regular line is already scanned; of the slots past n_lines -- further ALT indices, and
  // lines k_stream merely delimited -- k_head listed the ones that hold a scan (real_tasks), and wave w takes entries
  // w, w + stride, ..: the same number of scans per wave, give or take one.  (Striding over the task slots themselves -- three
  // quarters of configs[3]'s are placeholders of alleles k_head settled from class lists -- left the unluckiest wave 13 scans
  // where the mean is 5.)  Census path: every slot from 0 on.
  // A wave fetches 64 of its entries at a time, one per lane (k = k0 + lane), and walks those with something to scan; the next
  // 64 are in flight meanwhile.
  const bool listed = a.fused != 0;
  const uint32_t n_items = listed ? min(a.counters->n_real, a.max_tasks) : n_tasks;
  const uint32_t first = wave_in_grid();
  static_assert(sizeof(GtTask) == 32, "fetched as two 16-byte words: line, allele, s_begin, cend | cmap_off, pad[3]");
  auto fetch = [&](uint32_t k0, u32x4 &lo, u32x4 &hi, uint32_t &ti) {
    const unsigned long long it = (unsigned long long)first + (unsigned long long)(k0 + (uint32_t)lane) * stride;
    lo = hi = u32x4{0u, 0u, 0u, 0u};
    ti = 0xFFFFFFFFu;
    if (it < n_items) ti = listed ? a.real_tasks[it] : (uint32_t)it;
    if (ti < n_tasks && (!listed || ti >= n_lines)) {
      const u32x4 *p = reinterpret_cast<const u32x4 *>(&a.tasks[ti]);
      lo = p[0];
      hi = p[1];
    }
  };
  u32x4 nlo, nhi;
  uint32_t nti;
  fetch(0u, nlo, nhi, nti);
  for (uint32_t k0 = 0; (unsigned long long)first + (unsigned long long)k0 * stride < n_items; k0 += kWave) {
   const u32x4 lo = nlo, hi = nhi;
   const uint32_t ti_l = nti;
   fetch(k0 + kWave, nlo, nhi, nti);
   unsigned long long todo = __ballot(lo.y != 0u);  // (allele 0: rejected before getAlleles, or settled: nothing to scan)
   while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    GtTask t;
    t.line = lane_value(lo.x, src);
    t.allele = lane_value(lo.y, src);
    t.s_begin = lane_value(lo.z, src);
    t.cend = lane_value(lo.w, src);
    t.cmap_off = lane_value(hi.x, src);
    t.pad[0] = lane_value(hi.y, src);
    t.pad[1] = lane_value(hi.z, src);
    t.pad[2] = lane_value(hi.w, src);
    const uint32_t ti = lane_value(ti_l, src);
    uint8_t *cm = t.cmap_off != BVCF_NO_CMAP ? a.cmap + t.cmap_off : nullptr;
    GtStats st = {0, 0, 0, 0, 0};
    uint32_t n_fields;
    // regular region: 4 bytes per sample, every dword of a lane is one "x|y<TAB>" field
    bool regular = false;
    if ((t.pad[0] & 0xFFu) == kRawTask) {
      // a further allele of a line k_stream scanned: its carriers are among the entries saved with the line (raw_save)
      const uint8_t *area = a.cmap + t.s_begin;
      // (all three loads at once -- the area is there for kRawMax entries whatever the count says -- and the count applied
      // afterwards: one memory latency per task instead of two)
      u32x4 e = {0u, 0u, 0u, 0u};
      uint32_t idx = 0;
      if ((uint32_t)lane < kRawMax) {
        idx = reinterpret_cast<const uint32_t *>(area + 16u)[lane];
        e = reinterpret_cast<const u32x4 *>(area + 16u + 4u * (kRawMax + 1u))[lane];
      }
      const uint32_t n = min(*reinterpret_cast<const uint32_t *>(area), kRawMax);
      if ((uint32_t)lane >= n) {
        e = u32x4{0u, 0u, 0u, 0u};
        idx = 0;
      }
      const Alleles4 g = gather4(e.x, e.y, e.z, e.w);
      uint32_t LO, HI;
      classes4(g, t.allele <= 9u ? t.allele * 0x01010101u : 0x7F7F7F7Fu, &LO, &HI);  // (the fast scan's fields hold one digit)
      wave_sum3(__popc(LO & ~HI), __popc(HI & ~LO), __popc(LO & HI), ns, &st.n_het, &st.n_hom, &st.n_miss);
      st.ac = st.n_het + 2u * st.n_hom;
      st.an = 2u * (ns - st.n_miss);
      if (cm) {
        const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;
        const uint32_t byte_k = class_byte(LO, HI);
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += kStageChunks) {  // (a window of the stage at a time: 16 384 samples)
          const uint32_t nc = min(n_chunks - c0, kStageChunks);
          zero_stage(stage, nc);
          if (byte_k && idx / kStageBytes == c0 / kStageChunks) stage[idx % kStageBytes] = (uint8_t)byte_k;
          flush_stage(stage, cm, c0, nc * 64u, a.cmap_stride);
        }
      }
      n_fields = ns;
      regular = true;
      GT_KIND(0);
    } else if (a.wide && t.cend + 1u - t.s_begin == 4u * ns && a.results[ti].pad == 0) {
      // k_gt_wide scanned the region window by window and every window was regular: add up
      const GtResult part = a.results[ti];
      st.n_het = part.n_het;
      st.n_hom = part.n_hom;
      st.n_miss = part.n_miss;
      st.ac = st.n_het + 2u * st.n_hom;
      st.an = 2u * (ns - st.n_miss);
      n_fields = ns;
      regular = true;
    } else if (!a.wide && t.cend + 1u - t.s_begin == 4u * ns &&
               gt_scan_fast(a, t.s_begin, ns, t.allele, cm, stage, false, &st)) {
      n_fields = ns;
      regular = true;
      GT_KIND(1);
    } else if (a.wide && a.results[ti].pad == 2u) {
      // k_gt_wide_general summed the windows of this line
      const GtResult part = a.results[ti];
      st.ac = part.ac;
      st.an = part.an;
      st.n_het = part.n_het;
      st.n_hom = part.n_hom;
      st.n_miss = part.n_miss;
      n_fields = part.n_fields + 1u;
    } else {
      uint32_t tabs;
      gt_scan_general(a, t.s_begin, t.cend, ns, t.allele, cm, &st, &tabs);
      n_fields = tabs + 1u;
      GT_KIND(2);
    }
    if (lane == 0) {
      GtResult r;
      r.ac = st.ac;
      r.an = st.an;
      r.n_het = st.n_het;
      r.n_hom = st.n_hom;
      r.n_miss = st.n_miss;
      r.n_fields = n_fields;
      r.regular = regular ? 1u : 0u;
      r.pad = 0;
      a.results[ti] = r;
    }
    // streaming path, a further ALT index of a settled line: the counts go straight into its records (k_finish only sees
    // lines whose ALT #1 was scanned here)
    for (uint32_t j = (uint32_t)lane; j < (t.pad[0] >> kTaskRecShift); j += kWave) {
      const uint32_t slot = j == 0u ? t.pad[1] : t.pad[2] + j - 1u;
      if (slot < a.max_alleles) {
        bvcf_allele *rec = &a.alleles[slot];
        rec->ac = st.ac;
        rec->an = st.an;
        rec->n_het = st.n_het;
        rec->n_hom = st.n_hom;
        rec->n_miss = st.n_miss;
      }
    }
   }
  }
}
// ------------------------------------------------------------------ k_gt_wide: one wave per (task, window)
// Cohorts of tens of thousands of samples and more: a 64 MiB batch holds a few hundred lines of several hundred
// kilobytes each, far fewer than the GPU has waves.  The regular scan of a line is therefore split into windows of
// kStageChunks chunks (16 384 samples, the span of the LDS class-map stage), one wave each; a window adds its counts
// to the task's result with atomics (zeroed by the host before the launch) and writes its own stretch of the class
// map.  k_gt, launched after it, turns the sums into the task's result -- or, if some window met a field that is
// not regular, rescans the task with the general scan.
__global__ __launch_bounds__(kWgThreads) void k_gt_wide(KernelArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerWg][kStageBytes];
  uint8_t *stage = s_stage[wave_in_wg()];
  const int lane = lane_id();
  const uint32_t ns = a.n_samples;
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_tasks = min(n_lines + a.counters->n_tasks, a.max_tasks);
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;
  const uint32_t n_win = (n_chunks + kStageChunks - 1u) / kStageChunks;
  const unsigned long long n_items = (unsigned long long)n_tasks * n_win;
  const uint32_t stride = gridDim.x * kWavesPerWg;
  for (unsigned long long it = wave_in_grid(); it < n_items; it += stride) {
    const uint32_t ti = (uint32_t)(it / n_win), w = (uint32_t)(it % n_win);
    const GtTask t = a.tasks[ti];
    if (t.allele == 0 || t.cend + 1u - t.s_begin != 4u * ns) continue;  // k_gt's general scan
    uint8_t *cm = t.cmap_off != BVCF_NO_CMAP ? a.cmap + t.cmap_off : nullptr;
    GtStats st = {0, 0, 0, 0, 0};
    const bool ok = gt_scan_fast(a, t.s_begin, ns, t.allele, cm, stage, false, &st, w * kStageChunks, (w + 1u) * kStageChunks);
    if (lane == 0) {
      GtResult *r = &a.results[ti];
      if (!ok) {
        atomicOr(&r->pad, 1u);
      } else {
        if (st.n_het) atomicAdd(&r->n_het, st.n_het);
        if (st.n_hom) atomicAdd(&r->n_hom, st.n_hom);
        if (st.n_miss) atomicAdd(&r->n_miss, st.n_miss);
      }
    }
  }
}

// ------------------------------------------------------------------ wide lines, fields beyond GT
// The general scan of a line whose sample region is not the 4-byte grid, split over waves: windows of a.win_bytes of
// the region.  A field's sample index is the number of TABs before it, so the windows' TAB counts come first
// (k_tabs_wide, which also zeroes the class maps), and each window of k_gt_wide_general starts from the sum of the
// counts before it.  Counts are kept per line (the further ALT indices of a line share them) at
// win_tabs[s_begin / win_bytes + line + w]: consecutive lines cannot collide there.
__device__ __forceinline__ uint32_t wide_windows(const KernelArgs &a, const GtTask &t) {
  const uint32_t len = t.cend - t.s_begin;
  return len ? (len + a.win_bytes - 1u) / a.win_bytes : 1u;
}
__device__ __forceinline__ uint32_t wide_slot(const KernelArgs &a, const GtTask &t) { return t.s_begin / a.win_bytes + t.line; }
// tasks scanned window by window by the two kernels below (the others are k_gt_wide's, or have no scan)
__device__ __forceinline__ bool wide_general_task(const KernelArgs &a, const GtTask &t) {
  return t.allele != 0 && t.cend >= t.s_begin && t.cend + 1u - t.s_begin != 4u * a.n_samples &&
         wide_slot(a, t) + wide_windows(a, t) <= a.win_tabs_cap;
}

__global__ __launch_bounds__(kWgThreads) void k_tabs_wide(KernelArgs a) {
  const int lane = lane_id();
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_tasks = min(n_lines + a.counters->n_tasks, a.max_tasks);
  const uint32_t w_max = a.counters->pad[1] / a.win_bytes + 1u;  // pad[1]: the longest sample region (k_head)
  const unsigned long long n_items = (unsigned long long)n_tasks * w_max;
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  for (unsigned long long it = wave_in_grid(); it < n_items; it += stride) {
    const uint32_t ti = (uint32_t)(it / w_max), w = (uint32_t)(it % w_max);
    const GtTask t = a.tasks[ti];
    if (!wide_general_task(a, t)) continue;
    const uint32_t n_win = wide_windows(a, t);
    if (w >= n_win) continue;
    // this window's stretch of the task's class map
    if (t.cmap_off != BVCF_NO_CMAP) {
      const uint32_t words = a.cmap_stride / 4u;
      const uint32_t lo = (uint32_t)((unsigned long long)words * w / n_win), hi = (uint32_t)((unsigned long long)words * (w + 1u) / n_win);
      uint32_t *cm = reinterpret_cast<uint32_t *>(a.cmap + t.cmap_off);
      for (uint32_t i = lo + lane; i < hi; i += kWave) cm[i] = 0u;
    }
    if (ti >= n_lines) continue;  // a further ALT index: the line's own task counts (k_head always gives an evaluated
                                  // line its ALT #1 task on this path: the scan also settles the field count)
    const uint32_t lo = t.s_begin + w * a.win_bytes, hi = min(lo + a.win_bytes, t.cend);
    const uint32_t lb = lo & ~3u;
    uint32_t cnt = 0;
    for (uint32_t off = lb + 16u * lane; off < hi; off += kChunk) {
      const u32x4 v = ld_stream(a.buf + min(off, cap_off));
      uint32_t valid = bits_until(hi, off);
      if (off < lo) valid &= ~bits_until(lo, off);
      cnt += __popc(eq_mask16(v, '\t') & valid);
    }
    cnt = wave_sum(cnt);
    if (lane == 0) a.win_tabs[wide_slot(a, t) + w] = cnt;
  }
}

__global__ __launch_bounds__(kWgThreads) void k_gt_wide_general(KernelArgs a) {
  const int lane = lane_id();
  const uint32_t ns = a.n_samples;
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_tasks = min(n_lines + a.counters->n_tasks, a.max_tasks);
  const uint32_t w_max = a.counters->pad[1] / a.win_bytes + 1u;
  const unsigned long long n_items = (unsigned long long)n_tasks * w_max;
  const uint32_t stride = gridDim.x * kWavesPerWg;
  for (unsigned long long it = wave_in_grid(); it < n_items; it += stride) {
    const uint32_t ti = (uint32_t)(it / w_max), w = (uint32_t)(it % w_max);
    const GtTask t = a.tasks[ti];
    if (!wide_general_task(a, t)) continue;
    const uint32_t n_win = wide_windows(a, t);
    if (w >= n_win) continue;
    // TABs of the region before this window
    const uint32_t slot = wide_slot(a, t);
    uint32_t before = 0;
    for (uint32_t j = lane; j < w; j += kWave) before += a.win_tabs[slot + j];
    ScanWindow win;
    win.base = wave_sum(before);
    win.lo = t.s_begin + w * a.win_bytes;
    win.hi = w + 1u == n_win ? t.cend : win.lo + a.win_bytes;
    uint8_t *cm = t.cmap_off != BVCF_NO_CMAP ? a.cmap + t.cmap_off : nullptr;
    GtStats st = {0, 0, 0, 0, 0};
    uint32_t tabs = 0;
    gt_scan_general(a, t.s_begin, t.cend, ns, t.allele, cm, &st, &tabs, nullptr, &win);
    if (lane == 0) {
      GtResult *r = &a.results[ti];
      if (st.ac) atomicAdd(&r->ac, st.ac);
      if (st.an) atomicAdd(&r->an, st.an);
      if (st.n_het) atomicAdd(&r->n_het, st.n_het);
      if (st.n_hom) atomicAdd(&r->n_hom, st.n_hom);
      if (st.n_miss) atomicAdd(&r->n_miss, st.n_miss);
      if (tabs) atomicAdd(&r->n_fields, tabs);
      if (w == 0) atomicOr(&r->pad, 2u);  // "summed by windows": k_gt only finishes the result
    }
  }
}

// ------------------------------------------------------------------ k_dosage: one wave per output allele
// --dosageOutput (main.go:306-342,576-584): the int8 row of every alleles[] slot that holds a record.  Runs after
// k_finish, only when the ctx asks for it; every field goes through the general scan, which knows the allele
// count of any ploidy (the 2-bit class of the label path cannot tell "1" from "1|1").
template <bool share_of_row>
__device__ __forceinline__ void k_dosage_body(const KernelArgs &a) {
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_alleles = min(n_lines + a.counters->n_alleles, a.max_alleles);
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t w_max = share_of_row ? a.counters->pad[1] / a.win_bytes + 1u : 1u;
  const unsigned long long n_items = (unsigned long long)n_alleles * w_max;
  for (unsigned long long it = wave_in_grid(); it < n_items; it += stride) {
    const uint32_t k = (uint32_t)(it / w_max), w = (uint32_t)(it % w_max);
    const bvcf_allele r = a.alleles[k];
    const uint32_t li = k < n_lines ? k : r.line;
    if (li >= n_lines) continue;
    const bvcf_line L = a.lines[li];
    if (L.status != BVCF_LINE_OK || L.n_rec == 0) continue;
    // slot k belongs to line li if it is the line's own slot or one of its further alleles
    if (k >= n_lines && (k < L.rec_first || k - L.rec_first + 1u >= L.n_rec)) continue;
    int8_t *row = a.dosage + (size_t)k * a.dosage_stride;
    const int lane = lane_id();
    if (r.cmap_off != BVCF_NO_CMAP && r.gt_task < a.max_tasks && a.results[r.gt_task].regular) {
      if (share_of_row) continue;
      // the scan that produced this allele's class map was the regular one: the row is the map, 2 bits -> int8
      const uint8_t *cm = a.cmap + r.cmap_off;
      const uint32_t n_bytes = (a.n_samples + 3u) / 4u;
      auto expand = [](uint32_t byte) -> uint32_t {  // codes 0 1 2 3 -> bytes 0x00 0x01 0x02 0xFF
        const uint32_t e = (byte | (byte << 6) | (byte << 12) | (byte << 18)) & 0x03030303u;
        const uint32_t miss = (e & (e >> 1)) & 0x01010101u;
        return e | (miss * 0xFCu);
      };
      if (r.flags & BVCF_ALLELE_CMAP_SPARSE) {
        for (uint32_t i = lane; i < n_bytes; i += kWave) reinterpret_cast<uint32_t *>(row)[i] = 0u;
        __builtin_amdgcn_s_waitcnt(0);  // the zeros land before the few entries below overwrite them
        const uint32_t n = min(reinterpret_cast<const uint32_t *>(cm)[0], (uint32_t)BVCF_CMAP_SPARSE_MAX);
        if ((uint32_t)lane < n) {
          const uint32_t e = reinterpret_cast<const uint32_t *>(cm)[1 + lane];
          if ((e >> 8) < n_bytes) reinterpret_cast<uint32_t *>(row)[e >> 8] = expand(e & 0xFFu);
        }
      } else {
        for (uint32_t i = lane; i < n_bytes; i += kWave) reinterpret_cast<uint32_t *>(row)[i] = expand(cm[i]);
      }
      continue;
    }
    GtTask t;  // the line's sample region, as put_task described it
    t.line = li;
    t.allele = r.alt_idx + 1u;
    t.s_begin = L.off + L.fend[8] + 1u;
    t.cend = L.off + L.len;
    if (a.wide && wide_general_task(a, t)) {
      // split over waves like the scan that counted it; the TAB counts per share are still in win_tabs
      if (!share_of_row) continue;
    } else if (share_of_row) {
      continue;
    }
    GtStats st;
    uint32_t tabs;
    if (!share_of_row) {
      gt_scan_general(a, t.s_begin, t.cend, a.n_samples, t.allele, nullptr, &st, &tabs, row);
      continue;
    }
    const uint32_t n_win = wide_windows(a, t);
    if (w >= n_win) continue;
    const uint32_t slot = wide_slot(a, t);
    uint32_t before = 0;
    for (uint32_t j = lane; j < w; j += kWave) before += a.win_tabs[slot + j];
    ScanWindow win;
    win.base = wave_sum(before);
    win.lo = t.s_begin + w * a.win_bytes;
    win.hi = w + 1u == n_win ? t.cend : win.lo + a.win_bytes;
    gt_scan_general(a, t.s_begin, t.cend, a.n_samples, t.allele, nullptr, &st, &tabs, row, &win);
  }
}

__global__ __launch_bounds__(kWgThreads) void k_dosage(KernelArgs a) { k_dosage_body<false>(a); }
// wide lines with fields beyond GT: one wave per (output allele, share of the line), see k_gt_wide_general
__global__ __launch_bounds__(kWgThreads) void k_dosage_wide(KernelArgs a) { k_dosage_body<true>(a); }

}  // namespace bvcf_dev
