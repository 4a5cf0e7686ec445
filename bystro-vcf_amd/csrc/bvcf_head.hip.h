// bvcf_head.hip.h — k_head (fixed columns, FILTER gate, getAlleles, records + scan tasks) and k_finish
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include "bvcf_common.hip.h"
#include "bvcf_alleles.hip.h"
#include "bvcf_gtscan.hip.h"
#include "bvcf_stream.hip.h"

namespace bvcf_dev {

// ------------------------------------------------------------------ k_head: 16 lanes per line

constexpr int kGroup = 16;                       // lanes per line in k_head
constexpr int kGroupsPerWg = kWgThreads / kGroup;
constexpr uint32_t kWindow = kGroup * 16;        // bytes per group step
// Line-head bytes kept in LDS for the serial phase.  CHROM..FILTER of a 1000-Genomes line are ~40 bytes; what lies past
// the staged bytes (a long indel, INFO) is read from memory.  64 instead of 128 bytes: 36 KB of LDS per workgroup instead
// of 52, four workgroups per CU instead of three -- k_head_lean alone 109 -> 91 us per 262 144 rows of configs[3], 63 ->
// 54 us on biallelic rows (round 4, profiles/r04_k_head_*; the kernel is one latency-bound step per workgroup, so what
// counts is how many of its 1 024 workgroups are resident at once).
#ifndef BVCF_HEAD_STAGE
#define BVCF_HEAD_STAGE 64
#endif
constexpr uint32_t kHeadStage = BVCF_HEAD_STAGE;

__device__ __forceinline__ int glane() { return threadIdx.x & (kGroup - 1); }

// A 16-lane group is one DPP row: scans and sums stay in the VALU (row shifts / rotations) instead of going
// through the LDS crossbar five times per call (__shfl_up x 4 + __shfl), which was most of phase T's time.
// sum of the row, in every lane of the row
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);  // row_ror:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);  // row_ror:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);  // row_ror:1
  return v;
}

// exclusive prefix sum inside a 16-lane group; *total = group sum
__device__ __forceinline__ uint32_t group_excl_scan(uint32_t v, uint32_t *total) {
  uint32_t inc = v;
  inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);  // row_shr:1, zero fill
  inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
  inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
  inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
  *total = group_sum(v);
  return inc - v;
}

__device__ __forceinline__ uint32_t gbcast0(uint32_t v) { return __shfl(v, 0, kGroup); }

template <class B>
__device__ inline bool filter_in(const B &buf, Span f, const uint16_t *off, const uint16_t *len, uint32_t n,
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

// gr (optional): the genotype counts of the record, when they are already known (streaming path); otherwise
// k_finish copies them from the scan result of `task`
__device__ inline void write_allele(const KernelArgs &a, uint32_t idx, uint32_t line, uint32_t alt_idx,
                                    const AlleleEval &e, long long pos, uint8_t ref, uint8_t alt_base,
                                    uint8_t site_type, uint32_t task, uint32_t cmap_off, const GtResult *gr = nullptr) {
  bvcf_allele r;
  r.pos = pos;
  r.line = line;
  r.alt_idx = alt_idx;
  r.alt_off = e.alt_off;
  r.alt_len = e.mnp ? 1u : e.alt_len;
  r.ac = gr ? gr->ac : 0u;
  r.an = gr ? gr->an : 0u;
  r.n_het = gr ? gr->n_het : 0u;
  r.n_hom = gr ? gr->n_hom : 0u;
  r.n_miss = gr ? gr->n_miss : 0u;
  // offsets are multiples of 16; k_stream sets bit 0 when the slot holds a class list (bits 1-3: see finish_list)
  const bool sparse = cmap_off != BVCF_NO_CMAP && (cmap_off & 1u);
  r.cmap_off = cmap_off != BVCF_NO_CMAP ? cmap_off & ~15u : cmap_off;
  r.ref = ref;
  r.alt_base = alt_base;
  r.kind = e.mnp ? (uint8_t)BVCF_ALT_BASE : e.kind;
  r.site_type = site_type;
  r.trtv = (site_type == BVCF_SITE_MULTI || r.kind != BVCF_ALT_BASE) ? 0 : trtv_of(ref, alt_base);
  r.flags = ((!e.mnp && e.pos_text) ? BVCF_ALLELE_POS_TEXT : 0) | (sparse ? BVCF_ALLELE_CMAP_SPARSE : 0);
  r.pad[0] = r.pad[1] = 0;
  r.gt_task = task;
  r.pad2 = 0;
  a.alleles[idx] = r;
}

constexpr uint32_t kNoTask = 0xFFFFFFFFu;

// What the low bits of a line's class-map offset (k_stream: finish_list / finish_dense) say about its further ALT indices.
// Listed: ALT #1 is a class list with the lists of ALT #2..#kmax behind it in the slot, or a dense map whose samples carry
// no further allele (kmax = 1); nobody carries a higher index.  Not listed: no map, a map with nothing known, or kRawEnc.
__device__ __forceinline__ bool further_listed(uint32_t cm) { return cm != BVCF_NO_CMAP && (cm & 15u) != 0u && (cm & 15u) != kRawEnc; }
__device__ __forceinline__ uint32_t further_kmax(uint32_t cm) { return (cm & 1u) ? ((cm >> 1) & 7u) + 1u : 1u; }

// Write genotype-scan task `ti` (allele == 0 marks a slot without a scan).  Task i < n_lines is
// "line i, ALT #1"; tasks past n_lines are the further ALT indices of multiallelic lines.
// n_rec / rec0 / rec1 (streaming path): the allele records that take their counts from this scan -- record 0 at alleles[rec0],
// record j >= 1 at alleles[rec1 + j - 1] -- which k_gt then fills in itself (0: none, or left to k_finish)
__device__ inline void put_task(const KernelArgs &a, uint32_t ti, uint32_t line, uint32_t allele, uint32_t s_begin,
                                uint32_t cend, uint32_t cmap_off, uint32_t kind = 0u, uint32_t n_rec = 0u, uint32_t rec0 = 0u,
                                uint32_t rec1 = 0u) {
  if (ti < a.max_tasks) {
    GtTask t;
    t.line = line;
    t.allele = allele;
    t.s_begin = s_begin;
    t.cend = cend;
    t.cmap_off = cmap_off;
    t.pad[0] = kind | (n_rec << kTaskRecShift);  // (kRawTask: s_begin is the offset of the line's raw list in the class-map arena)
    t.pad[1] = rec0;
    t.pad[2] = rec1;
    a.tasks[ti] = t;
    // (wide lines: the longest sample region bounds the windows per task of the split scans)
    if (a.wide && allele != 0 && cend >= s_begin) atomicMax(&a.counters->pad[1], cend - s_begin);
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

#ifdef BVCF_EXP_TIMES
__device__ unsigned long long g_head_t[6][8192];
#define HSTAMP(k)                                                  \
  {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    hph_[k] += now_ - hlast_;                                      \
    hlast_ = now_;                                                 \
  }
#else
#define HSTAMP(k)
#endif
// -DBVCF_EXP_TIMES=3: the stamps split PART 2 instead (0: everything before it, 1: the owners' values over ds_bpermute, 2: the token,
// 3: sums over the line's lanes, 4: lists / tasks / records, 5: back to the lines and the line record)
#if defined(BVCF_EXP_TIMES) && BVCF_EXP_TIMES + 0 == 3
#define HSTAMP_A(k)
#define HSTAMP_B(k) HSTAMP(k)
#else
#define HSTAMP_A(k) HSTAMP(k)
#define HSTAMP_B(k)
#endif
template <bool kAllWindows>
__device__ __forceinline__ void k_head_body(const KernelArgs &a) {
#ifdef BVCF_EXP_TIMES
  unsigned long long hph_[6] = {0, 0, 0, 0, 0, 0}, hlast_ = __builtin_readcyclecounter();
#endif
  __shared__ uint32_t s_head[kLinesPerStep * kHeadRow];
  __shared__ uint32_t s_tab[kLinesPerStep * kTabRow];
  __shared__ uint32_t s_ls[kLinesPerStep], s_len[kLinesPerStep], s_found[kLinesPerStep], s_staged[kLinesPerStep],
      s_extra[kLinesPerStep];
  __shared__ uint32_t s_wave[kWavesPerWg][4];
  __shared__ uint32_t s_base[5];
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
    HSTAMP_A(0);

    // ================= phase T: 16 lanes per line =================
    // lane gl of a group fetches the offsets of the group's round-gl line, so the 16 rounds' offsets
    // are in flight together; the first window of round r + 1 is requested before round r is parsed
    uint32_t my_ls = 0, my_len = 0;
    uint32_t my_line_ls = 0, my_line_len = 0;  // (streaming chains: of line line0 + threadIdx.x)
    if (!kAllWindows) {
      const uint32_t l = line0 + threadIdx.x;
      if (l < n_lines) {
        my_line_ls = a.line_off[l];
        if (a.fused) {
          my_line_len = a.line_len[l];  // (bit 31 rides along: the line's head TAB bitmap is in line_bits)
        } else {
          const uint32_t le = a.line_off[l + 1];
          my_line_len = le - my_line_ls >= a.eol_chars ? le - my_line_ls - a.eol_chars : 0u;  // chomp, main.go:535
        }
      }
    } else {
      const uint32_t l = line0 + (uint32_t)gl * kGroupsPerWg + g;
      if (l < n_lines) {
        my_ls = a.line_off[l];
        if (a.fused) {
          my_len = a.line_len[l];  // (bit 31 rides along: the line's head TAB bitmap is in line_bits)
        } else {
          const uint32_t le = a.line_off[l + 1];
          my_len = le - my_ls >= a.eol_chars ? le - my_ls - a.eol_chars : 0u;  // chomp, main.go:535
        }
      }
    }
    // kAllWindows (k_head: a ctx of one slot): the first 256 B window of all 16 rounds is requested up front, unconditionally
    // (lines past the end read offset 0), so the rounds are paced by the parse, not by one memory latency each;
    // costs 64 registers.  Otherwise (k_head_lean: chains that run beside another batch's k_stream, where a wave of 225
    // registers would wait for a whole SIMD's worth): see the staging below.
    constexpr uint32_t kRounds = kLinesPerStep / kGroupsPerWg;
    u32x4 v_win[kAllWindows ? kRounds : 1u];
    uint32_t r_ls[kAllWindows ? kRounds : 1u], r_len[kAllWindows ? kRounds : 1u];
    if (kAllWindows) {
#pragma unroll
      for (uint32_t r = 0; r < kRounds; r++) {
        r_ls[r] = __shfl(my_ls, r, kGroup);
        r_len[r] = __shfl(my_len, r, kGroup);
        v_win[r] = *reinterpret_cast<const u32x4_u *>(a.buf + min(r_ls[r] + 16u * gl, a.cap - 16u));
      }
    }
    auto tokenise = [&](uint32_t r, uint32_t ls, uint32_t len_flag, const u32x4 &v_first) {
      const uint32_t ll = r * kGroupsPerWg + g;
      const uint32_t line = line0 + ll;
      if (line >= n_lines) return;
      const uint32_t len = len_flag & ~kHasHeadBits;
      if (len_flag & kHasHeadBits) {
        // k_stream found this line's TABs when it parsed the head window (line_bits): only stage the bytes here
        const uint32_t rel = 16u * gl;
        if (rel < kHeadStage) {
          uint32_t *row = &s_head[ll * kHeadRow + rel / 4];
          row[0] = v_first.x;
          row[1] = v_first.y;
          row[2] = v_first.z;
          row[3] = v_first.w;
        }
        if (gl == 0) {
          s_ls[ll] = ls;
          s_len[ll] = len_flag;
          s_found[ll] = 0;
          s_staged[ll] = min(len, kHeadStage);
          s_extra[ll] = 0;
        }
        return;
      }
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
        uint32_t m = eq_mask16(v, '\t') & bits_until(cend, off);
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
          extra += __popc(eq_mask16(v, '\t') & bits_until(cend, off));
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
    };
    // this thread's line in phase S: its TAB bitmap and ALT #1's counts are asked for here, beside the head bytes, not one
    // memory latency each later on
    u32x4 pre_b0 = {0u, 0u, 0u, 0u}, pre_b1 = {0u, 0u, 0u, 0u};
    GtResult pre_res = GtResult{};
    bool pre_res_ok = false;
    if (kAllWindows) {
#pragma unroll
      for (uint32_t r = 0; r < kRounds; r++) tokenise(r, r_ls[kAllWindows ? r : 0], r_len[kAllWindows ? r : 0], v_win[kAllWindows ? r : 0]);
    } else {
      // Lines whose TABs k_stream found -- all but the first of a wave's run, on the streaming path -- need only their first
      // kHeadStage bytes staged: kHeadStage / 16 lanes per line, every round's load in flight at once.  (Before: the 16-lane
      // rounds below for every line, one window ahead: sixteen memory latencies per step for 64 of each 256 bytes loaded.)
      constexpr uint32_t kLanesPerHead = kHeadStage / 16u, kLinesPerRound = kWgThreads / kLanesPerHead;
      constexpr uint32_t kHeadRounds = kLinesPerStep / kLinesPerRound;
      static_assert(kHeadStage % 16u == 0 && kWgThreads % kLanesPerHead == 0 && kLinesPerStep % kLinesPerRound == 0, "head staging");
      s_ls[threadIdx.x] = my_line_ls;
      s_len[threadIdx.x] = my_line_len;
      s_found[threadIdx.x] = 0;
      s_staged[threadIdx.x] = min(my_line_len & ~kHasHeadBits, kHeadStage);
      s_extra[threadIdx.x] = 0;
      __syncthreads();
      u32x4 hv[kHeadRounds];
      const uint32_t hq = threadIdx.x % kLanesPerHead;
#pragma unroll
      for (uint32_t r = 0; r < kHeadRounds; r++) {
        const uint32_t ll = r * kLinesPerRound + threadIdx.x / kLanesPerHead;
        hv[r] = u32x4{0u, 0u, 0u, 0u};
        if (line0 + ll < n_lines && (s_len[ll] & kHasHeadBits)) hv[r] = load16(a.buf, s_ls[ll] + 16u * hq, a.cap);
      }
      {
        const uint32_t line = line0 + threadIdx.x;
        if (line < n_lines && (my_line_len & kHasHeadBits)) {
          const u32x4 *bits = reinterpret_cast<const u32x4 *>(a.line_bits + (size_t)line * 8u);
          pre_b0 = bits[0];
          pre_b1 = bits[1];
        }
        if (line < n_lines && a.fused && ns > 0 && line < a.max_tasks) {
          pre_res = a.results[line];
          pre_res_ok = true;
        }
      }
#pragma unroll
      for (uint32_t r = 0; r < kHeadRounds; r++) {
        const uint32_t ll = r * kLinesPerRound + threadIdx.x / kLanesPerHead;
        if (line0 + ll < n_lines && (s_len[ll] & kHasHeadBits)) {
          uint32_t *row = &s_head[ll * kHeadRow + 4u * hq];
          row[0] = hv[r].x;
          row[1] = hv[r].y;
          row[2] = hv[r].z;
          row[3] = hv[r].w;
        }
      }
      // the others (every line of a k_stream_gen batch, the first line of a k_stream wave's run; the census path): tokenised from
      // the text, 16 lanes a line, the first window of the next round's line requested before this round's is parsed
      auto wants = [&](uint32_t r) -> bool {
        const uint32_t ll = r * kGroupsPerWg + g;
        return r < kRounds && line0 + ll < n_lines && !(s_len[ll] & kHasHeadBits);
      };
      u32x4 v_next = {0u, 0u, 0u, 0u};
      if (wants(0)) v_next = load16(a.buf, s_ls[g] + 16u * gl, a.cap);
#pragma nounroll
      for (uint32_t r = 0; r < kRounds; r++) {
        const u32x4 v_first = v_next;
        if (wants(r + 1u)) v_next = load16(a.buf, s_ls[(r + 1u) * kGroupsPerWg + g] + 16u * gl, a.cap);
        if (wants(r)) {
          const uint32_t ll = r * kGroupsPerWg + g;
          tokenise(r, s_ls[ll], s_len[ll], v_first);
        }
      }
    }
    __syncthreads();

    HSTAMP_A(1);
    // ================= phase S: one lane per line =================
    const uint32_t ll = threadIdx.x;
    const uint32_t line = line0 + ll;
    const bool active = line < n_lines;
    const uint32_t ls = active ? s_ls[ll] : 0u, len_flag = active ? s_len[ll] : 0u;
    const uint32_t len = len_flag & ~kHasHeadBits;
    uint32_t found = active ? s_found[ll] : 0u;
    const uint32_t cend = ls + len;
    uint32_t *tab = &s_tab[ll * kTabRow];
    if (len_flag & kHasHeadBits) {
      // the nine TABs from k_stream's bitmap of the head window (256 bits from the dword at or before ls)
      u32x4 b0 = pre_b0, b1 = pre_b1;
      if (kAllWindows) {
        const u32x4 *bits = reinterpret_cast<const u32x4 *>(a.line_bits + (size_t)line * 8u);
        b0 = bits[0];
        b1 = bits[1];
      }
      const uint32_t w[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const uint32_t base = ls & ~3u;
      uint32_t k = 0;
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) {
        uint32_t m = w[q];
        while (m && k < 9u) {
          tab[k++] = base + 32u * q + (uint32_t)__ffs(m) - 1u;
          m &= m - 1u;
        }
      }
      found = k;
    }
    Bytes hb;
    hb.g = a.buf;
    hb.lds = as_lds(&s_head[ll * kHeadRow]);
    hb.lo = ls;
    hb.sub = ls;
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

    uint32_t rec_first = 0, n_rec = 0, site_type = 0, first_task = line;
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

    HSTAMP_A(2);
    // ---- slot reservation, once per workgroup step: record slot `line` and task slot `line` are
    // the line's own; only further records / ALT indices draw from the batch counters.  Biallelic
    // lines — all of a 1KG-shaped file — never touch an atomic.
    const uint32_t want_rec = bound > 1 ? bound - 1 : 0u;
    // streaming path: a line whose ALT #1 scan k_stream left to k_gt takes one more task slot, so
    // that k_gt only has to visit the slots past n_lines there
    uint32_t res_fields = 0, res_miss = 0;
    if (eval && a.fused && ns > 0 && line < a.max_tasks) {
      if (!kAllWindows && pre_res_ok) {
        res_fields = pre_res.n_fields;
        res_miss = pre_res.n_miss;
      } else {
        res_fields = a.results[line].n_fields;
        res_miss = a.results[line].n_miss;
      }
    }
    const bool deferred = eval && a.fused && ns > 0 && line < a.max_tasks && res_fields == kDeferred;
    const uint32_t want_task = ((eval && ns > 0 && mode == 2) ? n_commas : 0u) + (deferred ? 1u : 0u);
    // streaming path: which of these slots will hold a scan -- three quarters of configs[3]'s are settled from class lists
    // below -- so that k_gt walks a dense list of them, the same number per wave (real_tasks; striding over the slots themselves
    // the unluckiest of 4 096 waves had 13 scans where the mean is 5, and that wave was the kernel's time).  ALT #1 of a deferred
    // line and all its further ALT indices; the further ALT indices of a line whose ALT #1 is a dense map (raw list or text);
    // of a line kept as class lists those past the last list when somebody is missing.
    uint32_t lcm = BVCF_NO_CMAP, want_real = 0;
    if (a.fused && want_task > 0) {
      if (deferred) {
        want_real = want_task;
      } else {
        if (maps) lcm = a.line_cmap[line];
        if (further_listed(lcm)) {
          const uint32_t kmax = further_kmax(lcm);
          if (res_miss != 0u && n_commas + 1u > kmax) want_real = n_commas + 1u - kmax;
        } else {
          want_real = n_commas;
        }
      }
    }
    // streaming path: a line whose ALT #1 is left to k_gt goes on k_finish's work list (the scan also settles its field
    // count); everything else is complete when this kernel -- and, for further ALT indices with a scan, k_gt -- ends
    // (a further ALT index of a line k_stream scanned: k_gt writes the counts into the records itself, see put_task)
    const unsigned long long fin_mask = __ballot(a.fused && deferred);
    uint32_t wt_rec, wt_task, wt_real;
    uint32_t extra_base = wave_excl_scan(want_rec, &wt_rec);
    uint32_t task_base = wave_excl_scan(want_task, &wt_task);
    uint32_t real_base = wave_excl_scan(want_real, &wt_real);
    uint32_t fin_at = __builtin_amdgcn_mbcnt_hi((uint32_t)(fin_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fin_mask, 0u));
    if (lane == 0) {
      s_wave[w][0] = wt_rec;
      s_wave[w][1] = wt_task;
      s_wave[w][2] = (uint32_t)__popcll(fin_mask);
      s_wave[w][3] = wt_real;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
      // one thread per counter, so that the five atomics are one round trip: records, tasks, k_finish's list, k_gt's list, and
      // -- streaming path -- the class maps of the extra tasks, which come from the cursor k_stream used (as many as tasks)
      const uint32_t col = threadIdx.x == 4 ? 1u : threadIdx.x;
      uint32_t sum = 0;
      for (int k = 0; k < kWavesPerWg; k++) sum += s_wave[k][col];
      if (threadIdx.x == 4 && !(a.fused && maps)) sum = 0;
      uint32_t *ctr = threadIdx.x == 0 ? &a.counters->n_alleles : (threadIdx.x == 1 ? &a.counters->n_tasks :
                      (threadIdx.x == 2 ? &a.counters->n_finish : (threadIdx.x == 3 ? &a.counters->n_real : &a.counters->cmap_maps)));
      uint32_t got = 0;
      if (sum) got = atomicAdd(ctr, sum);
      // s_base: 0 records, 1 tasks, 2 class maps, 3 k_finish's list, 4 k_gt's list
      s_base[threadIdx.x == 2 ? 3 : (threadIdx.x == 3 ? 4 : (threadIdx.x == 4 ? 2 : threadIdx.x))] = got;
    }
    __syncthreads();
    uint32_t task_rank = task_base;  // this line's first extra task, counted inside the workgroup
    for (int k = 0; k < w; k++) {
      extra_base += s_wave[k][0];
      task_rank += s_wave[k][1];
      fin_at += s_wave[k][2];
      real_base += s_wave[k][3];
    }
    real_base += s_base[4];
    if (a.fused && deferred && s_base[3] + fin_at < a.max_lines + a.max_alleles) a.finish_items[s_base[3] + fin_at] = line;
    extra_base += n_lines + s_base[0];
    task_base = n_lines + s_base[1] + task_rank;
    const uint32_t map_base = a.fused ? s_base[2] + task_rank : task_base;

    HSTAMP_A(3);
    HSTAMP_B(0);
    // ---- part 2: the ALT tokens -> records and scan tasks, ONE LANE PER (line, ALT token) PAIR.
    // The reference walks a line's ALT tokens one after the other (main.go:774-999).  With one lane per line a wave that
    // holds a single multiallelic line ran the token loop three times with one lane working -- k_head's time followed the
    // number of waves that hold such a line, not their number (47 -> 69 us per 262 144 rows at 2 % multiallelic lines).
    // Here the tokens of the wave's 64 lines are numbered (exclusive scan of the token counts) and dealt to the lanes,
    // 64 pairs per round: a biallelic wave is one round with lane == line, a wave of BASELINE configs[3] (20 % lines
    // with 2-3 ALTs) two rounds instead of three or four passes.  What the sequential loop carries from token to
    // token -- records emitted so far, task slots used, "Invalid POS" ending the loop -- becomes prefix sums over the
    // lanes of a line (segments of consecutive lanes); a line that straddles two rounds carries its sums in its own lane.
    const bool has_tok = eval && (mode == 1 || mode == 2);
    if (eval) {
      if (mode == 0) log_err(a, line, 0, BVCF_ERR_SAME);
      if (mode == 3) log_err(a, line, 0, BVCF_ERR_EMPTY_REF);
    }
    const bool fits = (unsigned long long)extra_base + want_rec <= a.max_alleles;
    // With samples, the scan for ALT #1 always runs: it also settles len(record) == len(header).
    // On the streaming path k_stream has already done it (results[line], line_cmap[line]).
    uint32_t cm0 = BVCF_NO_CMAP;
    uint32_t task0 = line;  // where ALT #1's counts are (to be) found
    uint32_t tasks_used = 0, emitted = 0, stype_line = 0;
    bool dropped = false;  // "Invalid POS" ended the line's token loop (main.go:826-829)
    if (eval) {
      if (ns > 0 && !a.fused) {
        cm0 = cmap_of(a, line, maps && has_tok);
        put_task(a, line, line, 1, s_begin, cend, cm0);
        task_written = true;
      }
      if (ns > 0 && a.fused) {
        if (deferred) {
          task0 = task_base;
          cm0 = cmap_of(a, map_base, maps);
          put_task(a, task0, line, 1, s_begin, cend, cm0);
          if (real_base < a.max_tasks) a.real_tasks[real_base] = task0;
          tasks_used = 1;
        } else if (maps) {
          cm0 = a.line_cmap[line];
        }
      }
      first_task = task0;
    }
    // Streaming path: the counts of ALT #1 of a line k_stream scanned are final (results[line]), as are those of the
    // further ALT indices resolved from class lists below: such records are complete here; those of further ALT indices
    // with a scan are completed by k_gt.  A line whose ALT #1 depends on k_gt is on k_finish's work list (above).
    const bool final0 = eval && a.fused && ns > 0 && !deferred && line < a.max_tasks;
    {
      const uint32_t n_tok = has_tok ? (mode == 1 ? 1u : n_commas + 1u) : 0u;
      uint32_t n_pairs;
      const uint32_t pair0 = wave_excl_scan(n_tok, &n_pairs);  // the line's tokens are pairs [pair0, pair0 + n_tok)
      const uint32_t lflags = mode | (deferred ? 4u : 0u) | (fits ? 8u : 0u) | (final0 ? 16u : 0u);
      uint32_t *const s_mark = &s_found[(uint32_t)w * kWave];  // (s_found was consumed above: this wave's 64 words)
#pragma nounroll
      for (uint32_t q0 = 0; q0 < n_pairs; q0 += kWave) {
        // ---- who owns lane p's pair: the lines with tokens in this round leave their lane number where those start
        const bool mine = n_tok && pair0 < q0 + kWave && pair0 + n_tok > q0;
        const uint32_t seg0 = pair0 > q0 ? pair0 - q0 : 0u;  // lane of the first of them
        s_mark[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (mine) s_mark[seg0] = (uint32_t)lane + 1u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_here = min(n_pairs - q0, (uint32_t)kWave);
        const bool pl = (uint32_t)lane < n_here;
        const uint32_t mk = wave_incl_scan_max(s_mark[lane]);
        const uint32_t o = (pl && mk) ? mk - 1u : (uint32_t)lane;  // the owner's lane
        auto from_owner = [&](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(o * 4u), (int)v); };
        const uint32_t o_pair0 = from_owner(pair0), o_flags = from_owner(lflags), o_commas = from_owner(n_commas);
        const uint32_t o_extra = from_owner(extra_base), o_task = from_owner(task_base), o_map = from_owner(map_base);
        const uint32_t o_task0 = from_owner(task0), o_cm0 = from_owner(cm0);
        const uint32_t o_real = from_owner(real_base), o_miss = from_owner(res_miss);
        // (streaming chains: ALT #1's counts of the owner's line were loaded with its head, see phase T)
        GtResult o_res = GtResult{};
        if (!kAllWindows) {
          o_res.ac = from_owner(pre_res.ac);
          o_res.an = from_owner(pre_res.an);
          o_res.n_het = from_owner(pre_res.n_het);
          o_res.n_hom = from_owner(pre_res.n_hom);
          o_res.n_miss = from_owner(pre_res.n_miss);
          o_res.n_fields = from_owner(pre_res.n_fields);
          o_res.regular = 1;
        }
        const uint32_t o_emitted = from_owner(emitted), o_used = from_owner(tasks_used), o_dropped = from_owner(dropped ? 1u : 0u);
        HSTAMP_B(1);
        const uint32_t k = q0 + (uint32_t)lane - o_pair0;  // ALT index of this lane's token
        const uint32_t o_mode = o_flags & 3u;
        const bool o_deferred = (o_flags & 4u) != 0, o_fits = (o_flags & 8u) != 0, o_final0 = (o_flags & 16u) != 0;
        // the owner's line: its bytes and fixed columns are in the LDS of this workgroup step
        const uint32_t oll = (uint32_t)w * kWave + o, o_line = line0 + oll;
        const uint32_t o_ls = s_ls[oll], o_cend = o_ls + (s_len[oll] & ~kHasHeadBits);
        const uint32_t *otab = &s_tab[oll * kTabRow];
        AlleleCtx c;
        c.buf.g = a.buf;
        c.buf.lds = as_lds(&s_head[oll * kHeadRow]);
        c.buf.lo = o_ls;
        c.buf.sub = o_ls;
        c.buf.n = pl ? s_staged[oll] : 0u;
        const Bytes &ob = c.buf;
        c.chrom = Span{o_ls, otab[0] - o_ls};
        c.pos = Span{otab[0] + 1u, otab[1] - otab[0] - 1u};
        c.ref = Span{otab[2] + 1u, otab[3] - otab[2] - 1u};
        c.alt = Span{otab[3] + 1u, (need > 4u ? otab[4] : o_cend) - otab[3] - 1u};
        c.int_pos = 0;
        c.pos_bad = false;
        c.line = o_line;
        const uint32_t o_sbegin = need == 9 ? otab[8] + 1u : o_cend;

        // ---- this lane's token
        AlleleEval e = AlleleEval{};
        Span t = c.alt;
        // (REF, ALT and POS as whole words from the staged head when they are short and lie inside it: eval_token_row)
        // (a SNP -- one REF byte, one ALT byte -- is two byte reads in eval_single: cheaper than fetching the words)
        const bool snp = o_mode == 1 && c.ref.len == 1u;
        if (pl && (snp || !eval_token_row<kHeadStage>(as_lds_words(&s_head[oll * kHeadRow]), c.buf.n, o_ls, c, k, o_mode == 1, &t, e))) {
          if (o_mode == 1) {
            eval_single(c, e);
          } else {
            uint32_t i = 0, seen = 0, start = 0;
#pragma nounroll
            for (; i < c.alt.len && seen < k; i++)
              if (ob[c.alt.off + i] == ',') {
                seen++;
                start = i + 1u;
              }
            uint32_t end = start;
#pragma nounroll
            while (end < c.alt.len && ob[c.alt.off + end] != ',') end++;
            t.off = c.alt.off + start;
            t.len = end - start;
            eval_token(c, t, e);
          }
        }
        HSTAMP_B(2);
        // ---- what the tokens before it in the line did (exclusive sums over the line's lanes of this round, plus what
        // the line carries from the round before)
        const uint32_t seg_lane = o_pair0 > q0 ? o_pair0 - q0 : 0u;
        auto seg_excl = [&](uint32_t v) -> uint32_t {
          const uint32_t ex = wave_incl_scan(v) - v;
          return ex - (uint32_t)__builtin_amdgcn_ds_bpermute((int)(seg_lane * 4u), (int)ex);
        };
        const uint32_t stop_me = (pl && e.stop) ? 1u : 0u;
        const uint32_t stops_before = seg_excl(stop_me);  // (wave-wide DPP scan + bpermute: every lane takes part, no short circuit)
        const bool dropped_me = o_dropped != 0u || stops_before != 0u;
        const bool live = pl && !dropped_me;
        const uint32_t n_me = live ? e.n : 0u;
        const uint32_t tk_me = (live && ns > 0 && k > 0 && e.n > 0) ? 1u : 0u;
        const uint32_t emitted_before = o_emitted + seg_excl(n_me);
        const uint32_t used_before = o_used + seg_excl(tk_me);
        uint32_t stype_me = 0;
        HSTAMP_B(3);
        // this token's place in k_gt's list (see want_real): written whether or not the token ends up with a scan
        uint32_t real_at = kNoTask, real_task = kNoTask;
        if (pl && a.fused && ns > 0 && k > 0 && o_mode == 2u) {
          if (o_deferred)
            real_at = o_real + k;
          else if (!further_listed(o_cm0))
            real_at = o_real + k - 1u;
          else if (o_miss != 0u && k >= further_kmax(o_cm0))
            real_at = o_real + k - further_kmax(o_cm0);
        }

        if (live) {
          if (e.err) log_err(a, o_line, (e.err == BVCF_ERR_POS) ? 0u : k + 1u, e.err, k);
          if (e.n) {
            uint32_t task = o_task0, cm_off = o_cm0;
            GtResult g0 = GtResult{};
            if (o_final0 && ns > 0) g0 = kAllWindows ? a.results[o_line] : o_res;
            const GtResult *gr = (k == 0 && o_final0) ? &g0 : nullptr;
            GtResult r = GtResult{};
            if (ns > 0 && k > 0) {
              task = o_task + used_before;
              // Streaming path, a line k_stream kept as a list of its few non-reference samples: the same entries gave
              // the class list of every further ALT index they carry (finish_list), so the line is not read again
              // (the reference rescans it once per allele, main.go:549-556).  The counts come from the list.
              bool resolved = false;
              const bool raw = a.fused && !o_deferred && o_cm0 != BVCF_NO_CMAP && (o_cm0 & 15u) == kRawEnc;
              if (a.fused && !o_deferred && further_listed(o_cm0) && task < a.max_tasks) {
                // bit 0: ALT #1 is a class list and the lists of ALT #2..#kmax follow it in the slot (kmax - 1 in bits 1-3).
                // Otherwise ALT #1 is a dense map and bits 1-3 = 1: no sample carries a further allele (finish_list,
                // finish_dense; kRawEnc: some do, and k_gt settles them from the entries saved behind the slot)
                const uint32_t kmax = further_kmax(o_cm0);
                r.ac = 0;
                r.an = g0.an;
                r.n_het = r.n_hom = 0;
                r.n_miss = g0.n_miss;
                r.n_fields = g0.n_fields;
                r.regular = 1;
                r.pad = 0;
                if (k + 1u <= kmax) {
                  cm_off = ((o_cm0 & ~15u) + 64u * k) | 1u;
                  const uint32_t *list = reinterpret_cast<const uint32_t *>(a.cmap + (cm_off & ~15u));
                  const uint32_t n = min(list[0], (uint32_t)BVCF_CMAP_SPARSE_MAX);
#pragma nounroll
                  for (uint32_t i = 0; i < n; i++) {
                    const uint32_t b = list[1u + i] & 0xFFu, lo = b & 0x55u, hi = (b >> 1) & 0x55u;
                    r.n_het += __popc(lo & ~hi);
                    r.n_hom += __popc(hi & ~lo);
                  }
                  r.ac = r.n_het + 2u * r.n_hom;
                  resolved = true;
                } else if (o_miss == 0u) {
                  cm_off = BVCF_NO_CMAP;  // nobody carries it and nobody is missing: ac == 0, the row is dropped (main.go:558-560)
                  resolved = true;
                }
                if (resolved) {
                  a.results[task] = r;
                  put_task(a, task, o_line, 0, o_cend, o_cend, BVCF_NO_CMAP);  // nothing to scan
                  gr = &r;
                }
              }
              // the records of this token (below): filled in by k_gt when the line itself is settled (a deferred line is on
              // k_finish's list: the scan of its ALT #1 may still reject it)
              const uint32_t t_rec = (a.fused && !o_deferred && o_fits) ? e.n : 0u;
              const uint32_t rec0 = emitted_before == 0u ? o_line : o_extra + emitted_before - 1u, rec1 = o_extra + emitted_before;
              if (raw) {
                // k_gt classifies the line's saved entries for this allele instead of reading the line again
                resolved = true;
                cm_off = cmap_of(a, o_map + used_before, maps);
                put_task(a, task, o_line, k + 1, (o_cm0 & ~15u) + a.cmap_stride, o_cend, cm_off, kRawTask, t_rec, rec0, rec1);
                real_task = task;
              }
              if (!resolved) {
                cm_off = cmap_of(a, o_map + used_before, maps);
                put_task(a, task, o_line, k + 1, o_sbegin, o_cend, cm_off, 0u, t_rec, rec0, rec1);
                real_task = task;
              }
            }
            if (ns == 0) task = kNoTask;
            if (o_fits) {
              // type call, main.go:1004-1037 (single-ALT path: main.go:743,764)
              uint8_t stype;
              if (o_commas > 0)
                stype = BVCF_SITE_MULTI;
              else if (!e.mnp && e.kind == BVCF_ALT_DEL)
                stype = BVCF_SITE_DEL;
              else if (!e.mnp && e.kind == BVCF_ALT_INS)
                stype = BVCF_SITE_INS;
              else
                stype = e.n > 1 ? BVCF_SITE_MNP : BVCF_SITE_SNP;
              stype_me = stype;
              // slot of the line's j-th record
              auto slot = [&](uint32_t j) -> uint32_t { return j == 0 ? o_line : o_extra + j - 1; };
              if (e.mnp) {
                uint32_t j = 0;
#pragma nounroll
                for (uint32_t i = 0; i < c.ref.len; i++) {
                  const uint8_t rb = ob[c.ref.off + i], ab = ob[t.off + i];
                  if (rb == ab) continue;
                  write_allele(a, slot(emitted_before + j), o_line, k, e, c.int_pos + (long long)i, rb, ab, stype, task, cm_off, gr);
                  j++;
                }
              } else {
                write_allele(a, slot(emitted_before), o_line, k, e, e.pos, e.ref, e.alt_base, stype, task, cm_off, gr);
              }
            }
          }
        }
        if (real_at < a.max_tasks) a.real_tasks[real_at] = real_task;
        HSTAMP_B(4);
        // ---- back to the lines: the sums up to and including the last of their tokens in this round
        const uint32_t seg_end = min(pair0 + n_tok, q0 + (uint32_t)kWave) - q0 - 1u;  // (meaningful where `mine`)
        const uint32_t from = (mine ? seg_end : (uint32_t)lane) * 4u;
        const uint32_t em_in = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)(emitted_before + n_me));
        const uint32_t us_in = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)(used_before + tk_me));
        const uint32_t dr_in = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)((dropped_me || stop_me) ? 1u : 0u));
        // (a line of one token takes that lane's type; with several tokens it is MULTIALLELIC, main.go:1004-1011)
        const uint32_t st_in = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)stype_me);
        if (mine) {
          emitted = em_in;
          tasks_used = us_in;
          dropped = dr_in != 0u;
          if (n_commas == 0) stype_line = st_in;
        }
      }
    }
    if (eval && n_commas > 0 && emitted && fits) stype_line = BVCF_SITE_MULTI;
    if (eval) {
      site_type = stype_line;
      auto slot = [&](uint32_t j) -> uint32_t { return j == 0 ? line : extra_base + j - 1; };
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
      if (ns == 0 || final0) n_fields = a.n_header;  // (k_stream only lists the lines its regular scan accepted: ns sample fields)
    }

    HSTAMP_A(4);
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
      L.gt_task = first_task;
      L.status = (uint8_t)status;
      L.site_type = (uint8_t)site_type;
      L.pad[0] = L.pad[1] = 0;
      a.lines[line] = L;
      // every line owns task slot `line` and record slot `line`: mark the ones it did not fill
      // (the streaming path's k_gt never looks at the first n_lines task slots)
      if (ns > 0 && !task_written && !a.fused) put_task(a, line, line, 0, cend, cend, BVCF_NO_CMAP);
      if (!primary_written && line < a.max_alleles) a.alleles[line].gt_task = kNoTask;
    }
    HSTAMP_A(5);
    HSTAMP_B(5);
  }
#ifdef BVCF_EXP_TIMES
  if (threadIdx.x == 0 && blockIdx.x < 8192)
    for (int k = 0; k < 6; k++) g_head_t[k][blockIdx.x] = hph_[k];
#endif
}

__global__ __launch_bounds__(kWgThreads) void k_head(KernelArgs a) { k_head_body<true>(a); }
// for chains that overlap with another batch's k_stream (see kAllWindows)
// (three waves per SIMD: 168 registers; without the bound the compiler takes 175 and a workgroup less fits a CU)
__global__ __launch_bounds__(kWgThreads) __attribute__((amdgpu_waves_per_eu(3))) void k_head_lean(KernelArgs a) { k_head_body<false>(a); }

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
  if (a.fused) {
    // streaming path: k_head completed everything that did not wait for k_gt; the rest is on its work list
    const uint32_t n_items = min(a.counters->n_finish, a.max_lines + a.max_alleles);
    for (uint32_t w = tid; w < n_items; w += nthreads) {
      const uint32_t li = a.finish_items[w];
      if (li >= n_lines) continue;
      bvcf_line *L = &a.lines[li];
      const uint32_t st = L->status;
      if (st == BVCF_LINE_OK || st == BVCF_LINE_NOALLELE) {
        if (L->gt_task < n_tasks) {
          const uint32_t nf = 9u + a.results[L->gt_task].n_fields;
          L->n_fields = nf;
          if (nf != a.n_header) {
            L->status = BVCF_LINE_FIELDS;
            L->n_rec = 0;
          }
        }
      }
      // the line's records (slot li, then rec_first ..): counts from the scan each one names
      const uint32_t n_rec = L->n_rec, rec_first = L->rec_first;
      for (uint32_t j = 0; j < n_rec; j++) {
        const uint32_t i = j ? rec_first + j - 1u : li;
        if (i >= n_alleles) break;
        bvcf_allele *r = &a.alleles[i];
        const uint32_t t = r->gt_task;
        if (t >= n_tasks) continue;
        const GtResult g = a.results[t];
        r->ac = g.ac;
        r->an = g.an;
        r->n_het = g.n_het;
        r->n_hom = g.n_hom;
        r->n_miss = g.n_miss;
      }
    }
    return;
  }
  for (uint32_t i = tid; i < n_lines; i += nthreads) {
    bvcf_line *L = &a.lines[i];
    const uint32_t st = L->status;
    if (st != BVCF_LINE_OK && st != BVCF_LINE_NOALLELE) continue;
    if (L->gt_task >= n_tasks) continue;
    const uint32_t nf = 9u + a.results[L->gt_task].n_fields;
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
