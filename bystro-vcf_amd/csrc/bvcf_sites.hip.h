// bvcf_sites.hip.h — k_sites: sites-only input (no sample columns) in one pass after the newline census
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// A sites-only line is ~150 bytes: a 1 KiB chunk holds seven of them, and per line the path needs the first TABs
// (strings.Split, main.go:535), the field count and the FILTER gate (linePasses, main.go:447-454), getAlleles
// (main.go:723-1038) and trTv (main.go:602-606) -- and writes 128 bytes of records.  The census chain spent its time
// re-reading: k_scatter_eol revisits the chunks for the line starts, k_head re-reads every line head with 16 lanes
// per line in 16 latency-bound rounds.  Here one wave walks a contiguous run of 8 KiB windows and reads every byte
// ONCE:
//   * each 1 KiB chunk goes from registers into a 16 KiB text ring in LDS (the current and the previous window);
//   * its terminator mask gives the line ends (appended to a FIFO in LDS), its TAB mask goes into a bit ring;
//   * whenever 64 line ends are pending -- and at the end of a window -- ONE LANE PER LINE walks the TAB bits of its
//     line (first `need` TABs, total count), runs the gate and getAlleles on the ring bytes and writes the line and
//     allele records.  A line belongs to the window its terminator is in; its start is the byte after the previous
//     terminator, which the wave carries along (only the first window of a run searches backwards for it).
// The line index of the first terminator of a run comes from the census prefix (k_count_eol / k_scan_*), so records
// land in input order without a second pass.  Lines longer than a window (their start has left the ring) take a
// wave-cooperative slow path for the TABs and read their head bytes from memory.
#pragma once

#include "bvcf_common.hip.h"
#include "bvcf_alleles.hip.h"
#include "bvcf_head.hip.h"

namespace bvcf_dev {

constexpr uint32_t kSitesWin = 8192;                       // bytes per window
constexpr uint32_t kSitesRing = 2 * kSitesWin;             // text ring per wave: previous + current window
constexpr uint32_t kSitesChunks = kSitesWin / kChunk;      // chunk registers per window
constexpr uint32_t kSitesFifo = 64 + kChunk;               // pending line ends: < 64 carried + one chunk's worth
constexpr int kSitesWaves = 2;                             // waves per workgroup (43 KiB of LDS: three per CU)
constexpr int kSitesThreads = kSitesWaves * kWave;

struct SitesLds {
  uint8_t text[kSitesRing];          // byte at block offset o lives at text[o & (kSitesRing - 1)]
  uint32_t tabs[kSitesRing / 32];    // bit o & (kSitesRing - 1): byte o is a TAB
  uint16_t fifo[kSitesFifo];         // pending terminators, relative to the current window's start
  uint32_t long_tab[10];             // slow path: the first 9 TAB offsets and the TAB count of a line longer than the ring
};

// last terminator at a position < limit, or kNone (wave-cooperative, backwards, 1 KiB per step)
__device__ inline uint32_t find_eol_before(const KernelArgs &a, uint32_t limit) {
  const int lane = lane_id();
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  uint32_t end = limit;
  while (end > 0) {
    const uint32_t base = end >= kChunk ? (end - kChunk) & ~15u : 0u;
    const uint32_t off = base + 16u * lane;
    const u32x4 v = *reinterpret_cast<const u32x4_u *>(a.buf + min(off, cap_off));
    const uint32_t m = eq_mask16(v, a.eol_byte) & bits_until(end, off) & (off <= cap_off ? 0xFFFFu : 0u);
    const unsigned long long b = __ballot(m != 0);
    if (b) {
      const int src = 63 - __clzll((long long)b);
      return lane_value(off + 31u - (uint32_t)__clz(m), src);
    }
    end = base;
  }
  return kNone;
}

__global__ __launch_bounds__(kSitesThreads) void k_sites(KernelArgs a, uint32_t n_chunks) {
  __shared__ __attribute__((aligned(16))) SitesLds s_lds[kSitesWaves];
  __shared__ FilterTable s_ft;
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.filters);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_ft);
    for (uint32_t i = threadIdx.x; i < sizeof(FilterTable) / 4; i += kSitesThreads) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = lane_id();
  const uint32_t wiw = bcast0(threadIdx.x >> 6);
  SitesLds &S = s_lds[wiw];
  const uint32_t wave = blockIdx.x * kSitesWaves + wiw;
  const uint32_t n_waves = gridDim.x * kSitesWaves;
  const uint32_t nb = a.nbytes;
  const uint32_t n_win = (uint32_t)(((unsigned long long)nb + kSitesWin - 1u) / kSitesWin);
  // balanced runs of windows
  const uint32_t q_win = n_win / n_waves, r_win = n_win % n_waves;
  const uint32_t win_lo = wave * q_win + min(wave, r_win);
  const uint32_t win_hi = win_lo + q_win + (wave < r_win ? 1u : 0u);
  if (win_lo >= win_hi) return;
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t need = min(9u, a.n_header - 1u);  // TABs that bound the fixed columns we read
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  constexpr uint32_t kRingMask = kSitesRing - 1u;

  auto chunk_load = [&](uint32_t win, uint32_t c) -> u32x4 {
    const unsigned long long off = (unsigned long long)win * kSitesWin + c * kChunk + 16u * lane;
    return ld_stream(a.buf + (off < cap_off ? (uint32_t)off : cap_off));
  };
  // stage 16 bytes of the lane and the TAB bits of those bytes
  auto stage = [&](uint32_t off, const u32x4 &v, uint32_t mt) {
    *reinterpret_cast<u32x4 *>(&S.text[off & kRingMask]) = v;
    reinterpret_cast<uint16_t *>(S.tabs)[(off & kRingMask) >> 4] = (uint16_t)mt;
  };

  uint32_t w0 = win_lo * kSitesWin;
  // ---- where the first line of the run starts, and its line index
  uint32_t ps = 0;       // start of the line whose terminator comes next
  uint32_t ring_lo = w0; // oldest block offset the ring holds
  // terminators before w0 = index of the line the first terminator of the run ends (census prefix of the chunk)
  const uint32_t c_first = w0 / kChunk;
  uint32_t rank = c_first < n_chunks ? a.census[c_first] + a.group_base[c_first / kScanGroup] : 0u;
  if (win_lo > 0) {
    // the previous window goes into the ring (a line that ends in this run may have started there)
    uint32_t last = kNone;
#pragma unroll
    for (uint32_t c = 0; c < kSitesChunks; c++) {
      const u32x4 v = chunk_load(win_lo - 1u, c);
      const uint32_t off = w0 - kSitesWin + c * kChunk + 16u * lane;
      stage(off, v, eq_mask16(v, '\t'));
      const uint32_t me = eq_mask16(v, a.eol_byte);
      const unsigned long long b = __ballot(me != 0);
      if (b) {
        const int src = 63 - __clzll((long long)b);
        last = lane_value(off + 31u - (uint32_t)__clz(me), src);
      }
    }
    ring_lo = w0 - kSitesWin;
    if (last == kNone) last = find_eol_before(a, ring_lo);  // a line longer than a window
    ps = last == kNone ? 0u : last + 1u;
  }

  uint32_t n_pending = 0;
  // one lane per pending line, lines [0, n) of the FIFO; `staged_end`: the ring holds [ring_lo, staged_end)
  auto flush = [&](uint32_t n, uint32_t staged_end) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool active = (uint32_t)lane < n;
    const uint32_t line = rank + (uint32_t)lane;
    uint32_t ls = 0, le = 0;
    if (active) {
      le = w0 + S.fifo[lane];
      ls = lane == 0 ? ps : w0 + S.fifo[lane - 1] + 1u;
    }
    const uint32_t len = active && le + 1u - ls >= a.eol_chars ? le + 1u - ls - a.eol_chars : 0u;  // chomp, main.go:535
    const uint32_t cend = ls + len;
    // ---- a line whose start has left the ring (only the first pending line can be one): TABs by the whole wave
    const bool is_long = bcast0((n > 0 && ps < ring_lo) ? 1u : 0u) != 0u;
    if (is_long) {
      const uint32_t l_ls = ps, l_cend = bcast0(cend);
      uint32_t found = 0;
      for (uint32_t base = l_ls & ~15u; base < l_cend; base += kChunk) {
        const uint32_t off = base + 16u * lane;
        const u32x4 v = *reinterpret_cast<const u32x4_u *>(a.buf + min(off, cap_off));
        uint32_t m = eq_mask16(v, '\t') & bits_until(l_cend, off) & (off <= cap_off ? 0xFFFFu : 0u);
        if (off < l_ls) m &= ~bits_until(l_ls, off);
        uint32_t tot;
        uint32_t rk = found + wave_excl_scan(__popc(m), &tot);
        while (m && rk < 9u) {
          S.long_tab[rk] = off + __ffs(m) - 1;
          m &= m - 1;
          rk++;
        }
        found += tot;
      }
      if (lane == 0) S.long_tab[9] = found;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- strings.Split(row, "\t") from the TAB bits: the first `need` TABs and the count (main.go:535)
    uint32_t tab[9];
    uint32_t found = 0, n_tabs = 0;
    if (active && !(is_long && lane == 0)) {
      uint32_t wi = ls >> 5;
      const uint32_t we = cend >> 5;
      uint32_t cur = ls < cend ? S.tabs[wi & (kSitesRing / 32 - 1u)] & (0xFFFFFFFFu << (ls & 31u)) : 0u;
      uint32_t k_next = 0;  // next entry of tab[] to fill
#pragma unroll
      for (uint32_t k = 0; k < 9; k++) tab[k] = cend;
      for (;;) {
        if (wi == we) cur &= (1u << (cend & 31u)) - 1u;  // bits below cend
        n_tabs += __popc(cur);
        // (tab[] is indexed by constants only: a register array)
#pragma unroll
        for (uint32_t k = 0; k < 9; k++) {
          if (k == k_next && cur && k < need) {
            tab[k] = wi * 32u + (uint32_t)__ffs(cur) - 1u;
            cur &= cur - 1u;
            k_next = k + 1u;
          }
        }
        if (wi >= we) break;
        wi++;
        cur = S.tabs[wi & (kSitesRing / 32 - 1u)];
      }
      found = min(n_tabs, need);
    } else if (active) {
#pragma unroll
      for (uint32_t k = 0; k < 9; k++) tab[k] = S.long_tab[k];
      n_tabs = S.long_tab[9];
      found = min(n_tabs, need);
    } else {
#pragma unroll
      for (uint32_t k = 0; k < 9; k++) tab[k] = 0;
    }

    Bytes hb;
    hb.g = a.buf;
    hb.lds = S.text;
    hb.lo = ring_lo;
    hb.n = staged_end - ring_lo;
    hb.sub = 0;
    hb.mask = kRingMask;

    uint32_t status = BVCF_LINE_OK;
    uint32_t n_fields = n_tabs + 1u;
    if (active && n_fields != a.n_header) status = BVCF_LINE_FIELDS;  // len(record) == len(header), main.go:449

    auto fspan = [&](uint32_t i) -> Span {
      Span sp;
      sp.off = i == 0 ? ls : tab[i - 1] + 1;
      const uint32_t e = i < need ? tab[i] : cend;
      sp.len = e - sp.off;
      return sp;
    };

    // ---- gate and what the line will need (as k_head, part 1)
    AlleleCtx c;
    uint32_t mode = 0, n_commas = 0, bound = 0;
    if (active && status == BVCF_LINE_OK && a.n_header > 6) {
      const FilterTable *ft = &s_ft;
      Span f = fspan(6);
      if (!ft->allow_nil && !filter_in(hb, f, ft->allow_off, ft->allow_len, ft->allow_n, ft->text))
        status = BVCF_LINE_FILTER;
      else if (!ft->deny_nil && filter_in(hb, f, ft->deny_off, ft->deny_len, ft->deny_n, ft->text))
        status = BVCF_LINE_FILTER;
    }
    const bool eval = active && status == BVCF_LINE_OK;
    if (eval) {
      c.buf = hb;
      c.chrom = fspan(0);
      c.pos = fspan(1);
      c.ref = fspan(3);
      c.alt = fspan(4);
      c.int_pos = 0;
      c.pos_bad = false;
      c.line = line;
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

    // ---- record slots past the line's own: one atomic per wave round, and only when a line needs them
    const uint32_t want_rec = bound > 1 ? bound - 1 : 0u;
    uint32_t wt_rec;
    uint32_t extra_base = wave_excl_scan(want_rec, &wt_rec);
    uint32_t got = 0;
    if (wt_rec) {
      if (lane == 0) got = atomicAdd(&a.counters->n_alleles, wt_rec);
      got = bcast0(got);
    }
    extra_base += n_lines + got;

    // ---- evaluate the ALT tokens, write the records (as k_head, part 2, without samples)
    uint32_t rec_first = 0, n_rec = 0, site_type = 0;
    bool primary_written = false;
    if (eval && line < n_lines) {
      if (mode == 0) log_err(a, line, 0, BVCF_ERR_SAME);
      if (mode == 3) log_err(a, line, 0, BVCF_ERR_EMPTY_REF);
      const bool fits = (unsigned long long)extra_base + want_rec <= a.max_alleles;
      auto slot = [&](uint32_t j) -> uint32_t { return j == 0 ? line : extra_base + j - 1; };
      uint32_t cur = 0, emitted = 0;
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
          if (fits) {
            uint8_t stype;  // type call, main.go:1004-1037 (single-ALT path: main.go:743,764)
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
                write_allele(a, slot(emitted + j), line, k, e, c.int_pos + (long long)i, rb, ab, stype, kNoTask, BVCF_NO_CMAP);
                j++;
              }
            } else {
              write_allele(a, slot(emitted), line, k, e, e.pos, e.ref, e.alt_base, stype, kNoTask, BVCF_NO_CMAP);
            }
          }
          emitted += e.n;
        }
      }
      // reserved but unused slots must not look like records
      if (fits)
#pragma nounroll
        for (uint32_t j = emitted > 1 ? emitted : 1; j < bound; j++) a.alleles[slot(j)].gt_task = kNoTask;
      if (emitted) primary_written = true;
      if (fits) rec_first = extra_base;
      if (emitted == 0)
        status = BVCF_LINE_NOALLELE;
      else if (fits)
        n_rec = emitted;
      n_fields = a.n_header;
    }

    // ---- line record
    if (active && line < n_lines) {
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
      if (!primary_written && line < a.max_alleles) a.alleles[line].gt_task = kNoTask;
    }

    // ---- pop
    const uint32_t last_rel = S.fifo[n - 1];
    ps = bcast0(w0 + last_rel + 1u);
    rank += n;
    __builtin_amdgcn_wave_barrier();
    if (n_pending > n) {  // (only after a chunk with more than 64 terminators)
      for (uint32_t i = 0; i < n_pending - n; i += kWave) {
        const uint16_t x = i + lane < n_pending - n ? S.fifo[n + i + lane] : (uint16_t)0;
        __builtin_amdgcn_wave_barrier();
        if (i + lane < n_pending - n) S.fifo[i + lane] = x;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    n_pending -= n;
  };

  // ---- the run, window by window.  The next window's eight chunks are requested when this one starts and are not
  // touched before it ends; the chunk loop is a real loop (one copy of the per-line code), so the chunk registers are
  // picked by a switch over constants.
  u32x4 cur[kSitesChunks], nxt[kSitesChunks];
#pragma unroll
  for (uint32_t c = 0; c < kSitesChunks; c++) cur[c] = chunk_load(win_lo, c);
  for (uint32_t win = win_lo; win < win_hi; win++) {
    w0 = win * kSitesWin;
    if (win > win_lo) ring_lo = w0 - kSitesWin;
    const uint32_t win_next = win + 1u < win_hi ? win + 1u : win;  // (the last window's prefetch is dropped)
#pragma unroll
    for (uint32_t c = 0; c < kSitesChunks; c++) nxt[c] = chunk_load(win_next, c);
#pragma nounroll
    for (uint32_t c = 0; c < kSitesChunks; c++) {
      u32x4 v;
      switch (c) {
        case 0: v = cur[0]; break;
        case 1: v = cur[1]; break;
        case 2: v = cur[2]; break;
        case 3: v = cur[3]; break;
        case 4: v = cur[4]; break;
        case 5: v = cur[5]; break;
        case 6: v = cur[6]; break;
        default: v = cur[7]; break;
      }
      static_assert(kSitesChunks == 8, "the switch above covers 8 chunks");
      const uint32_t off = w0 + c * kChunk + 16u * lane;
      const uint32_t valid = bits_until(nb, off);
      const uint32_t mt = eq_mask16(v, '\t') & valid;
      const uint32_t me = eq_mask16(v, a.eol_byte) & valid;
      stage(off, v, mt);
      if (__any(me != 0)) {
        uint32_t tot;
        uint32_t at = n_pending + wave_excl_scan(__popc(me), &tot);
        uint32_t m = me;
        while (m) {
          S.fifo[at++] = (uint16_t)(c * kChunk + 16u * lane + (uint32_t)__ffs(m) - 1u);
          m &= m - 1;
        }
        n_pending += tot;
      }
      // 64 pending line ends make a full round; the end of the window flushes the rest (FIFO entries are relative
      // to this window)
      const bool last = c + 1u == kSitesChunks;
      while (n_pending >= 64u || (last && n_pending)) flush(min(n_pending, 64u), w0 + (c + 1u) * kChunk);
    }
#pragma unroll
    for (uint32_t c = 0; c < kSitesChunks; c++) cur[c] = nxt[c];
  }
}

}  // namespace bvcf_dev
