// bvcf_sites.hip.h — k_sites: sites-only input (no sample columns) in one pass after the newline census
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// A sites-only line is ~150 bytes: a 1 KiB chunk holds seven of them, and per line the path needs the first TABs
// (strings.Split, main.go:535), the field count and the FILTER gate (linePasses, main.go:447-454), getAlleles
// (main.go:723-1038) and trTv (main.go:602-606) -- and writes 128 bytes of records.  The census chain spent its time
// re-reading: k_scatter_eol revisits the chunks for the line starts, k_head re-reads every line head with 16 lanes
// per line in 16 latency-bound rounds.  Here one wave walks a contiguous run of 1 KiB windows and reads every byte
// ONCE:
//   * each 1 KiB chunk goes from registers into a text ring in LDS (kSitesRing bytes: the last few windows);
//   * its terminator mask gives the line ends (appended to a FIFO in LDS), its TAB mask goes into a bit ring;
//   * whenever 64 line ends are pending -- or the oldest pending line is about to be overwritten in the ring, or the
//     run ends -- ONE LANE PER LINE walks the TAB bits of its line (first `need` TABs, total count), runs the gate
//     and getAlleles on the ring bytes and writes the line and allele records.  A line belongs to the run its
//     terminator is in; its start is the byte after the previous terminator, which the wave carries along.
// The line index of the first terminator of a run comes from the census prefix (k_count_eol / k_scan_*), so records
// land in input order without a second pass.  A line whose start is not in the ring -- the first line of every run,
// and lines longer than the ring's history -- takes a wave-cooperative path for its TABs and gets its first 256
// bytes staged on their own.
#pragma once

#include "bvcf_common.hip.h"
#include "bvcf_alleles.hip.h"
#include "bvcf_head.hip.h"

namespace bvcf_dev {

#ifndef BVCF_SITES_RING
#define BVCF_SITES_RING 8192
#endif
constexpr uint32_t kSitesWin = kChunk;                     // bytes per window (one chunk register, one more in flight)
constexpr uint32_t kSitesRing = BVCF_SITES_RING;           // text ring per wave: kSitesRing - kSitesWin bytes of history
constexpr uint32_t kSitesChunks = kSitesWin / kChunk;      // chunk registers per window
constexpr uint32_t kSitesFifo = 64 + kChunk;               // pending line ends: < 64 carried + one chunk's worth
constexpr uint32_t kSitesLongHead = 256;                   // bytes of a long line's head staged on their own
constexpr int kSitesWaves = 2;                             // waves per workgroup
constexpr int kSitesThreads = kSitesWaves * kWave;
static_assert((kSitesRing & (kSitesRing - 1)) == 0 && kSitesRing >= 2 * kSitesWin && kSitesRing <= 65536, "ring: power of two, u16 offsets");

struct SitesLds {
  uint8_t text[kSitesRing];          // byte at block offset o lives at text[o & (kSitesRing - 1)]
  uint32_t tabs[kSitesRing / 32];    // bit o & (kSitesRing - 1): byte o is a TAB
  uint16_t fifo[kSitesFifo];         // pending terminators: block offset & (kSitesRing - 1)
  uint8_t long_head[kSitesLongHead]; // the first bytes of a line whose start is not in the ring
  uint32_t long_tab[10];             // ... its first 9 TAB offsets and its TAB count
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

#ifdef BVCF_EXPERIMENTS  // round 2's kernel, slower than k_sites2: kept for A/B builds (make EXTRA=-DBVCF_EXPERIMENTS)
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
  // terminators before w0 = index of the line the first terminator of the run ends (census prefix of the chunk)
  const uint32_t c_first = w0 / kChunk;
  uint32_t rank = c_first < n_chunks ? a.census[c_first] + a.group_base[c_first / kScanGroup] : 0u;
  uint32_t ps = 0;  // start of the first pending line (of the line in progress when nothing is pending)
  if (win_lo > 0) {
    const uint32_t last = find_eol_before(a, w0);
    ps = last == kNone ? 0u : last + 1u;
  }
  uint32_t ring_lo = w0;     // oldest block offset the ring holds
  uint32_t staged_end = w0;  // the ring holds [ring_lo, staged_end)
  uint32_t n_pending = 0;

  // one lane per pending line, lines [0, n) of the FIFO
  auto flush = [&](uint32_t n) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool active = (uint32_t)lane < n;
    const uint32_t line = rank + (uint32_t)lane;
    // FIFO entries are ring offsets of bytes in [staged_end - kSitesRing, staged_end)
    const uint32_t ring_base = staged_end - kSitesRing;
    auto unring = [&](uint32_t rel) -> uint32_t { return ring_base + ((rel - ring_base) & kRingMask); };
    uint32_t ls = 0, le = 0;
    if (active) {
      le = unring(S.fifo[lane]);
      ls = lane == 0 ? ps : unring(S.fifo[lane - 1]) + 1u;
    }
    const uint32_t len = active && le + 1u - ls >= a.eol_chars ? le + 1u - ls - a.eol_chars : 0u;  // chomp, main.go:535
    const uint32_t cend = ls + len;
    // ---- a line whose start is not in the ring (only the first pending line can be one): its TABs by the whole
    // wave from memory, its first bytes into a buffer of their own
    const bool is_long = ps < ring_lo;
    if (is_long) {
      const uint32_t l_ls = ps, l_cend = bcast0(cend);
      uint32_t found = 0;
      for (uint32_t base = l_ls & ~15u; base < l_cend; base += kChunk) {
        const uint32_t off = base + 16u * lane;
        const u32x4 v = *reinterpret_cast<const u32x4_u *>(a.buf + min(off, cap_off));
        if (base == (l_ls & ~15u) && lane < (int)(kSitesLongHead / 16u + 1u)) {
          // bytes [l_ls, l_ls + kSitesLongHead) for lane 0's serial work (unaligned: byte-wise placement by the lanes
          // that hold them)
#pragma unroll
          for (uint32_t q = 0; q < 16; q++) {
            const uint32_t o = off + q;
            const uint32_t w = q < 4 ? v.x : (q < 8 ? v.y : (q < 12 ? v.z : v.w));
            if (o >= l_ls && o - l_ls < kSitesLongHead && off <= cap_off) S.long_head[o - l_ls] = (uint8_t)(w >> (8u * (q & 3u)));
          }
        }
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
#pragma unroll
    for (uint32_t k = 0; k < 9; k++) tab[k] = cend;
    if (active && !(is_long && lane == 0)) {
      if (ls < cend) {
        const uint32_t we = (cend - 1u) >> 5;  // last word with bytes of the line
        const uint32_t last_mask = (cend & 31u) ? (1u << (cend & 31u)) - 1u : 0xFFFFFFFFu;
        auto ldw = [&](uint32_t w) -> uint32_t { return S.tabs[w & (kSitesRing / 32u - 1u)] & (w == we ? last_mask : 0xFFFFFFFFu); };
        uint32_t wi = ls >> 5;
        uint32_t cur = ldw(wi) & (0xFFFFFFFFu << (ls & 31u));
#pragma unroll
        for (uint32_t k = 0; k < 9; k++) {
          if (k < need) {
            while (!cur && wi < we) cur = ldw(++wi);
            if (cur) {
              tab[k] = wi * 32u + (uint32_t)__ffs(cur) - 1u;
              cur &= cur - 1u;
              found = k + 1u;
            }
          }
        }
        n_tabs = found + __popc(cur);
        while (wi < we) n_tabs += __popc(ldw(++wi));
      }
    } else if (active) {
      n_tabs = S.long_tab[9];
      found = min(n_tabs, need);
#pragma unroll
      for (uint32_t k = 0; k < 9; k++)
        if (k < found) tab[k] = S.long_tab[k];
    }

    // ---- the serial work of the round, one lane per line, for the lanes with `on`: once for the lines that lie in
    // the ring (bytes straight from LDS, no test per byte), once more for a long first line (its staged head, then
    // memory)
    auto serial = [&](auto hb, const bool on) {
    const bool active = on;  // (shadows the round's: the lanes this pass works for)
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
    AlleleCtxT<decltype(hb)> c;
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
    };  // serial
    {
      BytesT<false> ring;
      ring.g = a.buf;
      ring.lds = as_lds(S.text);
      ring.lo = ring_lo;
      ring.n = staged_end - ring_lo;
      ring.sub = 0;
      ring.mask = kRingMask;
      serial(ring, active && !(is_long && lane == 0));
    }
    if (is_long) {
      BytesT<true> head;
      head.g = a.buf;
      head.lds = as_lds(S.long_head);
      head.lo = ls;
      head.n = kSitesLongHead;
      head.sub = ls;
      head.mask = 0xFFFFFFFFu;
      serial(head, active && lane == 0);
    }

    // ---- pop
    ps = bcast0(unring(S.fifo[n - 1]) + 1u);
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

  // ---- the run, window by window.  The next window's chunks are requested when this one starts and are not touched
  // before it ends.  One loop with one copy of the per-line code (flush): each turn either flushes a round of pending
  // lines or takes the next chunk; the chunk registers are picked by a switch over constants.
  u32x4 cur[kSitesChunks], nxt[kSitesChunks];
#pragma unroll
  for (uint32_t c = 0; c < kSitesChunks; c++) cur[c] = chunk_load(win_lo, c);
  uint32_t win = win_lo, c = 0;
  bool fresh = true;  // window `win` has not been started
  for (;;) {
    const bool at_end = win >= win_hi;
    // a full round; or what is pending when the run ends, or when the window about to be staged overwrites the ring
    // bytes the oldest pending line still lives in
    uint32_t n_flush = n_pending >= 64u ? 64u : 0u;
    if (!n_flush && n_pending) {
      const uint32_t next_end = win * kSitesWin + kSitesWin;
      if (at_end || (fresh && next_end >= kSitesRing && ps < next_end - kSitesRing)) n_flush = n_pending;
    }
    if (n_flush) {
      flush(n_flush);
      continue;
    }
    if (at_end) break;
    if (fresh) {
      w0 = win * kSitesWin;
      const uint32_t win_next = win + 1u < win_hi ? win + 1u : win;  // (the last window's prefetch is dropped)
#pragma unroll
      for (uint32_t k = 0; k < kSitesChunks; k++) nxt[k] = chunk_load(win_next, k);
      fresh = false;
      c = 0;
    }
    const u32x4 v = cur[0];
    static_assert(kSitesChunks == 1, "one chunk per window");
    const uint32_t off = w0 + c * kChunk + 16u * lane;
    uint32_t mt = eq_mask16(v, '\t'), me = eq_mask16(v, a.eol_byte);
    if (w0 + kSitesWin > nb) {  // (wave-uniform: the last window of the block)
      const uint32_t valid = bits_until(nb, off);
      mt &= valid;
      me &= valid;
    }
    stage(off, v, mt);
    staged_end = w0 + (c + 1u) * kChunk;
    if (staged_end - ring_lo > kSitesRing) ring_lo = staged_end - kSitesRing;
    if (__any(me != 0)) {
      uint32_t tot;
      uint32_t at = n_pending + wave_excl_scan(__popc(me), &tot);
      uint32_t m = me;
      while (m) {
        S.fifo[at++] = (uint16_t)((off + (uint32_t)__ffs(m) - 1u) & kRingMask);
        m &= m - 1;
      }
      n_pending += tot;
    }
    if (++c == kSitesChunks) {
#pragma unroll
      for (uint32_t k = 0; k < kSitesChunks; k++) cur[k] = nxt[k];
      win++;
      fresh = true;
    }
  }
}

#endif  // BVCF_EXPERIMENTS

}  // namespace bvcf_dev
