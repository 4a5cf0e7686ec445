// bvcf_stream.hip.h — the streaming path: k_stream (lines + ALT #1 scan in one pass) and k_order
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include <type_traits>

#include "bvcf_common.hip.h"
#include "bvcf_gtscan.hip.h"

namespace bvcf_dev {

// ------------------------------------------------------------------ k_stream: one wave per tile

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kDeferred = 0xFFFFFFFEu;  // StreamEntry.n_miss / GtResult.n_fields: scan left to k_gt

// first terminator byte at a position in [from, limit), or kNone; 4 KiB in flight per step
__device__ inline uint32_t find_eol(const KernelArgs &a, uint32_t from, uint32_t limit) {
  const int lane = lane_id();
  const uint32_t last_off = (a.cap - 16u) & ~3u;
  for (uint32_t base = from & ~3u; base < limit; base += 4u * kChunk) {  // dword-aligned loads (see realign)
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = ld_stream(a.buf + min(base + q * kChunk + 16u * lane, last_off));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t off = base + q * kChunk + 16u * lane;
      uint32_t m = eq_mask16(v[q], a.eol_byte) & bits_until(limit, off);
      if (off < from) m &= ~bits_until(from, off);  // the up to 3 bytes before `from`
      const unsigned long long b = __ballot(m != 0);
      if (b) {
        const int src = __ffsll((long long)b) - 1;
        return lane_value(off + __ffs(m) - 1, src);
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
  uint32_t valid = bits_until(a.nbytes, off);
  if ((uint32_t)lane >= n_lanes) valid = 0;
  const uint32_t me = eq_mask16(v, a.eol_byte) & valid;
  uint32_t mt = eq_mask16(v, '\t') & valid;
  const unsigned long long be = __ballot(me != 0);
  uint32_t eol_here = kNone;
  if (be) {
    const int src = __ffsll((long long)be) - 1;
    eol_here = lane_value(off + __ffs(me) - 1, src);
    mt &= bits_until(eol_here, off);  // TABs of this line only
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
    return lane_value(pos, src);
  }
  *found_io += tot;
  *eolp = eol_here;
  return kNone;
}

// The same for a 256 B window held by lanes 0..15 (the prefetched head of the next line): a DPP row
// scan replaces the 64-lane shuffle scan, and a terminator anywhere in the window simply declines
// (returns kNone: such a line is shorter than 256 B and goes through the general head scan).
// *mt_out: the lane's TAB mask (lanes 0..15 cover the window, 16 bytes each; bytes before `start` count as no TAB):
// handed to k_head with the line so that it does not tokenise the head again
__device__ __forceinline__ uint32_t head_window16(const KernelArgs &a, u32x4 v, uint32_t start, uint32_t *mt_out) {
  const int lane = lane_id();
  const uint32_t need = 9;
  // the window was loaded from the dword at or before `start`: blank the bytes of the previous line
  const uint32_t base = start & ~3u;
  if (lane == 0) v.x &= 0xFFFFFFFFu << (8u * (start & 3u));
  const uint32_t off = base + 16u * lane;
  uint32_t valid = bits_until(a.nbytes, off);
  if (lane >= 16) valid = 0;
  const uint32_t e4 = a.eol_byte * 0x01010101u;
  const uint32_t eol_any = (zero_bytes(v.x ^ e4) | zero_bytes(v.y ^ e4) | zero_bytes(v.z ^ e4) | zero_bytes(v.w ^ e4));
  if (__ballot(eol_any != 0 && lane < 16)) return kNone;
  const uint32_t mt = eq_mask16(v, '\t') & valid;
  *mt_out = mt;
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
  return lane_value(pos, src);
}

constexpr int kPipeChunks = 10;  // chunk registers of the cross-line pipeline: lines of <= 2560 samples

// -DBVCF_EXP_TIMES: per-wave start/end wall clock, hardware placement and time per pipeline phase
// (s_memtime stamps), read back through bvcf_debug_* by tools/wave_times.py.  Not part of the product.
#ifdef BVCF_EXP_TIMES
__device__ unsigned long long g_wave_t[2][32768];
__device__ unsigned long long g_phase_t[8][32768];
__device__ unsigned int g_wave_hw[2][32768];
// phase k = time from the previous stamp to STAMP(k); -DBVCF_EXP_TIMES=2 keeps only the per-wave start/end (the stamps
// cost registers, i.e. occupancy)
#if BVCF_EXP_TIMES + 0 == 2
#define STAMP(k)
#else
#define STAMP(k)                                                   \
  {                                                                \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    ph_[k] += now_ - last_;                                        \
    last_ = now_;                                                  \
  }
#endif
#else
#define STAMP(k)
#endif
__global__ __launch_bounds__(kWgThreads) void k_stream(KernelArgs a) {
#ifdef BVCF_EXP_TIMES
  if ((threadIdx.x & 63) == 0) {
    g_wave_t[0][wave_in_grid()] = wall_clock64();
    g_wave_hw[0][wave_in_grid()] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    g_wave_hw[1][wave_in_grid()] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
  unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWavesPerWg][kStageBytes];
  __shared__ __attribute__((aligned(16))) RawList s_sparse[kWavesPerWg];
  uint8_t *stage = s_stage[wave_in_wg()];
  RawList *sparse = &s_sparse[wave_in_wg()];
  const int lane = lane_id();
  // (threadIdx.x >> 6 is the same in all 64 lanes, which the compiler cannot know: without the broadcast everything
  // derived from the wave index -- tile numbers, run bounds, class-map slots -- lives in vector registers)
  const uint32_t wave = wave_in_grid();
  const uint32_t n_waves = gridDim.x * kWavesPerWg;
  const uint32_t ns = a.n_samples;
  const uint32_t nb = a.nbytes;
  const uint32_t T = a.tile_bytes;
  const bool maps = a.want_cmap != 0;
  const uint32_t n_chunks = (ns * 4u + kChunk - 1u) / kChunk;  // of a regular line
  const uint32_t table1 = (1u << 2) | (3u << 28);              // ALT #1
  // cross-line pipelining needs the whole line in the chunk registers and the in-scan terminator check
  const bool pipelined = n_chunks >= 1 && n_chunks <= (uint32_t)kPipeChunks && a.eol_chars == 1;
  uint32_t cm_next = 0, cm_end = 0;  // this wave's private block of class-map slots
  uint32_t seen = 0;                 // terminated lines this wave walked over
  uint32_t n_other = 0;              // listed lines that were not of the regular shape (left to k_gt)
  uint32_t n_listed = 0;             // lines listed in the tiles this wave has closed

  // A wave owns a contiguous run of tiles and walks it front to back, so only the first tile needs
  // a search for its first line start (those bytes are the previous wave's last line).  Entries
  // stay per tile: the quota argument is about bytes, not about who scans them.
  // (Claiming smaller runs from a counter instead evens out the waves' finish times but costs more
  // than it returns: every run start is a chain of dependent loads.  The kernel is bound by VALU
  // issue, not by the slowest wave: a SIMD arbitrates by age, so with equal runs the waves of the three
  // dispatch rounds finish at 161 / 196 / 216 us of a 240 us kernel, but runs weighted by those speeds
  // changed nothing -- two waves keep a SIMD as busy as three.)
  // Balanced: the first n_tiles % n_waves waves walk one tile more than the others (a plain ceil() split left 5 %
  // of the waves without work on the benchmark's 20 332 tiles).
  const uint32_t q_tiles = a.n_tiles / n_waves, r_tiles = a.n_tiles % n_waves;
  const uint32_t per_wave = q_tiles + (r_tiles ? 1u : 0u);  // the longest run, in tiles
  const uint32_t tile_lo = wave * q_tiles + min(wave, r_tiles);
  const uint32_t tile_hi = tile_lo + q_tiles + (wave < r_tiles ? 1u : 0u);
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

  // Class-map slots without atomics.  A line that is scanned here is regular, i.e. at least 4*ns + 8 bytes long,
  // so a run of B bytes holds at most B / (4*ns + 8) + 2 of them: wave w owns slots [w * W, (w + 1) * W) and the
  // arena stays within a few percent of compact.  (Asking a device counter for slots cost 8 % of this kernel
  // either way: 16 at a time, every request drains the loads in flight; a whole run at a time, 3 072 waves
  // hit one address when the kernel starts, and a single address serves about 90 atomics per microsecond.)
  // (a quarter more: a multiallelic line whose further alleles get class lists of their own takes a second slot; a wave
  // that runs out of them leaves such lines' further alleles to k_gt)
  const uint32_t lines_bound = (uint32_t)(((unsigned long long)per_wave * T) / (4ull * ns + 8ull)) + 2u;
  const uint32_t slots_per_wave = lines_bound + lines_bound / 4u;
  cm_next = wave * slots_per_wave;
  cm_end = cm_next + slots_per_wave;
  if (maps && wave == 0 && lane == 0)  // only waves that own tiles own slots; k_head's maps follow
    a.counters->cmap_maps = min(n_waves, a.n_tiles) * slots_per_wave;
  auto map_slot = [&]() -> uint32_t {
    if (!maps) return BVCF_NO_CMAP;
    if (cm_next >= cm_end) {  // unreachable by the bound above; never hand out a silent "no map"
      if (lane == 0) a.counters->pad[0] = 1;
      return BVCF_NO_CMAP;
    }
    return cmap_of(a, cm_next, true);
  };
  // list a line (in input order) in the tile it starts in
  // bits_ok: `bits` is the TAB mask of the line's head window (head_window16), one 16-bit piece per lane 0..15
  auto commit = [&](uint32_t ls, uint32_t cend, const GtStats &st, bool deferred, uint32_t cm_off, bool bits_ok = false,
                    uint32_t bits = 0u, uint32_t n_slots = 1u) {
    // (a deferred line gets its class map with its k_gt task, not here)
    while (ls >= (tile + 1) * T) {  // ls moved into a later tile of the run
      if (lane == 0) a.census[tile] = n_local;
      n_listed += n_local;
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
      en.len = (cend - ls) | (bits_ok ? kHasHeadBits : 0u);
      en.ac = st.ac;
      en.an = st.an;
      en.n_het = st.n_het;
      en.n_hom = st.n_hom;
      en.n_miss = deferred ? kDeferred : st.n_miss;
      en.cmap_off = cm_off;
      a.entries[(size_t)tile * a.tile_quota + n_local] = en;
    }
    if (bits_ok && lane < 16) a.head_bits[((size_t)tile * a.tile_quota + n_local) * 16u + (uint32_t)lane] = (uint16_t)bits;
    n_local++;
    if (maps && !deferred) cm_next += n_slots;
  };
  // chunk loads are dword-aligned and realigned in registers; geometries whose last field would
  // need a dword past the chunks (ns % 256 == 0) load unaligned instead (see gt_scan_fast)
  const uint32_t amask = (ns & 255u) ? 3u : 0u;
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  auto chunk_off = [&](uint32_t s_begin, uint32_t c) -> uint32_t {
    return min((s_begin & ~amask) + c * kChunk + 16u * lane, cap_off);
  };
  auto chunk_at = [&](uint32_t s_begin, uint32_t c) -> u32x4 { return ld_stream(a.buf + chunk_off(s_begin, c)); };
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
      //
      // Every load of the loop is issued unconditionally -- kLoads chunk loads per line, with a
      // harmless address when there is no B -- so that hipcc's s_waitcnt insertion can count the
      // loads in flight.  With loads behind `if (b_ok)` or `if (g < n_chunks)` its lower bound on
      // "loads younger than the one I need" is zero and every wait becomes vmcnt(0), which also
      // waits for the class-map and entry stores of the previous line (-10 % on k_stream).  The
      // loop is instantiated per chunk count, which also folds the last-chunk tests of the scan.
      auto run_pipeline = [&](auto loads_tag) {
        constexpr int kLoads = decltype(loads_tag)::value;
        constexpr uint32_t nc = kLoads;
        uint32_t pA = p, sA = s_begin, peA = (uint32_t)pred;
        u32x4 va[kPipeChunks];
        u32x4 hv = {0u, 0u, 0u, 0u};
        uint32_t mtA = 0, mtB = 0;  // head TAB masks of A (known once A came through the loop as B) and of B
        bool bitsA = false;
        auto head_at = [&](uint32_t start) -> u32x4 {  // 256 B from the dword at or before `start`, four times over
          return *reinterpret_cast<const u32x4_u *>(a.buf + min((start & ~3u) + 16u * (lane & 15), cap_off));
        };
        bool hv_ok = peA + 1u < r1;  // B starts inside this wave's run
        hv = head_at(hv_ok ? peA + 1u : sA);
#pragma unroll
        for (int g = 0; g < kPipeChunks; g++) va[g] = g < kLoads ? chunk_at(sA, g) : u32x4{0u, 0u, 0u, 0u};
        for (;;) {
          STAMP(0);
          // ---- B's head from the 256 B window
          uint32_t sB = 0, peB = 0;
          bool b_ok = false;
          if (hv_ok) {
            const uint32_t t9 = head_window16(a, hv, peA + 1u, &mtB);
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
          // C's head, ahead of B's chunks
          hv = head_at(hvc_ok ? peB + 1u : sA);
          STAMP(1);
          // ---- scan A, re-issuing each register for B
          const uint32_t cmA = bcast0(map_slot());  // (wave-uniform by construction; tell the compiler)
          uint8_t *cm = cmA != BVCF_NO_CMAP ? a.cmap + cmA : nullptr;
          // the line starts in list mode (nothing to zero, nothing classified per chunk) when a list fits the map's slot
#ifdef BVCF_EXP_NO_SPARSE
          const bool sparse_ok = false;
#else
          const bool sparse_ok = cm != nullptr && a.cmap_stride >= 4u * kSparseWords;
#endif
          FastAcc acc = {0, 0, 0, 0, sparse_ok ? 0u : kDenseMode};
          if (cm && !sparse_ok) zero_stage(stage, nc);
          const uint32_t rA = sA & amask;
          const uint32_t w0 = __builtin_amdgcn_alignbyte(__builtin_amdgcn_readfirstlane(va[0].y),
                                                         __builtin_amdgcn_readfirstlane(va[0].x), rA);
          const uint32_t sep = (w0 >> 8) & 0xFFu;
          if (sep != '|' && sep != '/') acc.bad = 1;
          const uint32_t kref = 0x09300030u | (sep << 8);
          const uint32_t term_xor = (a.eol_byte ^ 0x09u) << 24;
          const uint32_t s_next = b_ok ? sB : sA;  // (A again when there is no B: the loads are dropped)
          STAMP(2);
#pragma unroll
          for (int g = 0; g < kLoads; g++) {
            if ((uint32_t)g < nc) {
              const uint32_t nx = (uint32_t)g + 1u < nc
                                      ? (uint32_t)__builtin_amdgcn_readfirstlane(va[g + 1 < kPipeChunks ? g + 1 : g].x) : 0u;
              fast_chunk(realign(va[g], nx, rA), g, nc, ns, kref, table1, cm, stage, a.cmap_stride, term_xor, acc, sparse_ok ? sparse : nullptr);
            }
            va[g] = chunk_at(s_next, g);
          }
          STAMP(3);
          // a line that ends in list mode: its entries classified now, for ALT #1 and -- while ALT #1's list fits -- every
          // further ALT index; a line whose ALT #1 became a map (at its end, or on the way: it kept listing the lanes that
          // saw anything but 0 and 1) leaves its entries behind its slot for k_gt (a regular line is at least 4 ns + 8
          // bytes long: that bounds the lines that may follow in the run).
          uint32_t enc = 0, n_slots = 1;
          // the slots behind the line's own for its raw list (raw_save), while the range still covers one slot for every
          // line that may follow in the run and the arena covers them
          const uint32_t raw_slots = (kRawAreaBytes + a.cmap_stride - 1u) / a.cmap_stride;
          auto raw_area = [&]() -> uint8_t * {
            const uint32_t reserve = (r1 - min(r1, peA)) / (4u * ns + 8u) + 2u;
            if (cm_next + 1u + raw_slots + reserve > cm_end || cmap_of(a, cm_next + raw_slots, true) == BVCF_NO_CMAP) return nullptr;
            return cm + a.cmap_stride;
          };
          if (sparse_ok && acc.n_sp < kDenseMode) {
            enc = finish_list(sparse, acc, cm, min(kListAlleles, a.cmap_stride / (4u * kSparseWords)), stage, nc, a.cmap_stride,
                              acc.n_sp > BVCF_CMAP_SPARSE_MAX ? raw_area() : nullptr);
          } else if (sparse_ok && acc.n_oth == 0u) {
            enc = 1u << 1;
          } else if (sparse_ok && acc.n_oth < kDenseMode) {
            // the lanes that saw a digit >= 2 or a dot were listed on the way: k_gt settles the further alleles from them
            enc = finish_dense(sparse, acc.n_oth, raw_area());
          }
          if (enc == kRawEnc) n_slots += raw_slots;
          if (__any(acc.bad != 0)) {
            // A is not regular after all: B was predicted from a wrong line end.  Leave the
            // pipeline (the loads in flight are simply dropped) and take A the slow way.
            s_begin = sA;
            p = pA;
            break;
          }
          finish_stats(acc, &st);
          seen++;
          // (offsets are multiples of 16: the low bits tell k_head what finish_list left in the slot, see there)
          commit(pA, peA, st, false, cmA == BVCF_NO_CMAP ? cmA : cmA | enc, bitsA, mtA, n_slots);
          p = peA + 1u;
          if (!b_ok) {
            s_begin = kNone;  // nothing pending: rediscover from p
            break;
          }
          pA = peA + 1u;
          sA = sB;
          peA = peB;
          hv_ok = hvc_ok;
          mtA = mtB;
#ifdef BVCF_EXP_NO_HEAD_BITS  // (probe: what do the 32 bytes of TAB bitmap per line cost this kernel? k_head tokenises such lines itself)
          bitsA = false;
#else
          bitsA = true;
#endif
          STAMP(4);
        }
      };
      switch (n_chunks) {
        case 1: run_pipeline(std::integral_constant<int, 1>{}); break;
        case 2: run_pipeline(std::integral_constant<int, 2>{}); break;
        case 3: run_pipeline(std::integral_constant<int, 3>{}); break;
        case 4: run_pipeline(std::integral_constant<int, 4>{}); break;
        case 5: run_pipeline(std::integral_constant<int, 5>{}); break;
        case 6: run_pipeline(std::integral_constant<int, 6>{}); break;
        case 7: run_pipeline(std::integral_constant<int, 7>{}); break;
        case 8: run_pipeline(std::integral_constant<int, 8>{}); break;
        case 9: run_pipeline(std::integral_constant<int, 9>{}); break;
        default: run_pipeline(std::integral_constant<int, 10>{}); break;
      }
      static_assert(kPipeChunks == 10, "the dispatch above covers 1..10 chunks");
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
    // left to k_gt (k_head turns the entry into a task), which also settles its field count.  (A file made of such
    // lines is walked by k_stream_gen instead, see there and bvcf_core.hip.)
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
      if (cend - p + 1u >= a.n_header) {
        commit(p, cend, none, true, BVCF_NO_CMAP);
        n_other++;
      }
      p = cend + a.eol_chars;
    }
  }
  for (; tile < tile_hi; tile++) {  // the rest of the run has no line starts
    if (lane == 0) a.census[tile] = n_local;
    n_listed += n_local;
    n_local = 0;
  }
  if (lane == 0) a.run_lines[wave] = n_listed;  // (every wave of the grid, with or without tiles: k_order adds them up)
  if (lane == 0 && seen) atomicAdd(&a.counters->lines_seen, seen);
  if (lane == 0 && n_other) atomicAdd(&a.counters->n_other_shape, n_other);
#ifdef BVCF_EXP_TIMES
  if (lane == 0) {
    g_wave_t[1][wave] = wall_clock64();
    for (int k = 0; k < 8; k++) g_phase_t[k][wave] = ph_[k];
  }
#endif
}

}  // namespace bvcf_dev
#include "bvcf_streamgen.hip.h"
namespace bvcf_dev {

// tile-local entries -> input order, and the batch's line count.  One kernel (round 5; k_scan_groups + k_scan_top + a k_order
// that read their result before): with blocks in flight every kernel boundary of a chain is a wait for wave slots the other
// blocks' k_stream holds.  Workgroup b owns the tiles of a few consecutive waves of the one-pass kernel (their runs are
// contiguous); the lines before them are the sum of those waves' totals (run_lines: a few thousand words), and an exclusive
// scan of its own tiles' counts in LDS gives every entry its place.
__global__ __launch_bounds__(kWgThreads) void k_order(KernelArgs a) {
  __shared__ uint32_t s_part[2][kWavesPerWg];
  __shared__ uint32_t s_first[kWgThreads + 1];
  const int lane = lane_id();
  const uint32_t w = threadIdx.x >> 6;
  if (a.n_tiles == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters->n_lines = 0;
    return;
  }
  // the producer's split of the tiles over its waves (k_stream, k_stream_gen: the first n_tiles % n_waves runs are one longer)
  const uint32_t pw = a.prod_waves;
  const uint32_t q_tiles = a.n_tiles / pw, r_tiles = a.n_tiles % pw;
  auto run_first = [&](uint32_t wv) -> uint32_t { return wv * q_tiles + min(wv, r_tiles); };
  const uint32_t per_wg = (pw + gridDim.x - 1u) / gridDim.x;
  const uint32_t w_lo = min(blockIdx.x * per_wg, pw), w_hi = min(w_lo + per_wg, pw);
  const uint32_t t_lo = run_first(w_lo), t_hi = run_first(w_hi);
  // ---- lines before this range
  uint32_t sum = 0;
  {
    const uint32_t n4 = w_lo / 4u;
    const u32x4 *c4 = reinterpret_cast<const u32x4 *>(a.run_lines);
    for (uint32_t i = threadIdx.x; i < n4; i += kWgThreads) {
      const u32x4 v = c4[i];
      sum += v.x + v.y + v.z + v.w;
    }
    if (threadIdx.x < (w_lo & 3u)) sum += a.run_lines[n4 * 4u + threadIdx.x];
  }
  sum = wave_sum(sum);
  if (lane == 0) s_part[0][w] = sum;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t i = 0; i < kWavesPerWg; i++) base += s_part[0][i];
  if (w_hi == pw && w_lo < pw && t_lo >= t_hi && threadIdx.x == 0) {  // (a last range without tiles still owes the count)
    uint32_t rest = 0;
    for (uint32_t i = w_lo; i < w_hi; i++) rest += a.run_lines[i];
    a.counters->n_lines = base + rest;
  }
  if (t_lo >= t_hi) return;
  // ---- the range, kWgThreads tiles at a time
  for (uint32_t c0 = t_lo; c0 < t_hi; c0 += kWgThreads) {
    const uint32_t n_here = min(t_hi - c0, (uint32_t)kWgThreads);
    const uint32_t cnt = threadIdx.x < n_here ? min(a.census[c0 + threadIdx.x], a.tile_quota) : 0u;
    uint32_t wtot;
    const uint32_t pre = wave_excl_scan(cnt, &wtot);
    __syncthreads();  // (s_part[1], s_first: the round before is through)
    if (lane == 0) s_part[1][w] = wtot;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (uint32_t i = 0; i < kWavesPerWg; i++) {
      if (i < w) wbase += s_part[1][i];
      total += s_part[1][i];
    }
    s_first[threadIdx.x] = wbase + pre;
    if (threadIdx.x == 0) s_first[kWgThreads] = total;
    __syncthreads();
    const uint32_t pairs = n_here * a.tile_quota;
    for (uint32_t i = threadIdx.x; i < pairs; i += kWgThreads) {
      const uint32_t tl = i / a.tile_quota, k = i % a.tile_quota;
      const uint32_t first = s_first[tl];
      const uint32_t next = tl + 1u < kWgThreads ? s_first[tl + 1u] : s_first[kWgThreads];
      if (k >= next - first) continue;
      const uint32_t g = base + first + k;
      if (g >= a.max_lines) continue;
      const size_t ei = (size_t)(c0 + tl) * a.tile_quota + k;
      const StreamEntry en = a.entries[ei];
      a.line_off[g] = en.ls;
      a.line_len[g] = en.len & ~kNotRegular;  // (bit 31: line_bits[g] is valid)
      a.line_cmap[g] = en.cmap_off;
      if (en.len & kHasHeadBits) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.head_bits + ei * 16u);
        u32x4 *dst = reinterpret_cast<u32x4 *>(a.line_bits + (size_t)g * 8u);
        dst[0] = src[0];
        dst[1] = src[1];
      }
      if (g < a.max_tasks) {
        GtResult r;
        r.ac = en.ac;
        r.an = en.an;
        r.n_het = en.n_het;
        r.n_hom = en.n_hom;
        r.n_miss = en.n_miss;
        r.n_fields = en.n_miss == kDeferred ? kDeferred : a.n_header - 9u;
        // (k_stream only lists counts of lines its regular scan accepted; k_stream_gen marks lines with haploid / odd fields)
        r.regular = (en.n_miss == kDeferred || (en.len & kNotRegular)) ? 0u : 1u;
        r.pad = 0;
        a.results[g] = r;
      }
    }
    base += s_first[kWgThreads];  // (every thread reads it before the next round's first barrier)
  }
  if (t_hi == a.n_tiles && threadIdx.x == 0) a.counters->n_lines = base;
}


}  // namespace bvcf_dev
