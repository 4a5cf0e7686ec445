// bvcf_sites1.hip.h — k_sites2 and k_sites1: sites-only input (no sample columns) tile by tile
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// Both kernels share one body (s1_body): a wave loads a TILE of the text together with the bytes in front of it (an
// 8 KiB window = eight 1 KiB chunk loads, all in flight at once) into LDS -- text, a TAB bitmap and a terminator
// bitmap -- and owns the lines whose terminator lies in the tile.  The lead-in holds the start of the line that
// straddles the tile's start (a longer one takes the wave-cooperative path through memory, as in k_sites).  Then ONE
// LANE PER LINE, up to 64 lines per round, and the common lines never enter the general getAlleles code: with the
// line's first 64 TAB bits in a register pair the field ends are seven find-first-set steps; FILTER (up to four bytes
// against up to four keys) and a one-byte REF with a one-byte ACGT ALT that differs from it are tested on two unaligned
// dwords from LDS -- such a line is a SNP with its position taken verbatim (main.go:735-745) -- and lines that fail the
// field count or the FILTER gate are settled there as well.  Whatever is left in a round (indels, several ALTs,
// errors to log, TABs past byte 64, the long first line) goes through k_sites' per-line code with the other lanes
// switched off.  About 800 instructions per tile of ~50 lines where k_sites spent ~2 000 per round of ~40.
//
// What differs is where a tile's first line number comes from.
//
// k_sites2 (the default): from the newline census in front of it (k_count_eol + scans, as for k_sites).  Tiles are
//   7 chunks behind one chunk of lead-in, so a tile starts on a census entry; nothing depends on anything, the grid is
//   persistent, a wave strides through the tiles.  configs[1]: 70 us per 142 MB where k_sites takes 107; chain 106 us
//   (143).  The text is read twice (census, then here -- out of the Infinity Cache when the block fits it).
//
// k_sites1 (BVCF_SITES=3): in the same pass, the text is read once.  As soon as the terminators of its tile are counted
//   -- before anything else is done with the text -- the waves of a workgroup add their counts up in LDS and the
//   workgroup publishes ONE 16-bit word (count | published).  The line number of the workgroup's first line is the sum
//   of the words of all workgroups before it: wave 0 asks for the words of the last 1 024 workgroups (more than can be
//   in flight at once) and for the group sums of everything older, all in one go, right after publishing, and looks at
//   the answers after the per-line work of its tile's first round, whose results wait in registers meanwhile.  Group
//   sums are published by the last workgroup of each group from the words it has read anyway.  Plain relaxed
//   device-scope loads and stores of words that carry their own state.  Tile = workgroup number x waves per workgroup +
//   wave: a wave waits only for lower-numbered workgroups, and the dispatcher of each XCD starts its share of the
//   workgroups in order, so the lowest unfinished workgroup never waits for one that has not started (the assumption
//   every single-pass scan with static tile numbers makes; should it ever not hold, the look-back gives up after ~1 s
//   and flags the batch: an error, not a hang).  Records past a line's first take slots from max_lines up: the number
//   of lines is not known while the kernel runs.
//   It is correct (every test of k_sites2 runs through it as well) and it is NOT faster: 136-150 us per 142 MB.  What was
//   measured on the way (tools/s1_times.py, tools/s1bench.sh):
//     - returning atomics on one address, issued from all XCDs, complete at ~20 per microsecond: a ticket counter per
//       workgroup step (the deadlock-free way to order tiles in a persistent grid) cost 50-70 ns per ticket, serialised
//       (0.15-0.31 ms); accumulators shared by the 64 (4 096) tiles of a group: 0.2 ms (1 ms).  Hence no
//       read-modify-writes at all, static tile numbers, one workgroup per tile step;
//     - a wave that takes two tiles in a row publishes the second count only after finishing the first tile, which
//       waits for every earlier tile: a chain through the whole block, 8.5 ms;
//     - a device-scope load takes 1.5-3.5 us under load, and a tile cannot write its records before EVERY earlier tile's
//       text has arrived: workgroups that start together wait for the slowest load of the burst, finish together, and the
//       next generation repeats it (20 us per generation of 8 us of work).  Without that lockstep -- workgroups slowed
//       down at their start by a table copy + barrier -- the look-back itself is 1 us per tile, but then workgroup
//       turnover (5 us between a workgroup's end and its successor's first instruction with text) leaves 60 % of the wave
//       slots empty: 103-120 us.  Spreading the first generation's start over a tile's lifetime does not break the
//       lockstep (BVCF_S1_STAGGER_US).
//   The census costs 23 + 9 us and has none of these problems; k_sites1 stays selectable as the record of the attempt.
#pragma once

#include "bvcf_sites.hip.h"

namespace bvcf_dev {

constexpr uint32_t kS1Win = 8192;                      // bytes staged per tile step
constexpr uint32_t kS1Lead = 512;                      // k_sites1: of which lead-in (the tail of the previous tile)
constexpr uint32_t kS1Tile = kS1Win - kS1Lead;         // ... bytes a tile owns
constexpr uint32_t kS2Lead = kChunk;                   // k_sites2: tiles start on census chunks
constexpr uint32_t kS2Tile = kS1Win - kS2Lead;
constexpr uint32_t kS1Chunks = kS1Win / kChunk;        // chunk loads per tile step
#ifndef BVCF_S1_WAVES
#define BVCF_S1_WAVES 4
#endif
// k_sites2's waves per SIMD: three, by its 168 registers.  Measured with the general per-line code compiled out (100
// registers, no spills): four waves per SIMD are no faster (69.3 vs 69.9 us) -- the kernel is bound by what it moves,
// 153 MB in and 129 MB of records out per 142 MB of text; capped at 128 registers with that code in, it spills (84 us)
#ifndef BVCF_S2_WAVES_EU
#define BVCF_S2_WAVES_EU 3
#endif
// ONE tile per wave: a tile's count is published when the tile is staged.  (A wave that took two tiles in a row
// published the second one only after finishing the first -- which waits for every earlier tile, i.e. for the previous
// workgroup's second tile: a chain through the whole block, 8.5 ms instead of 0.1.)
constexpr int kS1Waves = BVCF_S1_WAVES;                // waves per workgroup
constexpr int kS1Threads = kS1Waves * kWave;
constexpr uint32_t kS1Group = 64;                      // workgroups per look-back group
constexpr uint32_t kS1LongHead = 256;

// state words (KernelArgs.census, zeroed by k_s1_zero before the launch): a word per group of 64 workgroups (sum |
// kS1StateA once published), then a 16-bit word per workgroup (count | 0x8000 once published)
constexpr uint32_t kS1StateA = 1u << 30, kS1Value = (1u << 30) - 1u;
constexpr uint32_t kS1LookLoads = 2;                           // 16-byte loads per lane
constexpr uint32_t kS1LookWgs = kS1LookLoads * kWave * 8u;     // workgroups whose own words are read (>= what can be in flight)
static_assert((uint32_t)kS1Waves * kS1Tile < 0x8000u, "a workgroup's count fits 15 bits");
__host__ __device__ inline uint32_t s1_n_tiles(uint32_t nbytes) { return (uint32_t)(((unsigned long long)nbytes + kS1Tile - 1u) / kS1Tile); }
__host__ __device__ inline uint32_t s2_n_tiles(uint32_t nbytes) { return (uint32_t)(((unsigned long long)nbytes + kS2Tile - 1u) / kS2Tile); }
// k_census_tiles (below): bundles of tiles, groups of bundles
constexpr uint32_t kS2BundleTiles = 16, kS2GroupBundles = 32, kS2GroupTiles = kS2BundleTiles * kS2GroupBundles;
__host__ __device__ inline uint32_t s2_n_bundles(uint32_t n_tiles) { return (n_tiles + kS2BundleTiles - 1u) / kS2BundleTiles; }
__host__ __device__ inline uint32_t s2_n_groups(uint32_t n_tiles) { return (n_tiles + kS2GroupTiles - 1u) / kS2GroupTiles; }
// (bundle totals follow the tile counts in KernelArgs.census, on a 64-entry boundary)
__host__ __device__ inline uint32_t s2_bundle_off(uint32_t n_tiles) { return (n_tiles + kS2BundleTiles + 63u) & ~63u; }
__host__ __device__ inline uint32_t s1_n_wgs(uint32_t n_tiles) { return (n_tiles + (uint32_t)kS1Waves - 1u) / (uint32_t)kS1Waves; }
__host__ __device__ inline uint32_t s1_l1_off() { return 4u; }
__host__ __device__ inline uint32_t s1_l0_off(uint32_t n_wgs) { return (s1_l1_off() + (n_wgs + kS1Group - 1u) / kS1Group + 3u) & ~3u; }  // 16-byte aligned
__host__ __device__ inline uint32_t s1_state_words(uint32_t nbytes) {
  const uint32_t nw = s1_n_wgs(s1_n_tiles(nbytes));
  return s1_l0_off(nw) + (nw + 1u) / 2u + kS1LookWgs / 2u + 8u;  // (the window's loads may start a little before and end a little behind the words in use)
}

// The terminator bitmap is read once per tile (the
// lane's four words, step 3), before the rounds start: the rounds' FIFO and the long first line's buffers live in its
// place.  A line's 64 TAB bits may be fetched from up to two words behind `tabs` -- whatever is there lies past the
// line's end and is masked off (a line of 64 bytes or more has its first 64 bytes inside the window).
constexpr uint32_t kS2Plane = 32u * 16u + 64u;  // one 16-byte piece of 32 records, and the skew
// (plain stores: the L2 gathers the pieces of a cache line; as non-temporal stores the old per-lane pieces took 123 us
// instead of 69, whole lines 67 instead of 64)
__device__ __forceinline__ void s2_store(u32x4 v, u32x4 *p) { *p = v; }
struct S1Lds {
  uint8_t text[kS1Win];                 // byte x of the window
  uint32_t tabs[kS1Win / 32];           // bit x: byte x is a TAB
  union {
    uint32_t eols[kS1Win / 32];         // bit x: byte x is the terminator
    struct {
      uint32_t fifo[kWave];             // the round's line ends: window offset | TABs before it << 16
      uint8_t long_head[kS1LongHead];   // the first bytes of a line that starts before the window
      uint32_t long_tab[10];            // ... its first 9 TAB offsets (block offsets) and its TAB count
    };
  };
  uint32_t xpose[4 * kS2Plane / 4];     // 32 line records on their way out, piece by piece (see the rounds)
};
static_assert(sizeof(S1Lds) == 10240 + 4 * kS2Plane, "three workgroups per CU (the registers allow no more)");

__device__ __forceinline__ uint32_t s1_load(const uint32_t *p) {
#ifdef BVCF_EXP_S1_RMW
  return atomicOr(const_cast<uint32_t *>(p), 0u);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void s1_store(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bits [0, n) for n in 0..32
__device__ __forceinline__ uint32_t low_bits(uint32_t n) { return n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u; }

#ifdef BVCF_EXPERIMENTS
// the state words of the next k_sites1 launch and the batch counters, zeroed by one small launch
__global__ __launch_bounds__(256) void k_s1_zero(KernelArgs a, uint32_t n_words) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) a.census[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x < sizeof(BatchCounters) / 4u) reinterpret_cast<uint32_t *>(a.counters)[threadIdx.x] = 0u;
}

#endif  // BVCF_EXPERIMENTS

// kCensus = false: k_sites1, the line numbers by look-back (one tile per wave, workgroup b takes tiles kS1Waves*b ..);
// kCensus = true:  k_sites2, the line numbers from the newline census in front of it (a persistent grid, tiles strided)
// kPacked (k_sites2p): the packed form of the results, a 32-byte bvcf_site per line
template <bool kCensus, bool kPacked = false>
__device__ __forceinline__ void s1_body(const KernelArgs &a, const uint32_t n_tiles, const uint32_t n_chunks) {
  constexpr uint32_t kLead = kCensus ? kS2Lead : kS1Lead;
  constexpr uint32_t kTile = kS1Win - kLead;
  __shared__ __attribute__((aligned(16))) S1Lds s_lds[kS1Waves];
  __shared__ uint32_t s_cnt[kS1Waves], s_wg_base;
  const int lane = lane_id();
  const uint32_t wiw = bcast0(threadIdx.x >> 6);
  S1Lds &S = s_lds[wiw];
  const uint32_t nb = a.nbytes;
  const uint32_t need = min(9u, a.n_header - 1u);  // TABs that bound the fixed columns we read
  const uint32_t cap_off = (a.cap - 16u) & ~3u;
  uint32_t *const st_l1 = a.census + s1_l1_off();
  uint16_t *const st_l0 = reinterpret_cast<uint16_t *>(a.census + s1_l0_off(gridDim.x));
  // the packed form (bvcf_params.packed_sites, k_sites2 only): a 32-byte bvcf_site per line, full records only for the
  // lines that leave the fast lanes, in slots handed out per round
  constexpr bool packed = kPacked;
  static_assert(kCensus || !kPacked, "the packed form is k_sites2's");
  // records past a line's first: behind the lines' slots -- whose number only the census knows in advance (packed: how
  // many lines get full records is not known either, so there too they follow slot max_lines)
  const uint32_t extras_at = (kCensus && !packed) ? min(a.counters->n_lines, a.max_lines) : a.max_lines;

  // ---- the FILTER gate of the common lines (linePasses, main.go:447-454): up to four allowed values of up to four
  // bytes as dwords, nothing excluded -- prepared on the host (KernelArgs.s1_*).  mode 0: no such table (every line that
  // reaches the gate goes the general way), 1: keys, 2: nothing to test
  uint32_t f_mode = a.s1_fmode;
  if (a.n_header != 8u && a.n_header != 9u) f_mode = 0xFFu;  // (the fast lanes know the 8- and 9-column layouts)

#ifdef BVCF_EXP_TIMES
#define S1STAMP(k)                                                   \
  {                                                                  \
    const unsigned long long now_ = __builtin_readcyclecounter();    \
    if (lane == 0) g_phase_t[k][t & 32767u] = now_ - s1_last_;       \
    s1_last_ = now_;                                                 \
  }
#else
#define S1STAMP(k)
#endif
  u32x4 v[kS1Chunks];
  auto load_window = [&](uint32_t t) {
    const unsigned long long ws = t ? (unsigned long long)t * kTile - kLead : 0ull;
#pragma unroll
    for (uint32_t c = 0; c < kS1Chunks; c++) {
      const unsigned long long off = ws + c * kChunk + 16u * lane;
      v[c] = ld_stream(a.buf + (off < cap_off ? (uint32_t)off : cap_off));
    }
  };
  // ---- one tile (its window is in v[] already)
  auto process_tile = [&](const uint32_t t) {
#ifdef BVCF_EXP_TIMES
    unsigned long long s1_last_ = __builtin_readcyclecounter();
    if (lane == 0) g_wave_t[0][t & 32767u] = wall_clock64();
#endif
    const bool has_tile = t < n_tiles;  // (the last workgroup's spare waves only keep the barriers company)
    const uint32_t tile_start = has_tile ? t * kTile : 0u;
    const uint32_t lead = t ? kLead : 0u;
    const uint32_t win_start = tile_start - lead;
    const uint32_t tile_end = (uint32_t)min((unsigned long long)tile_start + kTile, (unsigned long long)nb);
    // window coordinates: the tile's own bytes are [lead, hi_w); bytes of the block end at end_w
    const uint32_t hi_w = lead + (tile_end - tile_start);
    const uint32_t end_w = (uint32_t)min((unsigned long long)nb - win_start, (unsigned long long)kS1Win);

    // ---- 1. eight chunks in flight (load_window), then masks and staging
    uint32_t rank_c = 0, rank_g = 0;
    if constexpr (kCensus) {
      // terminators before the tile = the census prefix of its first chunk (k_count_eol / k_scan_*): asked for now, used
      // when the records are written
      const uint32_t c_first = tile_start / kChunk;
      if (n_chunks == 0u && !kPacked) {  // the census was taken per tile (k_count_tiles) and scanned in one go
        if (has_tile) rank_c = a.census[t];
      } else if (n_chunks == 0u) {
        // the census per tile (k_census_tiles), not scanned: every lane asks for its share of what lies in front of the
        // tile -- lanes 0..15 the tiles of its bundle, lanes 32..63 the bundles of its group, all lanes the groups before
        if (has_tile) {
          const uint32_t ti = t % kS2BundleTiles, bundle = t / kS2BundleTiles, bi = bundle % kS2GroupBundles, grp = t / kS2GroupTiles;
          const uint32_t l = (uint32_t)lane;
          const bool tile_lane = l < ti, bundle_lane = l >= 32u && l - 32u < bi;
          if (tile_lane | bundle_lane)
            rank_c = a.census[tile_lane ? t - ti + l : s2_bundle_off(n_tiles) + bundle - bi + (l - 32u)];
          if (l < grp) rank_g = a.s2_groups[l];  // (used where rank_c is: no wait here)
          if (grp > kWave)                       // (a batch of more than 224 MiB: the further groups, at a load's latency)
            for (uint32_t j = l + kWave; j < grp; j += kWave) rank_c += a.s2_groups[j];
        }
      } else if (c_first < n_chunks) {
        rank_c = a.census[c_first];
        rank_g = a.group_base[c_first / kScanGroup];
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the previous tile's rounds are done with the LDS)
    S1STAMP(0)
    // ---- 2. the terminators first: the tile's count is what every later tile waits for.  It goes out before anything
    // else is done with the text; the atomic that tells whether this tile completes its group is in flight while the
    // TAB masks are formed and the window is staged.
    uint32_t me[kS1Chunks];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t c = 0; c < kS1Chunks; c++) {
      me[c] = eq_mask16(v[c], a.eol_byte);
      const uint32_t x = c * kChunk + 16u * lane;  // window offset of the lane's bytes
      cnt += __popc(me[c] & bits_until(hi_w, x) & (x >= lead ? 0xFFFFu : 0u));
    }
    const uint32_t tile_eols = has_tile ? wave_sum(cnt) : 0u;
    uint32_t lines_before = 0, wg_total = 0;  // in this workgroup: in front of this wave's tile, in all of it
    if constexpr (!kCensus) {
      if (lane == 0) s_cnt[wiw] = tile_eols;
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < (uint32_t)kS1Waves; j++) {
        const uint32_t cj = s_cnt[j];
        wg_total += cj;
        lines_before += j < wiw ? cj : 0u;
      }
    }
    // wave 0 publishes the workgroup's count and asks for everything in front of it: the words of the workgroups of the
    // last kS1LookGroups groups, the sums of the groups before those
    const uint32_t wg = blockIdx.x, grp = wg / kS1Group;
    // the window: the 16-bit words of workgroups [look_lo, look_lo + kS1LookWgs), a multiple of 64 that ends behind this one
    const uint32_t look_lo = (grp + 1u) * kS1Group > kS1LookWgs ? (grp + 1u) * kS1Group - kS1LookWgs : 0u;
    const uint32_t n_old = look_lo / kS1Group;  // groups read as sums
    u32x4 xs[kS1LookLoads];
    uint32_t y0 = kS1StateA;
#pragma unroll
    for (uint32_t k = 0; k < kS1LookLoads; k++) xs[k] = u32x4{0u, 0u, 0u, 0u};
    auto ask = [&](uint32_t k) -> u32x4 {
      // (device-scope loads: two 8-byte ones per 16 bytes)
      const unsigned long long *p = reinterpret_cast<const unsigned long long *>(st_l0 + look_lo + 8u * (64u * k + lane));
      const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    };
    if (!kCensus && wiw == 0) {
      if (lane == 0) __hip_atomic_store(&st_l0[wg], (uint16_t)(0x8000u | wg_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (uint32_t k = 0; k < kS1LookLoads; k++) xs[k] = ask(k);
      if ((uint32_t)lane < n_old) y0 = s1_load(&st_l1[lane]);
    }
    S1STAMP(1)
#pragma unroll
    for (uint32_t c = 0; c < kS1Chunks; c++) {
      const uint32_t mt = eq_mask16(v[c], '\t');
      *reinterpret_cast<u32x4 *>(&S.text[c * kChunk + 16u * lane]) = v[c];
      reinterpret_cast<uint16_t *>(S.tabs)[c * 64u + lane] = (uint16_t)mt;
      reinterpret_cast<uint16_t *>(S.eols)[c * 64u + lane] = (uint16_t)me[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- 3. the lane's 128 bytes of the window: four words of each bitmap
    const u32x4 tw4 = *reinterpret_cast<const u32x4 *>(&S.tabs[4 * lane]);
    const u32x4 ew4 = *reinterpret_cast<const u32x4 *>(&S.eols[4 * lane]);
    uint32_t tw[4] = {tw4.x, tw4.y, tw4.z, tw4.w}, ew[4] = {ew4.x, ew4.y, ew4.z, ew4.w}, el[4];
    const uint32_t lane_w = 128u * lane;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t ws = lane_w + 32u * k;
      // bytes past the end of the block are not text (clamped loads)
      if (end_w < kS1Win) {
        const uint32_t okb = low_bits(end_w > ws ? end_w - ws : 0u);
        tw[k] &= okb;
        ew[k] &= okb;
      }
      // terminators of the lead-in (they place the first line's start) and of the tile proper
      el[k] = ws < lead ? ew[k] : 0u;
      uint32_t own = ws >= lead ? ew[k] : 0u;
      if (hi_w < kS1Win) own &= low_bits(hi_w > ws ? hi_w - ws : 0u);
      ew[k] = own;
    }
    if (end_w < kS1Win) {  // (the clamped bits are what the rounds read)
      *reinterpret_cast<u32x4 *>(&S.tabs[4 * lane]) = u32x4{tw[0], tw[1], tw[2], tw[3]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const uint32_t ce = __popc(ew[0]) + __popc(ew[1]) + __popc(ew[2]) + __popc(ew[3]);
    const uint32_t ct = __popc(tw[0]) + __popc(tw[1]) + __popc(tw[2]) + __popc(tw[3]);
    const uint32_t incl = wave_incl_scan(ce | (ct << 16));
    const uint32_t excl = incl - (ce | (ct << 16));

    S1STAMP(2)
    // ---- 4. where the first line of the tile starts: after the last terminator of the lead-in
    uint32_t ps_w = 0, ps_tc = 0;  // window offset of that start, TABs of the window before it
    bool long_first = false;
    uint32_t long_ps = 0;  // block offset of the first line's start when it lies before the window
    if (t > 0 && has_tile) {
      uint32_t hi_pos = 0, hi_tc = 0;
      bool have = false;
      uint32_t tc_run = excl >> 16;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        if (el[k]) {
          const uint32_t b = 31u - (uint32_t)__clz(el[k]);
          hi_pos = lane_w + 32u * k + b;
          hi_tc = tc_run + __popc(tw[k] & low_bits(b));
          have = true;
        }
        tc_run += __popc(tw[k]);
      }
      const unsigned long long bl = __ballot(have);
      if (bl) {
        const int src = 63 - __clzll((long long)bl);
        ps_w = lane_value(hi_pos, src) + 1u;
        ps_tc = lane_value(hi_tc, src);
      } else {
        long_first = true;
        const uint32_t last = find_eol_before(a, win_start);
        long_ps = last == kNone ? 0u : last + 1u;
      }
    }

    // ---- 5. the look-back: lines before this tile (once per tile, after the first round's per-line work)
    uint32_t base = 0;
    bool base_known = false;
    auto look_back = [&]() {
      if (base_known) return;
      base_known = true;
      if constexpr (kCensus) {
        base = (kPacked && n_chunks == 0u) ? wave_sum(rank_c + rank_g) : rank_c + rank_g;
        return;
      }
      if (wiw == 0) {
        uint32_t patience = 1u << 20;  // polls (~1 us each) before the wave gives up
        // the answers asked for above; whatever was not there yet is asked for again
        uint32_t old_sum = 0, near = 0, own_grp = 0;
        for (;;) {
          bool missing = (y0 >> 30) == 0u;
          old_sum = y0 & kS1Value;
          for (uint32_t h = kWave; h < n_old; h += kWave) {  // (blocks past ~190 MB: more than 64 old groups)
            const uint32_t y = h + lane < n_old ? s1_load(&st_l1[h + lane]) : kS1StateA;
            missing = missing || (y >> 30) == 0u;
            old_sum += y & kS1Value;
          }
          near = 0;
          own_grp = 0;
#pragma unroll
          for (uint32_t k = 0; k < kS1LookLoads; k++) {
            const uint32_t w4[4] = {xs[k].x, xs[k].y, xs[k].z, xs[k].w};
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
              const uint32_t i = look_lo + 8u * (64u * k + lane) + q;
              const uint32_t e = (w4[q >> 1] >> (16u * (q & 1u))) & 0xFFFFu;
              if (i < wg) {
                missing = missing || !(e & 0x8000u);
                near += e & 0x7FFFu;
                own_grp += i >= grp * kS1Group ? (e & 0x7FFFu) : 0u;
              }
            }
          }
          if (!__any(missing)) break;
          if (--patience == 0u) {
            if (lane == 0) a.counters->pad[0] = 1u;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
#pragma unroll
          for (uint32_t k = 0; k < kS1LookLoads; k++) xs[k] = ask(k);
          if ((uint32_t)lane < n_old && (y0 >> 30) == 0u) y0 = s1_load(&st_l1[lane]);
        }
        const uint32_t wg_base = wave_sum(old_sum + near);
        // the last workgroup of a group publishes the group's sum (its own group's words are among those it has read)
        if (wg % kS1Group == kS1Group - 1u) {
          const uint32_t gs = wave_sum(own_grp) + wg_total;
          if (lane == 0) s1_store(&st_l1[grp], kS1StateA | gs);
        }
        if (lane == 0) s_wg_base = wg_base;
      }
      __syncthreads();
      base = s_wg_base + lines_before;
    };

    if (!has_tile) {
      look_back();
      return;
    }

    // ---- 6. rounds of up to 64 lines, one lane per line
    uint32_t n_done = 0;
    uint32_t prev = ps_w | (ps_tc << 16);  // (window offset of the previous line's end + 1) | TABs before it << 16
    bool first_round = true;
    while (n_done < tile_eols) {
      const uint32_t n = min(tile_eols - n_done, (uint32_t)kWave);
      // the round's line ends, in order
      {
        uint32_t rk = (excl & 0xFFFFu) - n_done, tc = excl >> 16;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          uint32_t m = ew[k];
          while (m) {
            const uint32_t b = (uint32_t)__ffs(m) - 1u;
            m &= m - 1u;
            if (rk < (uint32_t)kWave) S.fifo[rk] = (lane_w + 32u * k + b) | ((tc + __popc(tw[k] & low_bits(b))) << 16);
            rk++;
          }
          tc += __popc(tw[k]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const bool active = (uint32_t)lane < n;
      const uint32_t e_me = active ? S.fifo[lane] : 0u;
      const uint32_t e_pv = lane == 0 ? prev : (active ? S.fifo[lane - 1] + 1u : 0u);  // (+1: the byte after that terminator)
      const bool is_long = first_round && long_first;  // lane 0's line starts before the window
      const uint32_t le_w = e_me & 0xFFFFu;
      const uint32_t ls_w = e_pv & 0xFFFFu;
      uint32_t n_tabs = ((e_me >> 16) - (e_pv >> 16)) & 0xFFFFu;
      const uint32_t ls = (is_long && lane == 0) ? long_ps : win_start + ls_w;  // block offsets
      const uint32_t le = win_start + le_w;
      const uint32_t len = active && le + 1u - ls >= a.eol_chars ? le + 1u - ls - a.eol_chars : 0u;  // chomp, main.go:535
      const uint32_t cend = ls + len;

      // ---- the common lines
      // first 64 TAB bits of the line (bit i: byte ls + i), cut at the line's end
      const uint32_t wi = ls_w >> 5, sh = ls_w & 31u;
      const uint32_t b0 = S.tabs[wi], b1 = S.tabs[wi + 1], b2 = S.tabs[wi + 2];
      unsigned long long W = (unsigned long long)__builtin_amdgcn_alignbit(b1, b0, sh) |
                             ((unsigned long long)__builtin_amdgcn_alignbit(b2, b1, sh) << 32);
      if (len < 64u) W &= (1ull << len) - 1ull;
      const uint32_t want_k = min(n_tabs, need);
      bool simple = active && !(is_long && lane == 0) && f_mode != 0xFFu && (uint32_t)__popcll(W) >= want_k;
      uint32_t fe[9];
#pragma unroll
      for (uint32_t k = 0; k < 9; k++) {
        fe[k] = len;
        if (k < 8) {  // (need <= 8 on this path)
          if (W && k < want_k) fe[k] = (uint32_t)__ffsll((long long)W) - 1u;
          W &= W - 1ull;
        }
      }
      uint32_t status = (n_tabs + 1u != a.n_header) ? (uint32_t)BVCF_LINE_FIELDS : (uint32_t)BVCF_LINE_OK;
      uint32_t ref_b = 0, alt_b = 0;
      if (__any(simple && status == BVCF_LINE_OK)) {
        // FILTER: field 6, up to four bytes of it as a dword
        const uint32_t f_off = ls_w + fe[5] + 1u, f_n = fe[6] - fe[5] - 1u;
        const uint32_t fa = f_off & ~3u;
        const uint32_t fw = __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t *>(&S.text[(fa + 4u) & (kS1Win - 1u)]),
                                                       *reinterpret_cast<const uint32_t *>(&S.text[fa & (kS1Win - 1u)]), f_off & 3u);
        const uint32_t fv = fw & low_bits(8u * min(f_n, 4u));
        bool pass = f_mode == 2u;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) pass = pass || (a.s1_flen[i] != 0u && f_n == a.s1_flen[i] && fv == a.s1_fkey[i]);
        if (f_mode == 0u && status == BVCF_LINE_OK) simple = false;
        if (simple && status == BVCF_LINE_OK && !pass) status = BVCF_LINE_FILTER;
        // REF, ALT: "<ref>\t<alt>\t" at field 3
        const uint32_t r_off = ls_w + fe[2] + 1u;
        const uint32_t ra = r_off & ~3u;
        const uint32_t rw = __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t *>(&S.text[(ra + 4u) & (kS1Win - 1u)]),
                                                       *reinterpret_cast<const uint32_t *>(&S.text[ra & (kS1Win - 1u)]), r_off & 3u);
        ref_b = rw & 0xFFu;
        alt_b = (rw >> 16) & 0xFFu;
        if (simple && status == BVCF_LINE_OK) {
          const bool snp = fe[3] - fe[2] == 2u && fe[4] - fe[3] == 2u && is_actg((uint8_t)alt_b) && alt_b != ref_b;
          if (!snp) simple = false;
        }
      }
      const bool general = active && !simple;

      // ---- the line number of the round's first line
      S1STAMP(3)
      if (!base_known) look_back();
      S1STAMP(4)
      const uint32_t line = base + n_done + (uint32_t)lane;

      // ---- the records of the common lines, written as whole cache lines.  A lane storing its own 64-byte records
      // touches 64 cache lines per store instruction, a quarter of each: with eight such stores per round the kernel
      // took 69 us per 142 MB, 43 without them.  The round's records are consecutive in memory (line = base + lane), so
      // they go out transposed -- lane L stores bytes [16 L, 16 L + 16) of a KiB, piece L & 3 of record L >> 2 --: the
      // line records through an LDS scratch, 32 at a time (planes of one 16-byte piece each, skewed by 64 bytes so that
      // the four pieces of a record sit in different banks); the allele records, constants but for three bytes, from a
      // ds_bpermute of those.
      // ---- the packed form: the round's 64 site records are 2 KiB of consecutive memory.  Every lane puts its record
      // into the LDS scratch in memory order, the wave copies the scratch out as whole cache lines (lane L stores bytes
      // [16 L, 16 L + 16) of each KiB).  A line that needs full records gets their slot first (one atomic per round that
      // has such lines), so that its site record -- "see lines[full_idx]" -- goes out with the others.
      uint32_t lslot = line;  // slot of the line's full records in lines[] / alleles[]
      if constexpr (packed) {
        const unsigned long long gm = __ballot(general && line < a.max_lines);
        if (gm) {
          uint32_t got = 0;
          if (lane == 0) got = atomicAdd(&a.counters->n_full, (uint32_t)__popcll(gm));
          lslot = bcast0(got) + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull));
        }
        uint32_t fb[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (simple) {
          fb[0] = fb[1] = 0u;
#pragma unroll
          for (uint32_t k = 0; k < 8; k++) fb[k >> 2] |= (k < want_k ? fe[k] : 0xFFu) << (8u * (k & 3u));
        }
        const uint32_t info = simple ? (ref_b | (alt_b << 8) | ((uint32_t)trtv_of((uint8_t)ref_b, (uint8_t)alt_b) << 16) | (status << 24))
                                     : ((uint32_t)BVCF_SITE_FULL << 24);
        uint8_t *const xp = reinterpret_cast<uint8_t *>(S.xpose);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4 *>(xp + 32u * (uint32_t)lane) = u32x4{ls, len, fb[0], fb[1]};
        *reinterpret_cast<u32x4 *>(xp + 32u * (uint32_t)lane + 16u) = u32x4{info, simple ? 0u : lslot, n_tabs + 1u, 0u};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t line0 = base + n_done;  // (wave-uniform)
        const uint32_t room = line0 < a.max_lines ? a.max_lines - line0 : 0u;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
          const uint32_t rec = 32u * h + ((uint32_t)lane >> 1);
          const u32x4 v = *reinterpret_cast<const u32x4 *>(xp + 1024u * h + 16u * (uint32_t)lane);
          if (rec < n && rec < room) s2_store(v, reinterpret_cast<u32x4 *>(&a.sites[line0]) + 64u * h + lane);
        }
      }
      const unsigned long long smask = packed ? 0ull : __ballot(simple && line < a.max_lines);
      if (!packed && smask) {
        const uint32_t line0 = base + n_done;  // (wave-uniform)
        const bool ok = status == BVCF_LINE_OK;
        const u32x4 pc[4] = {u32x4{ls, len, fe[0], fe[1]}, u32x4{fe[2], fe[3], fe[4], fe[5]}, u32x4{fe[6], fe[7], fe[8], 0u},  // .. rec_first
                             u32x4{ok ? 1u : 0u, n_tabs + 1u, line, status}};  // n_rec, n_fields, gt_task, status | site_type << 8
        uint8_t *const xp = reinterpret_cast<uint8_t *>(S.xpose);
        const uint32_t piece = (uint32_t)lane & 3u, sub = (uint32_t)lane >> 2;
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
          if (((uint32_t)(smask >> (32u * h))) == 0u) continue;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (((uint32_t)lane >> 5) == h) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) *reinterpret_cast<u32x4 *>(xp + kS2Plane * q + 16u * ((uint32_t)lane & 31u)) = pc[q];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (uint32_t k = 0; k < 2; k++) {
            const uint32_t src = 32u * h + 16u * k + sub;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(xp + kS2Plane * piece + 16u * (16u * k + sub));
            if ((smask >> src) & 1ull) s2_store(v, reinterpret_cast<u32x4 *>(&a.lines[line0 + 32u * h + 16u * k]) + lane);
          }
        }
        // alleles: {pos = 0 (text), line, alt_idx 0} {alt_off 0, alt_len 1, ac 0, an 0} {n_het, n_hom, n_miss 0, no class map}
        // {ref alt_base kind 0 site_type 0 | trtv flags | no task | 0}; a line that failed the gate gets the same record (all
        // that counts there is "no task")
        const uint32_t packed = ref_b | (alt_b << 8) | ((uint32_t)trtv_of((uint8_t)ref_b, (uint8_t)alt_b) << 16);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const uint32_t src = 16u * k + sub;
          const uint32_t info = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)packed);
          const uint32_t rec = line0 + src;
          u32x4 v = u32x4{0u, 0u, rec, 0u};
          if (piece == 1u) v = u32x4{0u, 1u, 0u, 0u};
          if (piece == 2u) v = u32x4{0u, 0u, 0u, BVCF_NO_CMAP};
          if (piece == 3u) v = u32x4{info & 0xFFFFu, (info >> 16) | ((uint32_t)BVCF_ALLELE_POS_TEXT << 8), kNoTask, 0u};
          if (((smask >> src) & 1ull) && rec < a.max_alleles) s2_store(v, reinterpret_cast<u32x4 *>(&a.alleles[line0 + 16u * k]) + lane);
        }
      }

      // ---- everything else: k_sites' per-line code for the lanes that are left
      if (__any(general)) {
        if (is_long) {
          // the first line starts before the window: its TABs by the whole wave from memory, its first bytes into a
          // buffer of their own
          const uint32_t l_ls = long_ps, l_cend = bcast0(cend);
          uint32_t found = 0;
          for (uint32_t bs = l_ls & ~15u; bs < l_cend; bs += kChunk) {
            const uint32_t off = bs + 16u * lane;
            const u32x4 vv = *reinterpret_cast<const u32x4_u *>(a.buf + min(off, cap_off));
            if (bs == (l_ls & ~15u) && lane < (int)(kS1LongHead / 16u + 1u)) {
#pragma unroll
              for (uint32_t q = 0; q < 16; q++) {
                const uint32_t o = off + q;
                const uint32_t w = q < 4 ? vv.x : (q < 8 ? vv.y : (q < 12 ? vv.z : vv.w));
                if (o >= l_ls && o - l_ls < kS1LongHead && off <= cap_off) S.long_head[o - l_ls] = (uint8_t)(w >> (8u * (q & 3u)));
              }
            }
            uint32_t m = eq_mask16(vv, '\t') & bits_until(l_cend, off) & (off <= cap_off ? 0xFFFFu : 0u);
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
        // strings.Split(row, "\t") from the TAB bits: the first `need` TABs (block offsets) and the count (main.go:535)
        uint32_t tab[9];
        uint32_t found = 0;
#pragma unroll
        for (uint32_t k = 0; k < 9; k++) tab[k] = cend;
        if (general && !(is_long && lane == 0)) {
          if (len) {
            const uint32_t cend_w = ls_w + len;
            const uint32_t we = (cend_w - 1u) >> 5;  // last word with bytes of the line
            const uint32_t last_mask = (cend_w & 31u) ? (1u << (cend_w & 31u)) - 1u : 0xFFFFFFFFu;
            auto ldw = [&](uint32_t w) -> uint32_t { return S.tabs[w] & (w == we ? last_mask : 0xFFFFFFFFu); };
            uint32_t wj = ls_w >> 5;
            uint32_t cur = ldw(wj) & (0xFFFFFFFFu << (ls_w & 31u));
#pragma unroll
            for (uint32_t k = 0; k < 9; k++) {
              if (k < need) {
                while (!cur && wj < we) cur = ldw(++wj);
                if (cur) {
                  tab[k] = win_start + wj * 32u + (uint32_t)__ffs(cur) - 1u;
                  cur &= cur - 1u;
                  found = k + 1u;
                }
              }
            }
          }
        } else if (general) {
          n_tabs = S.long_tab[9];
          found = min(n_tabs, need);
#pragma unroll
          for (uint32_t k = 0; k < 9; k++)
            if (k < found) tab[k] = S.long_tab[k];
        }

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
          AlleleCtxT<decltype(hb)> c;
          uint32_t mode = 0, n_commas = 0, bound = 0;
          if (active && status == BVCF_LINE_OK && a.n_header > 6) {
            const FilterTable *ft = a.filters;
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
          // record slots past the line's own, from max_lines up: one atomic per round, and only when a line needs them
          const uint32_t want_rec = bound > 1 ? bound - 1 : 0u;
          uint32_t wt_rec;
          uint32_t extra_base = wave_excl_scan(want_rec, &wt_rec);
          uint32_t got = 0;
          if (wt_rec) {
            if (lane == 0) got = atomicAdd(&a.counters->n_alleles, wt_rec);
            got = bcast0(got);
          }
          extra_base += extras_at + got;
          uint32_t rec_first = 0, n_rec = 0, site_type = 0;
          bool primary_written = false;
          if (eval && line < a.max_lines) {
            if (mode == 0) log_err(a, line, 0, BVCF_ERR_SAME);
            if (mode == 3) log_err(a, line, 0, BVCF_ERR_EMPTY_REF);
            const bool fits = (unsigned long long)extra_base + want_rec <= a.max_alleles;
            auto slot = [&](uint32_t j) -> uint32_t { return j == 0 ? lslot : extra_base + j - 1; };
            uint32_t cur = 0, emitted = 0;
            if (mode == 1 || mode == 2) {
#pragma nounroll
              for (uint32_t k = 0;; k++) {
                AlleleEval e;
                Span tk;
                if (mode == 1) {
                  if (k > 0) break;
                  eval_single(c, e);
                  tk = c.alt;
                } else {
                  if (!next_token(c, &cur, &tk)) break;
                  eval_token(c, tk, e);
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
                      const uint8_t rb = hb[c.ref.off + i], ab = hb[tk.off + i];
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
          if (active && line < a.max_lines) {
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
            a.lines[lslot] = L;
            if (!primary_written && lslot < a.max_alleles) a.alleles[lslot].gt_task = kNoTask;
          }
        };  // serial
        {
          BytesT<false> win;
          win.g = a.buf;
          win.lds = as_lds(S.text);
          win.lo = win_start;
          win.n = kS1Win;
          win.sub = win_start;
          win.mask = 0xFFFFFFFFu;
          serial(win, general && !(is_long && lane == 0));
        }
        if (is_long) {
          BytesT<true> head;
          head.g = a.buf;
          head.lds = as_lds(S.long_head);
          head.lo = ls;
          head.n = kS1LongHead;
          head.sub = ls;
          head.mask = 0xFFFFFFFFu;
          serial(head, general && lane == 0);
        }
      }

      // ---- next round
      prev = bcast0(S.fifo[n - 1]) + 1u;
      n_done += n;
      first_round = false;
      __builtin_amdgcn_wave_barrier();
    }
    S1STAMP(5)
    if (!base_known) look_back();  // (a tile without a line end still passes its group's prefix on)
#ifdef BVCF_EXP_TIMES
    if (lane == 0) g_wave_t[1][t & 32767u] = wall_clock64();
#endif
    if ((!kCensus || (kPacked && n_chunks == 0u)) && has_tile && t + 1u == n_tiles && lane == 0) {
      a.counters->n_lines = base + tile_eols;
      a.counters->lines_seen = base + tile_eols;
    }
  };

  if constexpr (kCensus) {
    // (tried: the next window's loads issued as soon as this one is staged -- 32 more live registers, spills at three
    // waves per SIMD, 70 -> 91 us; a one-dword-per-line LDS-DMA touch of the next window to warm the L2 -- no gain)
    for (uint32_t t = blockIdx.x * kS1Waves + wiw; t < n_tiles; t += gridDim.x * kS1Waves) {
      load_window(t);
      process_tile(t);
    }
  } else {
    const uint32_t t = blockIdx.x * kS1Waves + wiw;
    load_window(t < n_tiles ? t : 0u);
    process_tile(t);
  }
}

#ifdef BVCF_EXPERIMENTS  // the single-pass attempt, slower than k_sites2 behind its census: kept for A/B builds
__global__ __launch_bounds__(kS1Threads) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sites1(KernelArgs a, uint32_t n_tiles, uint32_t first_wgs, uint32_t stagger_ticks) {
  // (experiment: the workgroups that fill the GPU when the kernel starts do not start together -- all at once they ask
  // for 25 MB of text in one burst and every one of them then waits for the slowest load of the burst, whose count is
  // part of everybody's line numbers.  Spreading the first generation did not break the lockstep: off by default.)
  if (blockIdx.x < first_wgs && stagger_ticks) {
    const unsigned long long until = wall_clock64() + (unsigned long long)blockIdx.x * stagger_ticks / first_wgs;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
  }
  s1_body<false, false>(a, n_tiles, 0u);
}
#endif  // BVCF_EXPERIMENTS

// The census of the FULL form of the results (k_sites2; the packed form's k_sites2p takes k_census_tiles below, which
// needs no scan kernel -- the full form's kernel has no register to spare for summing the levels itself: 59.7 -> 67.2 us
// when it did): terminators per TILE, one wave per tile and seven chunk loads in flight; census[t] then goes
// through k_scan_top alone (20 k values per 142 MB: one workgroup's work), where the per-chunk census needs two scan levels
__global__ __launch_bounds__(kWgThreads) void k_count_tiles(KernelArgs a, uint32_t n_tiles) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t last_off = a.cap - 16u;
  constexpr uint32_t kN = kS2Tile / kChunk;
  for (uint32_t t = wave_in_grid(); t < n_tiles; t += stride) {
    const uint32_t base = t * kS2Tile;  // (n_tiles * kS2Tile < 2^32 + kS2Tile: blocks stay below 4 GiB)
    const uint32_t tile_end = (uint32_t)min((unsigned long long)base + kS2Tile, (unsigned long long)a.nbytes);
    u32x4 v[kN];
#pragma unroll
    for (uint32_t c = 0; c < kN; c++) v[c] = ld_stream(a.buf + min(base + c * kChunk + 16u * lane, last_off));
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t c = 0; c < kN; c++) cnt += __popc(eq_mask16(v[c], a.eol_byte) & bits_until(tile_end, base + c * kChunk + 16u * lane));
    cnt = wave_sum(cnt);
    if (lane == 0) a.census[t] = cnt;
  }
}

// ... and its scan: census[0, n) -> exclusive prefixes in place, one workgroup, 32 values per thread and step from eight
// independent 16-byte loads (k_scan_top's thread walks its share value by value -- fine for a few hundred group totals,
// 26 us for the 20 k tiles of 142 MB).  Sets the batch counters as k_scan_top does on the paths without k_stream.
__global__ __launch_bounds__(1024) void k_scan_flat(KernelArgs a, uint32_t n) {
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const int lane = lane_id();
  const uint32_t w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0u;
  __syncthreads();
  constexpr uint32_t kPer = 32, kStep = 1024u * kPer;
  u32x4 *const p4 = reinterpret_cast<u32x4 *>(a.census);  // (hipMalloc'ed: 16-byte aligned; its capacity covers the step's tail)
  for (uint32_t base = 0; base < n; base += kStep) {
    const uint32_t lo = base + threadIdx.x * kPer;
    u32x4 v[kPer / 4];
#pragma unroll
    for (uint32_t i = 0; i < kPer / 4; i++) v[i] = lo + 4u * i < n ? p4[(lo >> 2) + i] : u32x4{0u, 0u, 0u, 0u};
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < kPer / 4; i++) {
      const uint32_t at = lo + 4u * i;  // values past n (the buffer's stale tail) do not count
      if (at + 1u >= n + 1u) v[i].x = 0u;
      if (at + 1u >= n) v[i].y = 0u;
      if (at + 2u >= n) v[i].z = 0u;
      if (at + 3u >= n) v[i].w = 0u;
      sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    uint32_t wtot;
    const uint32_t pre = wave_excl_scan(sum, &wtot);
    if (lane == 0) s_wave[w] = wtot;
    __syncthreads();
    uint32_t run = s_carry + pre, total = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
      const uint32_t x = s_wave[j];
      run += j < w ? x : 0u;
      total += x;
    }
#pragma unroll
    for (uint32_t i = 0; i < kPer / 4; i++) {
      const u32x4 e = v[i];
      const u32x4 o = u32x4{run, run + e.x, run + e.x + e.y, run + e.x + e.y + e.z};
      run += e.x + e.y + e.z + e.w;
      if (lo + 4u * i < n) p4[(lo >> 2) + i] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t all = s_carry;
    a.counters->n_lines = all;
    a.counters->n_alleles = 0;
    a.counters->n_errs = 0;
    a.counters->n_tasks = 0;
    a.counters->lines_seen = all;
    a.counters->cmap_maps = 0;
    a.counters->pad[0] = a.counters->pad[1] = 0;
    a.counters->n_finish = 0;
    a.counters->n_full = 0;
    a.line_off[0] = 0u;
  }
}

// k_sites2's own census: line ends per TILE, and no scan kernel behind it (round 4; k_scan_flat took 7 of the chain's
// 77 us).  Three levels, none of them scanned: a workgroup counts one BUNDLE of 16 consecutive tiles (a wave four of
// them, the next one's seven chunk loads in flight while one is counted) and stores the 16 raw counts, the bundle's total, and adds that to
// its GROUP's total (32 bundles = 512 tiles = 3.5 MiB of text; one atomic per workgroup).  k_sites2 sums what lies in
// front of a tile itself: the raw counts of its bundle on lanes 0..15 and the bundle totals of its group on lanes
// 32..63 (one load), the totals of the groups before on all lanes (one load per 64 groups = 224 MiB), one wave sum.
// The group totals start from zero without a kernel of their own: a slot keeps two sets, and workgroup 0 clears the one
// the slot's NEXT batch will add to (nobody reads it meanwhile).  It also sets the batch counters, as k_scan_top does
// on the paths without k_stream (n_lines / lines_seen: the wave of k_sites2 that holds the last tile).

__global__ __launch_bounds__(kWgThreads) void k_census_tiles(KernelArgs a, uint32_t n_tiles, uint32_t groups_cap) {
  static_assert(kWavesPerWg * 4u == kS2BundleTiles, "a wave counts four tiles of its workgroup's bundle");
  __shared__ uint32_t s_cnt[kS2BundleTiles];
  const int lane = lane_id();
  const uint32_t wiw = wave_in_wg();
  const uint32_t last_off = a.cap - 16u;
  constexpr uint32_t kN = kS2Tile / kChunk;
  if (blockIdx.x == 0) {
    for (uint32_t j = threadIdx.x; j < groups_cap; j += kWgThreads) a.s2_groups_next[j] = 0u;
    if (threadIdx.x == 0) {
      a.counters->n_lines = 0;
      a.counters->n_alleles = 0;
      a.counters->n_errs = 0;
      a.counters->n_tasks = 0;
      a.counters->lines_seen = 0;
      a.counters->cmap_maps = 0;
      a.counters->pad[0] = a.counters->pad[1] = 0;
      a.counters->n_finish = 0;
      a.counters->n_full = 0;
      a.line_off[0] = 0u;
    }
  }
  const uint32_t t0 = blockIdx.x * kS2BundleTiles + wiw * 4u;
  // the wave's four tiles one after the other, the next one's chunks asked for before this one's are counted
  auto ask = [&](u32x4 *v, uint32_t t) {
    const uint32_t base = t < n_tiles ? t * kS2Tile : 0u;  // (n_tiles * kS2Tile < 2^32 + kS2Tile: blocks stay below 4 GiB)
#pragma unroll
    for (uint32_t c = 0; c < kN; c++) v[c] = ld_stream(a.buf + min(base + c * kChunk + 16u * lane, last_off));
  };
  u32x4 v[2][kN];
  ask(v[0], t0);
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t t = t0 + k;
    if (k < 3) ask(v[(k + 1) & 1], t + 1u);
    const uint32_t base = t * kS2Tile;
    const uint32_t tile_end = (uint32_t)min((unsigned long long)base + kS2Tile, (unsigned long long)a.nbytes);
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t c = 0; c < kN; c++) cnt += __popc(eq_mask16(v[k & 1][c], a.eol_byte) & bits_until(tile_end, base + c * kChunk + 16u * lane));
    cnt = t < n_tiles ? wave_sum(cnt) : 0u;
    if (lane == 0) s_cnt[wiw * 4u + k] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < kS2BundleTiles) {
    const uint32_t mine = s_cnt[threadIdx.x];
    a.census[blockIdx.x * kS2BundleTiles + threadIdx.x] = mine;  // (tiles past n_tiles: zeros, inside the bundle area's lead)
    uint32_t tot = 0;
#pragma unroll
    for (uint32_t j = 0; j < kS2BundleTiles; j++) tot += s_cnt[j];
    if (threadIdx.x == 0) {
      a.census[s2_bundle_off(n_tiles) + blockIdx.x] = tot;
      if (tot) atomicAdd(&a.s2_groups[blockIdx.x / kS2GroupBundles], tot);
    }
  }
}

__global__ __launch_bounds__(kS1Threads) __attribute__((amdgpu_waves_per_eu(BVCF_S2_WAVES_EU, BVCF_S2_WAVES_EU))) void k_sites2(KernelArgs a, uint32_t n_tiles, uint32_t n_chunks) {
  s1_body<true, false>(a, n_tiles, n_chunks);
}

// ... with the packed form of the results (bvcf_params.packed_sites): 32 bytes per line instead of 128
__global__ __launch_bounds__(kS1Threads) __attribute__((amdgpu_waves_per_eu(BVCF_S2_WAVES_EU, BVCF_S2_WAVES_EU))) void k_sites2p(KernelArgs a, uint32_t n_tiles, uint32_t n_chunks) {
  s1_body<true, true>(a, n_tiles, n_chunks);
}

}  // namespace bvcf_dev
