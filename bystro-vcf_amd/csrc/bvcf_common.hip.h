// bvcf_common.hip.h — shared constants, batch structures, wave/byte helpers
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvcf.h"

namespace bvcf_dev {


constexpr int kWave = 64;
constexpr int kWavesPerWg = 4;
constexpr int kWgThreads = kWave * kWavesPerWg;
constexpr uint32_t kChunk = 1024;      // bytes per wave-iteration (16 B x 64 lanes)
constexpr uint32_t kMaxProducerWaves = 32768;  // upper bound of KernelArgs.prod_waves (8 workgroups of 4 waves on 1024 CUs)
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
  uint32_t pad[2];       // [0]: internal error flag; [1]: wide ctxs, the longest sample region
  uint32_t n_finish;     // streaming path: entries of finish_items (what k_finish still has to settle)
  uint32_t n_full;       // sites-only input, packed form (KernelArgs.sites): lines whose full records were written (lines[0 .. n_full))
  uint32_t n_real;       // streaming path: entries of real_tasks (the task slots that hold a scan, for k_gt)
  uint32_t n_other_shape;  // streaming path: listed lines that were not of the shape the kernel is made for -- k_stream: not the
                           // 4-byte grid (left to k_gt); k_stream_gen: of the 4-byte grid.  The host picks the next batch's kernel by it.
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
  uint32_t regular;      // 1: every sample field was "x<sep>y" with single-digit alleles (the fast scan), so the dosage
                         // of a sample is its class (none 0, het 1, hom 2, missing -1)
  uint32_t pad;
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
  int8_t *dosage;          // want_dosage: one row of dosage_stride bytes per alleles[] slot, or null
  uint32_t dosage_stride;
  uint32_t cmap_stride;
  uint32_t max_lines, max_alleles, max_errs, max_tasks;
  unsigned long long max_cmap;
  const FilterTable *filters;
  uint32_t *census;      // [n_chunks] newline count per chunk -> exclusive prefix within group
  uint32_t *group_base;  // [n_groups]
  uint32_t *run_lines;   // [prod_waves] streaming path: lines listed by each wave of k_stream / k_stream_gen (its run of tiles)
  uint32_t prod_waves;   // waves of the one-pass kernel's grid
  uint32_t *s2_groups;       // k_census_tiles / k_sites2: line ends per group of kS2GroupTiles tiles, this batch's half ...
  uint32_t *s2_groups_next;  // ... and the half of the slot's next batch, zeroed meanwhile
  uint32_t *line_off;    // [max_lines + 1]
  bvcf_line *lines;
  bvcf_allele *alleles;
  bvcf_site *sites;      // sites-only input, bvcf_params.packed_sites: one 32-byte record per line; lines[] / alleles[] then
                         // only for the lines that need them, in slots handed out by counters->n_full.  Null: the full form
  bvcf_err *errs;
  uint8_t *cmap;
  GtTask *tasks;
  GtResult *results;
  BatchCounters *counters;
  // streaming path
  uint32_t fused;        // 1: k_stream found the lines and scanned ALT #1
  uint32_t gen_stream;   // 1: k_stream also scans lines that are not the 4-byte grid (bvcf_streamgen.hip.h); 0: leaves them to k_gt
  uint32_t wide;         // 1: census path, regular scans split into windows over several waves (k_gt_wide)
  uint32_t win_bytes;    // wide: bytes of a line's sample region per wave of the split general scan
  uint32_t win_tabs_cap; // entries of win_tabs
  uint32_t *win_tabs;    // wide: TABs per (line, window), see k_tabs_wide
  uint32_t tile_bytes;   // bytes of text a wave owns (lines belong to the tile they start in)
  uint32_t tile_quota;   // entries reserved per tile: a line that passes the field count is at
                         // least n_header - 1 + eol_chars bytes long
  uint32_t n_tiles;
  StreamEntry *entries;  // [n_tiles * tile_quota]
  uint32_t *line_len;    // [max_lines]; bit 31: line_bits holds the line's head TAB bitmap
  uint32_t *line_cmap;   // [max_lines] class map of ALT #1
  uint16_t *head_bits;   // [n_tiles * tile_quota][16] TAB mask per 16 bytes of a line's 256-byte head window (k_stream)
  uint32_t *line_bits;   // [max_lines][8] the same, in input order (k_order)
  uint32_t *finish_items;// [max_lines] streaming path: the lines k_finish settles (verdict + record counts)
  uint32_t *real_tasks;  // [max_tasks] streaming path: the task slots past n_lines that hold a scan (or kNoTask), see k_head
  // k_sites1: the FILTER gate of the common lines as dwords (see bvcf_sites1.hip.h; 0 = no such table, 1 = keys, 2 = no test)
  uint32_t s1_fmode;
  uint32_t s1_fkey[4], s1_flen[4];
};
constexpr uint32_t kHasHeadBits = 0x80000000u;
// StreamEntry.len only (k_order takes it out): the line's counts are right but its class map does not tell the dosage
// (haploid calls, fields of other ploidy) -- GtResult.regular = 0, k_dosage scans the line itself.  (A line is far
// shorter than 1 GiB.)
constexpr uint32_t kNotRegular = 0x40000000u;

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Cross-lane sums stay in the VALU: DPP row shifts inside each row of 16 lanes, then the two row
// broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  __shfl_* goes through
// the LDS crossbar (ds_bpermute + an lgkmcnt wait per step), which is what the per-line tail of
// k_stream used to spend most of its time on.
// inclusive prefix sum over the 64 lanes (lane 63 ends up with the wave total)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// running maximum over the 64 lanes (values >= 0; the same DPP steps as the sum)
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
  return v;
}

// value of lane `src` (wave-uniform index) in every lane's scalar view: v_readlane, no LDS
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return lane_value(wave_incl_scan(v), kWave - 1); }

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
  const uint32_t inc = wave_incl_scan(v);
  *total = lane_value(inc, kWave - 1);
  return inc - v;
}

__device__ __forceinline__ uint32_t bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The wave's index inside its workgroup / in the grid, as values the compiler knows to be wave-uniform: threadIdx.x >> 6
// is the same in all 64 lanes, but only a readfirstlane tells the compiler so -- without it everything derived from
// the index (task numbers, tile bounds, the task descriptor loaded with it) lives in vector registers and every test
// on it is compiled as a divergent branch.
__device__ __forceinline__ uint32_t wave_in_wg() { return bcast0(threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t wave_in_grid() { return blockIdx.x * kWavesPerWg + wave_in_wg(); }

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

// A 16 B-per-lane load whose address is not a multiple of 4 runs at 3/4 of the bandwidth of one
// that is (tools/membench.hip: 5.3 vs 6.9 TB/s on 1.3 GB; 4-, 8- and 16-byte alignment are all
// equal).  Text fields start anywhere, so the streaming loads start at the dword at or before the
// wanted byte (p & ~3) and the shift r = p & 3 is undone here: the lane's window becomes its own four
// dwords plus the first dword of the next lane -- of the next chunk's lane 0 (`next0`, wave-uniform)
// for lane 63.
__device__ __forceinline__ u32x4 realign(u32x4 v, uint32_t next0, uint32_t r) {
  // wave_shl:1 -- lane i reads lane i + 1; lane 63 has no source and keeps `old` = next0
  const uint32_t w4 = (uint32_t)__builtin_amdgcn_update_dpp((int)next0, (int)v.x, 0x130, 0xF, 0xF, false);
  u32x4 o;
  o.x = __builtin_amdgcn_alignbyte(v.y, v.x, r);
  o.y = __builtin_amdgcn_alignbyte(v.z, v.y, r);
  o.z = __builtin_amdgcn_alignbyte(v.w, v.z, r);
  o.w = __builtin_amdgcn_alignbyte(w4, v.w, r);
  return o;
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
// (zero_bytes() flags an equal byte with 0x80: a byte dot product with weights 1, 2, 4, 8 -- 16..128 for the odd
// dwords -- gathers eight flags into one number, scaled by 128.  13 instructions fewer than four multiply-gathers.)
__device__ __forceinline__ uint32_t eq_mask16(u32x4 v, uint32_t c) {
  const uint32_t c4 = c * 0x01010101u;
  uint32_t lo = __builtin_amdgcn_udot4(zero_bytes(v.x ^ c4), 0x08040201u, 0u, false);
  lo = __builtin_amdgcn_udot4(zero_bytes(v.y ^ c4), 0x80402010u, lo, false);
  uint32_t hi = __builtin_amdgcn_udot4(zero_bytes(v.z ^ c4), 0x08040201u, 0u, false);
  hi = __builtin_amdgcn_udot4(zero_bytes(v.w ^ c4), 0x80402010u, hi, false);
  return (lo >> 7) | (hi << 1);
}

// bits [0, end - off) of a 16-bit mask for two byte offsets in any order, over the whole 32-bit range (blocks reach
// past 2 GiB: a signed difference would wrap)
__device__ __forceinline__ uint32_t bits_until(uint32_t end, uint32_t off) {
  const uint32_t n = min(end - min(end, off), 16u);
  return (1u << n) - 1u;
}


// class map of map slot `mi` (census path: slot == task index), or BVCF_NO_CMAP past the arena
__device__ __forceinline__ uint32_t cmap_of(const KernelArgs &a, uint32_t mi, bool want) {
  return (want && ((unsigned long long)mi + 1ull) * a.cmap_stride <= a.max_cmap) ? mi * a.cmap_stride : BVCF_NO_CMAP;
}

}  // namespace bvcf_dev
