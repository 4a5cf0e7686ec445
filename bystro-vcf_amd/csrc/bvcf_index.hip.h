// bvcf_index.hip.h — newline census, scans, line offsets (the census path's line index)
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
#pragma once

#include "bvcf_common.hip.h"

namespace bvcf_dev {

// ------------------------------------------------------------------ line index

// newline census: a wave takes 4 consecutive 1 KiB chunks per step so that 4 KiB are in flight
__global__ __launch_bounds__(kWgThreads) void k_count_eol(KernelArgs a, uint32_t n_chunks) {
  const int lane = lane_id();
  const uint32_t wave = wave_in_grid();
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
      cnt[q] = __popc(eq_mask16(v[q], a.eol_byte) & bits_until(a.nbytes, off));
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
      a.counters->n_finish = 0;
      a.counters->n_full = 0;
      a.line_off[0] = 0u;
    }
  }
}

// line_off[i + 1] = offset just past line i's terminator.  A wave looks at 64 census entries at
// once (one per lane) and revisits only the chunks that hold a terminator.
__global__ __launch_bounds__(kWgThreads) void k_scatter_eol(KernelArgs a, uint32_t n_chunks) {
  const int lane = lane_id();
  const uint32_t wave = wave_in_grid();
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
      uint32_t idx = lane_value(mine, src);
      const uint32_t off = cc * kChunk + 16u * lane;
      u32x4 v = load16(a.buf, off, a.cap);
      uint32_t m = eq_mask16(v, a.eol_byte) & bits_until(a.nbytes, off);
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


}  // namespace bvcf_dev
