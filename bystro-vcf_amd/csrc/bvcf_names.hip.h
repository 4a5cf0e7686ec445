// bvcf_names.hip.h — the het / hom / missing sample-name lists of every output allele as text
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// processLines appends sample names to three slices per allele and joins each with fieldDelimiter
// (main.go:612-656, 1069-1190): up to 20 KB of text per row of a common variant, and the host formatter's one
// per-name loop.  With the sample names on the device the lists are rendered here from the class maps:
//   k_name_len    one wave per alleles[] slot: the byte length of its three lists (sum of name lengths + delimiters
//                 over the samples of each class)
//   k_name_scan   exclusive prefix over the slots -> arena offsets, total bytes
//   k_name_write  one wave per slot again: each lane places the names of its samples (class-map byte = 4 samples per
//                 lane and step; wave prefix sums of the lengths per class give the positions)
// The host then copies three strings per row.
#pragma once

#include "bvcf_common.hip.h"
#include "bvcf_gtscan.hip.h"

namespace bvcf_dev {

struct NameTable {
  const uint32_t *off;  // [n_samples + 1] name s = text[off[s], off[s + 1])
  const uint8_t *text;
  uint32_t delim_len;
  uint8_t delim[16];
};

struct NameArgs {
  NameTable nt;
  bvcf_names *lists;     // [max_alleles]
  uint32_t *tot;         // [max_alleles + 1] bytes of slot i's three lists -> exclusive prefix (k_name_scan)
  uint8_t *out;          // the arena
  unsigned long long cap;
  unsigned long long *total;  // bytes all lists need
};

// does alleles[] slot k hold a record whose row is printed?  (as k_dosage decides; ac > 0: main.go:558-560)
__device__ __forceinline__ bool name_slot_live(const KernelArgs &a, uint32_t k, uint32_t n_lines, const bvcf_allele &r) {
  const uint32_t li = k < n_lines ? k : r.line;
  if (li >= n_lines) return false;
  const bvcf_line L = a.lines[li];
  if (L.status != BVCF_LINE_OK || L.n_rec == 0) return false;
  if (k >= n_lines && (k < L.rec_first || k - L.rec_first + 1u >= L.n_rec)) return false;
  return r.ac != 0 && r.cmap_off != BVCF_NO_CMAP;
}

// the class-map byte this lane looks at in step `it` of allele r: its index and value (0 past the map)
__device__ __forceinline__ uint32_t name_map_byte(const KernelArgs &a, const bvcf_allele &r, uint32_t it, uint32_t *byte_idx,
                                                  uint32_t *n_steps) {
  const int lane = lane_id();
  const uint8_t *cm = a.cmap + r.cmap_off;
  if (r.flags & BVCF_ALLELE_CMAP_SPARSE) {
    *n_steps = 1;
    const uint32_t n = min(reinterpret_cast<const uint32_t *>(cm)[0], (uint32_t)BVCF_CMAP_SPARSE_MAX);
    if ((uint32_t)lane < n) {
      const uint32_t e = reinterpret_cast<const uint32_t *>(cm)[1 + lane];
      *byte_idx = e >> 8;
      return e & 0xFFu;
    }
    *byte_idx = 0;
    return 0;
  }
  const uint32_t n_bytes = (a.n_samples + 3u) / 4u;
  *n_steps = (n_bytes + kWave - 1u) / kWave;
  const uint32_t i = it * kWave + (uint32_t)lane;
  *byte_idx = i;
  return i < n_bytes ? cm[i] : 0u;
}

constexpr uint32_t kNameLdsSamples = 4095;   // the offset table of up to this many samples is copied into LDS (16 KiB)
constexpr uint32_t kNameLdsText = 40u << 10;  // ... and so is name text up to this size (k_name_write)

// name lengths / text: from LDS when the tables fit (kLds), from memory otherwise
template <bool kLds>
struct NameSrc {
  const uint32_t *off;   // LDS or global, by kLds
  const uint8_t *text;
  __device__ __forceinline__ uint32_t o(uint32_t s) const {
    if (kLds) return reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((const __attribute__((address_space(3))) uint8_t *)off)[s];
    return off[s];
  }
  __device__ __forceinline__ uint8_t t(uint32_t i) const {
    if (kLds) return ((const __attribute__((address_space(3))) uint8_t *)text)[i];
    return text[i];
  }
};

template <bool kLds>
__device__ __forceinline__ void k_name_len_body(const KernelArgs &a, const NameArgs &na, const NameSrc<kLds> &ns_) {
  const int lane = lane_id();
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_alleles = min(n_lines + a.counters->n_alleles, a.max_alleles);
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t dl = na.nt.delim_len;
  for (uint32_t k = wave_in_grid(); k < n_alleles; k += stride) {
    const bvcf_allele r = a.alleles[k];
    uint32_t len[3] = {0, 0, 0};
    if (name_slot_live(a, k, n_lines, r)) {
      uint32_t n_steps = 1;
      for (uint32_t it = 0; it < n_steps; it++) {
        uint32_t bi;
        const uint32_t b = name_map_byte(a, r, it, &bi, &n_steps);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          const uint32_t cls = (b >> (2u * q)) & 3u;
          const uint32_t s = bi * 4u + q;
          if (cls && s < a.n_samples) len[cls - 1u] += ns_.o(s + 1u) - ns_.o(s) + dl;
        }
      }
#pragma unroll
      for (uint32_t c = 0; c < 3; c++) {
        len[c] = wave_sum(len[c]);
        if (len[c]) len[c] -= dl;  // no delimiter after the last name
      }
    }
    if (lane == 0) {
      bvcf_names nl;
      nl.off[0] = nl.off[1] = nl.off[2] = 0;
      nl.len[0] = len[0];
      nl.len[1] = len[1];
      nl.len[2] = len[2];
      na.lists[k] = nl;
      na.tot[k] = len[0] + len[1] + len[2];
    }
  }
}

__global__ __launch_bounds__(kWgThreads) void k_name_len(KernelArgs a, NameArgs na) {
  __shared__ uint32_t s_off[kNameLdsSamples + 1];
  if (a.n_samples <= kNameLdsSamples) {
    for (uint32_t i = threadIdx.x; i <= a.n_samples; i += kWgThreads) s_off[i] = na.nt.off[i];
    __syncthreads();
    NameSrc<true> src{s_off, nullptr};
    k_name_len_body<true>(a, na, src);
  } else {
    NameSrc<false> src{na.nt.off, na.nt.text};
    k_name_len_body<false>(a, na, src);
  }
}

// exclusive prefix of tot[0 .. n) in place (one workgroup), total -> *na.total
__global__ __launch_bounds__(1024) void k_name_scan(KernelArgs a, NameArgs na) {
  __shared__ unsigned long long s_part[1024];
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n = min(n_lines + a.counters->n_alleles, a.max_alleles);
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per;
  unsigned long long sum = 0;
  for (uint32_t i = 0; i < per; i++)
    if (lo + i < n) sum += na.tot[lo + i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long t = threadIdx.x >= (unsigned)d ? s_part[threadIdx.x - d] : 0ull;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = s_part[threadIdx.x] - sum;
  for (uint32_t i = 0; i < per; i++) {
    if (lo + i < n) {
      const uint32_t v = na.tot[lo + i];
      // (offsets are 32-bit in bvcf_names: a batch whose lists pass 4 GiB reports the total and writes nothing)
      na.tot[lo + i] = (uint32_t)run;
      run += v;
    }
  }
  if (threadIdx.x == 1023) *na.total = s_part[1023];
}

template <bool kLds>
__device__ __forceinline__ void k_name_write_body(const KernelArgs &a, const NameArgs &na, const NameSrc<kLds> &ns_) {
  const int lane = lane_id();
  const uint32_t n_lines = min(a.counters->n_lines, a.max_lines);
  const uint32_t n_alleles = min(n_lines + a.counters->n_alleles, a.max_alleles);
  const uint32_t stride = gridDim.x * kWavesPerWg;
  const uint32_t dl = na.nt.delim_len;
  for (uint32_t k = wave_in_grid(); k < n_alleles; k += stride) {
    const bvcf_names nl = na.lists[k];
    if (nl.len[0] + nl.len[1] + nl.len[2] == 0) continue;
    const bvcf_allele r = a.alleles[k];
    const uint32_t base = na.tot[k];
    uint32_t at[3] = {base, base + nl.len[0], base + nl.len[0] + nl.len[1]};  // where the next name of each list goes
    if (lane == 0) {
      bvcf_names w = nl;
      w.off[0] = at[0];
      w.off[1] = at[1];
      w.off[2] = at[2];
      na.lists[k] = w;
    }
    const uint32_t end[3] = {at[0] + nl.len[0], at[1] + nl.len[1], at[2] + nl.len[2]};
    uint32_t n_steps = 1;
    for (uint32_t it = 0; it < n_steps; it++) {
      uint32_t bi;
      const uint32_t b = name_map_byte(a, r, it, &bi, &n_steps);
      // bytes this lane adds to each list, then the wave prefix: where its first name of each class starts
      uint32_t mine[3] = {0, 0, 0};
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t cls = (b >> (2u * q)) & 3u;
        const uint32_t s = bi * 4u + q;
        if (cls && s < a.n_samples) mine[cls - 1u] += ns_.o(s + 1u) - ns_.o(s) + dl;
      }
      uint32_t pos[3];
#pragma unroll
      for (uint32_t c = 0; c < 3; c++) {
        uint32_t tot;
        pos[c] = at[c] + wave_excl_scan(mine[c], &tot);
        at[c] += tot;
      }
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t cls = (b >> (2u * q)) & 3u;
        const uint32_t s = bi * 4u + q;
        if (cls && s < a.n_samples) {
          const uint32_t c = cls - 1u;
          const uint32_t n0 = ns_.o(s), nlen = ns_.o(s + 1u) - n0;
          uint32_t p = c == 0 ? pos[0] : (c == 1 ? pos[1] : pos[2]);
          const uint32_t e = c == 0 ? end[0] : (c == 1 ? end[1] : end[2]);
          for (uint32_t j = 0; j < nlen; j++) na.out[p + j] = ns_.t(n0 + j);
          p += nlen;
          for (uint32_t j = 0; j < dl && p + j < e; j++) na.out[p + j] = na.nt.delim[j];  // (not after the list's last name)
          p += dl;
          if (c == 0) pos[0] = p; else if (c == 1) pos[1] = p; else pos[2] = p;
        }
      }
    }
  }
}

__global__ __launch_bounds__(kWgThreads) void k_name_write(KernelArgs a, NameArgs na) {
  __shared__ uint32_t s_off[kNameLdsSamples + 1];
  __shared__ __attribute__((aligned(16))) uint8_t s_text[kNameLdsText];
  const unsigned long long total = *na.total;
  if (total > na.cap || total >= 0xFFFFFFF0ull) return;  // the host grows the arena and launches this kernel again
  const uint32_t text_bytes = a.n_samples ? na.nt.off[a.n_samples] : 0u;
  if (a.n_samples <= kNameLdsSamples && text_bytes <= kNameLdsText) {
    for (uint32_t i = threadIdx.x; i <= a.n_samples; i += kWgThreads) s_off[i] = na.nt.off[i];
    for (uint32_t i = threadIdx.x * 4u; i < text_bytes; i += kWgThreads * 4u)
      *reinterpret_cast<uint32_t *>(&s_text[i]) = *reinterpret_cast<const uint32_t *>(na.nt.text + i);  // (the table is padded)
    __syncthreads();
    NameSrc<true> src{s_off, s_text};
    k_name_write_body<true>(a, na, src);
  } else {
    NameSrc<false> src{na.nt.off, na.nt.text};
    k_name_write_body<false>(a, na, src);
  }
}

}  // namespace bvcf_dev
