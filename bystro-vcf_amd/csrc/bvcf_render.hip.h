// bvcf_render.hip.h — sites-only input, bvcf_params.render_sites: the TSV rows of the lines the packed form settles,
// written on the device (main.go:586-695 for the one shape of line the fast lanes of k_sites2p accept, main.go:735-745:
// a biallelic SNP of a file without samples)
// Part of the gfx950 device code of libbvcf; see bvcf_device.hip.h for the kernel map.
//
// Such a row is a pure function of the line's 32-byte site record and a few bytes of its text: ["chr"] CHROM <TAB> POS
// "\tSNP\t" REF <TAB> ALT <TAB> trTv, a tail that is the same for every such line (three empty lists, zero counts:
// "\t!\t0\t!\t0\t!\t0\t0\t0\t0" with the default --emptyField), the optional vcfPos / id / alleleIdx + info columns, "\n".
// A sites-only run through the CLI is bound by the host threads that put these rows together (20 M rows: 0.12 s of eight
// formatter threads against 0.03 s of device time per 20 M), so they are made here and the host only writes them out:
//   k_render_len   per group of 256 lines: the bytes of their rows, and how many of them the host still has to format
//                  (BVCF_SITE_FULL: indels, several ALTs, messages to log, TABs past byte 64)
//   k_render_scan  exclusive prefix over the groups (one workgroup), the batch's totals
//   k_render_rows  per group again: every lane writes its line's row at its place in the stream -- rows in input order,
//                  nothing between them -- and the lines of the host get a cut: (line, slot of its records, byte offset
//                  at which its rows belong)
// The lengths are computed twice rather than stored (4 bytes per line would be a fifth of what the site records weigh).
#pragma once

#include "bvcf_common.hip.h"

namespace bvcf_dev {

constexpr uint32_t kRenderGroup = 256;           // lines per workgroup step (== kWgThreads)
constexpr uint32_t kRenderFmtChr = 0, kRenderFmtSnp = 3, kRenderFmtTail = 8;  // offsets in RenderArgs.fmt

struct RenderArgs {
  const bvcf_site *sites;
  const uint8_t *text;
  uint8_t *rows;
  unsigned long long rows_cap;
  bvcf_row_cut *cuts;
  uint32_t cuts_cap;
  uint32_t n_groups_cap;
  uint32_t max_lines;               // site records there are room for (a batch with more lines is refused by bvcf_collect)
  unsigned long long *group_bytes;  // [n_groups_cap + 1]: bytes of a group's rows -> exclusive prefix (k_render_scan)
  uint32_t *group_full;             // [n_groups_cap + 1]: its lines for the host -> exclusive prefix
  unsigned long long *totals;       // [0] row bytes [1] lines for the host [2] rows rendered [3] bytes of the host's lines (cut_text)
  // bvcf_submit_bgzf batches (the text is on the device only): the bytes of the lines left to the host, packed, so that
  // the batch's whole text need not cross back; null for batches the caller submitted as text
  const bvcf_line *lines;
  uint8_t *cut_text;
  unsigned long long cut_text_cap;
  unsigned long long *group_ctext;  // [n_groups_cap + 1]
  const BatchCounters *counters;
  const uint8_t *fmt;               // "chr" | "\tSNP\t" | the constant tail
  uint32_t tail_len;
  uint32_t keep_pos, keep_id, keep_info;
};

struct SiteRow {
  uint32_t len;   // bytes of the row with its "\n"; 0: no row here (the line did not pass, or the host makes its rows)
  uint32_t full;  // 1: a line for the host
  uint32_t ctext; // ... and the bytes of its text that go back with it (RenderArgs.cut_text)
  uint32_t chr, f0, f1, f2, f6, n_pos, n_id, n_info;
};

// what line `li`'s site record says about its row
__device__ __forceinline__ SiteRow site_row(const RenderArgs &ra, const bvcf_site &s) {
  SiteRow r = {};
  if (s.status & BVCF_SITE_FULL) {
    r.full = 1;
    if (ra.cut_text && s.full_idx < ra.max_lines) r.ctext = ra.lines[s.full_idx].len;  // (a batch that overflowed its records is refused anyway)
    return r;
  }
  if (s.status != BVCF_LINE_OK) return r;
  auto fe = [&](int i) -> uint32_t { return s.fend[i] != 0xFFu ? (uint32_t)s.fend[i] : s.len; };
  r.f0 = s.fend[0];
  r.f1 = fe(1);
  r.f2 = fe(2);
  r.f6 = fe(6);
  const uint32_t f7 = fe(7);
  r.n_pos = r.f1 - r.f0 - 1u;
  r.n_id = ra.keep_id ? r.f2 - r.f1 - 1u : 0u;
  r.n_info = ra.keep_info ? f7 - r.f6 - 1u : 0u;
  r.chr = (r.f0 < 4u || ra.text[s.off] != 'c') ? 3u : 0u;  // main.go:570-574
  // (a block ends below 4 GiB - 1 MiB and a row is its line's bytes plus a hundred: 32 bits hold it)
  r.len = r.chr + r.f0 + 1u + r.n_pos + 10u + ra.tail_len + (ra.keep_pos ? 1u + r.n_pos : 0u) + (ra.keep_id ? 1u + r.n_id : 0u) +
          (ra.keep_info ? 3u + r.n_info : 0u) + 1u;
  return r;
}

// row bytes of the lanes before this one in the wave, and of the whole wave: 64-bit sums from two 32-bit scans (a row of
// --keepInfo can be megabytes long)
__device__ __forceinline__ unsigned long long wave_excl_scan_len(uint32_t len, unsigned long long *total) {
  uint32_t t_lo, t_hi;
  const uint32_t e_lo = wave_excl_scan(len & 0xFFFFFu, &t_lo), e_hi = wave_excl_scan(len >> 20, &t_hi);
  *total = (unsigned long long)t_lo + ((unsigned long long)t_hi << 20);
  return (unsigned long long)e_lo + ((unsigned long long)e_hi << 20);
}

__global__ __launch_bounds__(kWgThreads) void k_render_len(RenderArgs ra) {
  __shared__ unsigned long long s_len[kWavesPerWg], s_ct[kWavesPerWg];
  __shared__ uint32_t s_full[kWavesPerWg];
  const uint32_t n_lines = min(ra.counters->n_lines, min(ra.max_lines, ra.n_groups_cap * kRenderGroup));
  const uint32_t n_groups = (n_lines + kRenderGroup - 1u) / kRenderGroup;
  uint32_t n_ok = 0;
  for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t li = g * kRenderGroup + threadIdx.x;
    SiteRow r = {};
    if (li < n_lines) r = site_row(ra, ra.sites[li]);
    unsigned long long w_len, w_ct = 0;
    (void)wave_excl_scan_len(r.len, &w_len);
    const uint32_t w_full = wave_sum(r.full);
    if (ra.cut_text && w_full) (void)wave_excl_scan_len(r.ctext, &w_ct);
    __syncthreads();
    if (lane_id() == 0) {
      s_len[wave_in_wg()] = w_len;
      s_full[wave_in_wg()] = w_full;
      s_ct[wave_in_wg()] = w_ct;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long b = 0, ct = 0;
      uint32_t f = 0;
#pragma unroll
      for (int w = 0; w < kWavesPerWg; w++) {
        b += s_len[w];
        f += s_full[w];
        ct += s_ct[w];
      }
      ra.group_bytes[g] = b;
      ra.group_full[g] = f;
      if (ra.cut_text) ra.group_ctext[g] = ct;
    }
    n_ok += r.len ? 1u : 0u;
  }
  const uint32_t ok = wave_sum(n_ok);
  if (lane_id() == 0 && ok) atomicAdd(&ra.totals[2], (unsigned long long)ok);
}

// one workgroup of 1024: exclusive prefixes of both group arrays, in place; the totals
__global__ __launch_bounds__(1024) void k_render_scan(RenderArgs ra) {
  __shared__ unsigned long long s_b[1024], s_c[1024];
  __shared__ uint32_t s_f[1024];
  const bool ct = ra.cut_text != nullptr;
  const uint32_t n_lines = min(ra.counters->n_lines, min(ra.max_lines, ra.n_groups_cap * kRenderGroup));
  const uint32_t n_groups = (n_lines + kRenderGroup - 1u) / kRenderGroup;
  const uint32_t per = (n_groups + 1023u) / 1024u;
  const uint32_t lo = min(threadIdx.x * per, n_groups), hi = min(lo + per, n_groups);
  unsigned long long b = 0, cc = 0;
  uint32_t f = 0;
  for (uint32_t g = lo; g < hi; g++) {
    b += ra.group_bytes[g];
    f += ra.group_full[g];
    if (ct) cc += ra.group_ctext[g];
  }
  s_b[threadIdx.x] = b;
  s_f[threadIdx.x] = f;
  s_c[threadIdx.x] = cc;
  __syncthreads();
  // (1 024 partial sums: a plain doubling scan in LDS)
  for (uint32_t d = 1; d < 1024u; d <<= 1) {
    unsigned long long ab = 0, ac = 0;
    uint32_t af = 0;
    if (threadIdx.x >= d) {
      ab = s_b[threadIdx.x - d];
      af = s_f[threadIdx.x - d];
      ac = s_c[threadIdx.x - d];
    }
    __syncthreads();
    s_b[threadIdx.x] += ab;
    s_f[threadIdx.x] += af;
    s_c[threadIdx.x] += ac;
    __syncthreads();
  }
  unsigned long long run_b = s_b[threadIdx.x] - b, run_c = s_c[threadIdx.x] - cc;
  uint32_t run_f = s_f[threadIdx.x] - f;
  for (uint32_t g = lo; g < hi; g++) {
    const unsigned long long gb = ra.group_bytes[g];
    const uint32_t gf = ra.group_full[g];
    ra.group_bytes[g] = run_b;
    ra.group_full[g] = run_f;
    run_b += gb;
    run_f += gf;
    if (ct) {
      const unsigned long long gc = ra.group_ctext[g];
      ra.group_ctext[g] = run_c;
      run_c += gc;
    }
  }
  if (threadIdx.x == 1023u) {
    ra.totals[0] = s_b[1023];
    ra.totals[1] = s_f[1023];
    ra.totals[3] = s_c[1023];
  }
}

__device__ __forceinline__ void emit_bytes(uint8_t *&dst, const uint8_t *src, uint32_t n) {
#pragma nounroll
  for (uint32_t k = 0; k < n; k++) dst[k] = src[k];
  dst += n;
}

__global__ __launch_bounds__(kWgThreads) void k_render_rows(RenderArgs ra) {
  __shared__ unsigned long long s_len[kWavesPerWg], s_ct[kWavesPerWg];
  __shared__ uint32_t s_full[kWavesPerWg];
  // (the text of the host's lines goes back packed when it fits its buffer, and below 4 GiB; otherwise the whole text does)
  const bool ct = ra.cut_text != nullptr && ra.totals[3] <= ra.cut_text_cap && ra.totals[3] < 0xFFFFFFFFull;
  // the stream is sized for a typical file: a batch whose rows outgrow it writes nothing, the host grows it and launches
  // this kernel again (bvcf_collect)
  if (ra.totals[0] > ra.rows_cap || ra.totals[1] > ra.cuts_cap) return;
  const uint32_t n_lines = min(ra.counters->n_lines, min(ra.max_lines, ra.n_groups_cap * kRenderGroup));
  const uint32_t n_groups = (n_lines + kRenderGroup - 1u) / kRenderGroup;
  for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t li = g * kRenderGroup + threadIdx.x;
    bvcf_site s = {};
    SiteRow r = {};
    if (li < n_lines) {
      s = ra.sites[li];
      r = site_row(ra, s);
    }
    // my place: the group's, plus what the lanes before me in the workgroup take
    unsigned long long wtot_len;
    uint32_t wtot_full;
    const unsigned long long ex_len = wave_excl_scan_len(r.len, &wtot_len);
    const uint32_t ex_full = wave_excl_scan(r.full, &wtot_full);
    unsigned long long wtot_ct = 0, ex_ct = 0;
    if (ct && wtot_full) ex_ct = wave_excl_scan_len(r.ctext, &wtot_ct);
    __syncthreads();
    if (lane_id() == 0) {
      s_len[wave_in_wg()] = wtot_len;
      s_full[wave_in_wg()] = wtot_full;
      s_ct[wave_in_wg()] = wtot_ct;
    }
    __syncthreads();
    unsigned long long before_len = 0, before_ct = 0;
    uint32_t before_full = 0;
    for (uint32_t w = 0; w < wave_in_wg(); w++) {
      before_len += s_len[w];
      before_full += s_full[w];
      before_ct += s_ct[w];
    }
    const unsigned long long off = ra.group_bytes[g] + before_len + ex_len;
    if (r.full) {
      const uint32_t ci = ra.group_full[g] + before_full + ex_full;
      bvcf_row_cut c;
      c.line = li;
      c.slot = s.full_idx;
      c.off = off;
      c.text_off = BVCF_NO_TEXT_OFF;
      c.reserved = 0;
      if (ct) {
        const unsigned long long to = ra.group_ctext[g] + before_ct + ex_ct;
        if (to + r.ctext <= ra.cut_text_cap && s.full_idx < ra.max_lines) {
          c.text_off = (uint32_t)to;
          const uint8_t *src = ra.text + ra.lines[s.full_idx].off;
          uint8_t *p = ra.cut_text + to;
          emit_bytes(p, src, r.ctext);
        }
      }
      if (ci < ra.cuts_cap) ra.cuts[ci] = c;
    } else if (r.len && off + r.len <= ra.rows_cap) {  // (the bound holds by the check above; never write past the stream)
      const uint8_t *row = ra.text + s.off;
      uint8_t *p = ra.rows + off;
      emit_bytes(p, ra.fmt + kRenderFmtChr, r.chr);
      emit_bytes(p, row, r.f0 + 1u + r.n_pos);  // CHROM, the TAB, POS verbatim
      emit_bytes(p, ra.fmt + kRenderFmtSnp, 5u);
      p[0] = s.ref;
      p[1] = '\t';
      p[2] = s.alt_base;
      p[3] = '\t';
      p[4] = (uint8_t)('0' + s.trtv);  // main.go:602-606
      p += 5;
      emit_bytes(p, ra.fmt + kRenderFmtTail, ra.tail_len);
      if (ra.keep_pos) {  // main.go:674-692
        *p++ = '\t';
        emit_bytes(p, row + r.f0 + 1u, r.n_pos);
      }
      if (ra.keep_id) {
        *p++ = '\t';
        emit_bytes(p, row + r.f1 + 1u, r.n_id);
      }
      if (ra.keep_info) {
        p[0] = '\t';  // (alleleIdx of a biallelic line's one allele, main.go:687)
        p[1] = '0';
        p[2] = '\t';
        p += 3;
        emit_bytes(p, row + r.f6 + 1u, r.n_info);
      }
      *p = '\n';
    }
  }
}

}  // namespace bvcf_dev
