// membench — which read patterns of a 1.3 GB text buffer does the MI355X memory system like?
// (experiment harness, not part of the product or the tests)
//   P0 grid-stride: wave w reads burst (it * W + w)                 [k_count_eol-like]
//   P1 private runs: wave w streams its own contiguous 1/W of the buffer in bursts
//   P2 workgroup runs: a workgroup streams its own contiguous region, its waves taking
//      consecutive bursts round-robin
// burst = K x 1 KiB loads issued back to back (one dwordx4 per lane per load), then drained.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

template <int K, int PAT, bool NT>
__global__ __launch_bounds__(256) void k_read(const uint8_t *buf, size_t nbytes, uint32_t skew, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wpw = blockDim.x >> 6;
  const uint32_t wave = blockIdx.x * wpw + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * wpw;
  const size_t burst = (size_t)K * 1024;
  const size_t n_bursts = (nbytes - 64) / burst;
  uint32_t acc = 0;
  size_t b0, b1, step;
  if (PAT == 0) {
    b0 = wave; b1 = n_bursts; step = n_waves;
  } else if (PAT == 1) {
    const size_t per = (n_bursts + n_waves - 1) / n_waves;
    b0 = wave * per; b1 = b0 + per < n_bursts ? b0 + per : n_bursts; step = 1;
  } else {
    const size_t per = (n_bursts + gridDim.x - 1) / gridDim.x;
    b0 = blockIdx.x * per + (threadIdx.x >> 6);
    b1 = (blockIdx.x + 1) * per < n_bursts ? (blockIdx.x + 1) * per : n_bursts; step = wpw;
  }
  for (size_t b = b0; b < b1; b += step) {
    const uint8_t *p = buf + b * burst + skew + 16u * lane;
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const u32x4_u *q = (const u32x4_u *)(p + k * 1024);
      v[k] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// P3: line-structured trickle, the shape of k_stream's pipeline: a wave walks its private run line by
// line (fixed line length), keeps 10 chunk loads in flight and re-issues each register for the next
// line right after consuming it; HV: also one 256 B "head" load per line; ALU: dependent VALU ops
// per chunk standing in for the scan.
template <int ALU, bool HV, bool UNR>
__global__ __launch_bounds__(256) void k_lines(const uint8_t *buf, size_t nbytes, uint32_t line_len, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * 4;
  const size_t n_lines = (nbytes - 16384) / line_len;
  const size_t per = (n_lines + n_waves - 1) / n_waves;
  const size_t l0 = wave * per, l1 = l0 + per < n_lines ? l0 + per : n_lines;
  if (l0 >= l1) return;
  uint32_t acc = 0;
  u32x4 va[10];
  u32x4 hv = {0, 0, 0, 0};
  auto ld = [&](size_t off) -> u32x4 { return __builtin_nontemporal_load((const u32x4_u *)(buf + (off & ~(size_t)3) + 16u * lane)); };
  size_t p = l0 * line_len;
#pragma unroll
  for (int g = 0; g < 10; g++) va[g] = ld(p + 50 + g * 1024);
  for (size_t l = l0; l < l1; l++) {
    const size_t pn = p + line_len;
    if (HV) {
      acc += hv.x;
      if (lane < 16) hv = ld(pn);
    }
#pragma unroll
    for (int g = 0; g < 10; g++) {
      uint32_t x = va[g].x ^ va[g].y ^ va[g].z ^ va[g].w;
      if (UNR) {
#pragma unroll
        for (int i = 0; i < ALU; i++) x = x * 0x9E3779B1u + (x >> 15);
      } else {
#pragma unroll 1
        for (int i = 0; i < ALU; i += 4) {
          x = x * 0x9E3779B1u + (x >> 15);
          x = x * 0x9E3779B1u + (x >> 15);
          x = x * 0x9E3779B1u + (x >> 15);
          x = x * 0x9E3779B1u + (x >> 15);
        }
      }
      acc += x;
      va[g] = ld(pn + 50 + g * 1024);
    }
    p = pn;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int ALU, bool HV, bool UNR>
static float run_lines(const uint8_t *const *d, int nbuf, size_t n, uint32_t *sink, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_lines<ALU, HV, UNR>), dim3(wgs), dim3(256), 0, 0, d[0], n, 10164u, sink);
  hipEventRecord(e0);
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_lines<ALU, HV, UNR>), dim3(wgs), dim3(256), 0, 0, d[i % nbuf], n, 10164u, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

template <int K, int PAT, bool NT>
static float run(const uint8_t *d, size_t n, uint32_t skew, uint32_t *sink, int wgs, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_read<K, PAT, NT>), dim3(wgs), dim3(threads), 0, 0, d, n, skew, sink);
  hipEventRecord(e0);
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_read<K, PAT, NT>), dim3(wgs), dim3(threads), 0, 0, d, n, skew, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / iters;
}

int main() {
  const size_t n = 1332ull << 20;
  uint8_t *d;
  uint32_t *sink;
  if (hipMalloc(&d, n + 4096) != hipSuccess) return 1;
  hipMalloc(&sink, 64);
  hipMemset(d, 0x31, n + 4096);
  hipDeviceSynchronize();
  const int iters = 6;
#define ROW(K, PAT, NT, skew, wgs, thr)                                                               \
  {                                                                                                   \
    float ms = run<K, PAT, NT>(d, n, skew, sink, wgs, thr, iters);                                    \
    printf("K=%-2d pat=%d nt=%d skew=%-2u wgs=%-5d thr=%-4d  %.1f us  %.2f TB/s\n", K, PAT, (int)NT, skew, wgs, thr, \
           ms * 1e3, n / (ms * 1e-3) / 1e12);                                                         \
    fflush(stdout);                                                                                   \
  }
  uint8_t *dd[4] = {d, nullptr, nullptr, nullptr};
  for (int i = 1; i < 4; i++) {
    if (hipMalloc(&dd[i], n + 4096) != hipSuccess) return 1;
    hipMemset(dd[i], 0x31, n + 4096);
  }
  hipDeviceSynchronize();
#define LROW(ALU, HV, UNR, nbuf, wgs)                                                                       \
  {                                                                                                    \
    float ms = run_lines<ALU, HV, UNR>(dd, nbuf, n, sink, wgs, 8);                                          \
    printf("lines: alu=%-3d hv=%d unrolled=%d bufs=%d wgs=%-5d  %.1f us  %.2f TB/s\n", ALU, (int)HV, (int)UNR, nbuf, wgs, ms * 1e3, \
           n / (ms * 1e-3) / 1e12);                                                                    \
    fflush(stdout);                                                                                    \
  }
  for (int wgs = 256; wgs <= 1024; wgs *= 2) {
    LROW(64, true, true, 4, wgs) LROW(64, true, false, 4, wgs) LROW(128, true, true, 4, wgs) LROW(128, true, false, 4, wgs)
    LROW(32, true, true, 4, wgs) LROW(32, true, false, 4, wgs)
  }
  return 0;
}
