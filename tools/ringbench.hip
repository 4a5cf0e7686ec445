// ringbench — would k_stream's regular scan run faster on k_stream_gen's LDS-DMA ring?  (experiment harness, not product)
//
// Round 3's verdict asked for one structural experiment on k_stream: its ten-chunk REGISTER pipeline (every chunk
// register re-issued for the next line right after it is consumed, hipcc counting the loads in flight) against the
// LDS-DMA ring of k_stream_gen (global_load_lds_dwordx4 into a ring of 1 KiB slots, our own s_waitcnt vmcnt, one
// ds_read_b128 per chunk, a plain loop at ~70 registers and six workgroups per CU).  Both feed the same stand-in for
// the scan here: per 1 KiB chunk the skip test of the regular grid (four xors with the reference word, three ors, one
// wave-wide `any`) plus ALU dependent VALU operations, over 10 164-byte lines of a 1.3 GB buffer (configs[2]'s line),
// one private run of lines per wave as in k_stream.  What comes out is TB/s per (feed, ALU, workgroups per CU).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ringbench tools/ringbench.hip && ./tools/ringbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef const __attribute__((address_space(3))) uint8_t *lds_bytes_t;

constexpr uint32_t kLine = 10164;
constexpr uint32_t kRef = 0x09307C30u;  // "0|0<TAB>"

template <int ALU>
__device__ __forceinline__ uint32_t work(u32x4 v, uint32_t acc) {
  const uint32_t t0 = v.x ^ kRef, t1 = v.y ^ kRef, t2 = v.z ^ kRef, t3 = v.w ^ kRef;
  uint32_t x = t0 | t1 | t2 | t3;
  if (__any(x != 0)) {  // (the buffer is all "0|0<TAB>" but for one field per ~3 chunks: the skip rate of the 1KG spectrum's order)
#pragma unroll
    for (int i = 0; i < ALU; i++) x = x * 0x9E3779B1u + (x >> 15);
    acc += x;
  }
  return acc;
}

// ---- the register pipeline (k_stream): ten chunk registers, each re-issued for the next line once consumed
template <int ALU>
__global__ __launch_bounds__(256) void k_regs(const uint8_t *buf, size_t nbytes, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * 4;
  const size_t n_lines = (nbytes - 16384) / kLine;
  const size_t per = (n_lines + n_waves - 1) / n_waves;
  const size_t l0 = wave * per, l1 = l0 + per < n_lines ? l0 + per : n_lines;
  if (l0 >= l1) return;
  uint32_t acc = 0;
  u32x4 va[10];
  auto ld = [&](size_t off) -> u32x4 { return __builtin_nontemporal_load((const u32x4_u *)(buf + (off & ~(size_t)3) + 16u * lane)); };
  size_t p = l0 * kLine;
#pragma unroll
  for (int g = 0; g < 10; g++) va[g] = ld(p + 148 + g * 1024);
  for (size_t l = l0; l < l1; l++) {
    const size_t pn = p + kLine;
#pragma unroll
    for (int g = 0; g < 10; g++) {
      acc = work<ALU>(va[g], acc);
      va[g] = ld(pn + 148 + g * 1024);
    }
    p = pn;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// ---- the LDS-DMA ring (k_stream_gen): RING slots of 1 KiB per wave, one chunk read from LDS per step
constexpr int vmcnt_imm(int n) { return (n & 0xF) | 0x70 | 0xF00 | ((n >> 4) << 14); }
template <int ALU, int RING>
__global__ __launch_bounds__(256) void k_ring(const uint8_t *buf, size_t nbytes, uint32_t *sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_ring[];
  const int lane = threadIdx.x & 63;
  const uint32_t wiw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: tell the compiler)
  const uint32_t wave = blockIdx.x * 4 + wiw;
  const uint32_t n_waves = gridDim.x * 4;
  const size_t n_chunks_all = (nbytes - 16384) / 1024;
  const size_t per = (n_chunks_all + n_waves - 1) / n_waves;
  const size_t c0 = wave * per, c1 = c0 + per < n_chunks_all ? c0 + per : n_chunks_all;
  if (c0 >= c1) return;
  const uint8_t *ring = s_ring + wiw * (RING * 1024);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(lds_bytes_t)ring;
  auto issue = [&](size_t c) {
    const uint8_t *gsrc = buf + c * 1024 + 16u * lane;
    const uint32_t dst = ring_lds + (uint32_t)(c % RING) * 1024u;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < RING; j++) issue(c0 + j);
#pragma nounroll
  for (size_t c = c0; c < c1; c++) {
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(RING - 1));  // chunk c has landed
    const u32x4 v = *reinterpret_cast<const u32x4 *>(ring + (c % RING) * 1024 + 16u * lane);
    acc = work<ALU>(v, acc);
    issue(c + RING);  // into the slot just read (the ds_read above is complete: LDS operations of a wave stay in order)
  }
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
  if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float timed(F launch, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(0);
  hipEventRecord(e0);
  for (int i = 0; i < iters; i++) launch(i);
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
  uint8_t *dd[4];
  uint32_t *sink;
  hipMalloc(&sink, 64);
  // all-reference genotype text with one carrier per 3 KiB (a '1' in place of a '0'), so that ~1/3 of the chunks do the work
  uint8_t *h = (uint8_t *)malloc(n + 4096);
  for (size_t i = 0; i < n + 4096; i += 4) {
    h[i] = '0';
    h[i + 1] = '|';
    h[i + 2] = '0';
    h[i + 3] = '\t';
  }
  for (size_t i = 512; i < n; i += 3072) h[i] = '1';
  for (int i = 0; i < 4; i++) {
    if (hipMalloc(&dd[i], n + 4096) != hipSuccess) return 1;
    hipMemcpy(dd[i], h, n + 4096, hipMemcpyHostToDevice);
  }
  free(h);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("ringbench: %d CUs, %.2f GB per launch, line %u B; TB/s by feed, VALU per chunk on the worked third, workgroups per CU\n", cus, n / 1e9, kLine);
#define REGS(ALU, per_cu)                                                                                                   \
  {                                                                                                                          \
    const float ms = timed([&](int i) { hipLaunchKernelGGL((k_regs<ALU>), dim3(cus * per_cu), dim3(256), 0, 0, dd[i & 3], n, sink); }, 8); \
    printf("regs        alu=%-3d wg/cu=%d  %7.1f us  %.2f TB/s\n", ALU, per_cu, ms * 1e3, n / (ms * 1e-3) / 1e12);             \
    fflush(stdout);                                                                                                          \
  }
#define RINGR(ALU, RING, per_cu)                                                                                             \
  {                                                                                                                          \
    const float ms = timed([&](int i) { hipLaunchKernelGGL((k_ring<ALU, RING>), dim3(cus * per_cu), dim3(256), 4 * RING * 1024, 0, dd[i & 3], n, sink); }, 8); \
    printf("ring of %-2d  alu=%-3d wg/cu=%d  %7.1f us  %.2f TB/s\n", RING, ALU, per_cu, ms * 1e3, n / (ms * 1e-3) / 1e12);     \
    fflush(stdout);                                                                                                          \
  }
  REGS(16, 2) REGS(16, 3) REGS(48, 2) REGS(48, 3) REGS(96, 2) REGS(96, 3)
  RINGR(16, 4, 2) RINGR(16, 4, 4) RINGR(16, 4, 6) RINGR(16, 8, 2) RINGR(16, 8, 4)
  RINGR(48, 4, 2) RINGR(48, 4, 4) RINGR(48, 4, 6) RINGR(48, 8, 2) RINGR(48, 8, 4)
  RINGR(96, 4, 2) RINGR(96, 4, 4) RINGR(96, 4, 6) RINGR(96, 8, 2) RINGR(96, 8, 4)
  return 0;
}
