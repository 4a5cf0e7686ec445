// fetchbench — VALU issue cost by instruction form, whole chip busy, straight-line body of 2048
// instructions over 4 independent accumulators, 1/2/4 waves per SIMD.  cycles at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OP4(fmt) asm volatile(fmt : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n), "s"(s1), "s"(s2) : "vcc", "s20")
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a = threadIdx.x * seed + 1, b = a ^ 0x5bd1e995u, c = a + 77, d = b + 99;
  const uint32_t m = seed * 0x01010101u + threadIdx.x, n = m * 3 + 1, s1 = seed * 5, s2 = seed * 9 + 1;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 2048 / 4; r++) {
      if (KIND == 0) OP4("v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n v_xor_b32 %3, %4, %3");
      if (KIND == 1) OP4("v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4");
      if (KIND == 2) OP4("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5");
      if (KIND == 3) OP4("v_and_or_b32 %0, %0, %6, %5\n v_and_or_b32 %1, %1, %6, %5\n v_and_or_b32 %2, %2, %6, %5\n v_and_or_b32 %3, %3, %6, %5");
      if (KIND == 4) OP4("v_xor_b32 %0, 0x12345678, %0\n v_xor_b32 %1, 0x12345678, %1\n v_xor_b32 %2, 0x12345678, %2\n v_xor_b32 %3, 0x12345678, %3");
      if (KIND == 5) OP4("v_bfe_u32 %0, %0, 3, 30\n v_bfe_u32 %1, %1, 3, 30\n v_bfe_u32 %2, %2, 3, 30\n v_bfe_u32 %3, %3, 3, 30");
      if (KIND == 6) OP4("v_alignbyte_b32 %0, %0, %4, %6\n v_alignbyte_b32 %1, %1, %4, %6\n v_alignbyte_b32 %2, %2, %4, %6\n v_alignbyte_b32 %3, %3, %4, %6");
      if (KIND == 7) OP4("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf");
      if (KIND == 8) OP4("v_lshrrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshrrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshrrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshrrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD");
      if (KIND == 9) OP4("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x36\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x36\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x36\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x36");
      if (KIND == 10) OP4("v_min_u32 %0, 3, %0\n v_min_u32 %1, 3, %1\n v_min_u32 %2, 3, %2\n v_min_u32 %3, 3, %3");
      if (KIND == 11) OP4("v_lshl_or_b32 %0, %0, 2, %4\n v_lshl_or_b32 %1, %1, 2, %4\n v_lshl_or_b32 %2, %2, 2, %4\n v_lshl_or_b32 %3, %3, 2, %4");
      if (KIND == 12) OP4("v_bitop3_b32 %0, %0, %6, %5 bitop3:0x36\n v_bitop3_b32 %1, %1, %6, %5 bitop3:0x36\n v_bitop3_b32 %2, %2, %6, %5 bitop3:0x36\n v_bitop3_b32 %3, %3, %6, %5 bitop3:0x36");
      if (KIND == 13) OP4("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3");
      if (KIND == 14) OP4("v_cmp_ne_u32 vcc, 0, %0\n v_cmp_ne_u32 vcc, 0, %1\n v_cmp_ne_u32 vcc, 0, %2\n v_cmp_ne_u32 vcc, 0, %3");
      if (KIND == 15) OP4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc");
      if (KIND == 16) OP4("v_bcnt_u32_b32 %0, %0, 0\n v_bcnt_u32_b32 %1, %1, 0\n v_bcnt_u32_b32 %2, %2, 0\n v_bcnt_u32_b32 %3, %3, 0");
      if (KIND == 17) OP4("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3");
      if (KIND == 18) OP4("v_xor_b32 %0, %6, %0\n v_xor_b32 %1, %6, %1\n v_xor_b32 %2, %6, %2\n v_xor_b32 %3, %6, %3");
      if (KIND == 19) OP4("v_lshrrev_b32_e64 %0, %0, %6\n v_lshrrev_b32_e64 %1, %1, %6\n v_lshrrev_b32_e64 %2, %2, %6\n v_lshrrev_b32_e64 %3, %3, %6");
      if (KIND == 20) OP4("s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1");
      if (KIND == 21) OP4("v_xor_b32 %0, %4, %0\n s_add_u32 s20, s20, 1\n v_xor_b32 %2, %4, %2\n s_add_u32 s20, s20, 1");
      if (KIND == 22) OP4("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3");
      if (KIND == 23) OP4("v_perm_b32 %0, %0, %4, %6\n v_perm_b32 %1, %1, %4, %6\n v_perm_b32 %2, %2, %4, %6\n v_perm_b32 %3, %3, %4, %6");
      if (KIND == 24) OP4("v_and_or_b32 %0, %0, 3, 7\n v_and_or_b32 %1, %1, 3, 7\n v_and_or_b32 %2, %2, 3, 7\n v_and_or_b32 %3, %3, 3, 7");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int KIND>
static void run(const char *name, uint32_t *out) {
  printf("%-44s", name);
  for (int wpc = 1; wpc <= 4; wpc *= 2) {
    const int wgs = 256 * wpc, iters = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(256), 0, 0, out, 3u, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(wgs), dim3(256), 0, 0, out, 3u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  %dw: %.2f", wpc, ms * 1e-3 * 2.4e9 / ((double)iters * 2048 * wpc));
  }
  printf("   cyc/inst/SIMD\n");
  fflush(stdout);
}

int main() {
  uint32_t *out;
  hipMalloc(&out, 64 << 20);
  run<0>("v_xor_b32 v,v (VOP2)", out);
  run<22>("v_add_u32 v,v (VOP2)", out);
  run<18>("v_xor_b32 s,v (VOP2)", out);
  run<13>("v_lshlrev_b32 1,v (VOP2)", out);
  run<10>("v_min_u32 3,v (VOP2)", out);
  run<4>("v_xor_b32 literal,v (VOP2+lit 8B)", out);
  run<14>("v_cmp_ne_u32 vcc,0,v (VOPC)", out);
  run<15>("v_cndmask_b32 v,v,vcc (VOP2)", out);
  run<17>("v_readfirstlane_b32 (VOP1)", out);
  run<1>("v_and_or_b32 v,v,m,m (VOP3 3 vgpr, 2 same)", out);
  run<2>("v_and_or_b32 v,v,m,n (VOP3 3 vgpr)", out);
  run<3>("v_and_or_b32 v,v,s,n (VOP3 2 vgpr 1 sgpr)", out);
  run<24>("v_and_or_b32 v,v,3,7 (VOP3 1 vgpr 2 inline)", out);
  run<5>("v_bfe_u32 v,v,3,30 (VOP3 inline)", out);
  run<6>("v_alignbyte_b32 v,v,m,s (VOP3)", out);
  run<11>("v_lshl_or_b32 v,v,2,m (VOP3 2 vgpr)", out);
  run<19>("v_lshrrev_b32_e64 v,v,s (VOP3 1 vgpr)", out);
  run<9>("v_bitop3_b32 v,v,m,n (VOP3 3 vgpr)", out);
  run<12>("v_bitop3_b32 v,v,s,n (VOP3 2 vgpr)", out);
  run<16>("v_bcnt_u32_b32 v,v,0 (VOP3)", out);
  run<23>("v_perm_b32 v,v,m,s (VOP3)", out);
  run<7>("v_mov_b32_dpp row_shr:1 (DPP 8B)", out);
  run<8>("v_lshrrev_b32_sdwa (SDWA 8B)", out);
  run<20>("s_add_u32 (SALU)", out);
  run<21>("v_xor / s_add alternating", out);
  return 0;
}
