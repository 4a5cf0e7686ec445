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
  asm volatile("s_mov_b64 s[22:23], exec" ::: "s22", "s23");
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
      if (KIND == 25) OP4("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3");
      if (KIND == 26) OP4("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3");
      if (KIND == 27) OP4("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3");
      if (KIND == 28) OP4("v_or_b32 %0, %4, %0\n v_or_b32 %1, %4, %1\n v_or_b32 %2, %4, %2\n v_or_b32 %3, %4, %3");
      if (KIND == 29) OP4("v_min_u32 %0, %4, %0\n v_min_u32 %1, %4, %1\n v_min_u32 %2, %4, %2\n v_min_u32 %3, %4, %3");
      if (KIND == 30) OP4("v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %1, %1, %4, %5\n v_bfe_u32 %2, %2, %4, %5\n v_bfe_u32 %3, %3, %4, %5");
      if (KIND == 31) OP4("v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5");
      if (KIND == 32) OP4("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5");
      if (KIND == 33) OP4("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4");
      if (KIND == 34) OP4("v_cmp_ne_u32 vcc, %4, %0\n v_cmp_ne_u32 vcc, %4, %1\n v_cmp_ne_u32 vcc, %4, %2\n v_cmp_ne_u32 vcc, %4, %3");
      if (KIND == 35) OP4("v_lshl_or_b32 %0, %0, %4, %5\n v_lshl_or_b32 %1, %1, %4, %5\n v_lshl_or_b32 %2, %2, %4, %5\n v_lshl_or_b32 %3, %3, %4, %5");
      if (KIND == 36) OP4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5");
      if (KIND == 37) OP4("v_cndmask_b32 %0, %0, %4, s[22:23]\n v_cndmask_b32 %1, %1, %4, s[22:23]\n v_cndmask_b32 %2, %2, %4, s[22:23]\n v_cndmask_b32 %3, %3, %4, s[22:23]");
      if (KIND == 38) OP4("v_sub_u32 %0, %4, %0\n v_sub_u32 %1, %4, %1\n v_sub_u32 %2, %4, %2\n v_sub_u32 %3, %4, %3");
      if (KIND == 39) OP4("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4");
      if (KIND == 40) OP4("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0");
      if (KIND == 41) OP4("v_xor_b32 %0, 3, %0\n v_xor_b32 %1, 3, %1\n v_xor_b32 %2, 3, %2\n v_xor_b32 %3, 3, %3");
      if (KIND == 42) OP4("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5");
      if (KIND == 43) OP4("v_xad_u32 %0, %0, %4, %5\n v_xad_u32 %1, %1, %4, %5\n v_xad_u32 %2, %2, %4, %5\n v_xad_u32 %3, %3, %4, %5");
      if (KIND == 44) OP4("v_lshl_add_u32 %0, %0, %4, %5\n v_lshl_add_u32 %1, %1, %4, %5\n v_lshl_add_u32 %2, %2, %4, %5\n v_lshl_add_u32 %3, %3, %4, %5");
      if (KIND == 45) OP4("v_and_b32 %0, 0x7f7f7f7f, %0\n v_and_b32 %1, 0x7f7f7f7f, %1\n v_and_b32 %2, 0x7f7f7f7f, %2\n v_and_b32 %3, 0x7f7f7f7f, %3");
      if (KIND == 46) OP4("v_mul_u32_u24 %0, %4, %0\n v_mul_u32_u24 %1, %4, %1\n v_mul_u32_u24 %2, %4, %2\n v_mul_u32_u24 %3, %4, %3");
      if (KIND == 47) OP4("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x80\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x80\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x80\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x80");
      if (KIND == 48) OP4("v_bitop3_b32 %0, %0, 7, %5 bitop3:0x36\n v_bitop3_b32 %1, %1, 7, %5 bitop3:0x36\n v_bitop3_b32 %2, %2, 7, %5 bitop3:0x36\n v_bitop3_b32 %3, %3, 7, %5 bitop3:0x36");
      if (KIND == 49) OP4("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4");
      if (KIND == 50) OP4("v_dot4_u32_u8 %0, %0, %4, %5\n v_dot4_u32_u8 %1, %1, %4, %5\n v_dot4_u32_u8 %2, %2, %4, %5\n v_dot4_u32_u8 %3, %3, %4, %5");
      if (KIND == 51) OP4("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4");
      if (KIND == 52) OP4("v_sad_u8 %0, %0, %4, %5\n v_sad_u8 %1, %1, %4, %5\n v_sad_u8 %2, %2, %4, %5\n v_sad_u8 %3, %3, %4, %5");
      if (KIND == 53) OP4("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int KIND>
static void run(const char *name, uint32_t *out) {
  printf("%-44s", name);
  for (int wpc = 1; wpc <= 4; wpc *= 4) {
    const int wgs = 256 * wpc, iters = 256;
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
  run<25>("v_lshrrev_b32 v,v", out);
  run<26>("v_lshlrev_b32 v,v", out);
  run<27>("v_and_b32 v,v", out);
  run<28>("v_or_b32 v,v", out);
  run<29>("v_min_u32 v,v", out);
  run<30>("v_bfe_u32 v,v,v", out);
  run<31>("v_alignbyte_b32 v,v,v", out);
  run<32>("v_or3_b32 v,v,v", out);
  run<33>("v_bcnt_u32_b32 v,v", out);
  run<34>("v_cmp_ne_u32 vcc,v,v", out);
  run<35>("v_lshl_or_b32 v,v,v", out);
  run<36>("v_perm_b32 v,v,v", out);
  run<37>("v_cndmask_b32 v,v,s[22:23]", out);
  run<38>("v_sub_u32 v,v", out);
  run<39>("v_mov_b32 v,v", out);
  run<40>("v_mov_b32 v,0", out);
  run<41>("v_xor_b32 3,v (inline const)", out);
  run<42>("v_add3_u32 v,v,v", out);
  run<43>("v_xad_u32 v,v,v", out);
  run<44>("v_lshl_add_u32 v,v,v", out);
  run<45>("v_and_b32 literal,v", out);
  run<46>("v_mul_u32_u24 v,v", out);
  run<47>("v_bitop3 v,v,v (and3)", out);
  run<48>("v_bitop3 v,7,v (inline)", out);
  run<49>("v_mul_lo_u32 v,v", out);
  run<50>("v_dot4_u32_u8 v,v,v", out);
  run<51>("v_mul_hi_u32 v,v", out);
  run<52>("v_sad_u8 v,v,v", out);
  run<53>("v_mad_u32_u24 v,v,v", out);
  return 0;
}
