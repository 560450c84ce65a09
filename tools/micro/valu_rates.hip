// Issue rate of the vector instructions the f16x3 split is made of, one wave per SIMD and four (256 threads x 1024 workgroups), independent chains.
// hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(float* out, float s, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long d0 = threadIdx.x, d1 = 12345;
  uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) asm volatile(REP8("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 1) asm volatile(REP8("v_fma_mixlo_f16 %0, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %4, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %5, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %6, %8, %9, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %7, %8, %9, 0 op_sel_hi:[0,0,0]\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a0), "v"(s));
    if (OP == 2) asm volatile(REP8("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %8, %9\n v_cvt_pk_f16_f32 %2, %8, %9\n v_cvt_pk_f16_f32 %3, %8, %9\n v_cvt_pk_f16_f32 %4, %8, %9\n v_cvt_pk_f16_f32 %5, %8, %9\n v_cvt_pk_f16_f32 %6, %8, %9\n v_cvt_pk_f16_f32 %7, %8, %9\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a0), "v"(s));
    if (OP == 3) asm volatile(REP8("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h0));
    if (OP == 4) asm volatile(REP8("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h0));
    if (OP == 5) asm volatile(REP8("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[0,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[0,0,0]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(s));
#define OP8F(ins) asm volatile(REP8(ins " %0, %0, %8\n " ins " %1, %1, %8\n " ins " %2, %2, %8\n " ins " %3, %3, %8\n " ins " %4, %4, %8\n " ins " %5, %5, %8\n " ins " %6, %6, %8\n " ins " %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s))
    if (OP == 6) OP8F("v_mul_f32");
    if (OP == 7) OP8F("v_add_f32");
    if (OP == 8) OP8F("v_max_f32");
    if (OP == 9) OP8F("v_and_b32");
    if (OP == 10) OP8F("v_min_u32");
    if (OP == 11) OP8F("v_sub_f32");
    if (OP == 12) asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 13) asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    if (OP == 14) asm volatile(REP8("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 15) asm volatile(REP8("v_cmp_nge_f32 vcc, 0, %0\n v_cmp_nge_f32 vcc, 0, %1\n v_cmp_nge_f32 vcc, 0, %2\n v_cmp_nge_f32 vcc, 0, %3\n v_cmp_nge_f32 vcc, 0, %4\n v_cmp_nge_f32 vcc, 0, %5\n v_cmp_nge_f32 vcc, 0, %6\n v_cmp_nge_f32 vcc, 0, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    if (OP == 16) OP8F("v_fmac_f32");
    if (OP == 21) OP8F("v_mul_lo_u32");
    if (OP == 22) OP8F("v_add_u32");
    if (OP == 23) OP8F("v_or_b32");
    if (OP == 24) OP8F("v_xor_b32");
    if (OP == 25) OP8F("v_lshlrev_b32");
    if (OP == 26) asm volatile(REP8("v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 27) asm volatile(REP8("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 28) asm volatile(REP8("v_mad_u32_u24 %0, %0, %8, %8\n v_mad_u32_u24 %1, %1, %8, %8\n v_mad_u32_u24 %2, %2, %8, %8\n v_mad_u32_u24 %3, %3, %8, %8\n v_mad_u32_u24 %4, %4, %8, %8\n v_mad_u32_u24 %5, %5, %8, %8\n v_mad_u32_u24 %6, %6, %8, %8\n v_mad_u32_u24 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 29) asm volatile(REP8("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %0, %0, 1, %1\n") : "+v"(d0) : "v"(d1));
    if (OP == 30) OP8F("v_mul_u32_u24");
    // v_cndmask variants: e64 with an SGPR-pair mask; destination not an input; a relu as add |x| + mul 0.5
    if (OP == 17) asm volatile("s_mov_b64 s[10:11], 0x5555\n" REP8("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s10", "s11");
    if (OP == 18) asm volatile(REP8("v_cndmask_b32 %0, 0, %8, vcc\n v_cndmask_b32 %1, 0, %8, vcc\n v_cndmask_b32 %2, 0, %8, vcc\n v_cndmask_b32 %3, 0, %8, vcc\n v_cndmask_b32 %4, 0, %8, vcc\n v_cndmask_b32 %5, 0, %8, vcc\n v_cndmask_b32 %6, 0, %8, vcc\n v_cndmask_b32 %7, 0, %8, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
    if (OP == 19) asm volatile(REP8("v_add_f32 %0, %0, |%0|\n v_add_f32 %1, %1, |%1|\n v_add_f32 %2, %2, |%2|\n v_add_f32 %3, %3, |%3|\n v_add_f32 %4, %4, |%4|\n v_add_f32 %5, %5, |%5|\n v_add_f32 %6, %6, |%6|\n v_add_f32 %7, %7, |%7|\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    if (OP == 20) asm volatile(REP8("v_cmp_nge_f32 vcc, 0, %0\n v_cndmask_b32 %0, 0, %0, vcc\n v_cmp_nge_f32 vcc, 0, %1\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_nge_f32 vcc, 0, %2\n v_cndmask_b32 %2, 0, %2, vcc\n v_cmp_nge_f32 vcc, 0, %3\n v_cndmask_b32 %3, 0, %3, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7) + (float)d0;
}
template <int OP>
static void run(const char* name, float* out) {
  const int iters = 2000, blocks = 1024;      // 64 instructions per iteration and wave
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nt : {64, 256}) {
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(nt), 0, 0, out, 1.0001f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(nt), 0, 0, out, 1.0001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * (nt / 64) * iters * 64;      // wave-instructions
    printf("%-22s %3d threads / workgroup: %7.1f G wave-instructions/s  (%.2f per SIMD and ns at 1024 SIMDs)\n", name, nt, winstr / ms / 1e6, winstr / ms / 1e6 / 1024);
  }
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  run<0>("v_fma_f32", out); run<5>("v_fma_mix_f32", out); run<1>("v_fma_mixlo_f16", out); run<2>("v_cvt_pk_f16_f32", out); run<3>("v_ldexp_f32", out); run<4>("v_cvt_f32_f16", out);
  run<6>("v_mul_f32", out); run<7>("v_add_f32", out); run<11>("v_sub_f32", out); run<16>("v_fmac_f32", out); run<8>("v_max_f32", out); run<9>("v_and_b32", out); run<10>("v_min_u32", out); run<12>("v_mov_b32", out);
  run<21>("v_mul_lo_u32", out); run<30>("v_mul_u32_u24", out); run<28>("v_mad_u32_u24", out); run<22>("v_add_u32", out); run<27>("v_add3_u32", out); run<23>("v_or_b32", out); run<24>("v_xor_b32", out); run<25>("v_lshlrev_b32", out); run<26>("v_bfe_i32", out); run<29>("v_lshl_add_u64 (dependent)", out);
  run<13>("v_cndmask_b32", out); run<17>("v_cndmask_b32_e64 sgpr", out); run<18>("v_cndmask dst only", out); run<19>("v_add_f32 x, |x| (VOP3)", out); run<20>("cmp+cndmask pairs (x32)", out); run<14>("v_lshl_or_b32", out); run<15>("v_cmp_nge_f32", out);
  return 0;
}
