// What bit patterns do NaNs have on gfx950?  (Decides whether ReLU as v_max_i32(bits(x), 0) -- one instruction, like v_max_f32 -- carries
// the NaNs that arithmetic produces: it keeps positive-sign NaNs and quenches negative-sign ones.)
// hipcc --offload-arch=gfx950 -O2 tools/micro/nan_bits.hip -o tools/micro/nan_bits && tools/micro/nan_bits
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* in, uint32_t* out) {
  const float inf = in[0], zero = in[1], one = in[2], qnan = in[3], nnan = in[4], minus = in[5];
  int o = 0;
  if (threadIdx.x == 0) {
    out[o++] = __float_as_uint(inf - inf);
    out[o++] = __float_as_uint(zero * inf);
    out[o++] = __float_as_uint(__fsqrt_rn(minus));
    out[o++] = __float_as_uint(fmaf(inf, zero, one));
    out[o++] = __float_as_uint(qnan * minus);            // does a multiplication by -1 flip a NaN's sign?
    out[o++] = __float_as_uint(fmaf(qnan, minus, one));
    out[o++] = __float_as_uint(nnan + one);              // a negative-sign NaN through an add
    out[o++] = __float_as_uint(-qnan);                   // v_xor / neg modifier
    out[o++] = __float_as_uint(fmaxf(qnan, zero));       // today's ReLU
    out[o++] = (uint32_t)max((int)__float_as_uint(qnan), 0);
    out[o++] = (uint32_t)max((int)__float_as_uint(nnan), 0);
    out[o++] = __float_as_uint((float)(_Float16)qnan);
    out[o++] = __float_as_uint((float)(_Float16)inf - (float)(_Float16)inf);
  }
  // MFMA: inf * 0 inside a 16x16x32 f16 product; NaN operand times a negative weight
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)zero; b[i] = (_Float16)zero; }
  if (threadIdx.x == 0) { a[0] = (_Float16)inf; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f16x8 a2, b2;
  for (int i = 0; i < 8; ++i) { a2[i] = (_Float16)zero; b2[i] = (_Float16)minus; }
  if (threadIdx.x == 0) a2[0] = (_Float16)qnan;
  f32x4 c2 = {0, 0, 0, 0};
  c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, c2, 0, 0, 0);
  f16x8 a3, b3;      // inf + (-inf) inside the accumulation
  for (int i = 0; i < 8; ++i) { a3[i] = (_Float16)zero; b3[i] = (_Float16)one; }
  if (threadIdx.x == 0) { a3[0] = (_Float16)inf; a3[1] = (_Float16)(-inf); }
  f32x4 c3 = {0, 0, 0, 0};
  c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b3, c3, 0, 0, 0);
  if (threadIdx.x == 0) { out[16] = __float_as_uint(c[0]); out[17] = __float_as_uint(c2[0]); out[18] = __float_as_uint(c3[0]); }
  bf16x8 d, e;
  for (int i = 0; i < 8; ++i) { d[i] = (__bf16)zero; e[i] = (__bf16)zero; }
  if (threadIdx.x == 0) d[0] = (__bf16)inf;
  f32x4 c4 = {0, 0, 0, 0};
  c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d, e, c4, 0, 0, 0);
  f32x4 c5 = {0, 0, 0, 0};
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 z16 = {0};
  f32x16 c6 = __builtin_amdgcn_mfma_f32_32x32x2f32(threadIdx.x == 0 ? inf : zero, zero, z16, 0, 0, 0);
  c5[0] = c6[0];
  if (threadIdx.x == 0) { out[19] = __float_as_uint(c4[0]); out[20] = __float_as_uint(c5[0]); }
}

int main() {
  float h[6] = {INFINITY, 0.f, 1.f, 0.f, 0.f, -1.f};
  uint32_t q = 0x7FC00000u, n = 0xFFC00000u;
  memcpy(&h[3], &q, 4); memcpy(&h[4], &n, 4);
  float* din; uint32_t* dout;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 32 * 4);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(dout, 0, 32 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
  uint32_t r[32]; hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[] = {"inf - inf", "0 * inf", "sqrt(-1)", "fma(inf, 0, 1)", "(+qNaN) * -1", "fma(+qNaN, -1, 1)", "(-qNaN) + 1", "-(+qNaN)", "fmaxf(+qNaN, 0)",
                         "max_i32(+qNaN, 0)", "max_i32(-qNaN, 0)", "fp16(+qNaN) as f32", "fp16 inf - inf", "", "", "",
                         "mfma f16: inf * 0", "mfma f16: +qNaN * -1", "mfma f16: inf + -inf", "mfma bf16: inf * 0", "mfma f32: inf * 0"};
  for (int i = 0; i < 21; ++i) if (names[i][0]) printf("%-24s 0x%08X\n", names[i], r[i]);
  return 0;
}
