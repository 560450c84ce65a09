// Probe for the fp16 two-piece split (x = hi + lo, fp16 each, after a power-of-two scale) and the f16 MFMAs of gfx950:
//   * does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs (a lo piece below 2^-14), or flush them?
//   * the error of (hi, lo) x (hi, lo) products hh + hl + lh against fp64 on random data, beside the bf16x6 form.
// hipcc --offload-arch=gfx950 -O3 tools/micro/f16x3_probe.hip -o /tmp/f16x3 && /tmp/f16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& h, uint32_t& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  const float ra = a - (float)hh[0], rb = b - (float)hh[1];
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra, rb}, f16x2));
}

// C[32][32] = A[32][16] * B[16][32] on one wave; A row-major [32][16] fp32, B row-major [16][32] fp32; mode 0: hi only, 1: hh + hl + lh
__global__ void probe(const float* A, const float* B, float* C, float sa, float sb, int mode) {
  const int lane = threadIdx.x, c32 = lane & 31, half = lane >> 5;
  f16x8 ah, al, bh, bl;
  for (int q = 0; q < 8; q += 2) {
    uint32_t h, l;
    split2_pair(A[c32 * 16 + half * 8 + q] * sa, A[c32 * 16 + half * 8 + q + 1] * sa, h, l);
    f16x2 hv = __builtin_bit_cast(f16x2, h), lv = __builtin_bit_cast(f16x2, l);
    ah[q] = hv[0]; ah[q + 1] = hv[1]; al[q] = lv[0]; al[q + 1] = lv[1];
    split2_pair(B[(half * 8 + q) * 32 + c32] * sb, B[(half * 8 + q + 1) * 32 + c32] * sb, h, l);
    hv = __builtin_bit_cast(f16x2, h); lv = __builtin_bit_cast(f16x2, l);
    bh[q] = hv[0]; bh[q + 1] = hv[1]; bl[q] = lv[0]; bl[q + 1] = lv[1];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (mode == 1) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  }
  if (mode == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);      // lo x hi only
  else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  const float inv = 1.f / (sa * sb);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + c32] = c[r] * inv;
}

int main() {
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(512), B(512), C(1024);
  float *dA, *dB, *dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  auto run = [&](float sa, float sb, int mode) {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb, mode);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  };
  // 1. subnormal inputs: A = 2^-20 everywhere (fp16 subnormal: 16 ulps of 2^-24), B = 1
  for (auto& a : A) a = ldexpf(1.f, -20);
  for (auto& b : B) b = 1.f;
  run(1.f, 1.f, 0);
  printf("subnormal A (2^-20) x 1, K = 16: got %.6e, exact %.6e  -> f16 MFMA %s subnormal inputs\n", C[0], 16 * ldexp(1.0, -20), C[0] > 0 ? "HONOURS" : "FLUSHES");
  // lo piece subnormal: A = 1 + 2^-13 (hi = 1, lo = 2^-13... normal) vs A = 2^-6 (1 + 2^-13): lo = 2^-19 subnormal
  for (auto& a : A) a = ldexpf(1.f + ldexpf(1.f, -13), -6);
  run(1.f, 1.f, 2);
  printf("lo piece 2^-19 (subnormal) x 1, K = 16: lo-only product %.6e, exact %.6e\n", C[0], 16 * ldexp(1.0, -19));
  // 2. accuracy on random data
  for (int trial = 0; trial < 2; ++trial) {
    for (auto& a : A) a = nd(rng) * (trial ? 1e-3f : 1.f);
    for (auto& b : B) b = nd(rng) * 0.09f;
    std::vector<double> ref(1024, 0.0);
    double mx = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)A[i * 16 + k] * B[k * 32 + j]; ref[i * 32 + j] = s; mx = fmax(mx, fabs(s)); }
    for (int cfg = 0; cfg < 3; ++cfg) {
      const float sa = cfg == 0 ? 1.f : (trial ? 4194304.f : 4096.f), sb = cfg == 0 ? 1.f : (cfg == 1 ? 65536.f : 16.f);
      run(sa, sb, 1);
      double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - ref[i]));
      printf("random, |A| ~ %s, |B| ~ 0.09: scales (%g, %g): max |err| / max |ref| = %.3e\n", trial ? "1e-3" : "1", sa, sb, e / mx);
    }
  }
  return 0;
}
