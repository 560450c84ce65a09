// GPU micro-experiment: how does v_mfma_f32_32x32x16_bf16 round its accumulation?  A = ones (bf16), B = ones,
// C = a large fp32 value with a fractional ulp problem: c + 16 * (tiny) where the exact sum is not representable.
// Prints, for several C / product magnitudes, MFMA result vs exact (double) -> sign of the error tells RNE vs truncation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* cin, const float* av, const float* bv, float* out, int n) {
  for (int t = 0; t < n; ++t) {
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)av[t * 16 + (threadIdx.x >> 5) * 8 + q]; b[q] = (__bf16)bv[t * 16 + (threadIdx.x >> 5) * 8 + q]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = cin[t];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[t] = c[0];
  }
}
int main() {
  const int n = 4096;
  float *cin, *av, *bv, *out;
  hipMallocManaged(&cin, n * 4); hipMallocManaged(&av, n * 64); hipMallocManaged(&bv, n * 64); hipMallocManaged(&out, n * 4);
  srand(1);
  auto rb = [] { float v = (rand() / (float)RAND_MAX) * 2.f - 1.f; __bf16 h = (__bf16)v; return (float)h; };   // bf16-exact values
  for (int t = 0; t < n; ++t) {
    cin[t] = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 8.f;
    for (int q = 0; q < 16; ++q) { av[t * 16 + q] = rb(); bv[t * 16 + q] = rb(); }
  }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cin, av, bv, out, n);
  hipDeviceSynchronize();
  int toward0 = 0, away0 = 0, exact = 0, rne_ok = 0; double serr = 0, sabs = 0;
  for (int t = 0; t < n; ++t) {
    double ex = cin[t];
    for (int q = 0; q < 16; ++q) ex += (double)av[t * 16 + q] * (double)bv[t * 16 + q];
    float rne = (float)ex;
    double e = (double)out[t] - ex;
    if (out[t] == rne) ++rne_ok;
    if (e == 0) ++exact; else if ((e < 0) == (ex > 0)) ++toward0; else ++away0;
    serr += e * (ex > 0 ? 1 : -1); sabs += fabs(e);
  }
  printf("n=%d  result == RNE(exact): %d  exact: %d  error toward zero: %d  away from zero: %d  mean signed error (toward +|x|): %.3e  mean |error|: %.3e\n",
         n, rne_ok, exact, toward0, away0, serr / n, sabs / n);
  return 0;
}
