// Is the f16x3 split -- hi = fp16(x s), lo = fp16(x s - hi), s = 2^e -- the same bits when it is formed with FOUR v_fma_mix instructions per pair
// (the scaling inside the products, the subtraction inside the fma: both exact) as with today's 2 v_ldexp + v_cvt_pk_f16_f32 + 2 v_cvt_f32_f16 +
// 2 v_sub + v_cvt_pk_f16_f32?  Random values over the whole exponent range, every scale exponent the kernels use, subnormal fp16 results, Inf, NaN, 0.
// hipcc --offload-arch=gfx950 -O2 tools/micro/split_mix_probe.hip -o /tmp/split_mix_probe && /tmp/split_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_old(float a, float b, int e, uint32_t& h, uint32_t& l) {
  a = ldexpf(a, e); b = ldexpf(b, e);
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  const float ra = a - (float)hh[0], rb = b - (float)hh[1];
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra, rb}, f16x2));
}
__device__ __forceinline__ void split_mix(float a, float b, float s, uint32_t& h, uint32_t& l) {
  uint32_t hh, ll;
  asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hh), "=&v"(ll) : "v"(a), "v"(b), "s"(s));
  h = hh; l = ll;
}
__global__ void k(const float* x, int n, int e, uint32_t* out_old, uint32_t* out_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float s = __uint_as_float((uint32_t)(e + 127) << 23);
  uint32_t h0, l0, h1, l1;
  split_old(x[2 * i], x[2 * i + 1], e, h0, l0);
  split_mix(x[2 * i], x[2 * i + 1], __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s))), h1, l1);
  out_old[2 * i] = h0; out_old[2 * i + 1] = l0; out_new[2 * i] = h1; out_new[2 * i + 1] = l1;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    uint32_t bits = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((uint32_t)rand() << 31);
    if (i % 4 == 1) bits = (bits & 0x807fffffu) | ((uint32_t)(100 + rand() % 60) << 23);      // the magnitudes activations have
    memcpy(&x[i], &bits, 4);
  }
  const uint32_t special[] = {0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00000u, 0u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x477fe000u, 0x477ff000u, 0x33800000u};
  for (size_t i = 0; i < sizeof(special) / 4; ++i) memcpy(&x[2 * i], &special[i], 4);
  float* dx; uint32_t *d0, *d1;
  hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> a(n), b(n);
  long total = 0, diff = 0, diff_nan_only = 0;
  const int es[] = {0, 14, -14, 30, -30, 60, -60, 100, -100, 114, -114, 7, -3};
  for (int e : es) {
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, e, d0, d1);
    hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost);
    long de = 0;
    for (int i = 0; i < n; ++i) {
      ++total;
      if (a[i] != b[i]) {
        // NaN payload / sign differences are not differences of value: classify
        auto isnan16 = [](uint16_t v) { return (v & 0x7c00) == 0x7c00 && (v & 0x3ff); };
        const bool nan_lo = isnan16(a[i] & 0xffff) && isnan16(b[i] & 0xffff), nan_hi = isnan16(a[i] >> 16) && isnan16(b[i] >> 16);
        const bool same_lo = (a[i] & 0xffff) == (b[i] & 0xffff) || nan_lo, same_hi = (a[i] >> 16) == (b[i] >> 16) || nan_hi;
        // ... nor are the sign of a zero piece (fma(-0 s + 0) = +0 where v_ldexp keeps -0; a tiny x s - 0 rounds to a signed zero) or the pieces of a
        // value whose scaled magnitude overflows fp32 (v_ldexp rounds to Inf first: hi = Inf, lo = NaN; the fma forms x s exactly: hi = Inf, lo = -Inf)
        auto zero16 = [](uint16_t v) { return (v & 0x7fff) == 0; };
        auto nonfin16 = [](uint16_t v) { return (v & 0x7c00) == 0x7c00; };
        const bool z_lo = zero16(a[i] & 0xffff) && zero16(b[i] & 0xffff), z_hi = zero16(a[i] >> 16) && zero16(b[i] >> 16);
        const bool o_lo = nonfin16(a[i] & 0xffff) && nonfin16(b[i] & 0xffff), o_hi = nonfin16(a[i] >> 16) && nonfin16(b[i] >> 16);
        if ((same_lo || z_lo || o_lo) && (same_hi || z_hi || o_hi)) ++diff_nan_only;
        else { ++diff; ++de; if (de <= 3) printf("  e=%d word %d: old %08x new %08x (x = %08x %08x)\n", e, i, a[i], b[i], *(uint32_t*)&x[i & ~1], *(uint32_t*)&x[(i & ~1) + 1]); }
      }
    }
  }
  printf("split as 4 v_fma_mix per pair against the round-5 sequence: %ld words compared over %zu scale exponents, %ld differ in a finite non-zero piece, %ld in the sign of a zero / in which non-finite value an overflowed piece holds\n",
         total, sizeof(es) / sizeof(int), diff, diff_nan_only);
  return diff != 0;
}
