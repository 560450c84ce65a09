// GPU micro-experiment (round 5): the bf16x3 operand split with v_dot2c_f32_bf16 residuals.
//
// split3_pair (csrc/dss2_common.hpp) splits two fp32 values into three packed bf16 pieces each.  Its residuals
//   r = v - float(h)   are formed as  shift / mask (unpack h to fp32) + v_sub_f32: 2 VALU per value and level, 11 per pair.
// v_dot2c_f32_bf16 D, A, B computes D += A.lo * B.lo + A.hi * B.hi on packed bf16 pairs: with B = {-1, 0} / {0, -1} it is
//   r = v - h.lo  /  r = v - h.hi   in ONE instruction (7 per pair).  Two questions before it goes into the kernels:
//   (1) is the result bitwise the fp32 residual (exact products, one rounding that never rounds: the residual is representable),
//       also for denormal-range residuals, zeros, huge values, and does 0 * inf in the unused half poison it;
//   (2) does it share the packed-fp32 hazard of v_pk_fma_f32 beside a bf16 MFMA stream (tools/micro/pkfma_beside_mfma.hip:
//       wrong values in lanes 48..63 with two workgroups per CU).
// And (3) its throughput: pairs per ns per CU for both forms.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dot2_split_beside_mfma.hip -o /tmp/dot2 && /tmp/dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void split_old(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(ra, rb);
  l = cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
// (the selector constants go through registers the compiler cannot see into: with literal operands hipcc 7.2 encodes
//  {-1 bf16, 0} = 0x0000bf80 as the INLINE constant -1.0, which the instruction reads as fp32 -1.0 = {0, -1 bf16}: the first
//  run of this experiment subtracted the wrong half)
__device__ __forceinline__ uint32_t opaque(uint32_t c) { asm volatile("" : "+v"(c)); return c; }
__device__ __forceinline__ float sub_lo(float v, uint32_t pk) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, opaque(0x0000bf80u)), v, false);
}
__device__ __forceinline__ float sub_hi(float v, uint32_t pk) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), __builtin_bit_cast(bf16x2, opaque(0xbf800000u)), v, false);
}
__device__ __forceinline__ void split_new(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf16(a, b);
  const float ra = sub_lo(a, h), rb = sub_hi(b, h);
  m = cvt_pk_bf16(ra, rb);
  l = cvt_pk_bf16(sub_lo(ra, m), sub_hi(rb, m));
}

// NOISE: 0 none, 1 bf16 MFMA waves in OTHER workgroups of the CU, 2 bf16 MFMA waves of the same workgroup
template <int NOISE>
__global__ void __launch_bounds__(512) probe(const float* __restrict__ in, int n_in, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool noise = NOISE == 1 ? (blockIdx.x & 1) == 0 : (NOISE == 2 ? wave < 4 : false);
  if (noise) {
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)in[lane * 8 + q]; b[q] = (__bf16)in[512 + lane * 8 + q]; }
    f32x16 c0 = {}, c1 = {};
    for (int i = 0; i < iters * 4; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 12345.f) sink[0] = c0[1];
    return;
  }
  unsigned nbad = 0;
  for (int i = 0; i < iters; ++i) {
    const int idx = (int)(((unsigned)(blockIdx.x * 977 + i) * 64u + lane) * 2u % (unsigned)(n_in - 1));
    const float a = in[idx], b = in[idx + 1];
    uint32_t h0, m0, l0, h1, m1, l1;
    split_old(a, b, h0, m0, l0);
    split_new(a, b, h1, m1, l1);
    nbad += (h0 != h1) + (m0 != m1) + (l0 != l1);
  }
  atomicAdd(bad + (lane >> 4), nbad);
}

template <bool NEW>
__global__ void __launch_bounds__(256) rate(const float* __restrict__ in, float* __restrict__ sink, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint32_t h, m, l;
      if (NEW) split_new(a, b, h, m, l); else split_old(a, b, h, m, l);
      acc ^= h + m + l;
      a = __uint_as_float((__float_as_uint(a) ^ (l << 3)) & 0xbfffffffu);      // data dependence, stays finite
      b = __uint_as_float((__float_as_uint(b) ^ (m >> 5)) & 0xbfffffffu);
    }
  }
  if (acc == 0x12345u) sink[0] = a + b;
}

template <int N>
static void run(const char* what, const float* in, int n_in, unsigned* bad, float* sink, int grid) {
  unsigned long long tot[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 20; ++rep) {
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N>), dim3(grid), dim3(512), 0, 0, in, n_in, bad, sink, 4096);
    hipDeviceSynchronize();
    unsigned h[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    for (int g = 0; g < 4; ++g) tot[g] += h[g];
  }
  printf("%-66s grid %4d x 20 launches: differing pieces by lane group [0-15 | 16-31 | 32-47 | 48-63] = %llu %llu %llu %llu\n", what, grid,
         tot[0], tot[1], tot[2], tot[3]);
}

int main() {
  const int n = 1 << 18;
  float *in, *sink; unsigned* bad;
  hipMalloc(&in, n * 4); hipMalloc(&sink, 64); hipMalloc(&bad, 16);
  float* h = (float*)malloc(n * 4);
  srand(11);
  for (int i = 0; i < n; ++i) {
    // random sign / mantissa over the whole exponent range (incl. denormals), plus specials
    uint32_t bits = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((uint32_t)rand() << 31);
    const int kind = i & 15;
    if (kind < 8) { const uint32_t e = 100 + (uint32_t)(rand() % 56); bits = (bits & 0x807fffffu) | (e << 23); }      // "ordinary" magnitudes 2^-27 .. 2^28
    else if (kind == 8) bits &= 0x807fffffu;                                                                                // denormals
    else if (kind == 9) bits = (bits & 0x80000000u);                                                                        // +-0
    else if (kind == 10) bits = (bits & 0x807fffffu) | (1u << 23) | ((uint32_t)(rand() % 24) << 23);                       // tiny normals: residuals go denormal
    else if (kind == 11) bits = (bits & 0x807fffffu) | (0xfeu << 23);                                                       // near FLT_MAX (finite)
    memcpy(&h[i], &bits, 4);
    if (!std::isfinite(h[i])) h[i] = 1.f;
  }
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  run<0>("dot2c split vs shift/mask/sub split, alone", in, n, bad, sink, 1024);
  run<2>("... beside bf16 MFMA waves of the same workgroup (1 pair / SIMD)", in, n, bad, sink, 512);
  run<1>("... beside bf16 MFMA waves of ANOTHER workgroup on the CU", in, n, bad, sink, 512);
  run<1>("... grid 2048", in, n, bad, sink, 2048);
  // an infinite value in the UNUSED half of the packed operand: 0 * inf = NaN would poison the residual of the finite half
  {
    float two[4] = {1.2345678f, INFINITY, 1.2345678f, 3.0f};
    float* d; hipMalloc(&d, 16); hipMemcpy(d, two, 16, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<0>), dim3(1), dim3(64), 0, 0, d, 2, bad, sink, 1);
    hipDeviceSynchronize();
    unsigned hb[4]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    printf("pair (finite, +inf): pieces differing from the shift/mask/sub form: %u (non-finite in => non-finite out either way; NaN != NaN counts)\n", hb[0] + hb[1] + hb[2] + hb[3]);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nw = 0; nw < 2; ++nw) {
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (nw) hipLaunchKernelGGL((rate<true>), dim3(2048), dim3(256), 0, 0, in, sink, 2048);
      else hipLaunchKernelGGL((rate<false>), dim3(2048), dim3(256), 0, 0, in, sink, 2048);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double pairs = 2048.0 * 256 * 2048 * 8;
    printf("%s: %.3f ms for %.3g pair splits = %.1f pair splits per ns (chip), incl. the loop's 6 dependence ops per pair\n",
           nw ? "dot2c form (7 VALU / pair)      " : "shift/mask/sub form (11 / pair)", ms, pairs, pairs / (ms * 1e6));
  }
  return 0;
}
