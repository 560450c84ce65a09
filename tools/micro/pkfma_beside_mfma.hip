// GPU micro-experiment: does v_pk_fma_f32 with op_sel broadcasts return wrong values when a
// v_mfma_f32_32x32x16_bf16 stream of ANOTHER wave shares its SIMD?  (Round 2: the bf16x6 layer chain, built with packed fp32
// ops, returned run-to-run different values in lanes 48..63 of exactly such an instruction, only with two workgroups per CU;
// csrc/dss2_gemm_chain16.hip.)  Isolation: per CU, 4 or 8 "noise" waves issue back-to-back bf16 MFMAs while 4 or 8 "probe"
// waves evaluate  acc = a * b.hi + acc  as v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,1,1] and as two v_fma_f32, on the same
// random operands, 4096 iterations, and count lanes whose packed result differs bitwise from the scalar one.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pkfma_beside_mfma.hip -o /tmp/pkfma && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NOISE: 0 = none, 1 = v_mfma_f32_32x32x16_bf16, 2 = v_mfma_f32_32x32x2_f32, 3 = plain VALU (v_fma_f32) busy loop
template <int NOISE, bool SEPARATE_WG>
__global__ void __launch_bounds__(512) probe(const float* __restrict__ in, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // SEPARATE_WG: even workgroups are noise, odd ones probe (two workgroups per CU, as in the failing kernel);
  // otherwise waves 0..3 of every workgroup are noise and waves 4..7 probe (one pair per SIMD)
  const bool noise = SEPARATE_WG ? (blockIdx.x & 1) == 0 : wave < 4;
  if (noise) {
    if (NOISE == 0) return;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)in[lane * 8 + q]; b[q] = (__bf16)in[512 + lane * 8 + q]; }
    f32x16 c0 = {}, c1 = {};
    float fa = in[lane], fb = in[64 + lane];
    for (int i = 0; i < iters * 4; ++i) {
      if (NOISE == 1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
      } else if (NOISE == 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = __builtin_fmaf(c0[r], fa, fb);
      }
    }
    if (c0[0] + c1[3] == 12345.f) sink[0] = c0[1];      // keep the stream alive
    return;
  }
  unsigned nbad = 0;
  f32x2 accp = {in[lane], in[64 + lane]}, accs = accp;
  for (int i = 0; i < iters; ++i) {
    const float* p = in + ((i * 64 + lane) * 4 & 0xffff);
    const f32x2 a = {p[0], p[1]}, b = {p[2], p[3]};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(accp) : "v"(a), "v"(b));
    accs[0] = __builtin_fmaf(a[0], b[1], accs[0]);
    accs[1] = __builtin_fmaf(a[1], b[1], accs[1]);
    nbad += (__float_as_uint(accp[0]) != __float_as_uint(accs[0])) + (__float_as_uint(accp[1]) != __float_as_uint(accs[1]));
    accp = accs = f32x2{accs[0] * 0.5f + p[2], accs[1] * 0.5f + p[0]};      // keep the values bounded
  }
  atomicAdd(bad + (lane >> 4), nbad);      // mismatches per 16-lane group (round 2 saw them in lanes 48..63)
}

template <int N, bool S>
static void run(const char* what, const float* in, unsigned* bad, float* sink, int grid) {
  unsigned tot[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 20; ++rep) {
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((probe<N, S>), dim3(grid), dim3(512), 0, 0, in, bad, sink, 4096);
    hipDeviceSynchronize();
    unsigned h[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    for (int g = 0; g < 4; ++g) tot[g] += h[g];
  }
  printf("%-58s grid %4d x 20 launches: mismatching results by lane group [0-15 | 16-31 | 32-47 | 48-63] = %u %u %u %u\n", what, grid,
         tot[0], tot[1], tot[2], tot[3]);
}

int main() {
  float *in, *sink; unsigned* bad;
  hipMalloc(&in, 65536 * 4 + 64); hipMalloc(&sink, 64); hipMalloc(&bad, 16);
  float* h = (float*)malloc(65536 * 4 + 64);
  srand(7);
  for (int i = 0; i < 65536 + 16; ++i) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
  hipMemcpy(in, h, 65536 * 4 + 64, hipMemcpyHostToDevice);
  run<0, false>("v_pk_fma_f32 alone", in, bad, sink, 1024);
  run<1, false>("beside bf16 MFMA waves of the same workgroup (1 pair / SIMD)", in, bad, sink, 256);
  run<1, false>("... grid 512 (two workgroups per CU)", in, bad, sink, 512);
  run<1, false>("... grid 1024 (four workgroups per CU)", in, bad, sink, 1024);
  run<1, true>("beside bf16 MFMA waves of ANOTHER workgroup on the CU", in, bad, sink, 512);
  run<1, true>("... grid 2048", in, bad, sink, 2048);
  run<2, false>("beside fp32 MFMA waves (v_mfma_f32_32x32x2_f32), grid 1024", in, bad, sink, 1024);
  run<3, false>("beside plain VALU waves (v_fma_f32 loop), grid 1024", in, bad, sink, 1024);
  return 0;
}
