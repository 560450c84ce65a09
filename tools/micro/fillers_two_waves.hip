// GPU micro-experiment (round 5): how many of a wave's OWN vector instructions hide between its bf16 MFMAs when TWO such waves share a
// SIMD?  Round 4 (work_beside_mfma.hip): a wave that only issues VALU work beside ANOTHER wave's dense MFMA stream is starved (a
// dependent fma chain runs 9x slower, the MFMA stream not at all) -- so a phase-structured kernel (hops; then MFMAs) with two workgroups
// per CU pays MFMA time PLUS vector time.  A software-pipelined wave (chunk c's MFMAs interleaved with chunk c + 1's hops / splits)
// does not depend on the arbiter; the question is the filler budget per MFMA at one and at two such waves per SIMD, for plain VALU
// fillers and for LDS-gather + fma fillers (the hop mix).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fillers_two_waves.hip -o /tmp/f2w && /tmp/f2w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// F: VALU fillers per MFMA; G: per MFMA one ds_read_b128 gather feeding 4 of the fillers (0 / 1)
template <int F, int G>
__global__ void __launch_bounds__(512) k(const float* __restrict__ in, float* __restrict__ sink, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = in[i & 4095];
  __syncthreads();
  bf16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (__bf16)in[lane * 8 + q]; b[q] = (__bf16)in[512 + lane * 8 + q]; }
  f32x16 c[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  float v[12];
  for (int j = 0; j < 12; ++j) v[j] = in[lane + j];
  const float w = in[lane + 100];
  int goff = (lane * 16 + wave * 1024) & 32767;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (G) { z = *reinterpret_cast<const f32x4*>(lds + goff); goff = (goff + 4112) & 32767 & ~15; }
#pragma unroll
      for (int f = 0; f < F; ++f) v[(j * F + f) % 12] = __builtin_fmaf(v[(j * F + f) % 12], w, G && f < 4 ? z[f] : 1.0f);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (G) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (F) __builtin_amdgcn_sched_group_barrier(0x002, F + (G ? 2 : 0), 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 6; ++j) s += c[j][0] + c[j][5];
  for (int j = 0; j < 12; ++j) s += v[j];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int F, int G>
static void run(const float* in, float* sink, unsigned long long* cyc, int nwaves) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<F, G>), dim3(256), dim3(64 * nwaves), 32768, 0, in, sink, cyc, iters);
  hipLaunchKernelGGL((k<F, G>), dim3(256), dim3(64 * nwaves), 32768, 0, in, sink, cyc, iters);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; int n = 0;
  for (int bI = 0; bI < 256; ++bI) for (int wv = 0; wv < nwaves; ++wv) { s += (double)h[bI * 8 + wv]; ++n; }
  const double per_mfma = s / n / iters / 6.0;
  printf("  %d waves/SIMD  F=%2d fillers/MFMA%s: %6.1f cycles per MFMA and wave  -> SIMD pipe busy %5.1f %%\n", nwaves / 4, F, G ? " + 1 LDS gather" : "              ",
         per_mfma, 100.0 * 32.0 * (nwaves / 4) / per_mfma);
}

int main() {
  float *in, *sink; unsigned long long* cyc;
  hipMalloc(&in, 8192 * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 8 * 8);
  float* h = (float*)malloc(8192 * 4);
  srand(5);
  for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX) - 0.5f;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  printf("v_mfma_f32_32x32x16_bf16 (32 cycles of pipe each) with F own-wave v_fma_f32 fillers after each; 256 workgroups = one per CU\n");
  for (int nw = 4; nw <= 8; nw += 4) {
    run<0, 0>(in, sink, cyc, nw); run<2, 0>(in, sink, cyc, nw); run<4, 0>(in, sink, cyc, nw); run<6, 0>(in, sink, cyc, nw);
    run<8, 0>(in, sink, cyc, nw); run<12, 0>(in, sink, cyc, nw); run<16, 0>(in, sink, cyc, nw);
    run<4, 1>(in, sink, cyc, nw); run<6, 1>(in, sink, cyc, nw); run<8, 1>(in, sink, cyc, nw); run<12, 1>(in, sink, cyc, nw);
  }
  return 0;
}
