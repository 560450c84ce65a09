// GPU micro-experiment (round 4): does the chip hold a higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16
// in the layer chain's GEMM regime?  (/opt/skills/guides/MI355X_MICROARCH.md 'DVFS give-back' item 7: 1.12-1.15 x in bare loops.)
// One wave = a 64-row x 32-column output block of three matrices (96 accumulator registers, as gemm_chain_sp_kernel), A fragments
// of three bf16 planes re-read from LDS every k-step (ds_read_b128), B fragments in registers, the bf16x6 product order.  Random
// operands.  Two workgroups of four waves per CU.  Same FLOPs per launch for both shapes; wall time by events over many launches.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape_clock.hip -o /tmp/msc && /tmp/msc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ in, float* __restrict__ sink, unsigned long long* __restrict__ clk, int ksteps16) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // 3 planes x 64 rows x 128 k bf16 = 48 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 3 * 64 * 128 / 2; i += 256) {
    const float a = in[(i * 2) & 8191], b = in[(i * 2 + 1) & 8191];
    reinterpret_cast<__bf16*>(lds)[2 * i] = (__bf16)a; reinterpret_cast<__bf16*>(lds)[2 * i + 1] = (__bf16)b;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    bf16x8 b[3][3];
    for (int p = 0; p < 3; ++p) for (int m = 0; m < 3; ++m) for (int q = 0; q < 8; ++q) b[p][m][q] = (__bf16)in[(lane * 8 + q + 97 * (3 * p + m)) & 8191];
    f32x16 c[2][3];
    for (int rb = 0; rb < 2; ++rb) for (int m = 0; m < 3; ++m) for (int r = 0; r < 16; ++r) c[rb][m][r] = 0.f;
    for (int ks = 0; ks < ksteps16; ++ks) {
      const int kk = ks & 7;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        bf16x8 a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(lds + p * 16384 + (rb * 32 + (lane & 31)) * 256 + kk * 32 + (lane >> 5) * 16);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][m], c[rb][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][m], c[rb][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][m], c[rb][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][m], c[rb][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][m], c[rb][m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 3; ++m) c[rb][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][m], c[rb][m], 0, 0, 0);
      }
    }
    for (int rb = 0; rb < 2; ++rb) for (int m = 0; m < 3; ++m) s += c[rb][m][0] + c[rb][m][9];
  } else {
    bf16x8 b[2][3][3];
    for (int nb = 0; nb < 2; ++nb) for (int p = 0; p < 3; ++p) for (int m = 0; m < 3; ++m) for (int q = 0; q < 8; ++q) b[nb][p][m][q] = (__bf16)in[(lane * 8 + q + 97 * (9 * nb + 3 * p + m)) & 8191];
    f32x4 c[4][2][3];
    for (int mb = 0; mb < 4; ++mb) for (int nb = 0; nb < 2; ++nb) for (int m = 0; m < 3; ++m) c[mb][nb][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksteps16 / 2; ++ks) {
      const int kk = ks & 3;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        bf16x8 a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(lds + p * 16384 + (mb * 16 + (lane & 15)) * 256 + kk * 64 + (lane >> 4) * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[nb][0][m], c[mb][nb][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nb][1][m], c[mb][nb][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][2][m], c[mb][nb][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[nb][0][m], c[mb][nb][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][1][m], c[mb][nb][m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 3; ++m) c[mb][nb][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[nb][0][m], c[mb][nb][m], 0, 0, 0);
        }
      }
    }
    for (int mb = 0; mb < 4; ++mb) for (int nb = 0; nb < 2; ++nb) for (int m = 0; m < 3; ++m) s += c[mb][nb][m][0] + c[mb][nb][m][3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678f) sink[0] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const float* din, float* dsink, unsigned long long* dclk, int ksteps16, int launches) {
  const int grid = 512;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int i = 0; i < launches / 4; ++i) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 49152, 0, din, dsink, dclk, ksteps16);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 49152, 0, din, dsink, dclk, ksteps16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * grid);
  hipMemcpy(h.data(), dclk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
  double cyc = 0, ghz = 0;
  for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; }
  cyc /= grid; ghz /= grid;
  const double flop = (double)launches * grid * 4 * (double)ksteps16 * 36.0 * 32768.0;
  printf("%s: %8.3f ms per launch, %7.1f TFLOP/s (bf16 MFMA), %9.0f cycles in the loop (%.2f per 32x32x16-equivalent MFMA), in-kernel clock %.2f GHz\n",
         SHAPE == 0 ? "32x32x16" : "16x16x32", ms / launches, flop / (ms * 1e-3) / 1e12, cyc, cyc / (ksteps16 * 36.0), ghz);
}

int main() {
  float* din; float* dsink; unsigned long long* dclk;
  hipMalloc(&din, 8192 * 4); hipMalloc(&dsink, 64); hipMalloc(&dclk, 8 * 2 * 512);
  std::vector<float> h(8192);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(din, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  const int ksteps16 = 8 * 400;      // 400 "layers" of K = 128 per launch
  for (int rep = 0; rep < 3; ++rep) {
    run<0>(din, dsink, dclk, ksteps16, 60);
    run<1>(din, dsink, dclk, ksteps16, 60);
  }
  return 0;
}
