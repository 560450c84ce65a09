// GPU micro-experiment (round 4): how much of a wave's VALU / LDS work hides behind ANOTHER wave's bf16 MFMA stream on the same
// SIMD?  The layer chain and the weight gradient run ~2x their matrix-pipe time with two waves per SIMD whose phases (GEMM;
// hops + epilogue) are of equal length: in exact anti-phase the non-MFMA phase of one wave would have to run in the shadow of
// the other's MFMAs.  One workgroup of 8 waves per CU (LDS-limited): waves 0..3 = "gemm" role (back-to-back
// v_mfma_f32_32x32x16_bf16, NA independent accumulator chains, one ds_read_b128 per RD MFMAs), waves 4..7 = "work" role (a
// hop / epilogue-like mix: ds_read_b128 gathers, fma, bf16 splits, ds_write).  Each role is timed alone and beside the other
// (s_memtime, cycles per iteration).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/work_beside_mfma.hip -o /tmp/wbm && /tmp/wbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit 0: gemm waves active, bit 1: work waves active.  WORK: 0 = hop-like (gather + fma), 1 = epilogue-like (split + stores),
// 2 = pure VALU fma chain (no LDS)
template <int NA, int RD, int WORK>
__global__ void __launch_bounds__(512) k(const float* __restrict__ in, float* __restrict__ sink, unsigned long long* __restrict__ cyc, int mode, int iters, int prio, int swap) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = in[i & 4095];
  __syncthreads();
  const bool gemm = swap ? wave >= 4 : wave < 4;      // swap: the work role on the OLDER half of the workgroup
  const int wq = wave & 3;
  if (gemm ? !(mode & 1) : !(mode & 2)) return;
  if (!gemm) { if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3); }
  const unsigned long long T = (unsigned long long)iters;      // run for T cycles, count iterations
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int done = 0;
  if (gemm) {
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)in[lane * 8 + q]; b[q] = (__bf16)in[512 + lane * 8 + q]; }
    f32x16 c[NA];
    for (int j = 0; j < NA; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    const char* base = lds + (wq * 4096 + lane * 16);
    for (int i = 0; __builtin_amdgcn_s_memtime() - t0 < T; ++i, ++done) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        if (RD > 0 && (j % RD) == 0) a = *reinterpret_cast<const bf16x8*>(base + ((i * NA + j) & 3) * 1024);
        c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int j = 0; j < NA; ++j) s += c[j][0] + c[j][7];
    if (s == 12345.f) sink[0] = s;
  } else {
    float* Z = reinterpret_cast<float*>(lds + 16384);        // [64 rows][64 floats] fp32 slab
    char* P = lds + 16384 + 16384;                            // plane area
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int cg = lane & 15, r0 = wq * 16 + (lane >> 4) * 4;
    for (int i = 0; __builtin_amdgcn_s_memtime() - t0 < T; ++i, ++done) {
      if (WORK == 0) {        // a hop on 4 rows x 4 columns: 4 neighbours per row
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (r0 + u) & 63;
          f32x4 z[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) z[kk] = *reinterpret_cast<const f32x4*>(Z + ((row + kk * 5 + i) & 63) * 64 + 4 * cg);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(0.25f, z[kk][q], acc[q]);
        }
      } else if (WORK == 1) {  // epilogue: read 4 row pieces, split into 3 bf16 planes, store
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (r0 + u) & 63;
          const f32x4 v = *reinterpret_cast<const f32x4*>(Z + row * 64 + 4 * cg);
          unsigned hh[2], mm[2], ll[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float x0 = v[2 * q] + acc[0], x1 = v[2 * q + 1] + acc[1];
            unsigned h, m, l;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
            const float r0f = x0 - __uint_as_float(h << 16), r1f = x1 - __uint_as_float(h & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0f), "v"(r1f));
            const float s0 = r0f - __uint_as_float(m << 16), s1 = r1f - __uint_as_float(m & 0xffff0000u);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(s0), "v"(s1));
            hh[q] = h; mm[q] = m; ll[q] = l;
          }
          *reinterpret_cast<uint2*>(P + row * 128 + cg * 8) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(P + 8192 + row * 128 + cg * 8) = make_uint2(mm[0], mm[1]);
          *reinterpret_cast<uint2*>(P + 16384 + row * 128 + cg * 8) = make_uint2(ll[0], ll[1]);
          acc[0] += __uint_as_float(ll[0] << 16) * 1e-30f;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaf(acc[q], 0.999f, 0.001f);
      }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[1] = acc[0];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; cyc[2048 + blockIdx.x * 8 + wave] = done; }
}

template <int NA, int RD, int WORK>
static void run(const char* what, const float* in, float* sink, unsigned long long* cyc, int T, int prio, int swap) {
  double res[4][2] = {};
  static unsigned long long h[2 * 256 * 8];
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, sizeof(h));
      hipLaunchKernelGGL((k<NA, RD, WORK>), dim3(256), dim3(512), 100 * 1024, 0, in, sink, cyc, mode, T, prio, swap);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double g = 0, w = 0, gn = 0, wn = 0;
    for (int b = 0; b < 256; ++b) for (int v = 0; v < 8; ++v) {
      const bool isg = swap ? v >= 4 : v < 4;
      (isg ? g : w) += h[b * 8 + v]; (isg ? gn : wn) += h[2048 + b * 8 + v];
    }
    res[mode][0] = gn > 0 ? g / gn : 0; res[mode][1] = wn > 0 ? w / wn : 0;      // cycles per iteration
  }
  const double mf = NA * 32.0;
  printf("%-50s prio %d %s | gemm: alone %6.1f cyc/iter (pipe %3.0f %%), beside %6.1f (%3.0f %%) | work: alone %6.1f, beside %6.1f (x%.2f)\n",
         what, prio, swap ? "work=older" : "gemm=older", res[1][0], 100 * mf / res[1][0], res[3][0], 100 * mf / res[3][0], res[2][1], res[3][1], res[3][1] / res[2][1]);
}

int main() {
  float* in; float* sink; unsigned long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, 2 * 256 * 8 * 8);
  float hbuf[4096];
  srand(1);
  for (int i = 0; i < 4096; ++i) hbuf[i] = (rand() % 2001 - 1000) / 1000.0f;
  hipMemcpy(in, hbuf, sizeof(hbuf), hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<6, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  printf("one 8-wave workgroup per CU: waves 0-3 stream bf16 MFMAs (6 accumulator chains per iteration), waves 4-7 do hop / epilogue-like work;\n"
         "cycles per iteration of each role, alone and beside the other (s_memtime, mean over 256 CUs x 4 waves)\n");
  const int T = 2000000;      // cycles per launch and role
  for (int swap = 0; swap < 2; ++swap)
    for (int prio = 0; prio <= 3; prio += (prio == 0 ? 1 : 2)) {
      run<6, 0, 2>("MFMA from registers | pure VALU fma chain", in, sink, cyc, T, prio, swap);
      run<6, 2, 0>("MFMA + b128 read per 2 | hop (gathers + fma)", in, sink, cyc, T, prio, swap);
      run<6, 2, 1>("MFMA + b128 read per 2 | epilogue (split + stores)", in, sink, cyc, T, prio, swap);
    }
  return 0;
}
