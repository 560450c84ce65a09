// Weight gradient of TAGConv on the bf16 matrix pipe, fp32-accurate (bf16x6, dss2_common.hpp: split3):
//   dW_m = (P^m G)^T X,  db = colsum(G)   -- the same contract, slab layout and fixed-order reduction as dss2_wgrad.hip.
//
// The contraction runs over the ROWS of a tile, which is the slow dimension of the row-major slabs, while
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane.  So every element is split ONCE, where it is produced (the
// staging pass for G and X, the propagation pass for P G and P^2 G), and stored as three bf16 planes in a TRANSPOSED image
// [plane][column][64 rows]: a wave's operand fragment is then one ds_read_b128 per plane, and the MFMA phase has no VALU
// work at all (the first bf16x6 attempt split per wave and step: VALU-bound, slower than fp32 -- profiles/
// r02_chain_experiments.txt).  A thread owns a pair of rows of four columns, so a transposed store is one ds_write_b32 per
// column and plane.  Rows of a column are swizzled by 16-byte chunk (chunk ^ (column & 7)): conflict-free b128 reads
// without padding, which the 160 KB budget does not have:
//     fp32 G / P G (propagation ping-pong)   2 x 64 x 68 x 4     34.0 KB
//     transposed planes of G, P G, P^2 G     3 x 3 x 64 x 128 B  72.0 KB
//     transposed planes of X (128 columns)   3 x 128 x 128 B     48.0 KB
//     row scales, ELL slice (D <= 8)                              5.0 KB
// One workgroup of 8 waves per CU, persistent over tiles; a workgroup owns 128 output columns in two passes of 64 (the X
// planes are staged once per tile); wave w multiplies input block (w & 3) with output block (w >> 2) of the pass: 72 MFMAs
// per tile, pass and wave, against 768 fp32 MFMAs of twice the length per tile and wave in dss2_wgrad.hip.
// Two-row-block tiles (64 rows), K <= 2, ELL slices; everything else runs the fp32 kernel.
// Built without packed fp32 VALU ops like dss2_gemm_chain16.hip (see there).
#include <stdlib.h>

#include "dss2_wgrad_batch.hpp"

namespace dss2 {

constexpr int W16_TM = 64, W16_ZC = 64, W16_XW = 128, W16_NT = 512, W16_LDZF = 64, W16_DMAX = 8;      // (LDZF = 64: a row is one 256-byte bank row -- the b128 lane groups of the row-piece stores and gathers then cover all 16 slots whatever rows they touch; 68 cost a third of the LDS cycles in conflicts)

// byte offset of (column, row) inside one plane of a transposed image: 128 B per column, 16-byte chunks of 8 rows swizzled
// (the swizzle key ((col >> 1) ^ (col >> 4)) & 7 makes BOTH access patterns conflict-free: the transposed stores of a wave --
//  16 column groups four columns apart x 4 row pairs of one chunk -- and the b128 operand reads of 16 consecutive columns)
__device__ __forceinline__ int tp_key(int col) { return ((col >> 1) ^ (col >> 4)) & 7; }
__device__ __forceinline__ int tp_off(int col, int row) { return col * 128 + ((((row >> 3) ^ tp_key(col)) << 4) | ((row & 7) << 1)); }

// rows (2 rp, 2 rp + 1) x columns (c0 .. c0+3), c0 a multiple of 4 -> the three planes of a transposed image with NCOLS
// columns.  off0 = tp_off(c0, 2 rp), off2 = tp_off(c0 + 2, 2 rp); columns c0 + 1 / c0 + 3 share their keys (+128 bytes),
// so a unit costs two address registers and the rest are immediate offsets
template <int NCOLS>
__device__ __forceinline__ void store_planes(char* img, int off0, int off2, const f32x4 v0, const f32x4 v1) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, m, l;
    split3_pair(v0[q], v1[q], h, m, l);          // {row 2 rp, row 2 rp + 1} of column c0 + q: exactly the stored dword
    char* dst = img + ((q & 2) ? off2 : off0) + (q & 1) * 128;
    *reinterpret_cast<uint32_t*>(dst) = h;
    *reinterpret_cast<uint32_t*>(dst + NCOLS * 128) = m;
    *reinterpret_cast<uint32_t*>(dst + 2 * NCOLS * 128) = l;
  }
}

// NP: passes of 64 output columns per workgroup (the X planes are staged once for all of them).  RS2: layers with the extra
// scaled column sums (rowscale2, the folded first layer): 12 more accumulator registers per pass -- <3, 2, true> sits at exactly
// 256 VGPRs.
template <int NMAT, int NP, bool RS2>
__global__ void __launch_bounds__(W16_NT) wgrad16_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb) {
  constexpr int TM = W16_TM, ZC = W16_ZC, XW = W16_XW, NT = W16_NT, LDZF = W16_LDZF;
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[blockIdx.z] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[blockIdx.z] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[blockIdx.z] : p.slab;
  const float* __restrict__ rs2 = RS2 ? (wb.n > 0 ? wb.rowscale2[blockIdx.z] : p.rowscale2) : nullptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf0 = smem;
  float* Zf1 = Zf0 + TM * LDZF;
  char* ZT = reinterpret_cast<char*>(Zf1 + TM * LDZF);          // [NMAT][3 planes][ZC columns][128 B]
  char* XT = ZT + NMAT * 3 * ZC * 128;                            // [3 planes][XW columns][128 B]
  int2* ell = reinterpret_cast<int2*>(XT + 3 * XW * 128);        // [D][TM]
  const int D = p.ell_width;

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ibw = wave & 3, obh = wave >> 2;
  const int obg = blockIdx.y / nibg, ibg = blockIdx.y - obg * nibg;
  const int gcol0 = obg * (NP * ZC), xcol0 = ibg * XW;          // a workgroup owns NP x 64 output columns, one pass each
  const bool in_active = (xcol0 + ibw * 32) < p.hin;

  // staging units: a thread owns rows (2 rp, 2 rp + 1) of four columns.  G slab of a pass: 32 x 16 units, one per thread;
  // X slab: 32 x 32 units, two per thread
  const int g_cg = tid & 15, g_rp = tid >> 4;
  const int x_cg = tid & 31, x_rp0 = tid >> 5;          // second unit: x_rp0 + 16
  const int g_off0 = tp_off(4 * g_cg, 2 * g_rp), g_off2 = tp_off(4 * g_cg + 2, 2 * g_rp);
  const int x_off0[2] = {tp_off(4 * x_cg, 2 * x_rp0), tp_off(4 * x_cg, 2 * x_rp0 + 32)};
  const int x_off2[2] = {tp_off(4 * x_cg + 2, 2 * x_rp0), tp_off(4 * x_cg + 2, 2 * x_rp0 + 32)};
  // global rows of a tile through ONE uniform base per tile and 32-bit per-thread offsets (a tile spans < 2^31 bytes)
  const uint32_t g_goff = (uint32_t)((2 * g_rp) * p.ldg + 4 * g_cg) * 4u, x_goff = (uint32_t)((2 * x_rp0) * p.ldx + 4 * x_cg) * 4u;

  f32x16 acc[NP][NMAT];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps)
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ps][m][r] = 0.f;
  f32x4 bsum[NP];                             // running column sums of this thread's four G columns (its rows, all its tiles)
  f32x4 bs2[NP][RS2 ? NMAT : 1];              // ... scaled by rowscale2[row][m] (the folded layer's bias terms)
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    bsum[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < (RS2 ? NMAT : 1); ++m) bs2[ps][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // Register prefetch, staggered so that at most 16 prefetch registers are live inside an MFMA phase (96 accumulator
  // registers + operands leave no more under the 256-VGPR budget of an 8-wave workgroup): the next tile's X rows are
  // requested before the last pass's propagation, its first G slice after the last MFMA phase (covered by the X split of
  // the next tile), the G slice of pass ps + 1 right after pass ps is staged.
  f32x4 pg[2], px[4];
  auto load_g = [&](int tile, int ps) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const char* base = reinterpret_cast<const char*>(Gp + (size_t)ts * p.ldg + gcol0 + ps * ZC);      // uniform
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 2 * g_rp + u, c = gcol0 + ps * ZC + 4 * g_cg;
      pg[u] = (r < R && c < p.hout) ? *reinterpret_cast<const f32x4*>(base + g_goff + (uint32_t)(u * p.ldg) * 4u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto load_x = [&](int tile) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const char* base = reinterpret_cast<const char*>(Xp + (size_t)ts * p.ldx + xcol0);      // uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = 2 * (x_rp0 + 16 * i) + u, c = xcol0 + 4 * x_cg;
        px[2 * i + u] = (r < R && c < p.hin) ? *reinterpret_cast<const f32x4*>(base + x_goff + (uint32_t)((32 * i + u) * p.ldx) * 4u)
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  // one propagation hop: Zd = P Zs on this thread's unit; fp32 copy for the next hop (if any) and the transposed planes
  auto prop = [&](const float* Zs, float* Zd, char* img) {
    // four ELL entries of a row first, then their four gathers: independent entry -> row -> fma chains in flight (the
    // dependent pairs of a k-by-k loop left this phase latency-bound: 35 of the kernel's 98 us)
    f32x4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 2 * g_rp + u;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < D; k0 += 4) {
        int2 en[4];
        f32x4 z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) en[k] = k0 + k < D ? ell[(k0 + k) * TM + row] : make_int2(row, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = *reinterpret_cast<const f32x4*>(Zs + en[k].x * LDZF + 4 * g_cg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float w = __int_as_float(en[k].y);
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = fmaf(w, z[k][q], a[q]);
        }
      }
      s[u] = a;
      if (Zd) *reinterpret_cast<f32x4*>(Zd + row * LDZF + 4 * g_cg) = a;
    }
    store_planes<ZC>(img, g_off0, g_off2, s[0], s[1]);
  };

  if ((int)blockIdx.x < p.ntiles) { load_x(blockIdx.x); load_g(blockIdx.x, 0); }
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const int next = tile + gridDim.x;
    // ---- the tile's X planes (shared by all passes) and ELL slice
#pragma unroll
    for (int i = 0; i < 2; ++i) store_planes<XW>(XT, x_off0[i], x_off2[i], px[2 * i], px[2 * i + 1]);
    {
      const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
      for (int idx = tid; idx < D * TM; idx += NT) ell[idx] = src[idx];
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      // ---- stage this pass's G columns: fp32 (first hop's input) and planes; bias partial sums
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(Zf0 + (2 * g_rp + u) * LDZF + 4 * g_cg) = pg[u];
      store_planes<ZC>(ZT, g_off0, g_off2, pg[0], pg[1]);
      bsum[ps] += pg[0] + pg[1];
      if constexpr (RS2) {
        if (rs2) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int r = 2 * g_rp + u;
            if (r < R) {
              const f32x4 d = *reinterpret_cast<const f32x4*>(rs2 + (size_t)(ts + r) * 4);
#pragma unroll
              for (int m = 0; m < NMAT; ++m) bs2[ps][m] += pg[u] * d[m];
            }
          }
        }
      }
      __syncthreads();
      const bool last_pass = ps == NP - 1;      // (a pass beyond hout stages zeros and stores nothing)
      if (!last_pass) load_g(tile, ps + 1);
      else if (next < p.ntiles) load_x(next);
      // ---- P G, P^2 G
      if (NMAT > 1) {
        prop(Zf0, NMAT > 2 ? Zf1 : nullptr, ZT + 3 * ZC * 128);
        if (NMAT > 2) {
          __syncthreads();
          prop(Zf1, nullptr, ZT + 2 * 3 * ZC * 128);
        }
        __syncthreads();
      }
      // ---- MFMA phase: 4 steps of 16 rows, six bf16 MFMAs per matrix and step, operands straight from the planes
      if (in_active && gcol0 + ps * ZC + obh * 32 < p.hout) {
        const int nsteps = (R + 15) >> 4;
        const int zc = obh * 32 + c32, xc = ibw * 32 + c32;
        const int zkey = tp_key(zc), xkey = tp_key(xc);
        for (int ks = 0; ks < nsteps; ++ks) {
          const int choff = xc * 128 + (((2 * ks + half) ^ xkey) << 4);
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(XT + choff);
          const bf16x8 bm = *reinterpret_cast<const bf16x8*>(XT + XW * 128 + choff);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(XT + 2 * XW * 128 + choff);
          const int zoff = zc * 128 + (((2 * ks + half) ^ zkey) << 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) {
            const char* zi = ZT + m * 3 * ZC * 128 + zoff;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zi);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(zi + ZC * 128);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(zi + 2 * ZC * 128);
            f32x16 c = acc[ps][m];        // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            acc[ps][m] = c;
          }
        }
      }
      if (last_pass && next < p.ntiles) load_g(next, 0);
      __syncthreads();          // the Z planes and fp32 slabs are free for the next pass / tile
    }
  }

  // ---- one slab per workgroup column blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)blockIdx.x * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  if (in_active) {
    const int i = xcol0 + ibw * 32 + c32;
    if (i < p.hin) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = gcol0 + ps * ZC + obh * 32 + acc_row(r, half);
            if (o < p.hout) out[((size_t)m * p.hout + o) * p.hin + i] = acc[ps][m][r];
          }
    }
  }
  if (ibg == 0) {   // (uniform) column sums: the 32 threads that share a column group meet in LDS, fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [1 + NMAT][NT], one pass at a time
    const int nsum = rs2 ? 1 + NMAT : 1;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      __syncthreads();
      red[tid] = bsum[ps];
      if constexpr (RS2) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) red[(1 + m) * NT + tid] = bs2[ps][m];
      }
      __syncthreads();
      for (int j = tid; j < nsum * ZC; j += NT) {
        const int which = j / ZC, col = j - which * ZC;
        float s = 0.f;
        for (int rp = 0; rp < 32; ++rp) s += red[which * NT + rp * 16 + (col >> 2)][col & 3];
        const int o = gcol0 + ps * ZC + col;
        if (o < p.hout) out[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// 32-row tiles, TWO workgroups per CU (round 4).  The 64-row kernel above is one 8-wave workgroup per CU whose phases --
// staging, two hops, MFMA -- follow each other between barriers: the matrix pipe is busy 33 % of the time
// (profiles/r03_final_pmc_wgrad_batched.txt).  Here a workgroup is four waves that own 64 output x 128 input columns of
// every matrix on a 32-row tile; everything it keeps in LDS is half as tall:
//     fp32 G / P G (propagation ping-pong)   2 x 32 x 64 x 4     16.0 KB
//     transposed planes of G, P G, P^2 G     3 x 3 x 64 x 64 B   36.0 KB
//     transposed planes of X (128 columns)   3 x 128 x 64 B      24.0 KB
//     ELL slice (D <= 8)                                           2.0 KB
// = 78 KB, so two workgroups share a CU and one's staging / hops run beside the other's MFMA phase.  The other 64 output
// columns belong to the workgroup at blockIdx.y + 1, which stages the same X tile again (+25 % split work).  Wave w multiplies
// input block w with both output blocks: 72 MFMAs per tile and wave, the X fragment read once for six accumulator blocks.
// Transposed image: 64 bytes per column and plane (32 rows), 16-byte chunks of 8 rows XOR-swizzled with (column / 8) mod 4:
// the b128 operand reads of a lane group (16 columns) cover all 64 banks.  The transposed b32 stores of one instruction come
// from 16 column groups x 2 row pairs and all write the same column parity (= 16 of the 32 banks): the bank is
// 16 parity + 4 (chunk ^ key) + (row pair mod 4), the key spreads column-group bits 1-2, and ODD column groups own the row pair
// rp ^ 2, which spreads the last two bits: 16 distinct banks, 2-way conflicts (free for a b32 store; in thread order they
// were 4-way -- 150.6 us instead of 132.9 --, 8-way with a 32 x 1 lane shape).  The two lanes that split a row between them
// sit in the same wave, so a wave-wide load still covers whole rows.
constexpr int W16B_TM = 32, W16B_ZC = 64, W16B_XW = 128, W16B_NT = 256, W16B_LDZF = 64;
// (bits 3 and 4 of the column, swapped: the transposed stores of a half wave -- 16 column groups x 2 row pairs here, 4 x 8 in the
//  pipelined experiment profiles/experiments/r04_wgrad16c_pipelined.hip.txt -- then reach 16 banks; with the plain (col >> 3) & 3
//  this kernel took 124.6 us for the three C2 layers, with the swapped bits 120.6)
__device__ __forceinline__ int tpb_key(int col) { return (((col >> 3) & 1) << 1) | ((col >> 4) & 1); }
__device__ __forceinline__ int tpb_off(int col, int row) { return col * 64 + ((((row >> 3) ^ tpb_key(col)) << 4) | ((row & 7) << 1)); }

// rows (2 rp, 2 rp + 1) x columns (c0 .. c0+3), c0 a multiple of 4 (one key for the four columns): off0 = tpb_off(c0, 2 rp)
template <int NCOLS>
__device__ __forceinline__ void store_planes_b(char* img, int off0, const f32x4 v0, const f32x4 v1) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, m, l;
    split3_pair(v0[q], v1[q], h, m, l);
    char* dst = img + off0 + q * 64;
    *reinterpret_cast<uint32_t*>(dst) = h;
    *reinterpret_cast<uint32_t*>(dst + NCOLS * 64) = m;
    *reinterpret_cast<uint32_t*>(dst + 2 * NCOLS * 64) = l;
  }
}

template <int NMAT, bool RS2>
__global__ void __launch_bounds__(W16B_NT, 2) wgrad16b_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb) {
  constexpr int TM = W16B_TM, ZC = W16B_ZC, XW = W16B_XW, NT = W16B_NT, LDZF = W16B_LDZF;
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[blockIdx.z] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[blockIdx.z] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[blockIdx.z] : p.slab;
  const float* __restrict__ rs2 = RS2 ? (wb.n > 0 ? wb.rowscale2[blockIdx.z] : p.rowscale2) : nullptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf0 = smem;
  float* Zf1 = Zf0 + TM * LDZF;
  char* ZT = reinterpret_cast<char*>(Zf1 + TM * LDZF);          // [NMAT][3 planes][ZC columns][64 B]
  char* XT = ZT + NMAT * 3 * ZC * 64;                             // [3 planes][XW columns][64 B]
  int2* ell = reinterpret_cast<int2*>(XT + 3 * XW * 64);         // [D][TM]
  const int D = p.ell_width;

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w multiplies the input blocks 2 (w & 1), 2 (w & 1) + 1 with the output block w >> 1 of every matrix: 6 + 9 operand
  // fragments per k-step for its 36 MFMAs (one input block x both output blocks took 3 + 18: 29 % more LDS read bytes)
  const int xb0 = 2 * (wv & 1), obw = wv >> 1;
  // (Placing the workgroups that walk the same tile-list slice 8 linear ids apart -- same XCD, shared L2 for the X tile they both
  //  stage -- was 20 % SLOWER: such a pair lands on one CU and runs its phases in lockstep, MFMA beside MFMA and staging beside
  //  staging; with the slices n_split ids apart a CU's two workgroups are unrelated and drift out of phase.)
  const int slice = blockIdx.x, ysl = blockIdx.y;
  const int obg = ysl / nibg, ibg = ysl - obg * nibg;
  const int gcol0 = obg * ZC, xcol0 = ibg * XW;
  const bool x_on[2] = {(xcol0 + xb0 * 32) < p.hin, (xcol0 + (xb0 + 1) * 32) < p.hin};
  const bool in_active = x_on[0] && (gcol0 + obw * 32) < p.hout;

  // staging units: rows (2 rp, 2 rp + 1) of four columns; 16 column groups x 16 row pairs, one unit of G and two of X (columns
  // 4 cg and 64 + 4 cg) per thread
  const int cg = tid & 15, rp = (tid >> 4) ^ ((tid & 1) << 1);      // (odd column groups: row pair ^ 2, see the layout note above)
  const int g_off0 = tpb_off(4 * cg, 2 * rp);
  const int x_off0[2] = {tpb_off(4 * cg, 2 * rp), tpb_off(64 + 4 * cg, 2 * rp)};
  const uint32_t g_goff = (uint32_t)((2 * rp) * p.ldg + 4 * cg) * 4u;

  f32x16 acc[2][NMAT];      // [input block of the wave][matrix]
#pragma unroll
  for (int xb = 0; xb < 2; ++xb)
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xb][m][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs2[RS2 ? NMAT : 1];
#pragma unroll
  for (int m = 0; m < (RS2 ? NMAT : 1); ++m) bs2[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 pg[2], px[4], prs[RS2 ? 2 : 1];
  int2 pel;                                      // the tile's ELL entry of this thread (D <= 8: at most 256 entries), one tile ahead like its rows
  auto load_tile = [&](int tile) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const char* gb = reinterpret_cast<const char*>(Gp + (size_t)ts * p.ldg + gcol0);      // uniform
    const char* xb = reinterpret_cast<const char*>(Xp + (size_t)ts * p.ldx + xcol0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 2 * rp + u;
      pg[u] = (r < R && gcol0 + 4 * cg < p.hout) ? *reinterpret_cast<const f32x4*>(gb + g_goff + (uint32_t)(u * p.ldg) * 4u) : f32x4{0.f, 0.f, 0.f, 0.f};
      // X rows beyond the tile's R rows are read from its LAST row instead of being zeroed (finite values; they only ever meet
      // the zero rows of G / P G / P^2 G), columns beyond hin from column group 0 (their products are never stored)
      const uint32_t xro = (uint32_t)((r < R ? r : R - 1) * p.ldx) * 4u;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        px[2 * i + u] = *reinterpret_cast<const f32x4*>(xb + xro + ((xcol0 + 64 * i + 4 * cg < p.hin) ? (uint32_t)(64 * i + 4 * cg) * 4u : 0u));
      if constexpr (RS2) prs[u] = (rs2 && r < R) ? *reinterpret_cast<const f32x4*>(rs2 + (size_t)(ts + r) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    pel = tid < D * TM ? (reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM)[tid] : make_int2(tid & (TM - 1), 0);
  };
  auto prop = [&](const float* Zs, float* Zd, char* img) {
    f32x4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 2 * rp + u;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < D; k0 += 4) {      // (the slice is padded to four entries per row: zero weight, own row)
        int2 en[4];
        f32x4 z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) en[k] = ell[(k0 + k) * TM + row];
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = *reinterpret_cast<const f32x4*>(Zs + en[k].x * LDZF + 4 * cg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float w = __int_as_float(en[k].y);
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = fmaf(w, z[k][q], a[q]);
        }
      }
      s[u] = a;
      if (Zd) *reinterpret_cast<f32x4*>(Zd + row * LDZF + 4 * cg) = a;
    }
    store_planes_b<ZC>(img, g_off0, s[0], s[1]);
  };

  if (slice < p.ntiles) load_tile(slice);
  for (int tile = slice; tile < p.ntiles; tile += p.n_split) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const int next = tile + p.n_split;
    // ---- planes of X and G, the fp32 G slab (first hop's input), the ELL slice, bias partial sums
#pragma unroll
    for (int i = 0; i < 2; ++i) store_planes_b<XW>(XT, x_off0[i], px[2 * i], px[2 * i + 1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(Zf0 + (2 * rp + u) * LDZF + 4 * cg) = pg[u];
    store_planes_b<ZC>(ZT, g_off0, pg[0], pg[1]);
    if (tid < ((D + 3) & ~3) * TM) ell[tid] = pel;
    bsum += pg[0] + pg[1];
    if constexpr (RS2) {
      if (rs2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < NMAT; ++m) bs2[m] += pg[u] * prs[u][m];
      }
    }
    __syncthreads();
    if (next < p.ntiles) load_tile(next);      // in flight across the hops and the MFMA phase
    // ---- P G, P^2 G
    if (NMAT > 1) {
      prop(Zf0, NMAT > 2 ? Zf1 : nullptr, ZT + 3 * ZC * 64);
      if (NMAT > 2) {
        __syncthreads();
        prop(Zf1, nullptr, ZT + 2 * 3 * ZC * 64);
      }
      __syncthreads();
    }
    // ---- MFMA phase: 2 steps of 16 rows; a step's Z fragments (three per matrix) serve both input blocks of the wave
    if (in_active) {
      const int nsteps = (R + 15) >> 4;
      const int zc = obw * 32 + c32;
      const int zkey = tpb_key(zc);
      const int xc0 = xb0 * 32 + c32, xc1 = xc0 + 32;
      const int xkey0 = tpb_key(xc0), xkey1 = tpb_key(xc1);
      for (int ks = 0; ks < nsteps; ++ks) {
        const int ch = 2 * ks + half;
        bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          const int choff = (xb ? xc1 : xc0) * 64 + ((ch ^ (xb ? xkey1 : xkey0)) << 4);
          bh[xb] = *reinterpret_cast<const bf16x8*>(XT + choff);
          bm[xb] = *reinterpret_cast<const bf16x8*>(XT + XW * 64 + choff);
          bl[xb] = *reinterpret_cast<const bf16x8*>(XT + 2 * XW * 64 + choff);
        }
        const int zoff = zc * 64 + ((ch ^ zkey) << 4);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) {
          const char* zi = ZT + m * 3 * ZC * 64 + zoff;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zi);
          const bf16x8 am = *reinterpret_cast<const bf16x8*>(zi + ZC * 64);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(zi + 2 * ZC * 64);
#pragma unroll
          for (int xb = 0; xb < 2; ++xb) {
            if (xb == 1 && !x_on[1]) continue;      // (uniform)
            f32x16 c = acc[xb][m];        // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[xb], c, 0, 0, 0);
            acc[xb][m] = c;
          }
        }
      }
    }
    __syncthreads();          // planes and fp32 slabs are free for the next tile
  }

  // ---- one slab per tile-list slice blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)slice * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  if (in_active) {
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
      const int i = xcol0 + (xb0 + xb) * 32 + c32;
      if (i < p.hin) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = gcol0 + obw * 32 + acc_row(r, half);
            if (o < p.hout) out[((size_t)m * p.hout + o) * p.hin + i] = acc[xb][m][r];
          }
      }
    }
  }
  if (ibg == 0) {   // (uniform) column sums: the 16 threads that share a column group meet in LDS, fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [1 + NMAT][NT]
    const int nsum = rs2 ? 1 + NMAT : 1;
    __syncthreads();
    red[tid] = bsum;
    if constexpr (RS2) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m) red[(1 + m) * NT + tid] = bs2[m];
    }
    __syncthreads();
    for (int j = tid; j < nsum * ZC; j += NT) {
      const int which = j / ZC, col = j - which * ZC;
      float s = 0.f;
      for (int r16 = 0; r16 < 16; ++r16) s += red[which * NT + r16 * 16 + (col >> 2)][col & 3];
      const int o = gcol0 + col;
      if (o < p.hout) out[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Tall tiles (96 / 128 / 160 / 192 rows: graphs of 65 .. 192 nodes), round 4.  The contraction over rows can be cut anywhere;
// only the propagation needs a whole graph.  So P G of the WHOLE tile is kept in fp32 (one hop, gathered from the staged fp32
// G), and everything the MFMAs read is made per CHUNK of 32 rows exactly as in wgrad16b_kernel: transposed planes of G (read
// again from global memory -- L2 --, not from the fp32 image, so that image's space can be the planes' space), of P G (the
// chunk's rows of the fp32 image) and of P^2 G (one more hop, gathered from the P G image, never stored in fp32), and of the
// chunk's X rows.
//     fp32 P G of the tile                    32 NRB x 64 x 4              24 / 48 KB  (96 / 192 rows)
//     fp32 G of the tile, then the planes     max(32 NRB x 64 x 4, 60 KB)  60 KB
//     ELL slice (padded to 4 or 8 entries)    32 NRB x 8 x 8               6 / 12 KB
// One 8-wave workgroup per CU.  Waves 0-3 split the chunk's G and P G rows and the X columns 0..63, waves 4-7 gather and split
// P^2 G and split the X columns 64..127 (about the same instruction count); in the MFMA phase wave w multiplies input block
// w & 3 with output block w >> 2 of every matrix (18 MFMAs per 16 rows).  Chunks beyond the tile's last real row are skipped,
// the last chunk runs one k-step if that covers it (70-node graphs: 5 k-steps instead of 6).
constexpr int W16T_NT = 512;

template <int NRB, int NMAT, bool RS2, bool DB>
__global__ void __launch_bounds__(W16T_NT) wgrad16t_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb) {
  constexpr int TR = 32 * NRB, ZC = W16B_ZC, XW = W16B_XW, NT = W16T_NT, LDZF = W16B_LDZF;
  constexpr int PLANES = NMAT * 3 * ZC * 64 + 3 * XW * 64;
  constexpr int PBUF = TR * LDZF * 4 > PLANES ? TR * LDZF * 4 : PLANES;      // one set of planes; the tile's fp32 G fits in it
  constexpr int UBYTES = DB ? 2 * PBUF : PBUF;
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[blockIdx.z] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[blockIdx.z] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[blockIdx.z] : p.slab;
  const float* __restrict__ rs2 = RS2 ? (wb.n > 0 ? wb.rowscale2[blockIdx.z] : p.rowscale2) : nullptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf1 = smem;                                              // [TR][LDZF]  P G
  char* const U = reinterpret_cast<char*>(Zf1 + TR * LDZF);       // one or (DB) two sets of planes [NMAT][3 planes][ZC columns][64 B] + [3 planes][XW columns][64 B];
                                                                  // the tile's fp32 G [TR][LDZF] lives in the set the next chunk writes (dead after the first hop)
  int2* ell = reinterpret_cast<int2*>(U + UBYTES);                // [Dp][TR]
  int pb = 0;                                                     // (DB) the set the next chunk writes
  const int D = p.ell_width, Dp = (D + 3) & ~3;

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ibw = wave & 3, obh = wave >> 2, role = wave >> 2;
  const int slice = blockIdx.x, ysl = blockIdx.y;
  const int obg = ysl / nibg, ibg = ysl - obg * nibg;
  const int gcol0 = obg * ZC, xcol0 = ibg * XW;
  const bool in_active = (xcol0 + ibw * 32) < p.hin && (gcol0 + obh * 32) < p.hout;

  // whole-tile units: one row of four columns per row block (a wave covers four whole rows of the fp32 image)
  const int q16 = tid & 15, r32 = tid >> 4;
  const bool gcol_ok = gcol0 + 4 * q16 < p.hout;
  // chunk units: rows (2 rp, 2 rp + 1) of four columns, the thread map of wgrad16b_kernel within each half of the workgroup
  const int t8 = tid & 255, cg = t8 & 15, rp = (t8 >> 4) ^ ((t8 & 1) << 1);
  const int z_off = tpb_off(4 * cg, 2 * rp), x_off = tpb_off(64 * role + 4 * cg, 2 * rp);
  const uint32_t xcb = (xcol0 + 64 * role + 4 * cg < p.hin) ? (uint32_t)(64 * role + 4 * cg) * 4u : 0u;      // (columns beyond hin: group 0, never stored)
  const bool zcol_ok = gcol0 + 4 * cg < p.hout;

  f32x16 acc[NMAT];
#pragma unroll
  for (int m = 0; m < NMAT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs2[RS2 ? NMAT : 1];
#pragma unroll
  for (int m = 0; m < (RS2 ? NMAT : 1); ++m) bs2[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Everything a tile's prologue reads from global memory is requested one tile ahead (one workgroup per CU: nobody else
  // would cover the latency): its G rows, its ELL slice, the folded layer's row scales.
  constexpr int NEL = (W16_DMAX * TR + NT - 1) / NT;
  constexpr bool PRS = RS2 && NRB <= 4;      // (160- / 192-row tiles have no registers left for the row scales: read in place)
  // (round 5: every prefetch load is UNCONDITIONAL, from clamped rows / columns / entries, and masked where it is consumed -- the
  //  `cond ? load : zero` forms made the compiler join the two values right behind the load, an s_waitcnt vmcnt inside the prefetch;
  //  the row scales are 12-byte loads -- the dead fourth component's register was reused while the load was in flight -- and
  //  tile_start is read one tile ahead of the loads it addresses: profiles/experiments/r05_wgrad16h_phase_stamps.txt)
  typedef float f32x3_t __attribute__((ext_vector_type(3)));
  f32x4 pgw[NRB], px[2], pgc[2];
  f32x3_t prs[PRS ? NRB : 1];
  int2 pel[NEL];
  const uint32_t gw_col = gcol_ok ? (uint32_t)(4 * q16) * 4u : 0u, gc_col = zcol_ok ? (uint32_t)(4 * cg) * 4u : 0u;
  const float* __restrict__ rsb = RS2 ? (rs2 ? rs2 : Gp) : nullptr;      // (layers without row scales: any readable rows; never used)
  auto load_tile_g = [&](int tile, int ts, int R) {
    const char* gb = reinterpret_cast<const char*>(Gp + (size_t)ts * p.ldg + gcol0);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const int r = r32 + 32 * rb;
      const int rr = r < R ? r : R - 1;
      pgw[rb] = *reinterpret_cast<const f32x4*>(gb + (uint32_t)(rr * p.ldg) * 4u + gw_col);
      if constexpr (PRS) prs[rb] = *reinterpret_cast<const f32x3_t*>(rsb + (size_t)(ts + rr) * 4);
    }
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TR;
#pragma unroll
    for (int j = 0; j < NEL; ++j) {
      const int idx = tid + j * NT;
      pel[j] = src[idx < D * TR ? idx : 0];      // (entries beyond the slice: replaced by padding where they are stored)
    }
  };
  auto load_chunk = [&](int ts, int R, int c) {
    const char* xb = reinterpret_cast<const char*>(Xp + (size_t)ts * p.ldx + xcol0);
    const char* gb = reinterpret_cast<const char*>(Gp + (size_t)ts * p.ldg + gcol0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 32 * c + 2 * rp + u;
      const int rr = r < R ? r : R - 1;
      // X rows beyond the tile's R rows are read from its last row (finite; they only meet zero rows of G / P G / P^2 G)
      px[u] = *reinterpret_cast<const f32x4*>(xb + (uint32_t)(rr * p.ldx) * 4u + xcb);
      if (role == 0) pgc[u] = *reinterpret_cast<const f32x4*>(gb + (uint32_t)(rr * p.ldg) * 4u + gc_col);
    }
  };
  // one row of P Zs (four columns at c4) for a real row; the slice is padded to four entries per row (zero weight, own row)
  auto hop_row = [&](const float* Zs, int row, int c4) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Dp; k0 += 4) {
      int2 en[4];
      f32x4 z[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) en[k] = ell[(k0 + k) * TR + row];
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] = *reinterpret_cast<const f32x4*>(Zs + en[k].x * LDZF + c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = __int_as_float(en[k].y);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = fmaf(w, z[k][q], a[q]);
      }
    }
    return a;
  };

  int ts = 0, R = 0, ts_n = 0, R_n = 0;      // the current tile's rows and the next one's (read a tile ahead)
  if (slice < p.ntiles) {
    ts = p.tile_start[slice]; R = p.tile_start[slice + 1] - ts;
    load_tile_g(slice, ts, R); load_chunk(ts, R, 0);
    if (slice + p.n_split < p.ntiles) { ts_n = p.tile_start[slice + p.n_split]; R_n = p.tile_start[slice + p.n_split + 1] - ts_n; }
  }
  for (int tile = slice; tile < p.ntiles; tile += p.n_split) {
    const int nch = (R + 31) >> 5;
    const int next = tile + p.n_split;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      if (!(gcol_ok && r32 + 32 * rb < R)) pgw[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- fp32 G of the tile (first hop's input), the ELL slice, bias partial sums
    float* Zf0 = reinterpret_cast<float*>(U + pb * PBUF);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if (rb < nch) *reinterpret_cast<f32x4*>(Zf0 + (r32 + 32 * rb) * LDZF + 4 * q16) = pgw[rb];
      bsum += pgw[rb];
    }
    if constexpr (RS2) {
      if (rs2) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRS) d = f32x4{prs[rb][0], prs[rb][1], prs[rb][2], 0.f};
          else if (r32 + 32 * rb < R) d = *reinterpret_cast<const f32x4*>(rs2 + (size_t)(ts + r32 + 32 * rb) * 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) bs2[m] += pgw[rb] * d[m];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NEL; ++j) {
      const int idx = tid + j * NT;
      if (idx < Dp * TR) ell[idx] = idx < D * TR ? pel[j] : make_int2(idx % TR, 0);      // (padding entries: own row, zero weight)
    }
    __syncthreads();
    if (next < p.ntiles) load_tile_g(next, ts_n, R_n);      // in flight for the whole tile
    // ---- P G of the tile
    if (NMAT > 1) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        if (rb < nch) {
          const int row = r32 + 32 * rb;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (row < R) a = hop_row(Zf0, row, 4 * q16);
          *reinterpret_cast<f32x4*>(Zf1 + row * LDZF + 4 * q16) = a;
        }
      }
      __syncthreads();          // Zf0 is dead: the planes take its place
    }
    for (int c = 0; c < nch; ++c) {
      // ---- planes of the chunk
      char* ZT = U + pb * PBUF;
      char* XT = ZT + NMAT * 3 * ZC * 64;
      if (32 * c + 8 * (wave & 3) < ((R + 15) & ~15)) {      // (a wave's units are eight rows; rows beyond the last k-step are not read)
        store_planes_b<XW>(XT, x_off, px[0], px[1]);
        if (role == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (!(zcol_ok && 32 * c + 2 * rp + u < R)) pgc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          store_planes_b<ZC>(ZT, z_off, pgc[0], pgc[1]);
          if (NMAT > 1) {
            const float* src = Zf1 + (32 * c + 2 * rp) * LDZF + 4 * cg;
            store_planes_b<ZC>(ZT + 3 * ZC * 64, z_off, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + LDZF));
          }
        } else if (NMAT > 2) {
          f32x4 s[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = 32 * c + 2 * rp + u;
            s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < R) s[u] = hop_row(Zf1, row, 4 * cg);
          }
          store_planes_b<ZC>(ZT + 2 * 3 * ZC * 64, z_off, s[0], s[1]);
        }
      }
      __syncthreads();
      {      // ONE call site (two made the compiler load into temporaries and join them behind an s_waitcnt vmcnt(0))
        const bool same = c + 1 < nch;
        if (same || next < p.ntiles) load_chunk(same ? ts : ts_n, same ? R : R_n, same ? c + 1 : 0);
      }
      // ---- MFMA phase: up to 2 steps of 16 rows
      if (in_active) {
        const int left = R - 32 * c;
        const int nsteps = left > 16 ? 2 : 1;
        const int xc = ibw * 32 + c32, zc = obh * 32 + c32;
        const int xkey = tpb_key(xc), zkey = tpb_key(zc);
        for (int ks = 0; ks < nsteps; ++ks) {
          const int ch = 2 * ks + half;
          const int choff = xc * 64 + ((ch ^ xkey) << 4);
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(XT + choff);
          const bf16x8 bm = *reinterpret_cast<const bf16x8*>(XT + XW * 64 + choff);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(XT + 2 * XW * 64 + choff);
          const int zoff = zc * 64 + ((ch ^ zkey) << 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) {
            const char* zi = ZT + m * 3 * ZC * 64 + zoff;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zi);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(zi + ZC * 64);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(zi + 2 * ZC * 64);
            f32x16 cc = acc[m];        // smallest terms first
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, cc, 0, 0, 0);
            acc[m] = cc;
          }
        }
      }
      if constexpr (DB) pb ^= 1;          // the next chunk (or the next tile's fp32 G) writes the other set: no barrier here
      else __syncthreads();               // the planes are free for the next chunk / the next tile's fp32 G
    }
    ts = ts_n; R = R_n;
    if (next + p.n_split < p.ntiles) { ts_n = p.tile_start[next + p.n_split]; R_n = p.tile_start[next + p.n_split + 1] - ts_n; }
  }

  // ---- one slab per tile-list slice blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)slice * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  if (in_active) {
    const int i = xcol0 + ibw * 32 + c32;
    if (i < p.hin) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = gcol0 + obh * 32 + acc_row(r, half);
          if (o < p.hout) out[((size_t)m * p.hout + o) * p.hin + i] = acc[m][r];
        }
    }
  }
  if (ibg == 0) {   // (uniform) column sums: the 32 threads that share a column group meet in LDS, fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [1 + NMAT][NT]
    const int nsum = rs2 ? 1 + NMAT : 1;
    __syncthreads();
    red[tid] = bsum;
    if constexpr (RS2) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m) red[(1 + m) * NT + tid] = bs2[m];
    }
    __syncthreads();
    for (int j = tid; j < nsum * ZC; j += NT) {
      const int which = j / ZC, col = j - which * ZC;
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += red[which * NT + r * 16 + (col >> 2)][col & 3];
      const int o = gcol0 + col;
      if (o < p.hout) out[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
    }
  }
}

// two sets of planes (one barrier per chunk instead of two) wherever they fit
static bool wgrad16t_double(int nrb, int nmat, int ell_width) {
  static const int on = [] { const char* e = getenv("DSS2_WGRAD_TALL_DB"); return e ? atoi(e) : 1; }();
  const size_t fimg = (size_t)32 * nrb * W16B_LDZF * 4, planes = (size_t)nmat * 3 * W16B_ZC * 64 + 3 * (size_t)W16B_XW * 64;
  return on && fimg + 2 * (fimg > planes ? fimg : planes) + (size_t)((ell_width + 3) & ~3) * 32 * nrb * 8 <= (size_t)kMaxLdsBytes;
}

static size_t wgrad16t_lds_bytes(int nrb, int nmat, int ell_width) {
  const size_t fimg = (size_t)32 * nrb * W16B_LDZF * 4, planes = (size_t)nmat * 3 * W16B_ZC * 64 + 3 * (size_t)W16B_XW * 64;
  const size_t b = fimg + (wgrad16t_double(nrb, nmat, ell_width) ? 2 : 1) * (fimg > planes ? fimg : planes) + (size_t)((ell_width + 3) & ~3) * 32 * nrb * 8;
  const size_t red = (size_t)(1 + nmat) * W16T_NT * 16;
  return b > red ? b : red;
}


size_t wgrad16_lds_bytes(int nrb, int nmat, int hout, int hin, int ell_width) {
  if (nrb < 1 || nrb > 6 || nmat < 2 || nmat > 3 || ell_width < 1 || ell_width > W16_DMAX || hout <= 32 || (hout & 3) || (hin & 3)) return 0;
  if (nrb >= 3) {      // wgrad16t_kernel: tall tiles, chunks of 32 rows
    static const int tall = [] { const char* e = getenv("DSS2_WGRAD_TALL16"); return e ? atoi(e) : 1; }();
    return tall ? wgrad16t_lds_bytes(nrb, nmat, ell_width) : 0;
  }
  if (nrb == 1) {      // wgrad16b_kernel: two workgroups per CU
    const size_t b1 = 2 * (size_t)W16B_TM * W16B_LDZF * 4 + (size_t)nmat * 3 * W16B_ZC * 64 + 3 * (size_t)W16B_XW * 64 + (size_t)((ell_width + 3) & ~3) * W16B_TM * 8;
    const size_t red1 = (size_t)(1 + nmat) * W16B_NT * 16;
    return b1 > red1 ? b1 : red1;
  }
  size_t b = 2 * (size_t)W16_TM * W16_LDZF * 4 + (size_t)nmat * 3 * W16_ZC * 128 + 3 * (size_t)W16_XW * 128 + (size_t)ell_width * W16_TM * 8;
  const size_t red = (size_t)(1 + nmat) * W16_NT * 16;          // the final column-sum exchange reuses the front of the buffer
  return b > red ? b : red;
}

bool wgrad16_covers(const dss2_wgrad_args& a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return a.mfma_bf16 && !a.narrow && !a.rowscale && a.ell_tiles && al16(a.G) && al16(a.X) && (a.ldg & 3) == 0 && (a.ldx & 3) == 0 &&
         (!a.rowscale2 || al16(a.rowscale2)) && wgrad16_lds_bytes(a.nrb, a.nmat, a.hout, a.hin, a.ell_width) != 0 &&
         wgrad16_lds_bytes(a.nrb, a.nmat, a.hout, a.hin, a.ell_width) <= (size_t)kMaxLdsBytes;
}

int wgrad16_y_slices(int nrb, int hout, int hin) {
  return nrb != 2 ? ((hout + W16B_ZC - 1) / W16B_ZC) * ((hin + W16B_XW - 1) / W16B_XW) : ((hout + 127) / 128) * ((hin + 127) / 128);
}

template <int NMAT, bool RS2>
static int launch16b(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16b_kernel<NMAT, RS2>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(bf16x6, 32 rows)")) return 1;
  const int nobg = (a.hout + W16B_ZC - 1) / W16B_ZC, nibg = (a.hin + W16B_XW - 1) / W16B_XW;
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, wb.n > 0 ? wb.n : 1), dim3(W16B_NT),
                     wgrad16_lds_bytes(a.nrb, a.nmat, a.hout, a.hin, a.ell_width), stream, a, nibg, wb);
  return check_launch("wgrad(bf16x6, 32 rows)");
}

template <int NRB, int NMAT, bool RS2, bool DB>
static int launch16t(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16t_kernel<NRB, NMAT, RS2, DB>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(bf16x6, tall tiles)")) return 1;
  const int nobg = (a.hout + W16B_ZC - 1) / W16B_ZC, nibg = (a.hin + W16B_XW - 1) / W16B_XW;
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, wb.n > 0 ? wb.n : 1), dim3(W16T_NT),
                     wgrad16_lds_bytes(a.nrb, a.nmat, a.hout, a.hin, a.ell_width), stream, a, nibg, wb);
  return check_launch("wgrad(bf16x6, tall tiles)");
}

template <int NMAT, int NP, bool RS2>
static int launch16(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16_kernel<NMAT, NP, RS2>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(bf16x6)")) return 1;
  const int nobg = (a.hout + NP * W16_ZC - 1) / (NP * W16_ZC), nibg = (a.hin + W16_XW - 1) / W16_XW;
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, wb.n > 0 ? wb.n : 1), dim3(W16_NT),
                     wgrad16_lds_bytes(a.nrb, a.nmat, a.hout, a.hin, a.ell_width), stream, a, nibg, wb);
  return check_launch("wgrad(bf16x6)");
}

int launch_wgrad16(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  bool rs2 = a.rowscale2 != nullptr;
  for (int l = 0; l < wb.n; ++l) {
    if ((reinterpret_cast<uintptr_t>(wb.G[l]) | reinterpret_cast<uintptr_t>(wb.X[l]) | reinterpret_cast<uintptr_t>(wb.rowscale2[l])) & 15) {
      set_error("wgrad(bf16x6): layer %d has a misaligned operand", l); return 2;
    }
    rs2 = rs2 || wb.rowscale2[l] != nullptr;
  }
  if (a.nrb == 1) {
    if (a.nmat == 2) return rs2 ? launch16b<2, true>(a, stream, wb) : launch16b<2, false>(a, stream, wb);
    if (a.nmat == 3) return rs2 ? launch16b<3, true>(a, stream, wb) : launch16b<3, false>(a, stream, wb);
  }
#define DSS2_TALL(NRB, DB) \
  if (a.nrb == NRB && wgrad16t_double(a.nrb, a.nmat, a.ell_width) == DB) { \
    if (a.nmat == 2) return rs2 ? launch16t<NRB, 2, true, DB>(a, stream, wb) : launch16t<NRB, 2, false, DB>(a, stream, wb); \
    if (a.nmat == 3) return rs2 ? launch16t<NRB, 3, true, DB>(a, stream, wb) : launch16t<NRB, 3, false, DB>(a, stream, wb); \
  }
  DSS2_TALL(3, true) DSS2_TALL(4, true) DSS2_TALL(5, true) DSS2_TALL(6, true) DSS2_TALL(3, false) DSS2_TALL(4, false) DSS2_TALL(5, false) DSS2_TALL(6, false)
#undef DSS2_TALL
  if (a.nrb != 2) { set_error("wgrad(bf16x6): no kernel for nrb=%d nmat=%d", a.nrb, a.nmat); return 2; }
  if (a.nmat == 2) return rs2 ? launch16<2, 2, true>(a, stream, wb) : launch16<2, 2, false>(a, stream, wb);
  if (a.nmat == 3) return rs2 ? launch16<3, 2, true>(a, stream, wb) : launch16<3, 2, false>(a, stream, wb);
  set_error("wgrad(bf16x6): unsupported nmat=%d", a.nmat);
  return 2;
}

}  // namespace dss2
