// Weight gradient of the H -> H TAGConv layers on 64-row tiles whose INPUT activations arrive as transposed bf16x3 planes
// written by the kernel that produced them (round 5):  dW_m = (P^m G)^T X,  db = colsum(G), bf16x6 (dss2_common.hpp).
//
// wgrad16b_kernel (dss2_wgrad16.hip) spends ~220 of its ~500 vector instructions per thread and 32-row tile on operand
// splits -- X is split by BOTH workgroups that own its output halves -- and a SIMD's vector issue is one resource for the
// matrix pipe and the VALU: its time is MFMA time plus vector time.  The forward chain's epilogue splits every
// activation anyway (the next layer's A planes), so it now also stores the three pieces in the order THIS kernel's B operand
// wants, and the X side of the product costs no vector instruction here: a wave reads its fragments straight from global
// memory (L2), 16 bytes per lane, like the chain reads its weight fragments.
//
// X plane image of a layer (dss2_xplanes_bytes; written by dss2_gemm_prop_chain layers with x_planes != NULL and by
// dss2_edge_tile_fwd_xp): [tile][column block of 32][k-step 0..3][plane h, m, l][1 KB], the 1 KB being the B fragment of
// v_mfma_f32_32x32x16_bf16 in lane order: lane (n, kh) owns 16 bytes = 8 bf16 = the tile rows  r8 + 8 i, i = 0..7, with
// r8 = 2 kstep + kh, of column  32 block + ((n & 7) << 2 | n >> 3).  Any order of a contraction index is as good as another:
// the order is the one in which the producers' lanes hold their rows (the chain's epilogue lane owns rows r8 + 8 i of four
// columns: one 16-byte store per column and plane; the 32x32 accumulator of the edge MLP holds (r & 3) + 4 half = r8).
//
// Here: one workgroup = four waves, 64 output x 128 input columns of every matrix, two workgroups per CU (70 KB of LDS):
//     fp32 G of the tile (hop input, Z_0 source)   64 x 64 x 4       16 KB
//     fp32 P G of the tile (hop input, Z_1 source) 64 x 64 x 4       16 KB
//     transposed planes of G, P G, P^2 G, one CHUNK of 32 positions  36 KB    (positions 32 c .. 32 c + 31 = rows with (row & 7) >> 2 == c)
//     ELL slice, padded to 4 or 8 entries per row                     2 / 4 KB
// per tile: stage G, one hop for the whole tile, then per chunk: split G / P G rows, gather + split P^2 G rows, 2 k-steps of MFMAs
// (36 per wave and k-step, 9 fragments from LDS + 6 from global memory).  Vector instructions per thread and 64-row tile: ~500
// (wgrad16b: ~1000 for the same rows).
//
// Work split: the (layer, tile) pairs of the launch are ONE list cut into equal contiguous ranges, one per workgroup and output
// half -- at C2 3 x 1024 pairs over 256 workgroups = 12 each, where a grid slice per layer left 85 workgroups 12 or 13 tiles (the
// launch then takes 13).  A workgroup whose range crosses a layer boundary writes one slab per layer: slab id = workgroup +
// layer (unique and contiguous per layer; the caller reduces slabs [first_w(l) + l, last_w(l) + l] of layer l).
#include <stdlib.h>

#include "dss2_wgrad_batch.hpp"

namespace dss2 {

#ifdef DSS2_STAMPS
// Diagnostic build only (-DDSS2_STAMPS; tools/pstamps.py): per-wave phase stamps of the THIRD tile of every workgroup's range.
__device__ unsigned long long g_pstamps[512 * 4 * 16];
#define PSTAMP(slot)                                                                                   \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      unsigned long long t_;                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if (lane == 0 && blockIdx.x < 256) g_pstamps[((blockIdx.x * 2 + (blockIdx.y & 1)) * 4 + wv) * 16 + (slot)] = t_; \
    }                                                                                                  \
  } while (0)
#else
#define PSTAMP(slot) do {} while (0)
#endif

#ifndef W16P_EARLY
#define W16P_EARLY 0      // k-step 1's X fragments requested before the barrier that ends the plane building (24 more live registers there)
#endif
#ifndef W16P_TOUCH
#define W16P_TOUCH 1      // L2 prefetch of the X plane lines: 1 = the same chunk of the NEXT tile (two chunks ahead), 2 = the next chunk, 0 = none
#endif
#if W16P_TOUCH == 1
#define TOUCH_NEXT() touch(item + 1, c, c)
#elif W16P_TOUCH == 2
#define TOUCH_NEXT() do { if (c == 0) touch(item, 1, 1); else touch(item + 1, 0, 0); } while (0)
#else
#define TOUCH_NEXT() do { } while (0)
#endif
constexpr int W16P_TR = 64, W16P_ZC = 64, W16P_XW = 128, W16P_NT = 256, W16P_LDZF = 64, W16P_DMAX = 8;

// same transposed-image swizzle as wgrad16b_kernel (64 bytes per column and plane; "row" = position inside the chunk)
__device__ __forceinline__ int tpp_key(int col) { return (((col >> 3) & 1) << 1) | ((col >> 4) & 1); }
__device__ __forceinline__ int tpp_off(int col, int pos) { return col * 64 + ((((pos >> 3) ^ tpp_key(col)) << 4) | ((pos & 7) << 1)); }

template <int NCOLS>
__device__ __forceinline__ void store_planes_p(char* img, int off0, const f32x4 v0, const f32x4 v1) {
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
__global__ void __launch_bounds__(W16P_NT, 2) wgrad16p_kernel(const dss2_wgrad_args p, int nibg, const WgradPlanes wp) {
  constexpr int TR = W16P_TR, ZC = W16P_ZC, XW = W16P_XW, NT = W16P_NT, LDZF = W16P_LDZF;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf0 = smem;                                              // [TR][LDZF]  G
  float* Zf1 = Zf0 + TR * LDZF;                                   // [TR][LDZF]  P G
  char* ZT = reinterpret_cast<char*>(Zf1 + TR * LDZF);            // [NMAT][3 planes][ZC columns][64 B]
  int2* ell = reinterpret_cast<int2*>(ZT + NMAT * 3 * ZC * 64);   // [Dp][TR]
  const int D = p.ell_width, Dp = (D + 3) & ~3;
  f32x4* rsl = reinterpret_cast<f32x4*>(ell + Dp * TR);           // [TR] the folded layer's row scales of the tile (RS2)

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xb0 = 2 * (wv & 1), obw = wv >> 1;      // wave: input blocks xb0, xb0 + 1 x output block obw of every matrix
  const int ysl = blockIdx.y;
  const int obg = ysl / nibg, ibg = ysl - obg * nibg;
  // which range of the (layer, tile) list: y-slice 0 takes range blockIdx.x; the others are shifted (wp.pair: 1 = by half the ranges
  // + 3, the CU's two workgroups -- consecutive slices of one x land on one CU -- then walk unrelated tiles; 2 = x ^ 8; 0 = the same range)
  const int wg = ysl == 0 || wp.pair == 0 ? (int)blockIdx.x
               : (wp.pair == 1 ? (int)((blockIdx.x + (gridDim.x >> 1) * ysl + 3 * ysl) % gridDim.x)
                               : (((int)blockIdx.x ^ 8) < (int)gridDim.x ? ((int)blockIdx.x ^ 8) : (int)blockIdx.x));
  const int gcol0 = obg * ZC, xcol0 = ibg * XW;
  const bool x_on[2] = {(xcol0 + xb0 * 32) < p.hin, (xcol0 + (xb0 + 1) * 32) < p.hin};
  const bool in_active = x_on[0] && (gcol0 + obw * 32) < p.hout;

  // whole-tile units: rows r16 + 16 j (j < 4) of four columns 4 q16
  const int q16 = tid & 15, r16 = tid >> 4;
  const bool gcol_ok = gcol0 + 4 * q16 < p.hout;
  // chunk units: positions (2 pp, 2 pp + 1) of the chunk, four columns 4 q16.  Position 8 a + i of chunk c is tile row
  // (4 c + a) + 8 i, so the pair is rows rA, rA + 8 with rA = 4 c + (pp >> 2) + 16 (pp & 3).  (odd column groups own the position
  // pair pp ^ 2: the transposed b32 stores of a half wave then reach 16 banks -- see wgrad16b_kernel)
  const int pp = (tid >> 4) ^ ((tid & 1) << 1);
  const int z_off = tpp_off(4 * q16, 2 * pp);
  const int rA0 = (pp >> 2) + 16 * (pp & 3);

  // the launch's (layer, tile) list, this workgroup's range
  const long long total = (long long)wp.n_layers * p.ntiles;
  const long long it0 = (long long)wg * wp.ipw, it1 = (it0 + wp.ipw < total) ? it0 + wp.ipw : total;

  f32x4 pg[4], prs = {0.f, 0.f, 0.f, 0.f};
  int2 pel[2];
  auto load_item = [&](long long item) {
    const int L = (int)(item / p.ntiles), tile = (int)(item - (long long)L * p.ntiles);
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    const char* gb = reinterpret_cast<const char*>(wp.G[L] + (size_t)ts * p.ldg + gcol0 + 4 * q16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r16 + 16 * j;
      pg[j] = (r < R && gcol_ok) ? *reinterpret_cast<const f32x4*>(gb + (uint32_t)(r * p.ldg) * 4u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TR;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * NT;
      pel[e] = idx < D * TR ? src[idx] : make_int2(idx & (TR - 1), 0);      // (padding entries: own row, zero weight)
    }
    if constexpr (RS2) {      // the folded layer's row scales: one row per thread of the first wave, handed over in LDS
      const float* rs2n = wp.rowscale2[L];
      prs = (rs2n && tid < R) ? *reinterpret_cast<const f32x4*>(rs2n + (size_t)(ts + tid) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // L2 prefetch of a chunk's X plane lines (24 KB = 192 lines of 128 B for this workgroup's 128 input columns): one dword per line
  // into a register that is CONSUMED (added to a sink) the next time the slot is used -- a whole tile later, when it has long
  // arrived -- so no s_waitcnt ever stalls on it, and the chunk's fragment loads then hit L2 (a fragment load that misses to HBM
  // costs ~2 us against ~0.5 us of k-step to hide behind: without this the kernel is latency-bound at 150 us).
  [[maybe_unused]] uint32_t tch[2] = {0u, 0u}, tsink = 0u;
  [[maybe_unused]] auto touch = [&](long long it, int c, int slot) {
    if (it >= it1 || tid >= 192) return;
    const int L = (int)(it / p.ntiles), tile = (int)(it - (long long)L * p.ntiles);
    // line tid of the chunk: input block tid / 48, k-step 2 c + (tid % 48) / 24, then 24 lines = 3 pieces x 1 KB
    const int blk = tid / 48, rem = tid - blk * 48;
    const char* q = reinterpret_cast<const char*>(wp.XP[L]) + ((size_t)tile * wp.ncb + (size_t)(ibg * 4 + blk)) * 12288 + (size_t)(2 * c) * 3072 + (size_t)rem * 128;
    tsink += tch[slot];
    tch[slot] = *reinterpret_cast<const volatile uint32_t*>(q);
  };
  // one row of P Zs (four columns at c4); the slice is padded to four entries per row (zero weight, own row)
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

  long long item = it0;
  if (item < it1) load_item(item);
#if W16P_TOUCH
  touch(it0, 0, 0);
  touch(it0, 1, 1);
#endif
  while (item < it1) {
    // ---- one layer segment of the range: its own accumulators and slab
    const int L = (int)(item / p.ntiles);
    const long long seg_end = ((long long)(L + 1) * p.ntiles < it1) ? (long long)(L + 1) * p.ntiles : it1;
    const char* xp_layer = reinterpret_cast<const char*>(wp.XP[L]);
    [[maybe_unused]] const float* rs2 = RS2 ? wp.rowscale2[L] : nullptr;
    f32x16 acc[2][NMAT];
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

    for (; item < seg_end; ++item) {
      const int tile = (int)(item - (long long)L * p.ntiles);
      const int ts = p.tile_start[tile];
      const int R = p.tile_start[tile + 1] - ts;
      [[maybe_unused]] const bool stamp_on = item == it0 + 2;
      PSTAMP(0);
      // ---- fp32 G of the tile, the ELL slice, bias partial sums
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<f32x4*>(Zf0 + (r16 + 16 * j) * LDZF + 4 * q16) = pg[j];
        bsum += pg[j];
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (tid + e * NT < Dp * TR) ell[tid + e * NT] = pel[e];
      if constexpr (RS2) {
        if (tid < TR) rsl[tid] = prs;
      }
      __syncthreads();
      PSTAMP(1);
      if (item + 1 < it1) load_item(item + 1);      // in flight for the whole tile (the next item may belong to the next layer)
      if constexpr (RS2) {
        if (rs2) {      // (uniform; the folded layer only)
#pragma unroll
          for (int j = 0; j < 4; ++j) {      // (the thread's own G rows, read back from the image: pg already holds the next tile's)
            const f32x4 d = rsl[r16 + 16 * j];
            const f32x4 gj = *reinterpret_cast<const f32x4*>(Zf0 + (r16 + 16 * j) * LDZF + 4 * q16);
#pragma unroll
            for (int m = 0; m < NMAT; ++m) bs2[m] += gj * d[m];
          }
        }
      }
      // ---- P G of the tile (pad rows gather themselves with weight 0: zero rows)
      if (NMAT > 1) {
        f32x4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = hop_row(Zf0, r16 + 16 * j, 4 * q16);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(Zf1 + (r16 + 16 * j) * LDZF + 4 * q16) = a[j];
        __syncthreads();
      }
      const char* xp_tile = xp_layer + ((size_t)tile * wp.ncb + (size_t)(ibg * 4 + xb0)) * 12288 + lane * 16;
      for (int c = 0; c < 2; ++c) {
        if (4 * c >= R) break;      // (uniform) no real row in this chunk
        // ---- the chunk's X fragments of k-step 0: requested before the plane building that hides their latency
        bf16x8 xf[2][2][3];      // [k-step of the chunk][input block][plane]
        auto load_x = [&](int ksl) {
#pragma unroll
          for (int xb = 0; xb < 2; ++xb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              xf[ksl][xb][pl] = *reinterpret_cast<const bf16x8*>(xp_tile + (x_on[xb] ? xb : 0) * 12288 + (2 * c + ksl) * 3072 + pl * 1024);
        };
        if (in_active) load_x(0);
        // ---- planes of the chunk: rows rA, rA + 8
        {
          const int rA = 4 * c + rA0;
          const float* s0 = Zf0 + rA * LDZF + 4 * q16;
          store_planes_p<ZC>(ZT, z_off, *reinterpret_cast<const f32x4*>(s0), *reinterpret_cast<const f32x4*>(s0 + 8 * LDZF));
          if (NMAT > 1) {
            const float* s1 = Zf1 + rA * LDZF + 4 * q16;
            store_planes_p<ZC>(ZT + 3 * ZC * 64, z_off, *reinterpret_cast<const f32x4*>(s1), *reinterpret_cast<const f32x4*>(s1 + 8 * LDZF));
          }
          if (NMAT > 2) {
            const f32x4 u0 = hop_row(Zf1, rA, 4 * q16), u1 = hop_row(Zf1, rA + 8, 4 * q16);
            store_planes_p<ZC>(ZT + 2 * 3 * ZC * 64, z_off, u0, u1);
          }
        }
#if W16P_EARLY
        if (in_active) load_x(1);      // (requested before the barrier: its wait and k-step 0 cover the latency)
#endif
        PSTAMP(3 + 4 * c);
        __syncthreads();
        PSTAMP(4 + 4 * c);
        // ---- MFMA phase: 2 k-steps of 16 positions; a step's Z fragments (three per matrix) serve both input blocks of the wave
        if (in_active) {
          const int zc = obw * 32 + c32;
          const int zkey = tpp_key(zc);
          auto kstep = [&](const bf16x8 (&xk)[2][3], int ksl) {
            const int zoff = zc * 64 + (((2 * ksl + half) ^ zkey) << 4);
#pragma unroll
            for (int m = 0; m < NMAT; ++m) {
              const char* zi = ZT + m * 3 * ZC * 64 + zoff;
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(zi);
              const bf16x8 am = *reinterpret_cast<const bf16x8*>(zi + ZC * 64);
              const bf16x8 al = *reinterpret_cast<const bf16x8*>(zi + 2 * ZC * 64);
#pragma unroll
              for (int xb = 0; xb < 2; ++xb) {
                if (xb == 1 && !x_on[1]) continue;      // (uniform)
                f32x16 cc = acc[xb][m];        // smallest terms first
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xk[xb][0], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xk[xb][1], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xk[xb][2], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xk[xb][0], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xk[xb][1], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xk[xb][0], cc, 0, 0, 0);
                acc[xb][m] = cc;
              }
            }
          };
#if !W16P_EARLY
          load_x(1);                                   // in flight under k-step 0's MFMAs
#endif
          TOUCH_NEXT();
          __builtin_amdgcn_sched_barrier(0);
          kstep(xf[0], 0);
          __builtin_amdgcn_sched_barrier(0);           // (k-step 1's nine Z fragments are not hoisted over k-step 0: 36 registers)
          kstep(xf[1], 1);
        } else {
          TOUCH_NEXT();
        }
        PSTAMP(5 + 4 * c);
        __syncthreads();          // the planes are free for the next chunk, the fp32 images (after chunk 1) for the next tile
        PSTAMP(6 + 4 * c);
      }
    }

    // ---- the segment's slab: id = workgroup + layer; the y-slices tile the [nmat*hout, hin] matrix
    float* out = wp.slab + (size_t)(wg + L) * (size_t)wp.slab_len;
    if (in_active) {
      // (lane indices made opaque here: the store addresses are formed per segment, not hoisted out of the range loop as 43 spilled
      //  64-bit values)
      int cl = c32, hl = half, hin_l = p.hin, hout_l = p.hout;
      asm volatile("" : "+v"(cl), "+v"(hl), "+s"(hin_l), "+s"(hout_l));      // (the uniform strides too: 48 hoisted 64-bit scalar offsets otherwise)
      const int ncol = ((cl & 7) << 2) | (cl >> 3);      // accumulator column n -> input column of the block (the image's lane order)
      const int o0 = gcol0 + obw * 32 + 4 * hl;
#pragma unroll
      for (int xb = 0; xb < 2; ++xb) {
        const int i = xcol0 + (xb0 + xb) * 32 + ncol;
        if (i < hin_l) {
          float* ob = out + (size_t)o0 * hin_l + i;
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);      // acc_row(r, half) - 4 half
              if (o0 + dr < hout_l) ob[((size_t)m * hout_l + dr) * hin_l] = acc[xb][m][r];
            }
        }
      }
    }
    if (ibg == 0) {   // (uniform) column sums: the 16 threads that share a column group meet in LDS, fixed order
      f32x4* red = reinterpret_cast<f32x4*>(smem);          // [1 + NMAT][NT]
      const int nsum = rs2 ? 1 + NMAT : 1;
      red[tid] = bsum;      // (the last barrier of the tile loop freed the fp32 images)
      if constexpr (RS2) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) red[(1 + m) * NT + tid] = bs2[m];
      }
      __syncthreads();
      for (int j = tid; j < nsum * ZC; j += NT) {
        const int which = j / ZC, col = j - which * ZC;
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += red[which * NT + r * 16 + (col >> 2)][col & 3];
        const int o = gcol0 + col;
        if (o < p.hout) out[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
      }
      __syncthreads();      // (the next segment stages over the buffer)
    }
  }
  if (tsink + tch[0] + tch[1] == 0x9e3779b9u && p.ntiles < 0) wp.slab[0] = 0.f;      // (never true: keeps the prefetch registers' consumption alive)
}

size_t wgrad16p_lds_bytes(int nmat, int ell_width) {
  const size_t b = 2 * (size_t)W16P_TR * W16P_LDZF * 4 + (size_t)nmat * 3 * W16P_ZC * 64 + (size_t)((ell_width + 3) & ~3) * W16P_TR * 8 + (size_t)W16P_TR * 16;
  const size_t red = (size_t)(1 + nmat) * W16P_NT * 16;
  return b > red ? b : red;
}

bool wgrad16p_covers(int nrb, int nmat, int hout, int hin, int ell_width) {
  return nrb == 2 && nmat >= 2 && nmat <= 3 && ell_width >= 1 && ell_width <= W16P_DMAX && hout > 32 && (hout & 3) == 0 && hin >= 32 &&
         (hin & 31) == 0 && wgrad16p_lds_bytes(nmat, ell_width) <= (size_t)kMaxLdsBytes / 2;
}

template <int NMAT, bool RS2>
static int launch16p(const dss2_wgrad_args& a, const WgradPlanes& wp, int n_wg, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16p_kernel<NMAT, RS2>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(bf16x6, X planes)")) return 1;
  const int nobg = (a.hout + W16P_ZC - 1) / W16P_ZC, nibg = (a.hin + W16P_XW - 1) / W16P_XW;
  hipLaunchKernelGGL(kern, dim3(n_wg, nobg * nibg), dim3(W16P_NT), wgrad16p_lds_bytes(a.nmat, a.ell_width), stream, a, nibg, wp);
  return check_launch("wgrad(bf16x6, X planes)");
}

int launch_wgrad16p(const dss2_wgrad_args& a, const WgradPlanes& wp, int n_wg, hipStream_t stream) {
  bool rs2 = false;
  for (int l = 0; l < wp.n_layers; ++l) rs2 = rs2 || wp.rowscale2[l] != nullptr;
  if (a.nmat == 2) return rs2 ? launch16p<2, true>(a, wp, n_wg, stream) : launch16p<2, false>(a, wp, n_wg, stream);
  if (a.nmat == 3) return rs2 ? launch16p<3, true>(a, wp, n_wg, stream) : launch16p<3, false>(a, wp, n_wg, stream);
  set_error("wgrad(bf16x6, X planes): unsupported nmat=%d", a.nmat);
  return 2;
}

}  // namespace dss2

#ifdef DSS2_STAMPS
extern "C" int dss2_debug_read_pstamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_pstamps), sizeof(unsigned long long) * n);
}
#endif

extern "C" size_t dss2_xplanes_bytes(int64_t ntiles, int32_t ncols) {
  return (size_t)ntiles * (size_t)((ncols + 31) / 32) * 12288;
}

extern "C" int dss2_wgrad_xp_supported(int nrb, int nmat, int hout, int hin, int ell_width) {
  static const int on = [] { const char* e = getenv("DSS2_WGRAD_XP"); return e ? atoi(e) : 1; }();
  return on && dss2::wgrad16p_covers(nrb, nmat, hout, hin, ell_width) ? 1 : 0;
}

extern "C" int dss2_wgrad_xp_per_cu(int nrb, int nmat, int hout, int hin, int ell_width) {
  return dss2::wgrad16q_covers(nrb, nmat, hout, hin, ell_width) ? 1 : 2;      // the pipelined kernel takes the CU's whole LDS
}

extern "C" int dss2_wgrad_xp_y_slices(int hout, int hin) {
  return ((hout + dss2::W16P_ZC - 1) / dss2::W16P_ZC) * ((hin + dss2::W16P_XW - 1) / dss2::W16P_XW);
}

static int dss2_wgrad_batched_xp_launch(const dss2_wgrad_args* ap, const float* const* Gs, const void* const* Xps, float* slab, int64_t slab_len, const float* const* rowscale2s, int n_layers, int n_wg, void* stream);
extern "C" int dss2_wgrad_batched_xp(const dss2_wgrad_args* ap, const float* const* Gs, const void* const* Xps, float* slab, int64_t slab_len, const float* const* rowscale2s, int n_layers, int n_wg, void* stream) {
  if (!ap) { dss2::set_error("dss2_wgrad_batched_xp: null argument"); return 2; }
  DSS2_RECORD([a = *ap, g = dss2::plan_keep(Gs, (size_t)(n_layers > 0 ? n_layers : 0)), x = dss2::plan_keep(Xps, (size_t)(n_layers > 0 ? n_layers : 0)), slab, slab_len, r = dss2::plan_keep(rowscale2s, (size_t)(rowscale2s && n_layers > 0 ? n_layers : 0)), n_layers, n_wg](void* s_) { return dss2_wgrad_batched_xp_launch(&a, dss2::plan_ptr(g), dss2::plan_ptr(x), slab, slab_len, dss2::plan_ptr(r), n_layers, n_wg, s_); });
  return dss2_wgrad_batched_xp_launch(ap, Gs, Xps, slab, slab_len, rowscale2s, n_layers, n_wg, stream);
}
static int dss2_wgrad_batched_xp_launch(const dss2_wgrad_args* ap, const float* const* Gs, const void* const* Xps, float* slab, int64_t slab_len, const float* const* rowscale2s, int n_layers, int n_wg, void* stream) {
  using namespace dss2;
  if (n_layers < 1 || n_layers > WGRAD_MAX_BATCH) { set_error("wgrad_batched_xp: 1..%d layers, got %d", WGRAD_MAX_BATCH, n_layers); return 2; }
  if (!ap || !Gs || !Xps || !slab || n_wg < 1) { set_error("wgrad_batched_xp: null pointer / no workgroups"); return 2; }
  const dss2_wgrad_args& a = *ap;
  if (!wgrad16p_covers(a.nrb, a.nmat, a.hout, a.hin, a.ell_width) || !a.ell_tiles || a.rowscale || a.narrow || (a.ldg & 3)) {
    set_error("wgrad_batched_xp: shape not covered (nrb=%d nmat=%d hout=%d hin=%d ell=%d)", a.nrb, a.nmat, a.hout, a.hin, a.ell_width);
    return 2;
  }
  WgradPlanes wp = {};
  wp.n_layers = n_layers;
  wp.slab = slab;
  wp.slab_len = slab_len;
  wp.ncb = a.hin / 32;
  {      // experiment switch: DSS2_WGRAD_XP_MODE = which range the other output halves walk (see the kernel)
    static const int mode = [] { const char* e = getenv("DSS2_WGRAD_XP_MODE"); return e ? atoi(e) : 1; }();
    wp.pair = mode & 3;
  }
  const long long total = (long long)n_layers * a.ntiles;
  wp.ipw = (int)((total + n_wg - 1) / n_wg);
  const long long need = (long long)a.nmat * a.hout * a.hin + a.hout;
  for (int l = 0; l < n_layers; ++l) {
    if (!Gs[l] || !Xps[l]) { set_error("wgrad_batched_xp: layer %d has a null pointer", l); return 2; }
    wp.G[l] = Gs[l]; wp.XP[l] = Xps[l];
    wp.rowscale2[l] = rowscale2s ? rowscale2s[l] : nullptr;
    if ((reinterpret_cast<uintptr_t>(Gs[l]) | reinterpret_cast<uintptr_t>(Xps[l]) | reinterpret_cast<uintptr_t>(wp.rowscale2[l])) & 15) {
      set_error("wgrad_batched_xp: layer %d has a misaligned operand", l); return 2;
    }
    if (slab_len < need + (wp.rowscale2[l] ? (long long)a.nmat * a.hout : 0)) { set_error("wgrad_batched_xp: slab_len too small"); return 2; }
  }
  if (wgrad16q_covers(a.nrb, a.nmat, a.hout, a.hin, a.ell_width)) return launch_wgrad16q(a, wp, n_wg, as_stream(stream));
  return launch_wgrad16p(a, wp, n_wg, as_stream(stream));
}
