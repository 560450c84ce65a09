// Weight gradient of TAGConv on 96- .. 192-row tiles (graphs of 65 .. 192 nodes: C3, the 179-bus feeder) as f16x3 (end of round 5).
//   dW_m = (P^m G)^T X,  db = colsum(G)   -- contract, slab layout, thread maps and phases of wgrad16t_kernel (dss2_wgrad16.hip): the
//   propagation needs the whole graph, so P G of the WHOLE tile is kept in fp32; the contraction over rows is cut into chunks of 32
//   rows, whose transposed planes (G, P G, P^2 G, X) feed the matrix pipe.
// What changes against the bf16x6 form (same idea as dss2_wgrad16h.hip on 32-row tiles):
//   * every operand is TWO fp16 pieces after an exact power-of-two scale, a product is three v_mfma_f32_32x32x16_f16 instead of six
//     bf16 ones, a split is 7 vector instructions per pair instead of 11, a set of planes is 40 KB instead of 60;
//   * the scales are per TILE, running per workgroup: max |G| from the tile's G rows, which wait in registers a tile ahead as before,
//     and max |X| from the tile's X rows of ALL its chunks, which are now requested a whole tile ahead as well (8 / 16 more registers
//     at 96 rows, and three times the bytes in flight: the kernel waits on the arrival of its rows) -- so the
//     maxima need no barrier of their own: one partial per wave at the top of the tile, read behind the barrier that stages the
//     ELL slice; when a running exponent grows the accumulators are rescaled (v_ldexp, exact); G, P G, P^2 G carry hb headroom
//     bits for the gain of the hops (args.mfma_bf16 bits 8..15, from the host: ops._wgrad_mode);
//   * the slab is written with the scales taken out.
// C3 (ober_sub, B = 1024): 210.4 -> 173.8 us for the three layers (rocprofv3, one box); the per-chunk form below at 96 rows: replayed
// step 0.587-0.589 ms against 0.567-0.570 for this one.
//   * 128 .. 192 rows (PC; measured at 192): twelve row pieces of X per thread do not fit (19 spilled registers, 463 us against 457 for bf16x6), so there the X
//     rows stay one CHUNK ahead and X gets an exponent per chunk -- one partial per wave at the top of the chunk, one more barrier
//     (A) between building the G-side planes and splitting X; with two sets of planes that is two barriers per chunk, what the
//     bf16x6 kernel pays at this height with one set.  179-bus feeder, B = 1024: 462.8 -> 384.7 us, step 1.478 -> 1.408 ms.
// Errors of the size of fp32 arithmetic's own rounding (tests/test_gpu_f16x3.py).  Same inputs, same bits.
// Built without packed fp32 VALU ops like the other MFMA-beside-VALU translation units (build.sh).
#include <stdlib.h>

#include "dss2_wgrad_batch.hpp"

namespace dss2 {

constexpr int W16TH_ZC = 64, W16TH_XW = 128, W16TH_NT = 512, W16TH_LDZF = 64, W16TH_DMAX = 8;
// (the transposed image of wgrad16b_kernel / wgrad16h_kernel: 64 bytes per column and plane, 16-byte chunks of 8 rows swizzled)
__device__ __forceinline__ int tph2_key(int col) { return (((col >> 3) & 1) << 1) | ((col >> 4) & 1); }
__device__ __forceinline__ int tph2_off(int col, int row) { return col * 64 + ((((row >> 3) ^ tph2_key(col)) << 4) | ((row & 7) << 1)); }

// rows (2 rp, 2 rp + 1) x columns (c0 .. c0+3) scaled by 2^e -> the two planes of a transposed image with NCOLS columns
template <int NCOLS>
__device__ __forceinline__ void store_planes_h2(char* img, int off0, const f32x4 v0, const f32x4 v1, int e) {
  const float s = pow2f(e);      // (v_mul_f32: half the issue cycles of v_ldexp_f32, the same bits)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, l;
    split2_pair(v0[q] * s, v1[q] * s, h, l);
    char* dst = img + off0 + q * 64;
    *reinterpret_cast<uint32_t*>(dst) = h;
    *reinterpret_cast<uint32_t*>(dst + NCOLS * 64) = l;
  }
}

// PAIR (round 6; hout = hin = 32: the reference driver's model on 70-bus grids, seven H -> H layers per launch): at that width one wave of
// eight multiplies and half the staging lanes idle, while a tile still costs its barriers and latencies.  TWO layers share a workgroup's walk:
// columns 0..31 of the G-side image / of the X planes are layer 2 z's, columns 32..63 layer 2 z + 1's (per-thread base pointers); waves (0, 0)
// and (1, 1) multiply; the two layers share the running exponents.  157 -> ~90 us per launch of seven layers at 1024 tiles.
template <int NRB, int NMAT, bool RS2, bool PC, bool PAIR = false>
__global__ void __launch_bounds__(W16TH_NT) wgrad16th_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb, int hb) {
  constexpr int TR = 32 * NRB, ZC = W16TH_ZC, XW = W16TH_XW, NT = W16TH_NT, LDZF = W16TH_LDZF;
  constexpr int PLANES = NMAT * 2 * ZC * 64 + 2 * XW * 64;
  constexpr int PBUF = TR * LDZF * 4 > PLANES ? TR * LDZF * 4 : PLANES;      // one set of planes; the tile's fp32 G fits in it
  constexpr int UBYTES = 2 * PBUF;      // two sets of planes always fit (two planes per operand): one barrier per chunk
  const int zl = PAIR ? 2 * (int)blockIdx.z : (int)blockIdx.z;      // (first) layer of this workgroup
  const bool hasB = PAIR && zl + 1 < wb.n;                          // (uniform) the pair's second layer exists
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[zl] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[zl] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[zl] : p.slab;
  const float* __restrict__ rs2 = RS2 ? (wb.n > 0 ? wb.rowscale2[zl] : p.rowscale2) : nullptr;
  // the pair's second layer (PAIR only; a missing one reads the first layer's rows and stores nothing)
  const float* __restrict__ GpB = hasB ? wb.G[zl + 1] : Gp;
  const float* __restrict__ XpB = hasB ? wb.X[zl + 1] : Xp;
  float* __restrict__ slabB = hasB ? wb.slab[zl + 1] : slabp;
  const float* __restrict__ rs2B = (RS2 && hasB) ? wb.rowscale2[zl + 1] : nullptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf1 = smem;                                              // [TR][LDZF]  P G
  char* const U = reinterpret_cast<char*>(Zf1 + TR * LDZF);       // two sets of planes [NMAT][2 planes][ZC columns][64 B] + [2 planes][XW columns][64 B];
                                                                  // the tile's fp32 G [TR][LDZF] lives in the set the next chunk writes (dead after the first hop)
  int2* ell = reinterpret_cast<int2*>(U + UBYTES);                // [Dp][TR]
  int pb = 0;                                                     // the set the next chunk writes
  const int D = p.ell_width, Dp = (D + 3) & ~3;
  float* mxp = reinterpret_cast<float*>(ell + Dp * TR);           // [2][8 waves]: max |X|, max |G| of the tile whose rows wait in registers (PC: max |X| of the chunk)

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ibw = wave & 3, obh = wave >> 2, role = wave >> 2;
  const int slice = blockIdx.x, ysl = blockIdx.y;
  const int obg = ysl / nibg, ibg = ysl - obg * nibg;
  const int gcol0 = obg * ZC, xcol0 = ibg * XW;
  const bool in_active = PAIR ? (ibw == obh && ibw < 2 && (obh == 0 || hasB)) : ((xcol0 + ibw * 32) < p.hin && (gcol0 + obh * 32) < p.hout);

  // whole-tile units: one row of four columns per row block (a wave covers four whole rows of the fp32 image)
  const int q16 = tid & 15, r32 = tid >> 4;
  const int lyr_w = PAIR ? (q16 >> 3) : 0;      // (PAIR) which layer of the pair this thread's whole-tile columns belong to
  const bool gcol_ok = PAIR ? (4 * (q16 & 7) < p.hout && (lyr_w == 0 || hasB)) : (gcol0 + 4 * q16 < p.hout);
  // chunk units: rows (2 rp, 2 rp + 1) of four columns, the thread map of wgrad16b_kernel within each half of the workgroup
  const int t8 = tid & 255, cg = t8 & 15, rp = (t8 >> 4) ^ ((t8 & 1) << 1);
  const int z_off = tph2_off(4 * cg, 2 * rp), x_off = tph2_off(64 * role + 4 * cg, 2 * rp);
  const int lyr_c = PAIR ? (cg >> 3) : 0;
  const uint32_t xcb = PAIR ? ((role == 0 && 4 * (cg & 7) < p.hin && (lyr_c == 0 || hasB)) ? (uint32_t)(4 * (cg & 7)) * 4u : 0u)
                            : ((xcol0 + 64 * role + 4 * cg < p.hin) ? (uint32_t)(64 * role + 4 * cg) * 4u : 0u);      // (columns beyond hin: group 0, never stored)
  const bool zcol_ok = PAIR ? (4 * (cg & 7) < p.hout && (lyr_c == 0 || hasB)) : (gcol0 + 4 * cg < p.hout);
  // per-thread bases (PAIR: the layer the thread's columns belong to)
  const float* __restrict__ Gw_ = lyr_w ? GpB : Gp;
  const float* __restrict__ Gc_ = lyr_c ? GpB : Gp;
  const float* __restrict__ Xc_ = lyr_c ? XpB : Xp;

  f32x16 acc[NMAT];
#pragma unroll
  for (int m = 0; m < NMAT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs2[RS2 ? NMAT : 1];
#pragma unroll
  for (int m = 0; m < (RS2 ? NMAT : 1); ++m) bs2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  int Ex = -100, Eg = -100;      // running exponents of max |X|, max |G| over the tiles walked so far (uniform)

  // Everything a tile's prologue reads from global memory is requested one tile ahead (one workgroup per CU: nobody else
  // would cover the latency): its G rows, its ELL slice, the folded layer's row scales.
  constexpr int NEL = (W16TH_DMAX * TR + NT - 1) / NT;
  constexpr bool PRS = RS2 && NRB <= 4;      // (160- / 192-row tiles have no registers left for the row scales: read in place)
  // (round 5: every prefetch load is UNCONDITIONAL, from clamped rows / columns / entries, and masked where it is consumed -- the
  //  `cond ? load : zero` forms made the compiler join the two values right behind the load, an s_waitcnt vmcnt inside the prefetch;
  //  the row scales are 12-byte loads -- the dead fourth component's register was reused while the load was in flight -- and
  //  tile_start is read one tile ahead of the loads it addresses: profiles/experiments/r05_wgrad16h_phase_stamps.txt)
  typedef float f32x3_t __attribute__((ext_vector_type(3)));
  // px: the X rows of ALL the tile's chunks, requested a whole tile ahead (the tile's maximum needs them) -- or (PC, 192 rows: twelve
  // row pieces per thread do not fit) of ONE chunk, requested a chunk ahead, with a maximum and one more barrier per chunk
  constexpr int NPX = PC ? 1 : NRB;
  f32x4 pgw[NRB], px[NPX][2], pgc[2];
  f32x3_t prs[PRS ? NRB : 1];
  int2 pel[NEL];
  const uint32_t gw_col = gcol_ok ? (uint32_t)(4 * (PAIR ? (q16 & 7) : q16)) * 4u : 0u, gc_col = zcol_ok ? (uint32_t)(4 * (PAIR ? (cg & 7) : cg)) * 4u : 0u;
  const float* __restrict__ rs2_t = lyr_w ? rs2B : rs2;                      // the row scales of THIS thread's layer (NULL: a plain layer)
  const float* __restrict__ rsb = RS2 ? (rs2_t ? rs2_t : Gp) : nullptr;      // (layers without row scales: any readable rows; never used)
  auto load_tile_g = [&](int tile, int ts, int R) {
    const char* gb = reinterpret_cast<const char*>(Gw_ + (size_t)ts * p.ldg + gcol0);
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
  auto load_x = [&](int ts, int R, f32x4 (&dst)[2], int c) {
    const char* xb = reinterpret_cast<const char*>(Xc_ + (size_t)ts * p.ldx + xcol0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 32 * c + 2 * rp + u;
      const int rr = r < R ? r : R - 1;
      // X rows beyond the tile's R rows are read from its last row (finite; they only meet zero rows of G / P G / P^2 G)
      dst[u] = *reinterpret_cast<const f32x4*>(xb + (uint32_t)(rr * p.ldx) * 4u + xcb);
    }
  };
  auto load_gc = [&](int ts, int R, int c) {
    const char* gb = reinterpret_cast<const char*>(Gc_ + (size_t)ts * p.ldg + gcol0);
    if (role == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = 32 * c + 2 * rp + u;
        const int rr = r < R ? r : R - 1;
        pgc[u] = *reinterpret_cast<const f32x4*>(gb + (uint32_t)(rr * p.ldg) * 4u + gc_col);
      }
    }
  };
  // the maxima of the tile whose rows wait in px / pgw: one partial per wave, read by everybody behind the next barrier
  auto publish_max = [&]() {
    float mx = 0.f, mg = 0.f;
    if constexpr (!PC) {
#pragma unroll
      for (int c = 0; c < NPX; ++c) { mx = absmax4(mx, px[c][0]); mx = absmax4(mx, px[c][1]); }
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) mg = absmax4(mg, pgw[rb]);
    mx = wave_max(mx); mg = wave_max(mg);
    if (lane == 0) { if (!PC) mxp[wave] = mx; mxp[8 + wave] = mg; }
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
    load_tile_g(slice, ts, R);
#pragma unroll
    for (int c = 0; c < NPX; ++c) load_x(ts, R, px[c], c);
    load_gc(ts, R, 0);
    if (slice + p.n_split < p.ntiles) { ts_n = p.tile_start[slice + p.n_split]; R_n = p.tile_start[slice + p.n_split + 1] - ts_n; }
  }
  for (int tile = slice; tile < p.ntiles; tile += p.n_split) {
    const int nch = (R + 31) >> 5;
    const int next = tile + p.n_split;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      if (!(gcol_ok && r32 + 32 * rb < R)) pgw[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    publish_max();      // (read behind the barrier below)
    // ---- fp32 G of the tile (first hop's input), the ELL slice, bias partial sums
    float* Zf0 = reinterpret_cast<float*>(U + pb * PBUF);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if (rb < nch) *reinterpret_cast<f32x4*>(Zf0 + (r32 + 32 * rb) * LDZF + 4 * q16) = pgw[rb];
      bsum += pgw[rb];
    }
    if constexpr (RS2) {
      if (rs2 || rs2B) {      // (uniform; a thread whose layer has no row scales adds zeros)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRS) { if (rs2_t) d = f32x4{prs[rb][0], prs[rb][1], prs[rb][2], 0.f}; }
          else if (rs2_t && r32 + 32 * rb < R) d = *reinterpret_cast<const f32x4*>(rs2_t + (size_t)(ts + r32 + 32 * rb) * 4);
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
    // ---- this tile's scales: running exponents; the accumulators follow when one grows (dss2_wgrad16h.hip)
    {
      float mx = 0.f, mg = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { if (!PC) mx = fmaxf(mx, mxp[w]); mg = fmaxf(mg, mxp[8 + w]); }
      const int ex = PC ? Ex : __builtin_amdgcn_readfirstlane(exp_of(mx)), eg = __builtin_amdgcn_readfirstlane(exp_of(mg));
      const int nx = ex > Ex ? ex : Ex, ng = eg > Eg ? eg : Eg;
      const int d = (nx - Ex) + (ng - Eg);
      if (d != 0) {      // (uniform)
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] = ldexpf(acc[m][r], -d);
      }
      Ex = nx; Eg = ng;
    }
    int sxe = 14 - Ex;      // (PC: set per chunk)
    const int sze = 14 - hb - Eg;
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
#pragma unroll
    for (int c = 0; c < NRB; ++c) {      // (unrolled: px[c] is a register array; chunks beyond the tile's rows only refill their registers)
      const bool on = c < nch;           // (uniform)
      // ---- planes of the chunk
      char* ZT = U + pb * PBUF;
      char* XT = ZT + NMAT * 2 * ZC * 64;
      f32x4 (&pxc)[2] = px[PC ? 0 : c];
      if constexpr (PC) {
        if (on) {      // the chunk's maximum of X: one partial per wave, read behind barrier A below
          float mx = absmax4(absmax4(0.f, pxc[0]), pxc[1]);
          mx = wave_max(mx);
          if (lane == 0) mxp[wave] = mx;
        }
      }
      const bool rows_on = on && 32 * c + 8 * (wave & 3) < ((R + 15) & ~15);      // (a wave's units are eight rows; rows beyond the last k-step are not read)
      if constexpr (!PC) { if (rows_on) store_planes_h2<XW>(XT, x_off, pxc[0], pxc[1], sxe); }
      if (rows_on) {
        if (role == 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (!(zcol_ok && 32 * c + 2 * rp + u < R)) pgc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          store_planes_h2<ZC>(ZT, z_off, pgc[0], pgc[1], sze);
          if (NMAT > 1) {
            const float* src = Zf1 + (32 * c + 2 * rp) * LDZF + 4 * cg;
            store_planes_h2<ZC>(ZT + 2 * ZC * 64, z_off, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + LDZF), sze);
          }
        } else if (NMAT > 2) {
          f32x4 s[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = 32 * c + 2 * rp + u;
            s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < R) s[u] = hop_row(Zf1, row, 4 * cg);
          }
          store_planes_h2<ZC>(ZT + 2 * 2 * ZC * 64, z_off, s[0], s[1], sze);
        }
      }
      if constexpr (PC) {
        if (on) {
          __syncthreads();      // barrier A: the maxima are in
          float mx = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) mx = fmaxf(mx, mxp[w]);
          const int ex = __builtin_amdgcn_readfirstlane(exp_of(mx));
          if (ex > Ex) {      // (uniform)
            const int d = ex - Ex;
#pragma unroll
            for (int m = 0; m < NMAT; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[m][r] = ldexpf(acc[m][r], -d);
            Ex = ex;
          }
          sxe = 14 - Ex;
          if (rows_on) store_planes_h2<XW>(XT, x_off, pxc[0], pxc[1], sxe);
        }
      }
      if (on) __syncthreads();
      if constexpr (PC) {
        if (on) {      // ONE call site each
          const bool same = c + 1 < nch;
          if (same || next < p.ntiles) { load_x(same ? ts : ts_n, same ? R : R_n, pxc, same ? c + 1 : 0); load_gc(same ? ts : ts_n, same ? R : R_n, same ? c + 1 : 0); }
        }
      } else {
        // the NEXT TILE's rows of this chunk take the registers this chunk's rows have just left (one call site per chunk)
        if (next < p.ntiles) load_x(ts_n, R_n, pxc, c);
        if (on) {      // ONE call site (two made the compiler load into temporaries and join them behind an s_waitcnt vmcnt(0))
          const bool same = c + 1 < nch;
          if (same || next < p.ntiles) load_gc(same ? ts : ts_n, same ? R : R_n, same ? c + 1 : 0);
        }
      }
      // ---- MFMA phase: up to 2 steps of 16 rows; lo hi + hi lo + hi hi, smallest terms first
      if (on && in_active) {
        const int left = R - 32 * c;
        const int nsteps = left > 16 ? 2 : 1;
        const int xc = ibw * 32 + c32, zc = obh * 32 + c32;
        const int xkey = tph2_key(xc), zkey = tph2_key(zc);
        for (int ks = 0; ks < nsteps; ++ks) {
          const int ch = 2 * ks + half;
          const int choff = xc * 64 + ((ch ^ xkey) << 4);
          const f16x8 bh = *reinterpret_cast<const f16x8*>(XT + choff);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(XT + XW * 64 + choff);
          const int zoff = zc * 64 + ((ch ^ zkey) << 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) {
            const char* zi = ZT + m * 2 * ZC * 64 + zoff;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(zi);
            const f16x8 al = *reinterpret_cast<const f16x8*>(zi + ZC * 64);
            f32x16 cc = acc[m];
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, cc, 0, 0, 0);
            acc[m] = cc;
          }
        }
      }
      if (on) pb ^= 1;          // the next chunk (or the next tile's fp32 G) writes the other set: no barrier here
    }
    ts = ts_n; R = R_n;
    if (next + p.n_split < p.ntiles) { ts_n = p.tile_start[next + p.n_split]; R_n = p.tile_start[next + p.n_split + 1] - ts_n; }
  }

  // ---- one slab per tile-list slice blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  const size_t strideB = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2B ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)slice * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  float* outB = slabB + (size_t)slice * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : strideB);      // (PAIR: the second layer's slab)
  const int fin = Ex + Eg + hb - 28;      // acc = 2^(14 - Ex) 2^(14 - hb - Eg) dW
  if (in_active) {
    const int i = PAIR ? c32 : xcol0 + ibw * 32 + c32;
    float* ow = (PAIR && obh) ? outB : out;
    if (i < p.hin) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = PAIR ? acc_row(r, half) : gcol0 + obh * 32 + acc_row(r, half);
          if (o < p.hout) ow[((size_t)m * p.hout + o) * p.hin + i] = ldexpf(acc[m][r], fin);
        }
    }
  }
  if (ibg == 0) {   // (uniform) column sums: the 32 threads that share a column group meet in LDS, fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [1 + NMAT][NT]
    const int nsum = (rs2 || rs2B) ? 1 + NMAT : 1;
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
      if constexpr (PAIR) {      // columns 0..31: the first layer, 32..63: the second (its scaled sums only where it has row scales)
        const int lb = col >> 5, o = col & 31;
        float* ob = lb ? outB : out;
        if (o < p.hout && (lb == 0 || hasB) && (which == 0 || (lb ? rs2B : rs2) != nullptr))
          ob[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
        continue;
      }
      const int o = gcol0 + col;
      if (o < p.hout) out[(size_t)p.nmat * p.hout * p.hin + (which == 0 ? 0 : p.hout + (size_t)(which - 1) * p.hout) + o] = s;
    }
  }
}


size_t wgrad16th_lds_bytes(int nrb, int nmat, int ell_width) {
  const size_t fimg = (size_t)32 * nrb * W16TH_LDZF * 4, planes = (size_t)nmat * 2 * W16TH_ZC * 64 + 2 * (size_t)W16TH_XW * 64;
  const size_t b = fimg + 2 * (fimg > planes ? fimg : planes) + (size_t)((ell_width + 3) & ~3) * 32 * nrb * 8 + 64;      // (+ the maxima)
  const size_t red = (size_t)(1 + nmat) * W16TH_NT * 16;
  return b > red ? b : red;
}

// args.mfma_bf16 & 255 == 2 on 96- .. 192-row tiles (96 rows: X a tile ahead; 128 .. 192 rows: the per-chunk form).  hout = 32 is covered HERE (the
// bf16x6 kernels start above 32): one wave of eight multiplies, but the propagation, the splits and the prefetch are what a tile costs at
// that width -- the driver's model on ober_sub (dim_hid 32, 7 layers per launch): 206 -> ~150 us per launch against the fp32 kernel,
// replayed step 2.63 -> 2.37 ms
bool wgrad16th_covers(const dss2_wgrad_args& a) {
  static const int on = [] { const char* e = getenv("DSS2_WGRAD_TALL_F16"); return e ? atoi(e) : 1; }();
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return on && (a.mfma_bf16 & 255) == 2 && a.nrb >= 3 && a.nrb <= 6 && (a.nmat == 2 || a.nmat == 3) && !a.narrow && !a.rowscale && a.ell_tiles &&
         al16(a.G) && al16(a.X) && (a.ldg & 3) == 0 && (a.ldx & 3) == 0 && (!a.rowscale2 || al16(a.rowscale2)) && a.ell_width >= 1 &&
         a.ell_width <= W16TH_DMAX && a.hout >= 32 && (a.hout & 3) == 0 && (a.hin & 3) == 0 &&
         wgrad16th_lds_bytes(a.nrb, a.nmat, a.ell_width) <= (size_t)kMaxLdsBytes;
}

template <int NRB, int NMAT, bool RS2, bool PC, bool PAIR = false>
static int launch16th(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16th_kernel<NRB, NMAT, RS2, PC, PAIR>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(f16x3, tall tiles)")) return 1;
  const int nobg = (a.hout + W16TH_ZC - 1) / W16TH_ZC, nibg = (a.hin + W16TH_XW - 1) / W16TH_XW;
  const int hb = (a.mfma_bf16 >> 8) & 255;
  const int nz = wb.n > 0 ? (PAIR ? (wb.n + 1) / 2 : wb.n) : 1;
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, nz), dim3(W16TH_NT), wgrad16th_lds_bytes(a.nrb, a.nmat, a.ell_width), stream, a, nibg, wb, hb);
  return check_launch("wgrad(f16x3, tall tiles)");
}

int launch_wgrad16th(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  bool rs2 = a.rowscale2 != nullptr;
  for (int l = 0; l < wb.n; ++l) {
    if ((reinterpret_cast<uintptr_t>(wb.G[l]) | reinterpret_cast<uintptr_t>(wb.X[l]) | reinterpret_cast<uintptr_t>(wb.rowscale2[l])) & 15) {
      set_error("wgrad(f16x3, tall tiles): layer %d has a misaligned operand", l); return 2;
    }
    rs2 = rs2 || wb.rowscale2[l] != nullptr;
  }
  // two layers per workgroup where a layer is one 32-column block (DSS2_WGRAD_TALL_PAIR=0: one layer per workgroup as before)
  static const int pair_on = [] { const char* e = getenv("DSS2_WGRAD_TALL_PAIR"); return e ? atoi(e) : 1; }();
  if (pair_on && a.nrb == 3 && a.hout == 32 && a.hin == 32 && wb.n >= 2) {
    if (a.nmat == 2) return rs2 ? launch16th<3, 2, true, false, true>(a, stream, wb) : launch16th<3, 2, false, false, true>(a, stream, wb);
    return rs2 ? launch16th<3, 3, true, false, true>(a, stream, wb) : launch16th<3, 3, false, false, true>(a, stream, wb);
  }
#define DSS2_TALLH(NRB, PC) \
  if (a.nrb == NRB) { \
    if (a.nmat == 2) return rs2 ? launch16th<NRB, 2, true, PC>(a, stream, wb) : launch16th<NRB, 2, false, PC>(a, stream, wb); \
    return rs2 ? launch16th<NRB, 3, true, PC>(a, stream, wb) : launch16th<NRB, 3, false, PC>(a, stream, wb); \
  }
  DSS2_TALLH(3, false) DSS2_TALLH(4, true) DSS2_TALLH(5, true) DSS2_TALLH(6, true)
#undef DSS2_TALLH
  set_error("wgrad(f16x3, tall tiles): no kernel for nrb=%d", a.nrb);
  return 2;
}

}  // namespace dss2
