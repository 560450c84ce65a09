// Weight gradient of TAGConv on the fp16 matrix pipe, fp32-grade: "f16x3" (round 5).
//   dW_m = (P^m G)^T X,  db = colsum(G)   -- contract, slab layout, thread maps and phases of wgrad16b_kernel (dss2_wgrad16.hip).
//
// bf16x6 writes every operand as three bf16 pieces (8 + 8 + 8 bits) and needs six MFMAs per product; the matrix pipe does fp16
// products at the same rate, and an fp32 value is TWO fp16 pieces, hi = fp16(x), lo = fp16(x - hi) (11 + 11 bits and lo's sign:
// |x - hi - lo| <= 2^-23 |x|), so a product is THREE MFMAs: lo hi + hi lo + hi hi, the dropped lo lo term 2^-22 of the product --
// errors of the size of fp32 arithmetic itself (tools/micro/f16x3_probe.hip, tools/accuracy_bf16x6.py; bf16x6 sits ~10x below).
// What fp16 lacks is RANGE (2^-14 .. 65504 normal), so operands are scaled by powers of two -- exact -- before the split:
//   * per workgroup a running exponent Ex of max |X| and Eg of max |G| over the tiles it has walked (a tile's maxima are formed
//     from its rows while they wait in registers one tile ahead, exchanged through four LDS words at the barrier that is there
//     anyway); X is scaled to [2^14, 2^15) at its maximum, G, P G, P^2 G by 2^(14 - hb - Eg), hb = headroom bits for the
//     propagation's gain (host: ceil(log2(max row sum of |P^T| ^ (K)))), carried in args.mfma_bf16 bits 8..15);
//   * when a running exponent grows (a few times per workgroup at most) the accumulators are rescaled, exactly (v_ldexp);
//   * the slab is written with the scales taken out (v_ldexp).
// Elements down to 2^-18 (2^-(18 - hb) for G) of the running maximum keep all 22 bits; below that the absolute error stays at
// 2^-40 of the maximum (fp16 subnormals; the matrix pipe honours them: f16x3_probe).  The scales depend only on the data and the
// launch geometry: same inputs, same bits.  Inf / NaN inputs give Inf / NaN outputs as they should (the maxima ignore NaN).
// LDS: 2 x 8 KB fp32 G / P G + 24 KB planes of G, P G, P^2 G + 16 KB planes of X + ELL = 58 KB, two workgroups per CU.
// 36 MFMAs per tile and wave instead of 72, 7 VALU per split pair instead of 11.
// Built without packed fp32 VALU ops like the other MFMA-beside-VALU translation units (build.sh).
#include <stdlib.h>

#include "dss2_wgrad_batch.hpp"

namespace dss2 {

#ifdef DSS2_STAMPS
// Diagnostic build only (-DDSS2_STAMPS; tools/stamps.py wgradh): per-wave phase stamps of the THIRD tile of every workgroup's walk.
__device__ unsigned long long g_hstamps[512 * 4 * 16];
#define HSTAMP(slot)                                                                                   \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      unsigned long long t_;                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if (lane == 0 && blockIdx.x < 256 && blockIdx.z == 1) g_hstamps[((blockIdx.x * 2 + (blockIdx.y & 1)) * 4 + wv) * 16 + (slot)] = t_; \
    }                                                                                                  \
  } while (0)
#else
#define HSTAMP(slot) do {} while (0)
#endif

constexpr int W16H_TM = 32, W16H_ZC = 64, W16H_XW = 128, W16H_NT = 256, W16H_LDZF = 64, W16H_DMAX = 8;
// (the transposed image of wgrad16b_kernel: 64 bytes per column and plane, 16-byte chunks of 8 rows swizzled)
__device__ __forceinline__ int tph_key(int col) { return (((col >> 3) & 1) << 1) | ((col >> 4) & 1); }
__device__ __forceinline__ int tph_off(int col, int row) { return col * 64 + ((((row >> 3) ^ tph_key(col)) << 4) | ((row & 7) << 1)); }

// rows (2 rp, 2 rp + 1) x columns (c0 .. c0+3) scaled by 2^e -> the two planes of a transposed image with NCOLS columns
template <int NCOLS>
__device__ __forceinline__ void store_planes_h(char* img, int off0, const f32x4 v0, const f32x4 v1, int e) {
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

template <int NMAT, bool RS2>
__global__ void __launch_bounds__(W16H_NT, 2) wgrad16h_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb, int hb) {
  constexpr int TM = W16H_TM, ZC = W16H_ZC, XW = W16H_XW, NT = W16H_NT, LDZF = W16H_LDZF;
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[blockIdx.z] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[blockIdx.z] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[blockIdx.z] : p.slab;
  const float* __restrict__ rs2 = RS2 ? (wb.n > 0 ? wb.rowscale2[blockIdx.z] : p.rowscale2) : nullptr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zf0 = smem;
  float* Zf1 = Zf0 + TM * LDZF;
  char* ZT = reinterpret_cast<char*>(Zf1 + TM * LDZF);          // [NMAT][2 planes][ZC columns][64 B]
  char* XT = ZT + NMAT * 2 * ZC * 64;                             // [2 planes][XW columns][64 B]
  int2* ell = reinterpret_cast<int2*>(XT + 2 * XW * 64);         // [D][TM]
  float* mxp = reinterpret_cast<float*>(ell + W16H_DMAX * TM);   // [2][4 waves]: max |X|, max |G| of the tile in the registers
  const int D = p.ell_width;

  const int tid = threadIdx.x;
  const int lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xb0 = 2 * (wv & 1), obw = wv >> 1;
  const int slice = blockIdx.x, ysl = blockIdx.y;
  const int obg = ysl / nibg, ibg = ysl - obg * nibg;
  const int gcol0 = obg * ZC, xcol0 = ibg * XW;
  const bool x_on[2] = {(xcol0 + xb0 * 32) < p.hin, (xcol0 + (xb0 + 1) * 32) < p.hin};
  const bool in_active = x_on[0] && (gcol0 + obw * 32) < p.hout;

  const int cg = tid & 15, rp = (tid >> 4) ^ ((tid & 1) << 1);
  const int g_off0 = tph_off(4 * cg, 2 * rp);
  const int x_off0[2] = {tph_off(4 * cg, 2 * rp), tph_off(64 + 4 * cg, 2 * rp)};
  const uint32_t g_goff = (uint32_t)((2 * rp) * p.ldg + 4 * cg) * 4u;

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
  int Ex = -100, Eg = -100;      // running exponents of max |X|, max |G| (uniform)

  // (row scales: a 12-byte load -- of a 16-byte one the fourth component is dead, the register allocator hands its register to the
  //  hop's address arithmetic while the load is in flight, and the write-after-write wait stalls the hop for the whole prefetch)
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  f32x4 pg[2], px[4];
  f32x3 prs[RS2 ? 2 : 1];
  int2 pel;
  // The next tile's rows, requested one tile ahead.  Every load is UNCONDITIONAL, from clamped rows / columns, and masked where it is
  // consumed: with `cond ? load : 0` forms the compiler joined the loaded and the zero value right behind the load -- a vmcnt wait in
  // the middle of the prefetch, ~2400 cycles per tile (phase stamps: profiles/experiments/r05_wgrad16h_phase_stamps.txt).
  const uint32_t g_col = (gcol0 + 4 * cg < p.hout) ? (uint32_t)(4 * cg) * 4u : 0u;      // (columns beyond hout: column group 0, masked below)
  const bool g_ok = gcol0 + 4 * cg < p.hout;
  const uint32_t x_col[2] = {(xcol0 + 4 * cg < p.hin) ? (uint32_t)(4 * cg) * 4u : 0u, (xcol0 + 64 + 4 * cg < p.hin) ? (uint32_t)(64 + 4 * cg) * 4u : 0u};
  const float* __restrict__ rsb = RS2 ? (rs2 ? rs2 : Gp) : nullptr;      // (layers without row scales: any readable rows; never used)
  auto load_tile = [&](int tile, int ts, int R) {
    const char* gb = reinterpret_cast<const char*>(Gp + (size_t)ts * p.ldg + gcol0);      // uniform
    const char* xb = reinterpret_cast<const char*>(Xp + (size_t)ts * p.ldx + xcol0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 2 * rp + u;
      const int rr = r < R ? r : R - 1;      // rows beyond the tile's R rows: its last row (finite values; G's are zeroed at consumption)
      pg[u] = *reinterpret_cast<const f32x4*>(gb + (uint32_t)(rr * p.ldg) * 4u + g_col);
      const uint32_t xro = (uint32_t)(rr * p.ldx) * 4u;
#ifdef DSS2_ABLATE_XLOAD      // (diagnostic: X is never read)
#pragma unroll
      for (int i = 0; i < 2; ++i) px[2 * i + u] = f32x4{1.f, 1.f, 1.f, 1.f};
      (void)xro; (void)xb;
#else
#pragma unroll
      for (int i = 0; i < 2; ++i) px[2 * i + u] = *reinterpret_cast<const f32x4*>(xb + xro + x_col[i]);
#endif
      if constexpr (RS2) prs[u] = *reinterpret_cast<const f32x3*>(rsb + (size_t)(ts + rr) * 4);
    }
    pel = tid < D * TM ? (reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM)[tid] : make_int2(tid & (TM - 1), 0);
  };
  // the maxima of the tile whose rows wait in px / pg: one partial per wave, read by everybody after the next barrier
  auto publish_max = [&]() {
    float mx = 0.f, mg = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = absmax4(mx, px[i]);
#pragma unroll
    for (int u = 0; u < 2; ++u) mg = absmax4(mg, pg[u]);
    mx = wave_max(mx); mg = wave_max(mg);
    if (lane == 0) { mxp[wv] = mx; mxp[4 + wv] = mg; }
  };
  auto prop = [&](const float* Zs, float* Zd, char* img, int ez) {
    f32x4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 2 * rp + u;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < D; k0 += 4) {
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
    store_planes_h<ZC>(img, g_off0, s[0], s[1], ez);
  };

  // tile_start of the current tile (ts, R) and of the next one (ts_n, R_n: scalar loads, a tile ahead of the row loads they address)
  int ts = 0, R = 0, ts_n = 0, R_n = 0;
  if (slice < p.ntiles) {
    ts = p.tile_start[slice]; R = p.tile_start[slice + 1] - ts;
    load_tile(slice, ts, R);
    publish_max();
    if (slice + p.n_split < p.ntiles) { ts_n = p.tile_start[slice + p.n_split]; R_n = p.tile_start[slice + p.n_split + 1] - ts_n; }
  }
  __syncthreads();
  for (int tile = slice; tile < p.ntiles; tile += p.n_split) {
    const int next = tile + p.n_split;
#ifdef DSS2_STAMPS
    const bool stamp_on = tile == slice + 2 * p.n_split;
#endif
    HSTAMP(0);
    // ---- this tile's scales: running exponents; the accumulators follow when one grows
    {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(mxp), m1 = *reinterpret_cast<const f32x4*>(mxp + 4);
      const int ex = __builtin_amdgcn_readfirstlane(exp_of(fmaxf(fmaxf(m0[0], m0[1]), fmaxf(m0[2], m0[3]))));
      const int eg = __builtin_amdgcn_readfirstlane(exp_of(fmaxf(fmaxf(m1[0], m1[1]), fmaxf(m1[2], m1[3]))));
      const int nx = ex > Ex ? ex : Ex, ng = eg > Eg ? eg : Eg;
      const int d = (nx - Ex) + (ng - Eg);
      if (d != 0) {      // (uniform)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb)
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[xb][m][r] = ldexpf(acc[xb][m][r], -d);
      }
      Ex = nx; Eg = ng;
    }
    const int sxe = 14 - Ex, sze = 14 - hb - Eg;
    HSTAMP(1);
    // ---- planes of X and G, the fp32 G slab (first hop's input), the ELL slice, bias partial sums
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (!(g_ok && 2 * rp + u < R)) pg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef DSS2_ABLATE_XSPLIT      // (diagnostic builds, tools/wgrad_ablations.sh: what the split of X costs -- the planes then hold whatever LDS held)
#pragma unroll
    for (int i = 0; i < 2; ++i) store_planes_h<XW>(XT, x_off0[i], px[2 * i], px[2 * i + 1], sxe);
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(Zf0 + (2 * rp + u) * LDZF + 4 * cg) = pg[u];
    store_planes_h<ZC>(ZT, g_off0, pg[0], pg[1], sze);
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
    HSTAMP(2);
    __syncthreads();
    HSTAMP(3);
    if (next < p.ntiles) load_tile(next, ts_n, R_n);      // in flight across the hops and the MFMA phase
    HSTAMP(4);
    // ---- P G, P^2 G
#ifdef DSS2_ABLATE_HOPS      // (diagnostic: no propagation hops -- their planes hold whatever LDS held)
    if (false) {
#else
    if (NMAT > 1) {
#endif
      prop(Zf0, NMAT > 2 ? Zf1 : nullptr, ZT + 2 * ZC * 64, sze);
      HSTAMP(5);
      if (NMAT > 2) {
        __syncthreads();
        HSTAMP(6);
        prop(Zf1, nullptr, ZT + 2 * 2 * ZC * 64, sze);
        HSTAMP(7);
      }
      __syncthreads();
      HSTAMP(8);
    }
    // ---- MFMA phase: 2 steps of 16 rows; lo hi + hi lo + hi hi, smallest terms first
#ifdef DSS2_ABLATE_MFMA      // (diagnostic: no matrix instructions)
    if (false) {
#else
    if (in_active) {
#endif
      const int nsteps = (R + 15) >> 4;
      const int zc = obw * 32 + c32;
      const int zkey = tph_key(zc);
      const int xc0 = xb0 * 32 + c32, xc1 = xc0 + 32;
      const int xkey0 = tph_key(xc0), xkey1 = tph_key(xc1);
      for (int ks = 0; ks < nsteps; ++ks) {
        const int ch = 2 * ks + half;
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          const int choff = (xb ? xc1 : xc0) * 64 + ((ch ^ (xb ? xkey1 : xkey0)) << 4);
          bh[xb] = *reinterpret_cast<const f16x8*>(XT + choff);
          bl[xb] = *reinterpret_cast<const f16x8*>(XT + XW * 64 + choff);
        }
        const int zoff = zc * 64 + ((ch ^ zkey) << 4);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) {
          const char* zi = ZT + m * 2 * ZC * 64 + zoff;
          const f16x8 ah = *reinterpret_cast<const f16x8*>(zi);
          const f16x8 al = *reinterpret_cast<const f16x8*>(zi + ZC * 64);
#pragma unroll
          for (int xb = 0; xb < 2; ++xb) {
            if (xb == 1 && !x_on[1]) continue;      // (uniform)
            f32x16 c = acc[xb][m];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[xb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[xb], c, 0, 0, 0);
            acc[xb][m] = c;
          }
        }
      }
    }
    HSTAMP(9);
    if (next < p.ntiles) publish_max();      // (the next tile's rows have been in flight since the hops)
    ts = ts_n; R = R_n;
    if (next + p.n_split < p.ntiles) { ts_n = p.tile_start[next + p.n_split]; R_n = p.tile_start[next + p.n_split + 1] - ts_n; }
    HSTAMP(10);
    __syncthreads();          // planes, fp32 slabs and the maxima are free / valid for the next tile
    HSTAMP(11);
  }

  // ---- one slab per tile-list slice blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)slice * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  const int fin = Ex + Eg + hb - 28;      // acc = 2^(14 - Ex) 2^(14 - hb - Eg) dW
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
            if (o < p.hout) out[((size_t)m * p.hout + o) * p.hin + i] = ldexpf(acc[xb][m][r], fin);
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

size_t wgrad16h_lds_bytes(int nmat, int ell_width) {
  const size_t b = 2 * (size_t)W16H_TM * W16H_LDZF * 4 + (size_t)nmat * 2 * W16H_ZC * 64 + 2 * (size_t)W16H_XW * 64 + (size_t)W16H_DMAX * W16H_TM * 8 + 32;
  const size_t red = (size_t)(1 + nmat) * W16H_NT * 16;
  (void)ell_width;
  return b > red ? b : red;
}

bool wgrad16h_covers(const dss2_wgrad_args& a) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return (a.mfma_bf16 & 255) == 2 && a.nrb == 1 && (a.nmat == 2 || a.nmat == 3) && !a.narrow && !a.rowscale && a.ell_tiles && al16(a.G) && al16(a.X) &&
         (a.ldg & 3) == 0 && (a.ldx & 3) == 0 && (!a.rowscale2 || al16(a.rowscale2)) && a.ell_width >= 1 && a.ell_width <= W16H_DMAX &&
         a.hout > 32 && (a.hout & 3) == 0 && (a.hin & 3) == 0;
}

template <int NMAT, bool RS2>
static int launch16h(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad16h_kernel<NMAT, RS2>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad(f16x3, 32 rows)")) return 1;
  const int nobg = (a.hout + W16H_ZC - 1) / W16H_ZC, nibg = (a.hin + W16H_XW - 1) / W16H_XW;
  const int hb = (a.mfma_bf16 >> 8) & 255;
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, wb.n > 0 ? wb.n : 1), dim3(W16H_NT), wgrad16h_lds_bytes(a.nmat, a.ell_width), stream, a, nibg, wb, hb);
  return check_launch("wgrad(f16x3, 32 rows)");
}

int launch_wgrad16h(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  bool rs2 = a.rowscale2 != nullptr;
  for (int l = 0; l < wb.n; ++l) {
    if ((reinterpret_cast<uintptr_t>(wb.G[l]) | reinterpret_cast<uintptr_t>(wb.X[l]) | reinterpret_cast<uintptr_t>(wb.rowscale2[l])) & 15) {
      set_error("wgrad(f16x3): layer %d has a misaligned operand", l); return 2;
    }
    rs2 = rs2 || wb.rowscale2[l] != nullptr;
  }
  if (((a.mfma_bf16 >> 8) & 255) > 10) { set_error("wgrad(f16x3): %d headroom bits for the propagation leave no precision", (a.mfma_bf16 >> 8) & 255); return 2; }
  if (a.nmat == 2) return rs2 ? launch16h<2, true>(a, stream, wb) : launch16h<2, false>(a, stream, wb);
  if (a.nmat == 3) return rs2 ? launch16h<3, true>(a, stream, wb) : launch16h<3, false>(a, stream, wb);
  set_error("wgrad(f16x3): unsupported nmat=%d", a.nmat);
  return 2;
}

}  // namespace dss2

#ifdef DSS2_STAMPS
extern "C" int dss2_debug_read_hstamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dss2::g_hstamps), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
