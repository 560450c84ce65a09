// Weight gradient of TAGConv / Linear on gfx950:  dW_m = (P^m G)^T X,  db = colsum(G).
//
// Persistent workgroups walk the tile list (tile = whole graphs, <= 32*NRB rows).  Per tile the
// G slab (NB*32 output-feature columns) and the X slab (128 input-feature columns) are staged in
// LDS; P^m G is produced in LDS by a CSR segmented sum over the transposed graph (ping-pong
// buffers, no atomics), and every wave accumulates its 32x32 blocks of dW with fp32 MFMA
// (32x32x2; A = Z^T read column-wise from LDS, B = X) in registers ACROSS tiles.  One slab of
// partial sums per workgroup is written at the end; dss2_reduce_slabs adds them in fixed order.
#include <stdlib.h>

#include "dss2_common.hpp"
#include "dss2_wgrad_batch.hpp"

namespace dss2 {

constexpr int XW = 128;  // X columns per workgroup: one 32-column block per wave

// NB >= 2: 8 waves (two per SIMD).  Wave w multiplies input-feature block (w & 3) against the
// output-feature blocks of half (w >> 2).  Half 0 runs "MFMA, then its share of the propagation",
// half 1 "bias sums + propagation share, then MFMA", so on every SIMD the VALU/LDS propagation of
// one wave sits under the MFMAs of the other.  NB == 1 (narrow / tiny outputs): 4 waves.
#ifdef DSS2_STAMPS
// Diagnostic build only (-DDSS2_STAMPS): per-wave phase stamps of the SECOND tile of every workgroup.
__device__ unsigned long long g_wstamps[512 * 8 * 16];
#define WSTAMP(slot)                                                                                   \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      unsigned long long t_;                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      if (lane == 0 && blockIdx.x < 512 && blockIdx.z == 0) g_wstamps[(blockIdx.x * 8 + wave) * 16 + (slot)] = t_; \
    }                                                                                                  \
  } while (0)
#else
#define WSTAMP(slot) do {} while (0)
#endif

// propagate-first schedule (three G slabs in LDS): K = 2 on tiles of at most 64 rows
constexpr bool wgrad_pf(int nrb, int nmat) { return nmat == 3 && nrb <= 2; }

// grouped operand reads in the MFMA loop (see mma): where the register budget has room
// (the four excluded instantiations sit at the 256-register limit already and would spill 1-12 registers)
constexpr bool wgrad_fast_mma(int nrb, int nmat, int nb) {
  return !((nrb == 2 && nmat == 3 && nb == 4) || (nrb == 3 && nmat == 2 && nb == 2) || (nrb == 4 && nmat == 4 && nb == 2) ||
           (nrb == 4 && nmat == 1 && nb == 4));
}

// four-entries-at-a-time propagation (see prop): five instantiations have no room for its 24-48 transient registers
constexpr bool wgrad_prop4(int nrb, int nmat, int nb) {
  return !((nrb == 2 && nmat == 2 && (nb == 2 || nb == 4)) || (nrb == 3 && nmat == 2 && nb == 2) || (nrb == 4 && (nmat == 3 || nmat == 4) && nb == 2));
}

// W8 (tall tiles, NB == 1): eight waves all the same -- the two halves split the k steps (rows) of every MFMA phase instead of the
// output blocks, so that, as with NB >= 2, the propagation / staging of one wave of a SIMD sits under the MFMAs of the other.
// The 4-wave form runs one wave per SIMD there: 18.4 K cycles of MFMAs per 192-row tile out of 28 K (tools/wstamps.py).
template <int NB, bool W8 = false> struct WgradGeom {
  static constexpr int NW = (NB >= 2 || W8) ? 8 : 4;
  static constexpr int NT = NW * 64;
  static constexpr int NBW = NB >= 2 ? NB / 2 : 1;   // output blocks per wave
};


template <int NRB, int NMAT, int NB, bool W8 = false>
__global__ void __launch_bounds__((WgradGeom<NB, W8>::NT)) wgrad_kernel(const dss2_wgrad_args p, int nibg, const WgradBatch wb, int ksplit) {
  const float* __restrict__ Gp = wb.n > 0 ? wb.G[blockIdx.z] : p.G;
  const float* __restrict__ Xp = wb.n > 0 ? wb.X[blockIdx.z] : p.X;
  float* __restrict__ slabp = wb.n > 0 ? wb.slab[blockIdx.z] : p.slab;
  const float* __restrict__ rs2 = wb.n > 0 ? wb.rowscale2[blockIdx.z] : p.rowscale2;
  constexpr int TM = NRB * 32;
  constexpr int LDZ = NB * 32;
  constexpr int NW = WgradGeom<NB, W8>::NW, NT = WgradGeom<NB, W8>::NT, NBW = WgradGeom<NB, W8>::NBW;
  static_assert(!W8 || NB == 1, "W8 is the eight-wave form of the one-output-block kernel");
  constexpr int NG4 = TM * LDZ / 4 / NT;   // float4 of the G slab per thread
  constexpr int NX4 = TM * XW / 4 / NT;    // float4 of the X slab per thread
  static_assert(NG4 * NT * 4 == TM * LDZ && NX4 * NT * 4 == TM * XW, "slabs must tile the threads");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // PF ("propagate first"): all (P^T)^m G slabs side by side in LDS, both propagations up front at full VALU
  // speed, then ONE uninterrupted MFMA phase over the three slabs.  Phase stamps of the interleaved schedule
  // (tools/wstamps.py) showed MFMAs at 70 % and the propagation 3x slower when they share a SIMD, plus two
  // extra barriers per tile; PF needs a third slab, which fits for the hot shapes.
  constexpr bool PF = wgrad_pf(NRB, NMAT);
  float* Za = smem;
  float* Zb = Za + TM * LDZ;
  float* Zc = Zb + TM * LDZ;                                  // only with PF
  float* Xs = Za + (PF ? NMAT : (NMAT > 1 ? 2 : 1)) * TM * LDZ;
  const int D = p.ell_width;   // > 0: ELL [D][TM] slice of the transposed graph, else CSR slice
  // X slab row length: XW (four 32-column blocks, one per wave) -- except in the 4-wave kernels (NB == 1: narrow
  // hidden widths), where it shrinks to the input width so that more workgroups fit a CU (H = 32: 8 KB, not 32)
  const int xw = (NB == 1) ? min(XW, ((p.hin + 31) >> 5) << 5) : XW;
  f32x4* Dsc = reinterpret_cast<f32x4*>(Xs + TM * xw);   // [TM] row scales of the tile (rowscale2)
  f32x4* Bsum = reinterpret_cast<f32x4*>(Xs + TM * xw + TM * 4);   // [NT] running partial column sums (fast_bias)
  // (W8: no running bias partials -- the tall tile leaves no LDS for 512 more 16-byte slots; the per-tile pass runs instead)
  int2* ell = reinterpret_cast<int2*>(Xs + TM * xw + TM * 4 + (W8 ? 0 : NT * 4));
  int* lrow = reinterpret_cast<int*>(Xs + TM * xw + TM * 4 + (W8 ? 0 : NT * 4));
  int2* lent = reinterpret_cast<int2*>(lrow + TM + 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2;                 // which half of the workgroup (phase order, see phase())
  const int obh = NB >= 2 ? role : 0;         // output-block half (NB >= 2)
  const int obg = blockIdx.y / nibg, ibg = blockIdx.y - obg * nibg;
  const int gcol0 = obg * LDZ;
  const int xcol0 = ibg * XW;
  // K split (4-wave kernels, narrow inputs): with one or two 32-column input blocks only one or two of the four waves
  // would own an accumulator (H = 32: 96 MFMAs per tile on ONE wave).  The idle waves take a share of the tile's rows
  // (k steps j = ks, ks + KS, ..) for the same output block instead; the partial accumulators meet once, after the last
  // tile, in a fixed order through LDS.
  const int nib_act = min(4, (p.hin - xcol0 + 31) >> 5);
  constexpr int KSC = W8 ? 2 : 1;             // W8: the halves split the k steps, whatever the input width
  const int KS = W8 ? 2 : ((NB == 1 && ksplit) ? (nib_act == 1 ? 4 : (nib_act == 2 ? 2 : 1)) : 1);
  const int ibw = W8 ? (wave & 3) : (KS > 1 ? (wave & 3) % nib_act : (wave & 3));
  const int ks = W8 ? role : (KS > 1 ? (wave & 3) / nib_act : 0);
  const bool wave_active = (xcol0 + ibw * 32) < p.hin;

  f32x16 acc[NMAT][NBW];
#pragma unroll
  for (int m = 0; m < NMAT; ++m)
#pragma unroll
    for (int ob = 0; ob < NBW; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][ob][r] = 0.f;
  float dbacc = 0.f;
  float dbs[NMAT];   // scaled column sums per matrix (rowscale2), same threads as dbacc
#pragma unroll
  for (int m = 0; m < NMAT; ++m) dbs[m] = 0.f;

  const bool gvec = ((p.ldg & 3) == 0) && ((p.hout & 3) == 0) && ((reinterpret_cast<uintptr_t>(Gp) & 15) == 0);
  const bool xvec = ((p.ldx & 3) == 0) && ((p.hin & 3) == 0) && ((reinterpret_cast<uintptr_t>(Xp) & 15) == 0);
  // Plain bias gradient (column sums of G) without a per-tile pass: a thread's prefetched 16-byte pieces of the G
  // slab all sit in the same four columns (NT is a multiple of the row length), so it keeps a running partial sum
  // of those columns across ALL its tiles (in its own LDS slot: the register file is full); the partials meet
  // once, after the last tile.
  const bool fast_bias = gvec && !p.rowscale && !rs2 && ibg == 0 && (NT % (LDZ / 4) == 0);
  f32x4 bsum_reg = {0.f, 0.f, 0.f, 0.f};      // W8: the running partial sums in registers (no LDS left for 512 slots)
  if (fast_bias && !W8) Bsum[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr bool FASTMMA = wgrad_fast_mma(NRB, NMAT, NB);
  auto mma = [&](const float* Z, f32x16 (&a)[NBW], int R) {
    if (!wave_active) return;
    const int n2a = (R + 1) >> 1;
    const int n2e = n2a > ks ? (n2a - ks + KS - 1) / KS : 0;      // this wave's share of the k steps
    if (n2e == 0) return;
    const float* zp = Z + (half + 2 * ks) * LDZ + obh * NBW * 32 + c32;
    const float* xp = Xs + (half + 2 * ks) * xw + ibw * 32 + c32;
    const int zstep = KS * 2 * LDZ, xstep = KS * 2 * xw;
    // two operand register sets in ping-pong: the LDS reads of step n2+1 are in flight while the
    // MFMAs of step n2 issue (no register copies => the wait sits at the first use)
    float b0, b1, a0[NBW], a1[NBW];
    auto ld = [&](float& b, float (&av)[NBW], int n2) {
      b = xp[n2 * xstep];
#pragma unroll
      for (int ob = 0; ob < NBW; ++ob) av[ob] = zp[n2 * zstep + ob * 32];
    };
    auto mm = [&](float b, const float (&av)[NBW]) {
#pragma unroll
      for (int ob = 0; ob < NBW; ++ob) a[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ob], b, a[ob], 0, 0, 0);
    };
    int n2 = 0;
    if constexpr (FASTMMA) {
      // Groups of four k steps with compile-time strides (no K split, full-width X slab): the operand reads take
      // immediate offsets, so a step is 2 ds_read_b32 + 1 MFMA instead of ~14 instructions of address arithmetic,
      // clamping and waits -- one wave per SIMD issues ~4 cycles per instruction and the rolled loop ran at ~120
      // cycles per 64-cycle MFMA (tools/wstamps.py ober179: 11.5 K cycles for the 96 MFMAs of a phase).
      if (KS == KSC && xw == XW) {
        constexpr int U = (NB == 1 && NRB >= 4 && !W8) ? 8 : 4;      // (tall tiles: one wave per SIMD, registers to spare)
        const float* zq = Z + (half + 2 * ks) * LDZ + obh * NBW * 32 + c32;
        const float* xq = Xs + (half + 2 * ks) * XW + ibw * 32 + c32;
        float bq[2][U], aq[2][U][NBW];
        auto ldg = [&](float (&b)[U], float (&av)[U][NBW], int g) {
          const float* zg = zq + g * (U * 2 * KSC * LDZ);
          const float* xg = xq + g * (U * 2 * KSC * XW);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            b[u] = xg[u * 2 * KSC * XW];
#pragma unroll
            for (int ob = 0; ob < NBW; ++ob) av[u][ob] = zg[u * 2 * KSC * LDZ + ob * 32];
          }
        };
        auto mmg = [&](const float (&b)[U], const float (&av)[U][NBW]) {
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ob = 0; ob < NBW; ++ob) a[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][ob], b[u], a[ob], 0, 0, 0);
        };
        const int ng = n2e / U;
        if (ng > 0) {
          ldg(bq[0], aq[0], 0);
          int g = 0;
          for (; g + 2 <= ng; g += 2) {
            ldg(bq[1], aq[1], g + 1);
            __builtin_amdgcn_sched_barrier(0);
            mmg(bq[0], aq[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < ng) ldg(bq[0], aq[0], g + 2);
            __builtin_amdgcn_sched_barrier(0);
            mmg(bq[1], aq[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (g < ng) mmg(bq[0], aq[0]);
        }
        n2 = ng * U;
        if (n2 >= n2e) return;
      }
    }
    ld(b0, a0, n2);
    for (; n2 + 2 <= n2e; n2 += 2) {
      ld(b1, a1, n2 + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(b0, a0);
      __builtin_amdgcn_sched_barrier(0);
      ld(b0, a0, (n2 + 2 < n2e) ? n2 + 2 : n2 + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(b1, a1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n2 < n2e) mm(b0, a0);
  };
  auto prop = [&](const float* Zs, float* Zd) {
    if (D > 0) {
      // 16 bytes per lane: a thread owns 4 consecutive columns of its rows, so one ELL entry (broadcast) and one
      // ds_read_b128 serve 4 elements -- a quarter of the LDS instructions of a column-per-thread layout, which
      // matters because this VALU/LDS phase shares its SIMD with the other half's back-to-back MFMAs
      constexpr int Q = LDZ / 4;               // float4 groups per row
      constexpr int RSTEP = NT / Q;            // rows advanced per pass of the NT threads
      constexpr int NI = TM / RSTEP;           // rows owned by one thread
      static_assert(NT % Q == 0 && TM % RSTEP == 0, "float4 propagation must tile the slab exactly");
      const int c4 = (tid % Q) * 4;
      const int n0 = tid / Q;
      // rows in flight per thread: with PF the next-tile prefetch registers are not live during the propagation,
      // so all of a thread's rows go at once and the ELL entries of hop k+1 are requested before hop k's data
      // (the entry -> data -> fma chain is latency-bound); otherwise two rows (the accumulators own the budget)
      // (tall tiles on the 4-wave kernel run one wave per SIMD: the register file has room for all rows at once there too)
      constexpr int CH = (PF || (NB == 1 && NRB >= 6)) ? NI : ((NI % 2 == 0) ? 2 : 1);
      if constexpr (CH <= 2 && !PF && wgrad_prop4(NRB, NMAT, NB)) {
        // few rows in flight (the accumulators own the register budget): then four ELL entries of a row are requested
        // together and their four gathers after them -- two LDS latencies per four neighbours instead of one per neighbour
        // (tools/wstamps.py ober_sub: the propagation, 3.4-4.0 K cycles, outlasted the other half's 3.1 K of MFMAs)
        for (int i0 = 0; i0 < NI; i0 += CH) {
          f32x4 s[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int k0 = 0; k0 < D; k0 += 4) {
            int2 en[CH][4];
            f32x4 z[CH][4];
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int row = n0 + (i0 + i) * RSTEP;
                en[i][k] = k0 + k < D ? ell[(k0 + k) * TM + row] : make_int2(row, 0);
              }
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
              for (int k = 0; k < 4; ++k) z[i][k] = *reinterpret_cast<const f32x4*>(Zs + en[i][k].x * LDZ + c4);
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
              for (int k = 0; k < 4; ++k) s[i] += z[i][k] * __int_as_float(en[i][k].y);
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) *reinterpret_cast<f32x4*>(Zd + (n0 + (i0 + i) * RSTEP) * LDZ + c4) = s[i];
        }
      } else {
      for (int i0 = 0; i0 < NI; i0 += CH) {
        f32x4 s[CH];
        int2 en[CH], en_next[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) { s[i] = f32x4{0.f, 0.f, 0.f, 0.f}; en[i] = ell[n0 + (i0 + i) * RSTEP]; }
        for (int k = 0; k < D; ++k) {
          const int kn = k + 1 < D ? k + 1 : k;
#pragma unroll
          for (int i = 0; i < CH; ++i) en_next[i] = ell[kn * TM + n0 + (i0 + i) * RSTEP];
          f32x4 z[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) z[i] = *reinterpret_cast<const f32x4*>(Zs + en[i].x * LDZ + c4);
#pragma unroll
          for (int i = 0; i < CH; ++i) { s[i] += z[i] * __int_as_float(en[i].y); en[i] = en_next[i]; }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) *reinterpret_cast<f32x4*>(Zd + (n0 + (i0 + i) * RSTEP) * LDZ + c4) = s[i];
      }
      }
    } else {
      constexpr int RSTEP = NT / LDZ;
      const int c = tid % LDZ;
      const int n0 = tid / LDZ;
      for (int n = n0; n < TM; n += RSTEP) {
        float s = 0.f;
        const int e1 = lrow[n + 1];
        for (int e = lrow[n]; e < e1; ++e) {
          const int2 en = lent[e];
          s = fmaf(__int_as_float(en.y), Zs[en.x * LDZ + c], s);
        }
        Zd[n * LDZ + c] = s;
      }
    }
  };
  // bias gradient (column sums of G): the last LDZ threads of the workgroup (half 1 when 8 waves)
  // (rows >= R of the slab are zero, so the loops run over all TM rows with a fixed trip count, unrolled with
  //  four independent partial sums: the LDS latency is paid per batch instead of per row)
  auto bias_sums = [&](int ts, int R) {
    const int t = tid - (NT - LDZ);
    if (ibg != 0 || t < 0) return;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (rs2) {   // plain sums + one scaled sum per matrix, scales [s, P s, P^2 s, P^3 s] per row
      float sm[NMAT];
#pragma unroll
      for (int m = 0; m < NMAT; ++m) sm[m] = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) {
        const float z = Za[r * LDZ + t];
        const f32x4 d = Dsc[r];
        s4[r & 3] += z;
#pragma unroll
        for (int m = 0; m < NMAT; ++m) sm[m] = fmaf(z, d[m], sm[m]);
      }
#pragma unroll
      for (int m = 0; m < NMAT; ++m) dbs[m] += sm[m];
    } else if (p.rowscale) {
#pragma unroll 8
      for (int r = 0; r < TM; ++r) s4[r & 3] = fmaf(Za[r * LDZ + t], r < R ? p.rowscale[ts + r] : 0.f, s4[r & 3]);
    } else {
#pragma unroll 16
      for (int r = 0; r < TM; ++r) s4[r & 3] += Za[r * LDZ + t];
    }
    dbacc += (s4[0] + s4[1]) + (s4[2] + s4[3]);
  };
  // one MFMA phase + the propagation that feeds the next one, ordered per half (see kernel comment)
  bool stamp_on = false; (void)stamp_on;
  auto phase = [&](const float* Zs, float* Zd, f32x16 (&a)[NBW], int R, bool do_prop, bool do_bias, int ts, int s0) {
    (void)s0;
    if (NW == 8 && role == 1) {
      if (do_bias && !fast_bias) bias_sums(ts, R);
      WSTAMP(s0);
      if (do_prop) prop(Zs, Zd);
      WSTAMP(s0 + 1);
      mma(Zs, a, R);
      WSTAMP(s0 + 2);
    } else {
      mma(Zs, a, R);
      WSTAMP(s0);
      if (NW == 4 && do_bias && !fast_bias) bias_sums(ts, R);
      WSTAMP(s0 + 1);
      if (do_prop) prop(Zs, Zd);
      WSTAMP(s0 + 2);
    }
  };

  // register prefetch of the next tile's slabs (issued right after the staging barrier, consumed at
  // the top of the next iteration: the global-load latency hides under the MFMA phases)
  // (G and X take the 16-byte register-prefetch path independently; a slab whose shape or alignment
  // does not allow it is staged directly with scalar loads at the top of the iteration)
  f32x4 pg[NG4], px[NX4];
  auto issue_loads = [&](int tile) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    constexpr int QG = LDZ / 4;
    const int QX = W8 ? XW / 4 : (xw >> 2);
    // (W8, 256 registers: the per-load 64-bit addresses are loop invariants the compiler would carry -- spilled -- through the tile
    //  loop, and a scratch reload inside this burst waits for every load issued before it: 8.7 K cycles instead of 2.2 K)
    int tidl = tid;
    if (W8) asm volatile("" : "+v"(tidl));
    if (gvec) {
#pragma unroll
      for (int i = 0; i < NG4; ++i) {
        const int idx = tidl + i * NT;
        const int r = idx / QG, c = (idx - r * QG) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R && gcol0 + c < p.hout) v = *reinterpret_cast<const f32x4*>(Gp + (size_t)(ts + r) * p.ldg + gcol0 + c);
        pg[i] = v;
      }
    }
    if (xvec) {
#pragma unroll
      for (int i = 0; i < NX4; ++i) {
        const int idx = tidl + i * NT;
        const int r = idx / QX, c = (idx - r * QX) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < TM * QX && r < R && xcol0 + c < p.hin) v = *reinterpret_cast<const f32x4*>(Xp + (size_t)(ts + r) * p.ldx + xcol0 + c);
        px[i] = v;
      }
    }
  };
  auto write_slabs = [&](int ts, int R) {
    constexpr int QG = LDZ / 4;
    const int QX = xw >> 2;
    if (gvec) {
#pragma unroll
      for (int i = 0; i < NG4; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / QG, c = (idx - r * QG) * 4;
        *reinterpret_cast<f32x4*>(Za + r * LDZ + c) = pg[i];
      }
      if (fast_bias) {                      // rows >= R were loaded as zeros
        f32x4 sp = pg[0];
#pragma unroll
        for (int i = 1; i < NG4; ++i) sp += pg[i];
        if (W8) bsum_reg += sp; else Bsum[tid] += sp;
      }
    } else {
      for (int idx = tid; idx < TM * LDZ; idx += NT) {
        const int r = idx / LDZ, c = idx - r * LDZ;
        Za[idx] = (r < R && gcol0 + c < p.hout) ? Gp[(size_t)(ts + r) * p.ldg + gcol0 + c] : 0.f;
      }
    }
    if (xvec) {
#pragma unroll
      for (int i = 0; i < NX4; ++i) {
        const int idx = tid + i * NT;
        const int r = idx / QX, c = (idx - r * QX) * 4;
        if (idx < TM * QX) *reinterpret_cast<f32x4*>(Xs + r * xw + c) = px[i];
      }
    } else {
      for (int idx = tid; idx < TM * xw; idx += NT) {
        const int r = idx / xw, c = idx - r * xw;
        Xs[idx] = (r < R && xcol0 + c < p.hin) ? Xp[(size_t)(ts + r) * p.ldx + xcol0 + c] : 0.f;
      }
    }
  };

  if ((int)blockIdx.x < p.ntiles) issue_loads(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
#ifdef DSS2_STAMPS
    stamp_on = (tile == (int)blockIdx.x + (int)gridDim.x);
#endif
    WSTAMP(0);
    // ---- stage G slab, X slab and the transposed-graph slice
    write_slabs(ts, R);
    if (rs2 && tid < TM) Dsc[tid] = tid < R ? reinterpret_cast<const f32x4*>(rs2)[ts + tid] : f32x4{0.f, 0.f, 0.f, 0.f};
    if (NMAT > 1 || p.narrow) {
      if (D > 0 && p.ell_tiles != nullptr) {
        const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
        for (int idx = tid; idx < D * TM; idx += NT) ell[idx] = src[idx];
      } else if (D > 0) {
        for (int r = tid; r < TM; r += NT) {
          const int e0 = (r < R) ? p.rowptrT[ts + r] : 0;
          const int deg = (r < R) ? p.rowptrT[ts + r + 1] - e0 : 0;
          for (int k = 0; k < D; ++k)
            ell[k * TM + r] = (k < deg) ? make_int2(p.colT[e0 + k] - ts, __float_as_int(p.wT[e0 + k])) : make_int2(r, 0);
        }
      } else {
        const int base = p.rowptrT[ts];
        const int nnz = p.rowptrT[ts + R] - base;
        for (int r = tid; r <= TM; r += NT) lrow[r] = (r <= R) ? (p.rowptrT[ts + r] - base) : nnz;
        for (int k = tid; k < nnz; k += NT) lent[k] = make_int2(p.colT[base + k] - ts, __float_as_int(p.wT[base + k]));
      }
    }
    __syncthreads();
    WSTAMP(1);
    const int next = tile + gridDim.x;
    // Next tile's slabs -> registers.  Issuing a wave's sixteen 16-byte loads blocks it for 2-5 K cycles (the CU's
    // vector-memory pipe moves 64 B/clk; tools/wstamps.py), so with PF it is done inside the MFMA phase, the two
    // halves at different points: while one wave of a SIMD feeds the memory pipe the other keeps the MFMA pipe busy.
    // (W8: all eight waves at once, 4.4 K cycles with the MFMA pipe idle.  Issued inside the first phase instead, the halves at
    //  different points, the burst takes 7.9 K cycles beside the partner's MFMAs -- rolled loop, no spills -- and the tile 34.6 K
    //  instead of 32.0 K; as two inlined copies it spills 53 registers.)
    if (!PF && next < p.ntiles) issue_loads(next);
    WSTAMP(2);
    // ---- narrow mode: append P^m G as column blocks [m*hout, (m+1)*hout) of the same 32-wide slab
    if (p.narrow) {
      const int h = p.hout;
      for (int m = 1; m < p.nmat; ++m) {
        for (int idx = tid; idx < R * h; idx += NT) {
          const int row = idx / h, j = idx - row * h;
          const float* src = Za + (m - 1) * h + j;
          float sacc = 0.f;
          if (D > 0) {
            for (int k = 0; k < D; ++k) {
              const int2 en = ell[k * TM + row];
              sacc = fmaf(__int_as_float(en.y), src[en.x * LDZ], sacc);
            }
          } else {
            const int e1 = lrow[row + 1];
            for (int e = lrow[row]; e < e1; ++e) {
              const int2 en = lent[e];
              sacc = fmaf(__int_as_float(en.y), src[en.x * LDZ], sacc);
            }
          }
          Za[row * LDZ + m * h + j] = sacc;
        }
        __syncthreads();
      }
    }
    // ---- m = 0 .. NMAT-1, ping-pong propagation
    if constexpr (PF) {
      // bias sums: the two waves that own them run them inside the MFMA phase (their SIMD partners keep the
      // matrix pipe busy meanwhile) instead of holding everybody at the propagation barrier
      if (NW == 4 && !fast_bias) bias_sums(ts, R);
      WSTAMP(3);
      prop(Za, Zb);
      __syncthreads();
      WSTAMP(4);
      prop(Zb, Zc);
      __syncthreads();
      WSTAMP(5);
      if (role == 0 && next < p.ntiles) issue_loads(next);
      mma(Za, acc[0], R);
      if (role != 0 && next < p.ntiles) issue_loads(next);
      if (NW == 8 && !fast_bias) bias_sums(ts, R);
      mma(Zb, acc[1 % NMAT], R);
      mma(Zc, acc[2 % NMAT], R);
      WSTAMP(6);
    } else {
    phase(Za, Zb, acc[0], R, NMAT > 1, true, ts, 3);
    if (NMAT > 1) {
      __syncthreads();
      WSTAMP(6);
      phase(Zb, Za, acc[1 % NMAT], R, NMAT > 2, false, ts, 7);
    }
    if (NMAT > 2) {
      __syncthreads();
      WSTAMP(10);
      phase(Za, Zb, acc[2 % NMAT], R, NMAT > 3, false, ts, 11);
    }
    if (NMAT > 3) {
      __syncthreads();
      phase(Zb, Za, acc[3 % NMAT], R, false, false, ts, 11);
    }
    }
    __syncthreads();
    WSTAMP(14);
  }

  if (fast_bias) {   // (uniform) partial column sums -> LDS [NT / (LDZ/4)][LDZ] -> the owner threads
    constexpr int QG = LDZ / 4;
    *reinterpret_cast<f32x4*>(Za + (tid / QG) * LDZ + (tid % QG) * 4) = W8 ? bsum_reg : Bsum[tid];    // the slabs are dead: last tile done
    __syncthreads();
    const int t = tid - (NT - LDZ);
    if (t >= 0) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NT / QG; ++q) s += Za[q * LDZ + t];
      dbacc += s;
    }
  }
  if (KS > 1) {      // (uniform) K-split partial accumulators -> the ks = 0 wave of each input block, fixed order
    __syncthreads();  // the slabs are dead
    float* red = smem;   // [KS - 1][nib_act][16][64]
#pragma unroll
    for (int m = 0; m < NMAT; ++m) {
      if (ks > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((ks - 1) * nib_act + ibw) * 16 + r) * 64 + lane] = acc[m][0][r];
      }
      __syncthreads();
      if (ks == 0) {
        for (int q = 1; q < KS; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][0][r] += red[(((q - 1) * nib_act + ibw) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  // ---- one slab per workgroup column blockIdx.x; the y-slices tile the [nmat*hout, hin] matrix
  const size_t stride = (size_t)p.nmat * p.hout * p.hin + p.hout + (rs2 ? (size_t)p.nmat * p.hout : 0);
  float* out = slabp + (size_t)blockIdx.x * (wb.slab_stride > 0 ? (size_t)wb.slab_stride : stride);
  if (wave_active && ks == 0) {
    const int i = xcol0 + ibw * 32 + c32;
    if (i < p.hin) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int ob = 0; ob < NBW; ++ob)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = gcol0 + (obh * NBW + ob) * 32 + acc_row(r, half);
            // narrow mode: row o of the single block IS row (m', j) = (o / hout, o % hout) of dW_cat
            if (o < (p.narrow ? p.nmat * p.hout : p.hout)) out[((size_t)m * p.hout + o) * p.hin + i] = acc[m][ob][r];
          }
    }
  }
  {
    const int t = tid - (NT - LDZ);
    if (ibg == 0 && t >= 0 && gcol0 + t < p.hout) {
      out[(size_t)p.nmat * p.hout * p.hin + gcol0 + t] = dbacc;
      if (rs2) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) out[(size_t)p.nmat * p.hout * p.hin + p.hout + (size_t)m * p.hout + gcol0 + t] = dbs[m];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a very narrow layer (nmat * hout <= 8: the last TAGConv H -> 2), streaming variant.
// dW_cat[j][i] = sum_rows (P^m G)[row][o] * X[row][i] with j = m * hout + o is a rank-8 update per row: HBM-bound
// (X is read once), but the MFMA tile kernel stages a 34 KB X slab per workgroup and runs one 32-wide MFMA block
// for 6 useful columns.  Here a thread owns 4 consecutive input columns and keeps its 8 x 4 partial sums in
// registers over all tiles of the workgroup; rows stream straight from HBM (eight 16-byte loads in flight per
// thread), the propagated gradient rows [rows x 8] live in LDS and are read as broadcasts.
// ------------------------------------------------------------------------------------------
constexpr int WN_MAXO = 8;
template <int NRB>
__global__ void __launch_bounds__(256) wgrad_narrow_stream_kernel(const dss2_wgrad_args p) {      // (198 registers = two workgroups per CU; bounded to 170 / 128 the allocator spills 7 / 27)
  constexpr int TM = NRB * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int D = p.ell_width;
  const int h = p.hout, nm = p.nmat;
  const int tpr = p.hin >> 2;                 // threads per row (4 input columns each); divides 256
  const int nrg = 256 / tpr;                  // row groups
  const int c4 = (tid % tpr) * 4, rg = tid / tpr;
  float* Gp = smem;                           // [TM][WN_MAXO]
  int2* ell = reinterpret_cast<int2*>(Gp + TM * WN_MAXO);
  float* red = reinterpret_cast<float*>(ell + (nm > 1 ? D * TM : 0));   // [nrg][WN_MAXO][hin] at the end
  f32x4 acc[WN_MAXO];
#pragma unroll
  for (int j = 0; j < WN_MAXO; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbv = 0.f;                            // thread tid < hout: column sum of G[:, tid]
  // (TM <= 8 * nrg rows: one batch of eight 16-byte loads per thread covers the tile -- 64-row tiles at hin = 128 -- and is requested
  //  BEFORE the tile's G rows are staged and propagated, so the HBM latency of the only large operand runs under that work; round 5:
  //  13.7 -> ... us at C2.  Taller tiles request their remaining batches after the hops, as before.)
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    f32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = rg + u * nrg;
      xv[u] = (row < R) ? *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + row) * p.ldx + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int idx = tid; idx < TM * WN_MAXO; idx += 256) {
      const int row = idx / WN_MAXO, j = idx - row * WN_MAXO;
      Gp[idx] = (row < R && j < h) ? p.G[(size_t)(ts + row) * p.ldg + j] : 0.f;
    }
    if (nm > 1) {
      const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
      for (int idx = tid; idx < D * TM; idx += 256) ell[idx] = src[idx];
    }
    __syncthreads();
    for (int m = 1; m < nm; ++m) {            // (P^T)^m G appended as columns [m*h, (m+1)*h)
      for (int idx = tid; idx < R * h; idx += 256) {
        const int row = idx / h, j = idx - row * h;
        float s = 0.f;
        for (int k = 0; k < D; ++k) {
          const int2 en = ell[k * TM + row];
          s = fmaf(__int_as_float(en.y), Gp[en.x * WN_MAXO + (m - 1) * h + j], s);
        }
        Gp[row * WN_MAXO + m * h + j] = s;
      }
      __syncthreads();
    }
    if (tid < h) {
      float s = 0.f;
      for (int row = 0; row < R; ++row) s += Gp[row * WN_MAXO + tid];
      dbv += s;
    }
    for (int r0 = rg; r0 < TM; r0 += 8 * nrg) {   // eight rows (16-byte loads) in flight per thread
      if (r0 != rg) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = r0 + u * nrg;
          xv[u] = (row < R) ? *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + row) * p.ldx + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = r0 + u * nrg;
        if (row < TM) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gp + row * WN_MAXO), g1 = *reinterpret_cast<const f32x4*>(Gp + row * WN_MAXO + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { acc[j] += xv[u] * g0[j]; acc[4 + j] += xv[u] * g1[j]; }
        }
      }
    }
    __syncthreads();                          // Gp / ell are rewritten by the next tile
  }
  // ---- reduce the row groups through LDS, write the slab [nmat*hout*hin][hout]
#pragma unroll
  for (int j = 0; j < WN_MAXO; ++j) *reinterpret_cast<f32x4*>(red + ((size_t)rg * WN_MAXO + j) * p.hin + c4) = acc[j];
  __syncthreads();
  float* out = p.slab + (size_t)blockIdx.x * ((size_t)nm * h * p.hin + h);
  for (int idx = tid; idx < nm * h * p.hin; idx += 256) {
    const int j = idx / p.hin, i = idx - j * p.hin;
    float s = 0.f;
    for (int g = 0; g < nrg; ++g) s += red[((size_t)g * WN_MAXO + j) * p.hin + i];
    out[idx] = s;
  }
  if (tid < h) out[(size_t)nm * h * p.hin + tid] = dbv;
}

static bool wgrad_narrow_stream_ok(const dss2_wgrad_args& a) {
  static const int enabled = [] { const char* e = getenv("DSS2_NARROW_STREAM"); return e ? atoi(e) : 1; }();
  const int tpr = a.hin >> 2;
  return enabled && a.narrow && a.nmat * a.hout <= WN_MAXO && (a.hin & 3) == 0 && tpr > 0 && tpr <= 256 && (256 % tpr) == 0 &&
         (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && !a.rowscale && (a.nrb <= 4 || a.nrb == 6) &&
         (a.nmat == 1 || (a.ell_width > 0 && a.ell_tiles));
}

template <int NRB>
static int launch_wgrad_narrow_stream(const dss2_wgrad_args& a, hipStream_t stream) {
  const size_t lds = (size_t)NRB * 32 * WN_MAXO * 4 + (a.nmat > 1 ? (size_t)a.ell_width * NRB * 32 * 8 : 0) +
                     (size_t)256 * 4 * WN_MAXO * 4;     // red: nrg * 8 * hin floats = 256 * 4 * 8
  hipLaunchKernelGGL(wgrad_narrow_stream_kernel<NRB>, dim3(a.n_split), dim3(256), lds, stream, a);
  return check_launch("wgrad_narrow_stream");
}

// nmat here = number of MFMA matrix passes (1 in narrow mode); graph = a graph slice is staged
static size_t wgrad_lds(int nrb, int nmat, int nb, int max_nnz, int ell_width, bool graph, int hin = XW, bool w8 = false) {
  const size_t TM = (size_t)nrb * 32;
  const size_t xw = nb == 1 ? (size_t)((((hin + 31) >> 5) << 5) < XW ? (((hin + 31) >> 5) << 5) : XW) : (size_t)XW;
  size_t b = TM * (size_t)nb * 32 * 4 * (wgrad_pf(nrb, nmat) ? 3 : (nmat > 1 ? 2 : 1)) + TM * xw * 4 + TM * 16;
  if (!w8) b += (size_t)(nb >= 2 ? 512 : 256) * 16;   // running bias partials, one 16-byte slot per thread
  if (graph) b += ell_width > 0 ? TM * (size_t)ell_width * 8 : (TM + 2) * 4 + (size_t)max_nnz * 8;
  if (nb == 1 && b < 3 * 16 * 64 * 4) b = 3 * 16 * 64 * 4;     // the K-split's final reduction: three partial accumulators
  return b;
}

// (nrb, nmat, nb) instantiations whose 8-wave register budget (256 VGPRs: two waves per SIMD) the compiler misses, i.e. that
// spill into scratch (tools/kernel_resources.py, profiles/r02_kernel_resources.txt; tests/test_host_cpu.py keeps this list
// honest).  They are neither dispatched nor compiled: the next narrower NB re-reads the X slab once more per output-block
// group instead.
constexpr bool wgrad_spills(int nrb, int nmat, int nb) {
  constexpr int bad[][3] = {{1, 4, 4}, {2, 4, 4}, {3, 2, 4}, {3, 3, 4}, {3, 4, 4}, {4, 2, 2}, {4, 2, 4}, {4, 3, 4}, {4, 4, 4},
                            {6, 2, 2}, {6, 3, 2}, {6, 4, 2}};
  for (const auto& b : bad)
    if (b[0] == nrb && b[1] == nmat && b[2] == nb) return true;
  return false;
}

static int pick_nb(int nrb, int nmat, int hout, int max_nnz, int ell_width) {
  const int nob = (hout + 31) / 32;
  static const int nb_max = [] { const char* e = getenv("DSS2_WGRAD_NB"); return e ? atoi(e) : 4; }();   // tuning knob
  for (int nb = 4; nb >= 1; nb >>= 1) {
    if (nb > nb_max) continue;
    if (nb > 1 && nb / 2 >= nob) continue;  // do not over-allocate columns
    if (wgrad_spills(nrb, nmat, nb)) continue;
    if (wgrad_lds(nrb, nmat, nb, max_nnz, ell_width, nmat > 1) <= (size_t)kMaxLdsBytes) return nb;
  }
  return 0;
}

// the eight-wave form of the one-output-block kernel: 192-row tiles with full 128-column input groups
static bool wgrad_w8(const dss2_wgrad_args& a, int nb) {
  return a.nrb == 6 && nb == 1 && !a.narrow && a.nmat == 3 && (a.hin % 128) == 0 && a.ell_width > 0 &&      // (K = 1 spills 42 registers)
         wgrad_lds(6, a.nmat, 1, a.max_nnz, a.ell_width, true, a.hin, true) <= (size_t)kMaxLdsBytes;
}

template <int NRB, int NMAT, int NB, bool W8 = false>
static int launch_wgrad(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = wgrad_kernel<NRB, NMAT, NB, W8>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "wgrad")) return 1;
  const int nob = (a.hout + 31) / 32, nib = (a.hin + 31) / 32;
  const int nobg = (nob + NB - 1) / NB, nibg = (nib + 3) / 4;
  const size_t lds = wgrad_lds(NRB, NMAT, NB, a.max_nnz, a.ell_width, NMAT > 1 || a.narrow, a.hin, W8);
  static const int ksplit = [] { const char* e = getenv("DSS2_WGRAD_KSPLIT"); return e ? atoi(e) : 1; }();
  hipLaunchKernelGGL(kern, dim3(a.n_split, nobg * nibg, wb.n > 0 ? wb.n : 1), dim3(WgradGeom<NB, W8>::NT), lds, stream, a, nibg, wb, ksplit);
  return check_launch("wgrad");
}

}  // namespace dss2

extern "C" size_t dss2_wgrad_lds_bytes_ex(int nrb, int nmat, int hout, int hin, int max_nnz, int ell_width, int mfma_bf16);

extern "C" size_t dss2_wgrad_lds_bytes(int nrb, int nmat, int hout, int hin, int max_nnz, int ell_width) {
  if (nmat > 1 && nmat * hout <= 32) return dss2::wgrad_lds(nrb, 1, 1, max_nnz, ell_width, true, hin);   // narrow mode
  const int nb = dss2::pick_nb(nrb, nmat, hout, max_nnz, ell_width);
  return nb ? dss2::wgrad_lds(nrb, nmat, nb, max_nnz, ell_width, nmat > 1, hin) : (size_t)-1;
}

extern "C" size_t dss2_wgrad_lds_bytes_ex(int nrb, int nmat, int hout, int hin, int max_nnz, int ell_width, int mfma_bf16) {
  if (mfma_bf16) {
    const size_t b = dss2::wgrad16_lds_bytes(nrb, nmat, hout, hin, ell_width);
    if (b != 0 && b <= (size_t)dss2::kMaxLdsBytes) return b;
  }
  return dss2_wgrad_lds_bytes(nrb, nmat, hout, hin, max_nnz, ell_width);
}

// Workgroups a launch puts on EACH tile-list slice (grid.y): callers that want one workgroup per CU divide their n_split by
// it.  The bf16x6 kernel is one workgroup per CU by LDS, so its column groups over grid.y (H > 128; H > 64 for layers with
// rowscale2) would otherwise run in rounds, each round writing its own slabs.
extern "C" int dss2_wgrad_y_slices(int nrb, int nmat, int hout, int hin, int ell_width, int mfma_bf16, int has_rowscale2) {
  if (!mfma_bf16) return 1;
  const size_t b = dss2::wgrad16_lds_bytes(nrb, nmat, hout, hin, ell_width);
  if (b == 0 || b > (size_t)dss2::kMaxLdsBytes) return 1;
  (void)has_rowscale2;
  return dss2::wgrad16_y_slices(nrb, hout, hin);          // 64-row tiles: a workgroup owns 128 output x 128 input columns; 32-row tiles: 64 x 128
}

// workgroup groups along grid.z of a batched launch of n_layers layers: one per layer, except where the f16x3 tall-tile kernel walks TWO
// 32-column layers per workgroup (dss2_wgrad16th.hip, PAIR) -- the host sizes n_split (slabs per layer) with it so that the launch fills the chip
extern "C" int dss2_wgrad_batched_groups(int nrb, int hout, int hin, int mfma_bf16, int n_layers) {
  static const int pair_on = [] { const char* e = getenv("DSS2_WGRAD_TALL_PAIR"); return e ? atoi(e) : 1; }();
  static const int tall_on = [] { const char* e = getenv("DSS2_WGRAD_TALL_F16"); return e ? atoi(e) : 1; }();
  if (pair_on && tall_on && (mfma_bf16 & 255) == 2 && nrb == 3 && hout == 32 && hin == 32 && n_layers >= 2) return (n_layers + 1) / 2;
  return n_layers;
}

static int wgrad_dispatch(const dss2_wgrad_args& a, void* stream, const dss2::WgradBatch& wb) {
  using namespace dss2;
  if (a.n_split <= 0 || !a.slab) { set_error("wgrad: n_split/slab missing"); return 2; }
  if (a.nmat > 1 && (!a.rowptrT || !a.colT || !a.wT)) { set_error("wgrad: nmat > 1 needs the transposed CSR"); return 2; }
  if (a.ell_width < 0 || a.ell_width > 32) { set_error("wgrad: ell_width %d out of range 0..32", a.ell_width); return 2; }
  if (a.rowscale2 && a.narrow) { set_error("wgrad: rowscale2 is not supported in narrow mode"); return 2; }
  if (a.narrow) {
    if (a.nmat * a.hout > 32) { set_error("wgrad: narrow mode needs nmat*hout <= 32"); return 2; }
    hipStream_t sn = as_stream(stream);
    if (wb.n == 0 && wgrad_narrow_stream_ok(a)) {
      switch (a.nrb) {
        case 1: return launch_wgrad_narrow_stream<1>(a, sn);
        case 2: return launch_wgrad_narrow_stream<2>(a, sn);
        case 3: return launch_wgrad_narrow_stream<3>(a, sn);
        case 6: return launch_wgrad_narrow_stream<6>(a, sn);
        default: return launch_wgrad_narrow_stream<4>(a, sn);
      }
    }
    switch (a.nrb) {
      case 1: return launch_wgrad<1, 1, 1>(a, sn, wb);
      case 2: return launch_wgrad<2, 1, 1>(a, sn, wb);
      case 3: return launch_wgrad<3, 1, 1>(a, sn, wb);
      case 4: return launch_wgrad<4, 1, 1>(a, sn, wb);
      case 6: return launch_wgrad<6, 1, 1>(a, sn, wb);
      default: set_error("wgrad(narrow): unsupported nrb=%d", a.nrb); return 2;
    }
  }
  if (wgrad16h_covers(a)) return launch_wgrad16h(a, as_stream(stream), wb);        // f16x3 kernel (dss2_wgrad16h.hip): args.mfma_bf16 & 255 == 2, 32-row tiles
  if (wgrad16th_covers(a)) return launch_wgrad16th(a, as_stream(stream), wb);      // f16x3 kernel of 96- / 192-row tiles (dss2_wgrad16th.hip)
  if (wgrad16_covers(a)) return launch_wgrad16(a, as_stream(stream), wb);          // bf16x6 kernel (dss2_wgrad16.hip)
  const int nb = pick_nb(a.nrb, a.nmat, a.hout, a.max_nnz, a.ell_width);
  if (!nb) { set_error("wgrad: tile of %d rows does not fit LDS (nmat=%d nnz=%d)", a.nrb * 32, a.nmat, a.max_nnz); return 3; }
  hipStream_t s = as_stream(stream);
  if (wgrad_w8(a, nb)) return launch_wgrad<6, 3, 1, true>(a, s, wb);
#define DSS2_CASE(NRB, NMAT, NB) \
  if constexpr (!wgrad_spills(NRB, NMAT, NB)) { if (a.nrb == NRB && a.nmat == NMAT && nb == NB) return launch_wgrad<NRB, NMAT, NB>(a, s, wb); }
#define DSS2_NMATS(NRB, NB) DSS2_CASE(NRB, 1, NB) DSS2_CASE(NRB, 2, NB) DSS2_CASE(NRB, 3, NB) DSS2_CASE(NRB, 4, NB)
  DSS2_NMATS(1, 1) DSS2_NMATS(1, 2) DSS2_NMATS(1, 4)
  DSS2_NMATS(2, 1) DSS2_NMATS(2, 2) DSS2_NMATS(2, 4)
  DSS2_NMATS(3, 1) DSS2_NMATS(3, 2) DSS2_NMATS(3, 4)
  DSS2_NMATS(4, 1) DSS2_NMATS(4, 2) DSS2_NMATS(4, 4)
  DSS2_NMATS(6, 1) DSS2_NMATS(6, 2)
#undef DSS2_NMATS
#undef DSS2_CASE
  set_error("wgrad: unsupported (nrb=%d, nmat=%d, nb=%d)", a.nrb, a.nmat, nb);
  return 2;
}

static int dss2_wgrad_launch(const dss2_wgrad_args* ap, void* stream);
extern "C" int dss2_wgrad(const dss2_wgrad_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_wgrad: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_wgrad_launch(&a, s_); });
  return dss2_wgrad_launch(ap, stream);
}
static int dss2_wgrad_launch(const dss2_wgrad_args* ap, void* stream) {
  dss2::WgradBatch wb = {};
  return wgrad_dispatch(*ap, stream, wb);
}

static int dss2_wgrad_batched_launch(const dss2_wgrad_args* ap, const float* const* Gs, const float* const* Xs, float* const* slabs, const float* const* rowscale2s, int64_t slab_stride, int n_layers, void* stream);
extern "C" int dss2_wgrad_batched(const dss2_wgrad_args* ap, const float* const* Gs, const float* const* Xs, float* const* slabs, const float* const* rowscale2s, int64_t slab_stride, int n_layers, void* stream) {
  if (!ap) { dss2::set_error("dss2_wgrad_batched: null argument"); return 2; }
  DSS2_RECORD([a = *ap, g = dss2::plan_keep(Gs, (size_t)(n_layers > 0 ? n_layers : 0)), x = dss2::plan_keep(Xs, (size_t)(n_layers > 0 ? n_layers : 0)), sl = dss2::plan_keep(slabs, (size_t)(n_layers > 0 ? n_layers : 0)), r = dss2::plan_keep(rowscale2s, (size_t)(rowscale2s && n_layers > 0 ? n_layers : 0)), slab_stride, n_layers](void* s_) { return dss2_wgrad_batched_launch(&a, dss2::plan_ptr(g), dss2::plan_ptr(x), dss2::plan_ptr(sl), dss2::plan_ptr(r), slab_stride, n_layers, s_); });
  return dss2_wgrad_batched_launch(ap, Gs, Xs, slabs, rowscale2s, slab_stride, n_layers, stream);
}
static int dss2_wgrad_batched_launch(const dss2_wgrad_args* ap, const float* const* Gs, const float* const* Xs, float* const* slabs, const float* const* rowscale2s, int64_t slab_stride, int n_layers, void* stream) {
  if (n_layers < 1 || n_layers > dss2::WGRAD_MAX_BATCH) { dss2::set_error("wgrad_batched: 1..%d layers, got %d", dss2::WGRAD_MAX_BATCH, n_layers); return 2; }
  if (!Gs || !Xs || !slabs) { dss2::set_error("wgrad_batched: null pointer table"); return 2; }
  dss2::WgradBatch wb = {};
  wb.n = n_layers;
  wb.slab_stride = slab_stride;
  for (int l = 0; l < n_layers; ++l) {
    if (!Gs[l] || !Xs[l] || !slabs[l]) { dss2::set_error("wgrad_batched: layer %d has a null pointer", l); return 2; }
    wb.G[l] = Gs[l]; wb.X[l] = Xs[l]; wb.slab[l] = slabs[l];
    wb.rowscale2[l] = rowscale2s ? rowscale2s[l] : nullptr;
    if (wb.rowscale2[l] && slab_stride <= 0) { dss2::set_error("wgrad_batched: per-layer rowscale2 needs an explicit slab_stride"); return 2; }
  }
  dss2_wgrad_args a = *ap;
  a.G = Gs[0]; a.X = Xs[0]; a.slab = slabs[0];
  a.rowscale2 = nullptr;   // per layer, from the table
  return wgrad_dispatch(a, stream, wb);
}

#ifdef DSS2_STAMPS
extern "C" int dss2_debug_read_wstamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_wstamps), sizeof(unsigned long long) * n);
}
#endif
