// Layer-chained variant of gemm_prop (keep a graph resident in LDS across layers).
//
// Tiles hold whole graphs, so layer l+1 of a tile depends only on layer l of the SAME tile: a chain of
// H -> H TAGConv layers (forward) or of their data-gradients (backward) runs inside one workgroup with the
// activation tile staying in LDS.  Every layer's output still goes to HBM once (the backward pass and the
// weight gradients read it), but it is never read back, and the per-launch ramp/drain is paid once per
// chain instead of once per layer.  Per layer: MFMA over the X tile -> barrier (everyone is done reading
// X) -> Horner in the wave-private stage -> epilogue to HBM and, in place, into the X tile -> barrier.
//
// Restricted to what the hot configurations need (the general kernel covers the rest): ELL tile slices,
// 16-byte aligned operands, one 32-column group per wave (4 waves for H <= 128, 8 waves up to H = 256),
// H_in == H_out for every layer.
#pragma once
#include "dss2_common.hpp"

namespace dss2 {

#ifdef DSS2_CHAIN_STAMPS
// Diagnostic build only (csrc/build.sh -DDSS2_CHAIN_STAMPS, tools/cstamps.py): s_memtime stamps of every wave of the first
// 2048 workgroups at the phase boundaries of every layer.  They go to a buffer no kernel reads.
__device__ unsigned long long g_cstamps[2048 * 8 * 64];
#define CSTAMP(slot)                                                                                         \
  do {                                                                                                       \
    unsigned long long t_;                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (lane == 0 && blockIdx.x < 2048 && (slot) < 64) g_cstamps[((size_t)blockIdx.x * 8 + wave) * 64 + (slot)] = t_; \
  } while (0)
// the 100 MHz wall clock (s_memrealtime) into a slot: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define CSTAMP_RT(slot)                                                                                      \
  do {                                                                                                       \
    unsigned long long t_;                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (lane == 0 && blockIdx.x < 2048) g_cstamps[((size_t)blockIdx.x * 8 + wave) * 64 + (slot)] = t_;       \
  } while (0)
#else
#define CSTAMP(slot) do {} while (0)
#define CSTAMP_RT(slot) do {} while (0)
#endif

constexpr int CHAIN_MAX = 8;
struct ChainTable {
  dss2_chain_layer l[CHAIN_MAX]; int n;
  // diagnostic, NULL in normal runs (dss2_debug_chain_clock_probe): workgroups 0, 256, 512, 768 of a split-plane chain launch leave
  // {s_memtime, s_memrealtime} at their start and end -- shader cycles over 100 MHz ticks = the clock the chip HELD under the kernel
  unsigned long long* clock_probe;
};

constexpr int chain_waves_per_simd(int nrb, int nmat) { return nrb * nmat * 16 <= 128 ? 2 : 1; }
constexpr bool chain_rm(int nrb, int rs, bool b16, int nmat) { return b16 && rs == 2 && nrb == 2 && nmat <= 3; }      // (K = 3 would spill)
constexpr int CHAIN_RM_STRIDE = 36;      // floats per stage row in the RM layout (16-byte aligned, rows 4 apart on different banks)

// NW: waves per workgroup the kernel is compiled for.  RS: row split -- RS waves share one 32-column group, each owning
// NRB / RS of the tile's row blocks (its accumulators, its rows of the group's stage and of the epilogue); the
// Horner gather reads the other waves' rows, so with RS > 1 the stage hand-offs are workgroup barriers.  RS = 2
// gives narrow layers (H <= 64: one or two column groups) enough waves to hide their latencies.
//
// B16: the tile GEMM on the bf16 matrix pipe, fp32-accurate ("bf16x6", dss2_common.hpp: split3).  The weights come as
// bf16x3 fragments (dss2_pack_weights, transpose | 2: [m][col group][k/16][plane][lane][8]); the activation tile stays fp32
// in LDS (same footprint, two workgroups per CU as before) and a wave splits its A fragment -- 8 consecutive k of its row
// -- into the three bf16 pieces in registers (VALU work that issues in the shadow of the MFMAs).  Per 16 k and
// (row block, matrix): six v_mfma_f32_32x32x16_bf16 = 192 cycles against 512 for eight v_mfma_f32_32x32x2_f32.
template <int NRB, int NMAT, int NW, int RS, bool B16>
__global__ void __launch_bounds__(NW * 64, RS == 3 ? 3 : chain_waves_per_simd(NRB / RS, NMAT)) gemm_chain_kernel(const dss2_gemm_prop_args p, const ChainTable ct) {
  static_assert(NRB % RS == 0, "row split must divide the row blocks");
  constexpr int TM = NRB * 32;
  constexpr int NRW = NRB / RS;          // row blocks per wave
  constexpr int PF = 8;
  constexpr int LDA = TM + 4;
  // RM ("row-major Horner", narrow bf16x6 layers: two waves per 32-column group, one row block each): every matrix of the
  // group goes to its own stage slot once, and the hops run on 16-byte row pieces (a lane owns 4 columns of 4 rows: one ELL
  // entry and one ds_read_b128 per neighbour instead of 4 + 4 LDS reads in the accumulator layout), ending in the lanes --
  // and the row-major form -- the epilogue wants.  At H = 32 a layer is a chain of LDS / L2 latencies, not MFMA time.
  constexpr bool RM = chain_rm(NRB, RS, B16, NMAT);
  constexpr int SST = RM ? CHAIN_RM_STRIDE : 32;      // stage row stride (floats)
  // (prefetching the next layer's first weight fragments across the Horner / epilogue phases -- PFB -- measured slower: the
  //  36 extra live registers spill once the barriers no longer drain the loads; kept as a switch for the record)
  constexpr bool PFB = false;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int nw = nthreads >> 6;
  const int LDX = p.kpad + 4;
  const int tile = blockIdx.x;
  const uint64_t drop_seed = p.drop_state ? p.drop_state[0] : 0, drop_off = p.drop_state ? p.drop_state[1] : 0;
  float* Xs = smem;
  float* stage = Xs + TM * LDX;
  const int D = p.ell_width;
  int2* ell = reinterpret_cast<int2*>(stage + (RM ? (nw / RS) * NMAT * TM * SST : (nw / RS) * 32 * LDA));   // one stage per column group (RM: per group and matrix)
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int kq = p.kpad >> 2;

  // ---- stage the first layer's input tile (zero padded to TM x kpad) and the tile's ELL slice
  if (TM * kq <= PF * nthreads) {
    f32x4 px[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = tid + i * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TM * kq && r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      px[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = tid + i * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      if (idx < TM * kq) *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = px[i];
    }
  } else {
    for (int idx = tid; idx < TM * kq; idx += nthreads) {
      const int r = idx / kq, c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = v;
    }
  }
  {
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
    for (int idx = tid; idx < D * TM; idx += nthreads) ell[idx] = src[idx];
  }
  CSTAMP(0);
  __syncthreads();
  CSTAMP(1);
#ifdef DSS2_CHAIN_DEPHASE
  // Diagnostic build only (-DDSS2_CHAIN_DEPHASE=n): the workgroups of every second dispatch round start n x 8 K cycles late, so
  // the two workgroups that share a CU are out of phase (one's Horner / epilogue beside the other's MFMA phase).
  if ((blockIdx.x >> 8) & 1)
    for (int i = 0; i < DSS2_CHAIN_DEPHASE; ++i) __builtin_amdgcn_s_sleep(127);
#endif

  const int c32 = lane & 31, half = lane >> 5;
  const int nkk = p.kpad >> 3;
  const float* xa = Xs + c32 * LDX + half * 4;
  const int cg = wave / RS, rs = wave - cg * RS;   // column group, and which share of its row blocks
  float* st = RM ? stage + cg * NMAT * (TM * SST) : stage + cg * (32 * LDA);      // per column group [TM][32] row-major (wave-private when RS == 1)
  auto stage_sync = [&]() { if (RS == 1) wave_lds_sync(); else __syncthreads(); };
  const int ecol0 = cg * 32 + (lane & 7) * 4;
  const int cq = (lane & 7) * 4, r8 = lane >> 3;
  const bool ecol_ok = ecol0 < p.hout;

  bf16x8 bnext[3][NMAT];      // PFB: the next layer's first weight fragments
  if constexpr (PFB) {
    const bf16x8* __restrict__ src = reinterpret_cast<const bf16x8*>(ct.l[0].Bp);
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bnext[pl][m] = src[(((size_t)(m * p.ncg + cg) * (p.kpad >> 4)) * 3 + pl) * 64 + lane];
  }
  for (int li = 0; li < ct.n; ++li) {
    const dss2_chain_layer& L = ct.l[li];    // uniform: scalar loads from the kernel-argument segment
    const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(L.Bp);
    f32x16 acc[NRW][NMAT];
#pragma unroll
    for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][m][r] = 0.f;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (L.bias && ecol_ok) bias4 = *reinterpret_cast<const f32x4*>(L.bias + ecol0);

    if constexpr (B16) {
      // ---- bf16x6: 16 k per step; B fragments (L2) ping-pong, A fragment read as fp32 from LDS and split in registers
      const bf16x8* __restrict__ bp16 = reinterpret_cast<const bf16x8*>(L.Bp);
      const int nks = p.kpad >> 4;
      const float* xa16 = Xs + c32 * LDX + half * 8;
      bf16x8 b0[3][NMAT], b1[3][NMAT];
      auto load_bp = [&](const bf16x8* __restrict__ src, bf16x8 (&b)[3][NMAT], int ks) {
        const int kc = ks < nks ? ks : nks - 1;
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[pl][m] = src[(((size_t)(m * p.ncg + cg) * nks + kc) * 3 + pl) * 64 + lane];
      };
      auto load_b = [&](bf16x8 (&b)[3][NMAT], int ks) { load_bp(bp16, b, ks); };
      auto step16 = [&](const bf16x8 (&b)[3][NMAT], int ks) {
#pragma unroll
        for (int rb = 0; rb < NRW; ++rb) {
          const float* src = xa16 + (rs * NRW + rb) * 32 * LDX + ks * 16;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          uint32_t sh[4], sm[4], sl[4];
          split3_pair(v0[0], v0[1], sh[0], sm[0], sl[0]);
          split3_pair(v0[2], v0[3], sh[1], sm[1], sl[1]);
          split3_pair(v1[0], v1[1], sh[2], sm[2], sl[2]);
          split3_pair(v1[2], v1[3], sh[3], sm[3], sl[3]);
          const u32x4 uh = {sh[0], sh[1], sh[2], sh[3]}, um = {sm[0], sm[1], sm[2], sm[3]}, ul = {sl[0], sl[1], sl[2], sl[3]};
          const bf16x8 ah = __builtin_bit_cast(bf16x8, uh), am = __builtin_bit_cast(bf16x8, um), al = __builtin_bit_cast(bf16x8, ul);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) {      // smallest terms first
            f32x16 c = acc[rb][m];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0][m], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[1][m], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2][m], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0][m], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[1][m], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0][m], c, 0, 0, 0);
            acc[rb][m] = c;
          }
        }
      };
      if constexpr (PFB) {
        // the first fragments of a layer were requested during the previous layer's Horner / epilogue
#pragma unroll
        for (int m = 0; m < NMAT; ++m)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b0[pl][m] = bnext[pl][m];
      } else {
        load_b(b0, 0);
      }
      int ks = 0;
      if constexpr (RS == 3) {
        // three waves per SIMD (168-register budget): one fragment set, the other waves cover its L2 latency
        step16(b0, 0);
        for (ks = 1; ks < nks; ++ks) {
          load_b(b0, ks);
          step16(b0, ks);
        }
      } else {
      for (; ks + 2 <= nks; ks += 2) {
        load_b(b1, ks + 1);
        step16(b0, ks);
        if (!RM || ks + 2 < nks) load_b(b0, ks + 2);
        step16(b1, ks + 1);
      }
      if (ks < nks) step16(b0, ks);
      }
      if constexpr (PFB) {
        if (li + 1 < ct.n) load_bp(reinterpret_cast<const bf16x8*>(ct.l[li + 1].Bp), bnext, 0);
      }
    } else {
    // ---- MFMA over the X tile: 8 k per step, A (LDS) / B (packed weights, L2) ping-pong prefetch
    f32x4 a0[NRW] = {}, a1[NRW] = {}, b0[NMAT] = {}, b1[NMAT] = {};
    auto load_ab = [&](f32x4 (&a)[NRW], f32x4 (&b)[NMAT], int kk) {
      const int kc = kk < nkk ? kk : nkk - 1;
#pragma unroll
      for (int rb = 0; rb < NRW; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(xa + (rs * NRW + rb) * 32 * LDX + kc * 8);
#pragma unroll
      for (int m = 0; m < NMAT; ++m) b[m] = bp[((size_t)(m * p.ncg + cg) * nkk + kc) * 64 + lane];
    };
    auto mma_ab = [&](const f32x4 (&a)[NRW], const f32x4 (&b)[NMAT]) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
            acc[rb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], b[m][s], acc[rb][m], 0, 0, 0);
    };
    load_ab(a0, b0, 0);
    int kk = 0;
    for (; kk + 2 <= nkk; kk += 2) {
      load_ab(a1, b1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_ab(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_ab(a0, b0, kk + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma_ab(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk < nkk) mma_ab(a0, b0);
    }
    CSTAMP(2 + li * 6 + 0);      // GEMM phase done
    // every wave is done with this layer's X tile: the epilogue below overwrites it in place
    if (!RM && li + 1 < ct.n) __syncthreads();      // (RM: the barrier after the stage writes below orders the same accesses)
    CSTAMP(2 + li * 6 + 1);

    if constexpr (RM) {
      // ---- Horner on row pieces: slot m holds G_m; U = G_m + P (slot m+1), written back to slot m for the next hop
      // (no barrier needed before these writes: the previous layer's gathers ended before its closing barrier)
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[m * (TM * SST) + (rs * 32 + acc_row(r, half)) * SST + c32] = acc[0][m][r];
      stage_sync();
      const int rrow = rs * 32 + r8;      // this lane's rows: rrow + 8 i
      f32x4 U[4];
#pragma unroll
      for (int m = NMAT - 2; m >= 0; --m) {
        const float* src = st + (m + 1) * (TM * SST) + cq;
        float* own = st + m * (TM * SST) + rrow * SST + cq;
        int2 en[4], en_next[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { en[i] = ell[rrow + 8 * i]; U[i] = *reinterpret_cast<const f32x4*>(own + 8 * i * SST); }
        for (int k = 0; k < D; ++k) {
          const int kn = k + 1 < D ? k + 1 : k;
#pragma unroll
          for (int i = 0; i < 4; ++i) en_next[i] = ell[kn * TM + rrow + 8 * i];
          f32x4 z[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = *reinterpret_cast<const f32x4*>(src + en[i].x * SST);
#pragma unroll
          for (int i = 0; i < 4; ++i) { U[i] += z[i] * __int_as_float(en[i].y); en[i] = en_next[i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(own + 8 * i * SST) = U[i];      // (m == 0: the epilogue's rows, same lanes)
        if (m > 0) stage_sync();
      }
    }
    // ---- Horner: T = G_{NMAT-1}; T = G_m + P T   (ELL slice, wave-private stage)
    f32x16 T[NRW];
    if constexpr (!RM) {
#pragma unroll
    for (int rb = 0; rb < NRW; ++rb) T[rb] = acc[rb][NMAT - 1];
#pragma unroll
    for (int m = NMAT - 2; m >= 0; --m) {
      stage_sync();
#pragma unroll
      for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((rs * NRW + rb) * 32 + acc_row(r, half)) * 32 + c32] = T[rb][r];
      stage_sync();
#pragma unroll
      for (int rb = 0; rb < NRW; ++rb) T[rb] = acc[rb][m];
      for (int k = 0; k < D; ++k) {
        const int2* ek = ell + k * TM + 4 * half;
#pragma unroll
        for (int rb = 0; rb < NRW; ++rb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int2 en = ek[(rs * NRW + rb) * 32 + acc_row(r, 0)];
            T[rb][r] = fmaf(__int_as_float(en.y), st[en.x * 32 + c32], T[rb][r]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    }

    CSTAMP(2 + li * 6 + 2);      // Horner done
    // ---- epilogue: T -> stage -> 16-byte rows -> HBM (and the next layer's X tile)
    f32x4 pb4[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) pb4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (L.prebias && ecol_ok) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m) pb4[m] = *reinterpret_cast<const f32x4*>(L.prebias + (size_t)m * p.hout + ecol0);
    }
    if constexpr (!RM) {
    stage_sync();      // (RS > 1: the other waves are done gathering from the stage)
#pragma unroll
    for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[((rs * NRW + rb) * 32 + acc_row(r, half)) * 32 + c32] = T[rb][r];
    wave_lds_sync();   // the epilogue of a wave reads only the rows it wrote itself
    }
    CSTAMP(2 + li * 6 + 5);      // T staged for the epilogue
    const int col0 = cg * 32 + cq;
    const bool keep = li + 1 < ct.n;
    const int rlo = rs * NRW * 32, rhi = min(R, (rs + 1) * NRW * 32);     // this wave's rows
    // Common case (no folded bias, mask tensor, residual or in-kernel dropout: the plain forward layers and every
    // data-gradient layer): fixed trip count, every LDS / global read of the wave's rows issued before the first use, the
    // uniform flags tested once.  (The general loop below re-tests seven uniform flags per 16 rows through short basic
    // blocks and waits for its reads one pair at a time: 6.4 K of a layer's 29 K cycles, tools/cstamps.py.)
    // (only where a wave owns at most two row blocks: with more, the batch's 32 registers push the tall-tile instantiations into spills)
    // (instantiations with one wave per SIMD have the register file to themselves: the fast path also serves their three or
    //  four row blocks -- the 96-row tiles of C3 -- ; at two waves per SIMD it is limited to two row blocks)
    constexpr bool SIMPLE_OK = NRW <= 2 || (NW == 4 && chain_waves_per_simd(NRW, NMAT) == 1);
    const bool simple = SIMPLE_OK && !L.prebias && !L.dmask && !L.add_src && !L.drop_id;
    if (col0 < p.hout && simple) {
      constexpr int NIT = NRW * 2;      // 16 rows per pass of the wave's 64 lanes; two passes per batch
      const bool has_rs = L.relu_src != nullptr, has_bias = L.bias != nullptr, do_relu = (L.relu & 1) != 0;
#pragma unroll 1
      for (int it0 = 0; it0 < NIT; it0 += 2) {
        f32x4 y[2][2], gate[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = rlo + r8 + 16 * (it0 + it) + 8 * u;
            const int rr = row < rhi ? row : rlo;
            y[it][u] = *reinterpret_cast<const f32x4*>(st + rr * SST + cq);
            if (has_rs) gate[it][u] = *reinterpret_cast<const f32x4*>(L.relu_src + (size_t)(ts + rr) * p.ld_relu + col0);
          }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = rlo + r8 + 16 * (it0 + it) + 8 * u;
            if (row >= rhi) continue;
            f32x4 v = y[it][u];
            if (has_bias) v += bias4;
            if (do_relu) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_nan(v[q]);
            }
            if (has_rs) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_open(gate[it][u][q]) ? v[q] : 0.f;
            }
            *reinterpret_cast<f32x4*>(L.Y + (size_t)(ts + row) * p.ldy + col0) = v;
            if (keep) *reinterpret_cast<f32x4*>(Xs + row * LDX + col0) = v;
          }
      }
    } else if (col0 < p.hout) {
      constexpr int EU = RS == 3 ? 1 : 2;      // rows in flight per lane (three waves per SIMD: the smaller register budget)
      for (int row0 = rlo + r8; row0 < rhi; row0 += 8 * EU) {
        f32x4 y[EU], rs[EU], dm[EU], ad[EU], ps[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
          const int row = row0 + 8 * u;
          const bool ok = row < rhi;
          const size_t grow = (size_t)(ts + (ok ? row : rlo));
          y[u] = *reinterpret_cast<const f32x4*>(st + (ok ? row : rlo) * SST + cq);
          if (L.prebias) ps[u] = *reinterpret_cast<const f32x4*>(p.pre_rowscale + grow * 4);
          if (L.dmask) dm[u] = *reinterpret_cast<const f32x4*>(L.dmask + grow * p.ld_dmask + col0);
          if (L.relu_src) rs[u] = *reinterpret_cast<const f32x4*>(L.relu_src + grow * p.ld_relu + col0);
          if (L.add_src) ad[u] = *reinterpret_cast<const f32x4*>(L.add_src + grow * p.ld_add + col0);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
          const int row = row0 + 8 * u;
          if (row >= rhi) continue;
          f32x4 v = y[u];
          if (L.bias) v += bias4;
          if (L.prebias) {
#pragma unroll
            for (int m = 0; m < NMAT; ++m) v += pb4[m] * ps[u][m];
          }
          if (L.dmask) v *= dm[u];
          if (L.drop_id) v *= dropout_mult4(drop_seed, drop_off, (uint32_t)L.drop_id, (uint32_t)(ts + row), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
          if (L.relu & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = relu_nan(v[q]);
          }
          if (L.relu_src) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = relu_open(rs[u][q]) ? v[q] : 0.f;
          }
          if (L.add_src) v += ad[u];
          *reinterpret_cast<f32x4*>(L.Y + (size_t)(ts + row) * p.ldy + col0) = v;
          if (keep) *reinterpret_cast<f32x4*>(Xs + row * LDX + col0) = v;
        }
      }
    }
    CSTAMP(2 + li * 6 + 3);      // epilogue done
    if (keep) __syncthreads();   // the next layer's X tile is complete
    CSTAMP(2 + li * 6 + 4);
  }
}

inline size_t chain_lds_bytes(int nrb, int kpad, int ncg, int ell_width, int rm_nmat = 0) {      // rm_nmat > 0: the RM stage (one slot per matrix)
  const size_t TM = (size_t)nrb * 32;
  const size_t stage = rm_nmat > 0 ? (size_t)ncg * rm_nmat * TM * CHAIN_RM_STRIDE * 4 : (size_t)ncg * 32 * (TM + 4) * 4;
  return TM * (size_t)(kpad + 4) * 4 + stage + TM * (size_t)ell_width * 8;
}

template <int NRB, int NMAT, int NW, int RS, bool B16 = false>
inline int launch_chain(const dss2_gemm_prop_args& a, const ChainTable& ct, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_chain_kernel<NRB, NMAT, NW, RS, B16>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_prop_chain")) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * a.ncg * RS), chain_lds_bytes(NRB, a.kpad, a.ncg, a.ell_width, chain_rm(NRB, RS, B16, NMAT) ? NMAT : 0), stream, a, ct);
  return check_launch("gemm_prop_chain");
}


// the bf16x6 instantiations live in their own translation unit (dss2_gemm_chain16.hip, compiled without packed fp32 ops)
int launch_chain16(const dss2_gemm_prop_args& a, const ChainTable& ct, int rsplit, hipStream_t s);
// 64-row tiles, one wave per column group: the tile kept in LDS as split bf16 planes (dss2_gemm_chain_sp.hip; DSS2_CHAIN_SP=0: off)
bool chain_sp_supported(const dss2_gemm_prop_args& a);
int launch_chain_sp(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head* head, hipStream_t s);
// 192-row tiles, six row blocks per wave (dss2_gemm_chain_sp6.hip)
bool chain_sp6_supported(const dss2_gemm_prop_args& a);
int launch_chain_sp6(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head* head, hipStream_t s);

}  // namespace dss2
