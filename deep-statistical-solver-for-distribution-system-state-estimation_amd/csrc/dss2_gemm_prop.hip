// Fused tile GEMM (fp32 MFMA 32x32x2) + Horner graph propagation in LDS, for gfx950.
//
// One workgroup owns one tile of <= 32*NRB consecutive node rows made of whole graphs.
//   1. the tile's X rows are staged once in LDS (row stride kpad+4 floats: conflict-free
//      ds_read_b128 of the A fragments);
//   2. wave w owns 32 output columns `cg` of EVERY matrix m < NMAT and all NRB row blocks:
//      acc[rb][m] += X[rb] . B_m[:, cg]   with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
//      k is consumed 8 at a time: lane l holds A[row l&31][8kk + 4(l>>5) + s] and
//      B[8kk + 4(l>>5) + s][col l&31], s = 0..3, so both operands arrive as one 16-byte load
//      (A from LDS, B from the fragment-packed weights, 1 KiB contiguous per wave);
//   3. Horner epilogue on the accumulators: T = G_{NMAT-1}; T = G_m + P T for m = NMAT-2..0,
//      where P T is a CSR segmented sum over the tile's rows read from a wave-private LDS
//      stage (column on the lane => conflict-free, no atomics, fixed order);
//   4. bias / dropout mask / ReLU / ReLU-mask / residual, one 128-B row segment per half wave.
#include <stdlib.h>

#include "dss2_weightspace.hpp"

namespace dss2 {

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weights_kernel(const dss2_pack_desc* __restrict__ descs) {
  pack_weights_body(descs[blockIdx.y], (int)blockIdx.x);      // (dss2_weightspace.hpp)
}

// ------------------------------------------------------------------------------------------
// fused GEMM + propagation
// ------------------------------------------------------------------------------------------
#ifdef DSS2_STAMPS
// Diagnostic build only (csrc/build.sh -DDSS2_STAMPS -> libdss2_hip_stamps.so): per-wave phase stamps.
// Values go to a buffer of their own that no kernel reads; no output depends on them.
__device__ unsigned long long g_stamps[8192 * 8];
#define DSS2_STAMP(slot)                                                                       \
  do {                                                                                         \
    unsigned long long t_;                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (lane == 0 && (stamp_tile * 4 + wave) < 8192) g_stamps[(stamp_tile * 4 + wave) * 8 + (slot)] = t_; \
  } while (0)
#define DSS2_STAMP2(slot)                                                                      \
  do {                                                                                         \
    unsigned long long t_;                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (lane == 0 && (slot) < 16) g_stamps[(blockIdx.x * 8 + wave) * 16 + (slot)] = t_;         \
  } while (0)
#else
#define DSS2_STAMP(slot) do {} while (0)
#define DSS2_STAMP2(slot) do {} while (0)
#endif

// waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument):
// the accumulators take NRB*NMAT*16 registers of the 512 per lane.
constexpr int gemm_waves_per_simd(int nrb, int nmat) {
  return nrb * nmat * 16 <= 128 ? 2 : 1;
}

// B16 (matrix-sequential tall tiles with K-halved staging only): the tile GEMM as bf16x6 -- weights packed as bf16x3
// fragments (dss2_pack_weights, transpose | 2), the wave's A fragments (8 consecutive k of its rows, fp32 in LDS) split into
// three bf16 pieces in registers, six v_mfma_f32_32x32x16_bf16 per 16 k and row block, smallest terms first (the layer
// chain's scheme, dss2_gemm_chain_kernel.hpp).  One workgroup per CU at these LDS sizes, one wave per SIMD.
// RSP (bf16x6 tall tiles only): row split -- RSP waves share a column group, each owning NRB / RSP of the tile's row blocks (its
// accumulators, its rows of the group's stage and of the epilogue); the Horner gather reads the other wave's rows, so the
// stage hand-offs become workgroup barriers.  Two waves per SIMD instead of one: the VALU-issue-bound phases (A split,
// Horner, epilogue, staging) of one wave run beside the MFMAs of the other.
template <int NRB, int NMAT, bool B16 = false, int RSP = 1>
__global__ void __launch_bounds__(256 * RSP, RSP > 1 ? 2 : gemm_waves_per_simd(NRB, NMAT)) gemm_prop_kernel(const dss2_gemm_prop_args p) {
  static_assert(RSP == 1 || (B16 && NRB % RSP == 0), "row split: bf16x6 tall tiles only");
  constexpr int NRW = NRB / RSP;      // row blocks per wave
  constexpr int TM = NRB * 32;
  constexpr int PF = 8;    // float4 registers per thread for the batched X staging (64 x 128 floats / 256 threads)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int nw = nthreads >> 6;
  // K-halved staging (matrix-sequential mode only, flag bit 24 of p.relu set by the launcher): the X tile is staged in
  // two halves of kpad/2 columns per matrix pass, so that a tall tile (192 rows x 128 columns = 101 KB) leaves room for
  // four wave stages instead of two -- twice the waves, for 3x the (L2-resident) X reads.
  const bool khalf = ((p.relu >> 24) & 1) != 0;
  const int LDX = (khalf ? (p.kpad >> 1) : p.kpad) + 4;
  const int tile = blockIdx.x;
  const uint64_t drop_seed = p.drop_id ? p.drop_state[0] : 0, drop_off = p.drop_id ? p.drop_state[1] : 0;   // uniform scalar loads
  int stamp_tile = tile; (void)stamp_tile;
  DSS2_STAMP(0);
  const int dbg = (p.relu >> 8) & 0xff;   // diagnostics only (tools/ablate.py): 1 no MFMA, 2 no Horner, 4 no stores, 8 no X staging
  float* Xs = smem;
  float* stage = Xs + TM * LDX;
  // graph slice of the tile: ELL [D][TM] {local src, weight} when the batch's max degree D is small
  // (fixed trip count => independent LDS chains), else CSR (local row pointers + entries)
  const int D = p.ell_width;
  int2* ell = reinterpret_cast<int2*>(stage + (nw / RSP) * 32 * (TM + 4));
  int* lrow = reinterpret_cast<int*>(stage + (nw / RSP) * 32 * (TM + 4));
  int2* lent = reinterpret_cast<int2*>(lrow + TM + 2);
  const bool need_graph = NMAT > 1 || p.prop_in > 0;
  constexpr int LDA = TM + 4;
  constexpr bool SEQ = NMAT > 1 && NRB * NMAT >= 16;   // one matrix at a time (see below)

  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  DSS2_STAMP(1);
  // with input-side propagation only the first kin columns come from memory
  const int kin = p.prop_in > 0 ? p.kreal / (p.prop_in + 1) : p.kreal;
  const int kq = p.kpad >> 2;
  const bool vec_ok = ((kin & 3) == 0) && ((p.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);

  // ---- stage the X tile (zero padded to TM x kpad).  All of a thread's 16-byte loads are issued
  //      before the first LDS write (one exposed memory latency instead of one per load: the rolled
  //      load->wait->write loop cost 13-20 K cycles per tile under load, measured with stamps)
  if ((dbg & 8) || khalf) {
  } else if (vec_ok && TM * kq <= PF * nthreads) {
    f32x4 px[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = tid + i * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < TM * kq && r < R && c < kin) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      px[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = tid + i * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      if (idx < TM * kq) *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = px[i];
    }
  } else if (vec_ok) {
    for (int idx = tid; idx < TM * kq; idx += nthreads) {
      const int r = idx / kq;
      const int c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < R && c < kin) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = v;
    }
  } else {
    for (int idx = tid; idx < TM * p.kpad; idx += nthreads) {
      const int r = idx / p.kpad;
      const int c = idx - r * p.kpad;
      float v = 0.f;
      if (r < R && c < kin) v = p.X[(size_t)(ts + r) * p.ldx + c];
      Xs[r * LDX + c] = v;
    }
  }
  // ---- stage the tile's graph slice (precomputed per-tile ELL: one coalesced copy)
  if (need_graph) {
    if (D > 0 && p.ell_tiles != nullptr) {
      const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
      for (int idx = tid; idx < D * TM; idx += nthreads) ell[idx] = src[idx];
    } else if (D > 0) {
      for (int r = tid; r < TM; r += nthreads) {
        const int e0 = (r < R) ? p.rowptr[ts + r] : 0;
        const int deg = (r < R) ? p.rowptr[ts + r + 1] - e0 : 0;
        for (int k = 0; k < D; ++k)
          ell[k * TM + r] = (k < deg) ? make_int2(p.col[e0 + k] - ts, __float_as_int(p.w[e0 + k])) : make_int2(r, 0);
      }
    } else {
      const int base = p.rowptr[ts];
      const int nnz = p.rowptr[ts + R] - base;
      for (int r = tid; r <= TM; r += nthreads) lrow[r] = (r <= R) ? (p.rowptr[ts + r] - base) : nnz;
      for (int k = tid; k < nnz; k += nthreads) lent[k] = make_int2(p.col[base + k] - ts, __float_as_int(p.w[base + k]));
    }
  }
  __syncthreads();
  // ---- input-side propagation (narrow inputs): X_h = P X_{h-1} appended as columns [h*kin, (h+1)*kin)
  for (int hop = 1; hop <= p.prop_in; ++hop) {
    for (int idx = tid; idx < R * kin; idx += nthreads) {
      const int row = idx / kin, j = idx - row * kin;
      const float* src = Xs + (hop - 1) * kin + j;
      float sacc = 0.f;
      if (D > 0) {
        for (int k = 0; k < D; ++k) {
          const int2 en = ell[k * TM + row];
          sacc = fmaf(__int_as_float(en.y), src[en.x * LDX], sacc);
        }
      } else {
        const int e1 = lrow[row + 1];
        for (int e = lrow[row]; e < e1; ++e) {
          const int2 en = lent[e];
          sacc = fmaf(__int_as_float(en.y), src[en.x * LDX], sacc);
        }
      }
      Xs[row * LDX + hop * kin + j] = sacc;
    }
    __syncthreads();
  }

  DSS2_STAMP(2);
  const int c32 = lane & 31;   // A row inside a row block == output column inside a column group
  const int half = lane >> 5;
  const int nkk = p.kpad >> 3;
  const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(p.Bp);
  const float* xa = Xs + c32 * LDX + half * 4;
  const int cgw = wave / RSP, rb0 = (wave - cgw * RSP) * NRW;      // column group of this wave; its first row block
  float* st = stage + cgw * (32 * LDA);   // wave-private (RSP > 1: shared by the group's waves): [TM][32] row-major, or [32][TM+4] transposed
  auto stage_sync = [&]() { if (RSP == 1) wave_lds_sync(); else __syncthreads(); };
  const int rlo = rb0 * 32, rhi = min(R, (rb0 + NRW) * 32);      // this wave's rows in the epilogue

  for (int cg = cgw; cg < p.ncg; cg += nw / RSP) {
    // epilogue constants requested now so that their latency hides under the MFMA loop
    const int ecol0 = cg * 32 + (lane & 7) * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && ecol0 < p.hout && (p.hout & 3) == 0) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ecol0);
    f32x16 T[NRB];
    if constexpr (SEQ) {
      // Matrix-sequential mode for tall tiles (NRB * NMAT accumulators would not fit the register file: 6 x 3 spilled
      // 359 registers and ran at 14 % of peak).  One matrix at a time, highest hop first:
      //   T <- X B_{K};   T <- X B_m + P T  (m = K-1 .. 0)
      // with T parked in the wave's LDS stage during the MFMA pass of the next matrix (the gather needs it there
      // anyway).  Same MFMA count; the X tile is read NMAT times from LDS instead of once.
      constexpr int NPH = NRB * 2 / RSP;      // 16-byte pieces per thread of one half of the X tile (256 RSP threads, kpad <= 128)
      f32x4 pxh[NPH];
      // one pass (matrix m) as a function of its accumulator: the last pass (m = 0) is peeled off the loop and accumulates
      // straight into T -- a T assigned on the last trip of a rolled loop holds its 16 NRB registers through every trip
      auto seq_pass = [&](const int m, f32x16 (&accm)[NRB]) {
#pragma unroll
        for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) accm[rb][r] = 0.f;
        f32x4 sa0[NRB] = {}, sa1[NRB] = {}, sb0 = {}, sb1 = {};
        auto smma = [&](const f32x4 (&a)[NRB], const f32x4& b) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) accm[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], b[s], accm[rb], 0, 0, 0);
        };
        for (int hf = 0; hf < (khalf ? 2 : 1); ++hf) {
          const int nkh = khalf ? (nkk >> 1) : nkk;      // k-steps in this half
          const int kb = hf * nkh;                      // first k-step (B fragment index) of this half
          if (khalf) {                                  // (uniform: every wave has exactly one column group)
            const int kh4 = p.kpad >> 3;                // 16-byte pieces per row in a half
            // the OTHER half's pieces were requested right after the previous staging (register prefetch under the MFMA
            // pass): only the very first staging of a tile pays its memory latency in the open
            // (not with a row split: the 24 prefetch registers spill at two waves per SIMD, and the other wave of the SIMD covers the latency)
            const bool pre_ok = RSP == 1 && TM * kh4 <= NPH * nthreads;
            auto load_half = [&](int h) {
#pragma unroll
              for (int i = 0; i < NPH; ++i) {
                const int idx = tid + i * nthreads;
                const int r = idx / kh4, c = (idx - r * kh4) << 2;
                const int cglob = h * (p.kpad >> 1) + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (idx < TM * kh4 && r < R && cglob < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + cglob);
                pxh[i] = v;
              }
            };
            __syncthreads();                              // everyone is done with the previous half
            if (pre_ok) {
              if (m == NMAT - 1 && hf == 0) load_half(0);
#pragma unroll
              for (int i = 0; i < NPH; ++i) {
                const int idx = tid + i * nthreads;
                const int r = idx / kh4, c = (idx - r * kh4) << 2;
                if (idx < TM * kh4) *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = pxh[i];
              }
              __syncthreads();
              if (!(m == 0 && hf == 1)) load_half(1 - hf);
            } else {
            for (int base = 0; base < TM * kh4; base += PF * nthreads) {
              f32x4 px[PF];
#pragma unroll
              for (int i = 0; i < PF; ++i) {
                const int idx = base + tid + i * nthreads;
                const int r = idx / kh4, c = (idx - r * kh4) << 2;
                const int cglob = hf * (p.kpad >> 1) + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (idx < TM * kh4 && r < R && cglob < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + cglob);
                px[i] = v;
              }
#pragma unroll
              for (int i = 0; i < PF; ++i) {
                const int idx = base + tid + i * nthreads;
                const int r = idx / kh4, c = (idx - r * kh4) << 2;
                if (idx < TM * kh4) *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = px[i];
              }
            }
            __syncthreads();
            }
          }
          if constexpr (B16) {
            // 16 k per step; the B fragments (three planes, L2) one step ahead; A read as fp32 and split in registers
            const bf16x8* __restrict__ bp16 = reinterpret_cast<const bf16x8*>(p.Bp);
            const int nks = p.kpad >> 4, nsh = nkh >> 1, sb = kb >> 1;      // k16 steps: per matrix, in this half, first of this half
            const float* xa16 = Xs + c32 * LDX + half * 8;
            bf16x8 bq0[3], bq1[3];
            auto load_b16 = [&](bf16x8 (&b)[3], int s) {
              const int sc = s < nsh ? s : nsh - 1;
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) b[pl] = bp16[(((size_t)(m * p.ncg + cg) * nks + sb + sc) * 3 + pl) * 64 + lane];
            };
            auto step16 = [&](const bf16x8 (&b)[3], int s) {
#pragma unroll
              for (int rb = 0; rb < NRW; ++rb) {
                const float* src = xa16 + (rb0 + rb) * 32 * LDX + s * 16;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                uint32_t sh[4], sm[4], sl[4];
                split3_pair(v0[0], v0[1], sh[0], sm[0], sl[0]);
                split3_pair(v0[2], v0[3], sh[1], sm[1], sl[1]);
                split3_pair(v1[0], v1[1], sh[2], sm[2], sl[2]);
                split3_pair(v1[2], v1[3], sh[3], sm[3], sl[3]);
                const u32x4 uh = {sh[0], sh[1], sh[2], sh[3]}, um = {sm[0], sm[1], sm[2], sm[3]}, ul = {sl[0], sl[1], sl[2], sl[3]};
                const bf16x8 ah = __builtin_bit_cast(bf16x8, uh), am = __builtin_bit_cast(bf16x8, um), al = __builtin_bit_cast(bf16x8, ul);
                f32x16 c = accm[rb];        // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0], c, 0, 0, 0);
                accm[rb] = c;
              }
            };
            load_b16(bq0, 0);
            int s = 0;
            // (sched_barrier: without it the scheduler sinks the next step's requests below this step's MFMAs to share the
            //  registers of the two buffers, and every step then waits for its own L2 round trip)
            for (; s + 2 <= nsh; s += 2) {
              load_b16(bq1, s + 1);
              __builtin_amdgcn_sched_barrier(0);
              step16(bq0, s);
              __builtin_amdgcn_sched_barrier(0);
              load_b16(bq0, s + 2);
              __builtin_amdgcn_sched_barrier(0);
              step16(bq1, s + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (s < nsh) step16(bq0, s);
          } else {
          auto sload2 = [&](f32x4 (&a)[NRB], f32x4& b, int kk) {
            const int kc = kk < nkh ? kk : nkh - 1;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(xa + rb * 32 * LDX + kc * 8);
            b = bp[((size_t)(m * p.ncg + cg) * nkk + kb + kc) * 64 + lane];
          };
          sload2(sa0, sb0, 0);
          int kk = 0;
          for (; kk + 2 <= nkh; kk += 2) {
            sload2(sa1, sb1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            smma(sa0, sb0);
            __builtin_amdgcn_sched_barrier(0);
            sload2(sa0, sb0, kk + 2);
            __builtin_amdgcn_sched_barrier(0);
            smma(sa1, sb1);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (kk < nkh) smma(sa0, sb0);
        }
          }
        if (m < NMAT - 1) {            // T (in the stage since the end of the previous pass) <- accm + P T
          if (D > 0) {
            for (int k = 0; k < D; ++k) {
              const int2* ek = ell + k * TM + 4 * half;
#pragma unroll
              for (int rb = 0; rb < NRW; ++rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const int2 en = ek[(rb0 + rb) * 32 + acc_row(r, 0)];
                  accm[rb][r] = fmaf(__int_as_float(en.y), st[en.x * 32 + c32], accm[rb][r]);
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else {
#pragma unroll
            for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = (rb0 + rb) * 32 + acc_row(r, half);
                float s = accm[rb][r];
                const int e1 = lrow[row + 1];
                for (int e = lrow[row]; e < e1; ++e) {
                  const int2 en = lent[e];
                  s = fmaf(__int_as_float(en.y), st[en.x * 32 + c32], s);
                }
                accm[rb][r] = s;
              }
          }
        }
      };
#pragma unroll 1
      for (int m = NMAT - 1; m >= 1; --m) {
        f32x16 accm[NRB];
        seq_pass(m, accm);
        // park T in the stage for the next pass
        stage_sync();
#pragma unroll
        for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[((rb0 + rb) * 32 + acc_row(r, half)) * 32 + c32] = accm[rb][r];
        stage_sync();
      }
      seq_pass(0, T);
    } else {
    f32x16 acc[NRB][NMAT];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][m][r] = 0.f;

    // k loop, 8 k-values per step, A (LDS) and B (packed weights, L2) fragments prefetched one step
    // ahead in two register buffers in ping-pong.  No register copies, and sched_barrier(0) pins "issue
    // the next buffer's loads, then this buffer's MFMAs" (left alone, the scheduler sinks the loads to
    // the end of the MFMA block and the next block starts on vmcnt(0)); every wait is then a counted
    // s_waitcnt right before the first MFMA that needs the data.
    f32x4 a0[NRB] = {}, a1[NRB] = {}, b0[NMAT] = {}, b1[NMAT] = {};
    auto load_ab = [&](f32x4 (&a)[NRB], f32x4 (&b)[NMAT], int kk) {
      const int kc = kk < nkk ? kk : nkk - 1;   // clamped: harmless reload past the end
      if (!(dbg & 32)) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(xa + rb * 32 * LDX + kc * 8);
      }
      if (!(dbg & 16)) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) b[m] = bp[((size_t)(m * p.ncg + cg) * nkk + kc) * 64 + lane];
      }
    };
    auto mma_ab = [&](const f32x4 (&a)[NRB], const f32x4 (&b)[NMAT]) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
            acc[rb][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], b[m][s], acc[rb][m], 0, 0, 0);
    };
    load_ab(a0, b0, 0);
    int kk = (dbg & 1) ? nkk : 0;
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
    if (kk < nkk) mma_ab(a0, b0);   // odd number of k steps

    DSS2_STAMP(3);
    // ---- Horner propagation: T = G_{NMAT-1}; T = G_m + P T
    {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) T[rb] = acc[rb][NMAT - 1];
    if (NMAT > 1 && !(dbg & 2)) {
#pragma unroll
      for (int m = NMAT - 2; m >= 0; --m) {
        wave_lds_sync();  // previous round's reads of the stage are done
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[(rb * 32 + acc_row(r, half)) * 32 + c32] = T[rb][r];
        wave_lds_sync();
        if (D > 0) {
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) T[rb] = acc[rb][m];
          for (int k = 0; k < D; ++k) {   // uniform trip count; the 16 row chains of a block are independent
            const int2* ek = ell + k * TM + 4 * half;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int2 en = ek[rb * 32 + acc_row(r, 0)];
                T[rb][r] = fmaf(__int_as_float(en.y), st[en.x * 32 + c32], T[rb][r]);
              }
              // one row block (16 chains, ~48 temporaries) at a time: unbounded, the scheduler hoists all
              // NRB*32 LDS reads and the allocator spills into scratch in this phase
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        } else {
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rb * 32 + acc_row(r, half);
              float s = acc[rb][m][r];
              const int e1 = lrow[row + 1];
              for (int e = lrow[row]; e < e1; ++e) {
                const int2 en = lent[e];
                s = fmaf(__int_as_float(en.y), st[en.x * 32 + c32], s);
              }
              T[rb][r] = s;
            }
        }
      }
    }
    }
    }   // !SEQ
    DSS2_STAMP(4);
    // ---- epilogue: T -> wave-private LDS stage -> rolled, row-coalesced store loop (keeps the
    //      address arithmetic of the five optional operands out of the unrolled register code)
    // row-scaled pre-bias (a Linear folded into this layer): sum_m P^m (s (x) pb_m) = sum_m (P^m s) (x) pb_m,
    // a rank-NMAT term of the output; pre_rowscale holds the rows [s, P s, P^2 s, P^3 s]
    f32x4 pb4[NMAT];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) pb4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.prebias && ecol0 < p.hout && (p.hout & 3) == 0) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m) pb4[m] = *reinterpret_cast<const f32x4*>(p.prebias + (size_t)m * p.hout + ecol0);
    }
    stage_sync();      // (RSP > 1: the group's other wave is done gathering from the stage)
#pragma unroll
    for (int rb = 0; rb < NRW; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[((rb0 + rb) * 32 + acc_row(r, half)) * 32 + c32] = T[rb][r];
    wave_lds_sync();      // the row loops below read only the rows this wave wrote itself
    const bool vec_epi = ((p.hout & 3) == 0) && ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.Y) & 15) == 0) &&
                         (!p.relu_src || (((p.ld_relu & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.relu_src) & 15) == 0))) &&
                         (!p.dmask || (((p.ld_dmask & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.dmask) & 15) == 0))) &&
                         (!p.add_src || (((p.ld_add & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.add_src) & 15) == 0)));
    if (dbg & 4) {
    } else if (vec_epi) {
      // 16 B per lane: 8 lanes cover the 32 columns of a row, 8 rows (8 x 128 B) per wave instruction
      const int cq = (lane & 7) * 4, r8 = lane >> 3;
      const int col0 = cg * 32 + cq;
      // Tall tiles, common case (no folded bias, mask tensor, residual, row scale or in-kernel dropout): the uniform flags
      // tested once, the LDS / gate reads of 4 rows per lane issued before the first use.  One wave per SIMD issues an
      // instruction every ~4 cycles, and the general loop below spends most of them on flag tests and short waits (the
      // epilogue is 64 of this kernel's 299 us at the 179-bus shape, tools/ablate.py).
      // (also the single-matrix GEMMs of tall tiles -- the narrow head's data gradient, K = 8: pure epilogue, 72 -> 60 us at 192 rows)
      const bool simple = (SEQ || (NMAT == 1 && NRB >= 3)) && !p.prebias && !p.dmask && !p.add_src && !p.drop_id && !p.rowscale;
      if (col0 < p.hout && simple) {
        const bool has_rs = p.relu_src != nullptr, has_bias = p.bias != nullptr, do_relu = (p.relu & 1) != 0;
        constexpr int NP4 = 2;      // passes of 16 rows per batch (four spill 13 registers in the 192-row instantiations)
#pragma unroll 1
        for (int rowb = rlo + r8; rowb < rhi; rowb += 16 * NP4) {
          f32x4 y[NP4][2], gate[NP4][2];
#pragma unroll
          for (int it = 0; it < NP4; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int row = rowb + 16 * it + 8 * u;
              const int rr = row < rhi ? row : rlo;
              y[it][u] = *reinterpret_cast<const f32x4*>(st + rr * 32 + cq);
              if (has_rs) gate[it][u] = *reinterpret_cast<const f32x4*>(p.relu_src + (size_t)(ts + rr) * p.ld_relu + col0);
            }
#pragma unroll
          for (int it = 0; it < NP4; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int row = rowb + 16 * it + 8 * u;
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
              *reinterpret_cast<f32x4*>(p.Y + (size_t)(ts + row) * p.ldy + col0) = v;
            }
        }
      } else if (col0 < p.hout) {
        // rows r8, r8+8, ...: operands of 2 rows are requested together, then finished and stored
        for (int row0 = rlo + r8; row0 < rhi; row0 += 16) {
          f32x4 y[2], rs[2], dm[2], ad[2], ps[2];
          float rsc[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = row0 + 8 * u;
            const bool ok = row < rhi;
            const size_t grow = (size_t)(ts + (ok ? row : rlo));
            y[u] = *reinterpret_cast<const f32x4*>(st + (ok ? row : rlo) * 32 + cq);
            if (p.rowscale) rsc[u] = p.rowscale[grow];
            if (p.prebias) ps[u] = *reinterpret_cast<const f32x4*>(p.pre_rowscale + grow * 4);
            if (p.dmask) dm[u] = *reinterpret_cast<const f32x4*>(p.dmask + grow * p.ld_dmask + col0);
            if (p.relu_src) rs[u] = *reinterpret_cast<const f32x4*>(p.relu_src + grow * p.ld_relu + col0);
            if (p.add_src) ad[u] = *reinterpret_cast<const f32x4*>(p.add_src + grow * p.ld_add + col0);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int row = row0 + 8 * u;
            if (row >= rhi) continue;
            f32x4 v = y[u];
            if (p.bias) v += p.rowscale ? bias4 * rsc[u] : bias4;
            if (p.prebias) {
#pragma unroll
              for (int m = 0; m < NMAT; ++m) v += pb4[m] * ps[u][m];
            }
            if (p.dmask) v *= dm[u];
            if (p.drop_id) v *= dropout_mult4(drop_seed, drop_off, (uint32_t)p.drop_id, (uint32_t)(ts + row), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
            if (p.relu & 1) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_nan(v[q]);
            }
            if (p.relu_src) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_open(rs[u][q]) ? v[q] : 0.f;
            }
            if (p.add_src) v += ad[u];
            *reinterpret_cast<f32x4*>(p.Y + (size_t)(ts + row) * p.ldy + col0) = v;
          }
        }
      }
    } else {
      const int colg = cg * 32 + c32;
      if (colg < p.hout) {
        const float bias = p.bias ? p.bias[colg] : 0.f;
        for (int row = rlo + half; row < rhi; row += 2) {
          const size_t grow = (size_t)(ts + row);
          float y = st[row * 32 + c32];
          if (p.bias) y += p.rowscale ? bias * p.rowscale[grow] : bias;
          if (p.prebias) {
#pragma unroll
            for (int m = 0; m < NMAT; ++m) y = fmaf(p.prebias[(size_t)m * p.hout + colg], p.pre_rowscale[grow * 4 + m], y);
          }
          if (p.dmask) y *= p.dmask[grow * p.ld_dmask + colg];
          if (p.drop_id) y *= dropout_mult4(drop_seed, drop_off, (uint32_t)p.drop_id, (uint32_t)grow, (uint32_t)(colg >> 2), p.drop_thr, p.drop_scale)[colg & 3];
          if (p.relu & 1) y = relu_nan(y);
          if (p.relu_src) y = relu_open(p.relu_src[grow * p.ld_relu + colg]) ? y : 0.f;
          if (p.add_src) y += p.add_src[grow * p.ld_add + colg];
          p.Y[grow * p.ldy + colg] = y;
        }
      }
    }
    DSS2_STAMP(5);
  }
}

#ifdef DSS2_STAMPS
}  // namespace dss2
extern "C" int dss2_debug_read_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_stamps), sizeof(unsigned long long) * n);
}
namespace dss2 {
#endif

// ------------------------------------------------------------------------------------------
// narrow outputs (nmat * h <= 32, e.g. the last TAGConv H -> 2): all matrices side by side in ONE
// 32-column MFMA block (a third of the MFMA work of the column-group-per-matrix layout), all 256
// threads stage the tile, wave w multiplies row block w, and the Horner recurrence
// T_m = G_m + P T_{m+1} runs in place across the column blocks of a shared LDS stage.
// ------------------------------------------------------------------------------------------
template <int NRB>
__global__ void __launch_bounds__(256) gemm_narrow_kernel(const dss2_gemm_prop_args p) {
  constexpr int TM = NRB * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int LDX = p.kpad + 4;
  const int D = p.ell_width;
  const int h = p.narrow_h, nm = p.nmat;
  float* Xs = smem;
  float* st = Xs + TM * LDX;                       // [TM][32] shared stage
  int2* ell = reinterpret_cast<int2*>(st + TM * 32);
  int* lrow = reinterpret_cast<int*>(st + TM * 32);
  int2* lent = reinterpret_cast<int2*>(lrow + TM + 2);

  const bool vec_ok = ((p.kreal & 3) == 0) && ((p.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);
  if (vec_ok) {
    // all of a thread's 16-byte loads are issued before the first LDS write (one exposed memory latency instead
    // of one per load: the rolled load -> write loop made this HBM-bound kernel latency-bound, 21.6 us at C2)
    constexpr int PFN = 8;
    const int kq = p.kpad >> 2;
    for (int base = 0; base < TM * kq; base += PFN * 256) {
      f32x4 px[PFN];
#pragma unroll
      for (int i = 0; i < PFN; ++i) {
        const int idx = base + tid + i * 256;
        const int r = idx / kq, c = (idx - r * kq) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < TM * kq && r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
        px[i] = v;
      }
#pragma unroll
      for (int i = 0; i < PFN; ++i) {
        const int idx = base + tid + i * 256;
        const int r = idx / kq, c = (idx - r * kq) << 2;
        if (idx < TM * kq) *reinterpret_cast<f32x4*>(Xs + r * LDX + c) = px[i];
      }
    }
  } else {
    for (int idx = tid; idx < TM * p.kpad; idx += 256) {
      const int r = idx / p.kpad, c = idx - r * p.kpad;
      Xs[r * LDX + c] = (r < R && c < p.kreal) ? p.X[(size_t)(ts + r) * p.ldx + c] : 0.f;
    }
  }
  if (nm > 1) {
    if (D > 0 && p.ell_tiles != nullptr) {
      const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
      for (int idx = tid; idx < D * TM; idx += 256) ell[idx] = src[idx];
    } else if (D > 0) {
      for (int r = tid; r < TM; r += 256) {
        const int e0 = (r < R) ? p.rowptr[ts + r] : 0;
        const int deg = (r < R) ? p.rowptr[ts + r + 1] - e0 : 0;
        for (int k = 0; k < D; ++k)
          ell[k * TM + r] = (k < deg) ? make_int2(p.col[e0 + k] - ts, __float_as_int(p.w[e0 + k])) : make_int2(r, 0);
      }
    } else {
      const int base = p.rowptr[ts];
      const int nnz = p.rowptr[ts + R] - base;
      for (int r = tid; r <= TM; r += 256) lrow[r] = (r <= R) ? (p.rowptr[ts + r] - base) : nnz;
      for (int k = tid; k < nnz; k += 256) lent[k] = make_int2(p.col[base + k] - ts, __float_as_int(p.w[base + k]));
    }
  }
  __syncthreads();

  const int c32 = lane & 31, half = lane >> 5;
  const int nkk = p.kpad >> 3;
  const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(p.Bp);
  for (int rb = wave; rb < NRB; rb += 4) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xa = Xs + (rb * 32 + c32) * LDX + half * 4;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(xa), b0 = bp[lane], a1 = a0, b1 = b0;
    int kk = 0;
    for (; kk + 2 <= nkk; kk += 2) {
      a1 = *reinterpret_cast<const f32x4*>(xa + (kk + 1) * 8);
      b1 = bp[(size_t)(kk + 1) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const int kn = (kk + 2 < nkk) ? kk + 2 : kk + 1;
      a0 = *reinterpret_cast<const f32x4*>(xa + kn * 8);
      b0 = bp[(size_t)kn * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kk < nkk) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) st[(rb * 32 + acc_row(r, half)) * 32 + c32] = acc[r];
  }
  __syncthreads();
  // Horner across column blocks, in place: block m <- G_m + P * block (m+1)
  for (int m = nm - 2; m >= 0; --m) {
    float tmp[2];   // R*h <= 192*10 items over 256 threads: at most a few per thread
    int cnt = 0;
    for (int idx = tid; idx < R * h; idx += 256) {
      const int row = idx / h, j = idx - row * h;
      const float* src = st + (m + 1) * h + j;
      float sacc = st[row * 32 + m * h + j];
      if (D > 0) {
        for (int k = 0; k < D; ++k) {
          const int2 en = ell[k * TM + row];
          sacc = fmaf(__int_as_float(en.y), src[en.x * 32], sacc);
        }
      } else {
        const int e1 = lrow[row + 1];
        for (int e = lrow[row]; e < e1; ++e) {
          const int2 en = lent[e];
          sacc = fmaf(__int_as_float(en.y), src[en.x * 32], sacc);
        }
      }
      st[row * 32 + m * h + j] = sacc;   // block m is read only by its own item: no hazard
      (void)tmp; (void)cnt;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < R * h; idx += 256) {
    const int row = idx / h, j = idx - row * h;
    const size_t grow = (size_t)(ts + row);
    float y = st[row * 32 + j];
    if (p.bias) y += p.rowscale ? p.bias[j] * p.rowscale[grow] : p.bias[j];
    if (p.dmask) y *= p.dmask[grow * p.ld_dmask + j];
    if (p.relu & 1) y = relu_nan(y);
    if (p.relu_src) y = relu_open(p.relu_src[grow * p.ld_relu + j]) ? y : 0.f;
    if (p.add_src) y += p.add_src[grow * p.ld_add + j];
    p.Y[grow * p.ldy + j] = y;
  }
}

// ------------------------------------------------------------------------------------------
// Very narrow outputs (nmat * h <= 8, e.g. the last TAGConv H -> 2 with K = 2): a streaming variant of the
// narrow kernel.  The work is a skinny product (6 outputs per row), i.e. HBM-bound, but the tile kernel above
// keeps a 34 KB X tile in LDS (3 workgroups per CU) and feeds 16 dependent MFMA steps from L2, which makes it
// latency-bound (20 us for 31 MB).  Here nothing of X is staged: four threads own a row, each loads eight
// 16-byte pieces straight into registers (all in flight), multiplies them with the weight rows held in LDS,
// and a 2-step butterfly finishes the dot products; only the [rows x 8] result goes through LDS for the
// Horner recurrence across column blocks.  LDS is ~8 KB, so many workgroups per CU hide the memory latency.
// ------------------------------------------------------------------------------------------
constexpr int NS_MAXO = 8;      // outputs per row (nmat * h), zero padded
template <int NRB>
__global__ void __launch_bounds__(256) gemm_narrow_stream_kernel(const dss2_gemm_prop_args p) {
  constexpr int TM = NRB * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int D = p.ell_width;
  const int h = p.narrow_h, nm = p.nmat, no = nm * h;
  const int kp = p.kpad;
  float* Ws = smem;                               // [NS_MAXO][kp]
  float* Gs = Ws + NS_MAXO * kp;                  // [TM][NS_MAXO]
  int2* ell = reinterpret_cast<int2*>(Gs + TM * NS_MAXO);
  // weights: unpack the MFMA fragment order [k/8][lane][4] of the single 32-column group
  for (int idx = tid; idx < NS_MAXO * kp; idx += 256) {
    const int j = idx / kp, k = idx - j * kp;
    Ws[idx] = j < no ? p.Bp[(((size_t)(k >> 3) * 64 + ((k & 7) >> 2) * 32 + j) << 2) + (k & 3)] : 0.f;
  }
  if (nm > 1) {
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
    for (int idx = tid; idx < D * TM; idx += 256) ell[idx] = src[idx];
  }
  __syncthreads();
  const int cq = kp >> 2;                          // columns per thread (a quarter of the row), multiple of 4
  for (int item = tid; item < TM * 4; item += 256) {
    const int row = item >> 2, q = item & 3;
    float acc[NS_MAXO];
#pragma unroll
    for (int j = 0; j < NS_MAXO; ++j) acc[j] = 0.f;
    const float* xr = p.X + (size_t)(ts + (row < R ? row : 0)) * p.ldx + q * cq;
    for (int c0 = 0; c0 < cq; c0 += 32) {          // eight 16-byte loads in flight per pass
      f32x4 xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + 4 * i;
        xv[i] = (row < R && c < cq && q * cq + c < p.kreal) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + 4 * i;
        if (c < cq) {
#pragma unroll
          for (int j = 0; j < NS_MAXO; ++j) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(Ws + j * kp + q * cq + c);
            acc[j] = fmaf(xv[i][0], wv[0], fmaf(xv[i][1], wv[1], fmaf(xv[i][2], wv[2], fmaf(xv[i][3], wv[3], acc[j]))));
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NS_MAXO; ++j) {             // the four quarters of a row sit in adjacent lanes
      acc[j] += __shfl_xor(acc[j], 1);
      acc[j] += __shfl_xor(acc[j], 2);
    }
    if (q == 0) {
#pragma unroll
      for (int j = 0; j < NS_MAXO; ++j) Gs[row * NS_MAXO + j] = acc[j];
    }
  }
  __syncthreads();
  // Horner across column blocks, in place: block m <- G_m + P * block (m+1)
  for (int m = nm - 2; m >= 0; --m) {
    for (int idx = tid; idx < R * h; idx += 256) {
      const int row = idx / h, j = idx - row * h;
      const float* src = Gs + (m + 1) * h + j;
      float sacc = Gs[row * NS_MAXO + m * h + j];
      for (int k = 0; k < D; ++k) {
        const int2 en = ell[k * TM + row];
        sacc = fmaf(__int_as_float(en.y), src[en.x * NS_MAXO], sacc);
      }
      Gs[row * NS_MAXO + m * h + j] = sacc;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < R * h; idx += 256) {
    const int row = idx / h, j = idx - row * h;
    const size_t grow = (size_t)(ts + row);
    float y = Gs[row * NS_MAXO + j];
    if (p.bias) y += p.rowscale ? p.bias[j] * p.rowscale[grow] : p.bias[j];
    if (p.dmask) y *= p.dmask[grow * p.ld_dmask + j];
    if (p.relu & 1) y = relu_nan(y);
    if (p.relu_src) y = relu_open(p.relu_src[grow * p.ld_relu + j]) ? y : 0.f;
    if (p.add_src) y += p.add_src[grow * p.ld_add + j];
    p.Y[grow * p.ldy + j] = y;
  }
}

static bool narrow_stream_ok(const dss2_gemm_prop_args& a) {
  static const int enabled = [] { const char* e = getenv("DSS2_NARROW_STREAM"); return e ? atoi(e) : 1; }();
  return enabled && a.nmat * a.narrow_h <= NS_MAXO && (a.kpad & 15) == 0 && (a.kreal & 3) == 0 && (a.ldx & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(a.X) & 15) == 0 && (a.nrb <= 4 || a.nrb == 6) && (a.nmat == 1 || (a.ell_width > 0 && a.ell_tiles));
}

template <int NRB>
static int launch_narrow_stream(const dss2_gemm_prop_args& a, hipStream_t stream) {
  const size_t lds = (size_t)NS_MAXO * a.kpad * 4 + (size_t)NRB * 32 * NS_MAXO * 4 +
                     (a.nmat > 1 ? (size_t)NRB * 32 * a.ell_width * 8 : 0);
  hipLaunchKernelGGL(gemm_narrow_stream_kernel<NRB>, dim3(a.ntiles), dim3(256), lds, stream, a);
  return check_launch("gemm_narrow_stream");
}

static size_t narrow_lds_bytes(int nrb, int nmat, int kpad, int max_nnz, int ell_width) {
  const size_t TM = (size_t)nrb * 32;
  size_t b = TM * (size_t)(kpad + 4) * 4 + TM * 32 * 4;
  if (nmat > 1) b += ell_width > 0 ? TM * (size_t)ell_width * 8 : (TM + 2) * 4 + (size_t)max_nnz * 8;
  return b;
}

template <int NRB>
static int launch_narrow(const dss2_gemm_prop_args& a, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_narrow_kernel<NRB>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_narrow")) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(256), narrow_lds_bytes(NRB, a.nmat, a.kpad, a.max_nnz, a.ell_width),
                     stream, a);
  return check_launch("gemm_narrow");
}

static size_t lds_bytes_nw(int nrb, int nmat, int kpad, int nw, int max_nnz, int ell_width) {
  const size_t TM = (size_t)nrb * 32;
  size_t b = TM * (size_t)(kpad + 4) * 4 + (size_t)nw * 32 * (TM + 4) * 4;   // X tile + wave stages
  if (nmat > 1) b += ell_width > 0 ? TM * (size_t)ell_width * 8 : (TM + 2) * 4 + (size_t)max_nnz * 8;
  return b;
}

// waves per workgroup: one per 32-column group up to 4, fewer if the wave-private stages would not
// fit the 160 KiB LDS next to the X tile (large tiles)
static int gemm_waves(int ncg, int nrb, int nmat, int kpad, int max_nnz, int ell_width) {
  int nw = ncg < 4 ? ncg : 4;
  while (nw > 1 && lds_bytes_nw(nrb, nmat, kpad, nw, max_nnz, ell_width) > (size_t)kMaxLdsBytes) --nw;
  return nw;
}

static size_t lds_bytes(int nrb, int nmat, int kpad, int ncg, int max_nnz, int ell_width) {
  return lds_bytes_nw(nrb, nmat, kpad, gemm_waves(ncg, nrb, nmat, kpad, max_nnz, ell_width), max_nnz, ell_width);
}

// bf16x6 variant: the matrix-sequential, K-halved configuration only (every column group its own wave)
static bool gemm16_shape_ok(int nrb, int nmat, int kreal, int hout, int max_nnz, int ell_width) {
  const int kpad = (kreal + 15) / 16 * 16, ncg = (hout + 31) / 32;
  const bool seq = nmat > 1 && nrb * nmat >= 16;
  return seq && (nrb == 6 || nrb == 4) && ncg <= 4 && (kpad & 31) == 0 && (kreal & 3) == 0 && (hout & 3) == 0 && ell_width > 0 && ell_width <= 32 &&
         gemm_waves(ncg, nrb, nmat, kpad, max_nnz, ell_width) < ncg &&
         lds_bytes_nw(nrb, nmat, kpad / 2, ncg, max_nnz, ell_width) <= (size_t)kMaxLdsBytes;
}

template <int NRB, int NMAT, bool B16 = false, int RSP = 1>
static int launch(const dss2_gemm_prop_args& a_in, hipStream_t stream) {
  dss2_gemm_prop_args a = a_in;
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_prop_kernel<NRB, NMAT, B16, RSP>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_prop")) return 1;
  size_t lds = lds_bytes(NRB, a.prop_in > 0 ? 2 : NMAT, a.kpad, a.ncg, a.max_nnz, a.ell_width);
  int nw = gemm_waves(a.ncg, NRB, a.prop_in > 0 ? 2 : NMAT, a.kpad, a.max_nnz, a.ell_width);
  // tall tiles in matrix-sequential mode: stage the X tile in two K halves if that lets every column group have its wave
  constexpr bool seq = NMAT > 1 && NRB * NMAT >= 16;
  static const int kh_env = [] { const char* e = getenv("DSS2_GEMM_KHALF"); return e ? atoi(e) : 1; }();
  const bool vec = ((a.kreal & 3) == 0) && ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  if (seq && kh_env && a.prop_in == 0 && a.ncg <= 4 && nw < a.ncg && (a.kpad & 15) == 0 && vec && a.ell_width > 0 &&
      lds_bytes_nw(NRB, NMAT, a.kpad / 2, a.ncg, a.max_nnz, a.ell_width) <= (size_t)kMaxLdsBytes) {
    nw = a.ncg;
    lds = lds_bytes_nw(NRB, NMAT, a.kpad / 2, a.ncg, a.max_nnz, a.ell_width);
    a.relu |= 1 << 24;
  } else if (B16) {
    set_error("gemm_prop(bf16x6): needs the K-halved matrix-sequential configuration (nrb=%d nmat=%d kpad=%d ncg=%d)", NRB, NMAT, a.kpad, a.ncg);
    return 2;
  }
  // persistent over tiles: at most two workgroups per CU are co-resident at the LDS sizes of the
  // compute-heavy shapes, so 512 workgroups cover the chip; each walks tiles blockIdx.x, +grid, ...
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * nw * RSP), lds, stream, a);
  return check_launch("gemm_prop");
}

}  // namespace dss2

extern "C" size_t dss2_gemm_prop_lds_bytes(int nrb, int nmat, int kpad, int ncg, int max_nnz, int ell_width) {
  return dss2::lds_bytes(nrb, nmat, kpad, ncg, max_nnz, ell_width);
}

extern "C" int dss2_gemm_prop16_supported(int nrb, int nmat, int kreal, int hout, int max_nnz, int ell_width) {
  if (!((nrb == 6 && (nmat == 3 || nmat == 4)) || (nrb == 4 && nmat == 4))) return 0;
  return dss2::gemm16_shape_ok(nrb, nmat, kreal, hout, max_nnz, ell_width) ? 1 : 0;
}

static int dss2_pack_weights_launch(const dss2_pack_desc* descs, int n_desc, int max_elems, void* stream);
extern "C" int dss2_pack_weights(const dss2_pack_desc* descs, int n_desc, int max_elems, void* stream) {
  DSS2_RECORD([descs, n_desc, max_elems](void* s_) { return dss2_pack_weights_launch(descs, n_desc, max_elems, s_); });
  return dss2_pack_weights_launch(descs, n_desc, max_elems, stream);
}
static int dss2_pack_weights_launch(const dss2_pack_desc* descs, int n_desc, int max_elems, void* stream) {
  if (n_desc <= 0) return 0;
  dim3 grid((max_elems + 255) / 256, n_desc);
  hipLaunchKernelGGL(dss2::pack_weights_kernel, grid, dim3(256), 0, dss2::as_stream(stream), descs);
  return dss2::check_launch("pack_weights");
}

static int dss2_gemm_prop_launch(const dss2_gemm_prop_args* ap, void* stream);
extern "C" int dss2_gemm_prop(const dss2_gemm_prop_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_gemm_prop: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_gemm_prop_launch(&a, s_); });
  return dss2_gemm_prop_launch(ap, stream);
}
static int dss2_gemm_prop_launch(const dss2_gemm_prop_args* ap, void* stream) {
  using namespace dss2;
  const dss2_gemm_prop_args& a = *ap;
  if (a.ntiles <= 0) return 0;
  if ((a.kpad & 7) || a.kpad < a.kreal || a.kpad <= 0) { set_error("gemm_prop: bad kpad %d (kreal %d)", a.kpad, a.kreal); return 2; }
  if (a.ncg * 32 < a.hout || a.ncg <= 0) { set_error("gemm_prop: ncg %d too small for hout %d", a.ncg, a.hout); return 2; }
  if ((a.nmat > 1 || a.prop_in > 0) && (!a.rowptr || !a.col || !a.w)) { set_error("gemm_prop: propagation needs a CSR"); return 2; }
  if (a.prop_in > 0 && (a.nmat != 1 || a.kreal % (a.prop_in + 1) != 0)) { set_error("gemm_prop: prop_in needs nmat == 1 and kreal divisible by prop_in+1"); return 2; }
  if (a.prebias && (!a.pre_rowscale || a.narrow_h > 0 || a.prop_in > 0)) { set_error("gemm_prop: prebias needs pre_rowscale and the general kernel"); return 2; }
  if (a.drop_id && (!a.drop_state || a.narrow_h > 0)) { set_error("gemm_prop: in-kernel dropout needs drop_state and the general kernel"); return 2; }
  if (a.narrow_h > 0) {
    if (a.nmat * a.narrow_h > 32 || a.hout != a.narrow_h || a.prop_in) { set_error("gemm_prop: narrow mode needs nmat*narrow_h <= 32 and hout == narrow_h"); return 2; }
    if (narrow_lds_bytes(a.nrb, a.nmat, a.kpad, a.max_nnz, a.ell_width) > (size_t)kMaxLdsBytes) { set_error("gemm_prop(narrow): tile does not fit LDS"); return 3; }
    hipStream_t sn = as_stream(stream);
    if (narrow_stream_ok(a)) {
      switch (a.nrb) {
        case 1: return launch_narrow_stream<1>(a, sn);
        case 2: return launch_narrow_stream<2>(a, sn);
        case 3: return launch_narrow_stream<3>(a, sn);
        case 6: return launch_narrow_stream<6>(a, sn);
        default: return launch_narrow_stream<4>(a, sn);
      }
    }
    switch (a.nrb) {
      case 1: return launch_narrow<1>(a, sn);
      case 2: return launch_narrow<2>(a, sn);
      case 3: return launch_narrow<3>(a, sn);
      case 4: return launch_narrow<4>(a, sn);
      case 6: return launch_narrow<6>(a, sn);
      default: set_error("gemm_prop(narrow): unsupported nrb=%d", a.nrb); return 2;
    }
  }
  if (a.ell_width < 0 || a.ell_width > 32) { set_error("gemm_prop: ell_width %d out of range 0..32", a.ell_width); return 2; }
  if (lds_bytes(a.nrb, a.prop_in > 0 ? 2 : a.nmat, a.kpad, a.ncg, a.max_nnz, a.ell_width) > (size_t)kMaxLdsBytes) {
    set_error("gemm_prop: tile needs %zu B of LDS (> 160 KiB): nrb=%d kpad=%d nnz=%d",
              lds_bytes(a.nrb, a.prop_in > 0 ? 2 : a.nmat, a.kpad, a.ncg, a.max_nnz, a.ell_width), a.nrb, a.kpad, a.max_nnz);
    return 3;
  }
  hipStream_t s = as_stream(stream);
  if (a.b_format == 1) {
    if (!gemm16_shape_ok(a.nrb, a.nmat, a.kreal, a.hout, a.max_nnz, a.ell_width) || a.prop_in || a.rowscale || (a.kpad & 15) ||
        a.kpad != (a.kreal + 15) / 16 * 16 || (a.ldx & 3) || (reinterpret_cast<uintptr_t>(a.X) & 15)) {
      set_error("gemm_prop(bf16x6): unsupported shape (nrb=%d nmat=%d k=%d kpad=%d hout=%d)", a.nrb, a.nmat, a.kreal, a.kpad, a.hout);
      return 2;
    }
    static const int rs_env = [] { const char* e = getenv("DSS2_GEMM_RS"); return e ? atoi(e) : 2; }();
    if (a.nrb == 6 && a.nmat == 3) return rs_env == 2 ? launch<6, 3, true, 2>(a, s) : launch<6, 3, true>(a, s);
    if (a.nrb == 6 && a.nmat == 4) return rs_env == 2 ? launch<6, 4, true, 2>(a, s) : launch<6, 4, true>(a, s);
    if (a.nrb == 4 && a.nmat == 4) return rs_env == 2 ? launch<4, 4, true, 2>(a, s) : launch<4, 4, true>(a, s);
    set_error("gemm_prop(bf16x6): no instantiation for nrb=%d nmat=%d", a.nrb, a.nmat);
    return 2;
  }
  if (a.b_format != 0) { set_error("gemm_prop: unknown b_format %d", a.b_format); return 2; }
#define DSS2_CASE(NRB, NMAT) \
  if (a.nrb == NRB && a.nmat == NMAT) return launch<NRB, NMAT>(a, s);
  DSS2_CASE(1, 1) DSS2_CASE(1, 2) DSS2_CASE(1, 3) DSS2_CASE(1, 4)
  DSS2_CASE(2, 1) DSS2_CASE(2, 2) DSS2_CASE(2, 3) DSS2_CASE(2, 4)
  DSS2_CASE(3, 1) DSS2_CASE(3, 2) DSS2_CASE(3, 3) DSS2_CASE(3, 4)
  DSS2_CASE(4, 1) DSS2_CASE(4, 2) DSS2_CASE(4, 3) DSS2_CASE(4, 4)
  DSS2_CASE(6, 1) DSS2_CASE(6, 2) DSS2_CASE(6, 3) DSS2_CASE(6, 4)
#undef DSS2_CASE
  set_error("gemm_prop: unsupported (nrb=%d, nmat=%d); nrb in {1,2,3,4,6}, nmat in 1..4", a.nrb, a.nmat);
  return 2;
}
