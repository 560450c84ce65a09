// Layer chain for 64-row tiles with one wave per 32-column group (C2: H = 128), "split planes" form of the bf16x6 chain.
//
// The bf16x6 chain of dss2_gemm_chain_kernel.hpp keeps the activation tile as fp32 in LDS and every wave splits the A
// fragments it reads into their three bf16 pieces in registers: all four column-group waves split the same rows (4 x
// redundant, ~770 of a wave's ~1950 VALU instructions per layer), and the Horner hops gather scalar by scalar in the MFMA
// accumulator layout (512 LDS reads per wave and layer).  Round 2's PMC pass: matrix pipe busy 42 %, 6.8 VALU per MFMA.
//
// Here the tile lives in LDS already split: per 32-column stripe three bf16 planes [64 rows][32 k] (80-byte rows: the
// ds_read_b128 lane groups of an A fragment fall on 16 distinct 16-byte slots).  An A fragment is three ds_read_b128, the
// GEMM phase has no VALU work besides addresses; every element is split ONCE, by the wave that produces it in the epilogue.
// The stripe of column group w is written only by wave w, and it aliases wave w's Horner slots (two [64][32] fp32 slots,
// 16 KB per wave): after the barrier that ends the GEMM phase nobody reads the planes any more, the hops are wave-private
// (a wave owns all 64 rows of its columns), and the epilogue puts the next layer's planes over its own slots.  Two barriers
// per layer as before; LDS 64 KB + the ELL slice, two workgroups per CU.
// The hops run on 16-byte row pieces (a lane owns 4 columns of 8 rows: one ELL entry + one ds_read_b128 per neighbour) and
// end in the lanes and the row-major form the epilogue wants: 144 LDS reads per wave and layer instead of 520.
// Same MFMA sequence per accumulator and same fma order per hop as the fp32-tile form: bitwise the same results.
//
// Compiled without packed fp32 ops like the other bf16x6 translation units (build.sh, dss2_gemm_chain16.hip).
#include <stdlib.h>

#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

constexpr int SP_TM = 64;
constexpr int SP_RS = 40;                      // bf16 per plane row: 32 k + 8 pad
constexpr int SP_PLANE = SP_TM * SP_RS;        // bf16 per plane
constexpr int SP_REGION = 4096;                // floats per column group: two [64][32] fp32 slots = 16 KB >= three planes (15 KB)
constexpr int SP_SLOT = SP_TM * 32;

__device__ __forceinline__ void sp_barrier() {      // LDS-only hand-off: the Y stores of the epilogue stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// the three bf16 pieces of 4 consecutive values -> the three planes (8 bytes each)
__device__ __forceinline__ void sp_store_split(__bf16* dst, const f32x4 v) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(v[0], v[1], h0, m0, l0);
  split3_pair(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(dst + SP_PLANE) = u32x2{m0, m1};
  *reinterpret_cast<u32x2*>(dst + 2 * SP_PLANE) = u32x2{l0, l1};
}

// f16x3 form (MS = 2): the two fp16 pieces of 4 consecutive values scaled by 2^e -> planes 0, 1
__device__ __forceinline__ void sp_store_split_h(__bf16* dst, const f32x4 v, int e) {
  uint32_t h0, l0, h1, l1;
  const float s = pow2f(e);      // (v_mul_f32: half the issue cycles of v_ldexp_f32, the same bits)
  split2_pair(v[0] * s, v[1] * s, h0, l0);
  split2_pair(v[2] * s, v[3] * s, h1, l1);
  *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(dst + SP_PLANE) = u32x2{l0, l1};
}

// HM (fused narrow head, dss2_gemm_prop_chain_head): 0 none; 1 forward -- after the last chained layer the tile is still in
// the waves' registers: the head TAGConv (hid -> nout <= 4) is computed from them (per wave the partial products of its 32
// columns, summed over the waves in LDS, two hops on nout-wide rows by wave 0) instead of a launch that re-reads [N, hid];
// 2 backward -- the chain's input tile (the head's data gradient, gated by the head's input activation / dropout mask) is
// computed in the staging from the nout-wide upstream gradient instead of being written and re-read by a launch of its own.
constexpr int SP_HEAD_MAX = 4;      // nout

// MS: the tile GEMM's MFMA shape.  0: v_mfma_f32_32x32x16_bf16, two 32-row blocks x NMAT accumulator blocks per wave (six dependent MFMAs
// per block and k-step).  1: v_mfma_f32_16x16x32_bf16 (round 4), four 16-row x two 16-column blocks per matrix -- 24 independent
// accumulator chains of NMAT interleaved: in the chain's GEMM regime in isolation the 16x16 shape keeps the pipe at 92 % with two
// waves per SIMD, the 32x32 shape at 73 % (tools/micro/mfma_shape_clock.hip).  Same operands: the A fragment of a 16-row block is
// rows x 8 k at k = 32 kb + 8 (lane >> 4) of the same planes, the B fragment is read out of the SAME packed weights (fragment order of
// the 32x32x16 form: k-group 2 kb + (lane >> 5), source lane ((lane >> 4) & 1) * 32 + 16 nb + (lane & 15)).  Only the GEMM phase and the
// accumulator hand-off (put) differ; sums are formed in another order, so results differ from MS = 0 by rounding.  K a multiple of 32.
// MS = 2 (round 5): MS = 1's structure as f16x3 -- the tile lives in LDS as TWO fp16 planes per stripe, scaled by 2^ea (ea: a per-tile,
// per-layer exponent that puts the tile's largest |value| into [2^14, 2^15)); the weights come as two fp16 planes scaled by 2^ew[m][j] per
// matrix and output column (dss2_pack_desc, transpose bit 3: the exponents follow the planes); three v_mfma_f32_16x16x32_f16 per product
// group (lo hi + hi lo + hi hi) instead of six; the accumulators leave the GEMM phase multiplied by 2^-(ea + ew[m][j]) (v_ldexp in the hand-off: exact),
// so the hops and the epilogue are those of the bf16x6 form.  The tile maximum: every wave's maximum over its stripe of the finished
// layer output (DPP), four LDS words, ONE more barrier per layer.  Errors of the size of fp32 arithmetic's own (dss2_wgrad16h.hip).
// Layers gated by fp32 activations and X plane images: not in this form (the host routes them to MS = 1).
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
template <int NMAT, int NW, int HM, int MS>
__global__ void __launch_bounds__(NW * 64, 2) gemm_chain_sp_kernel(const dss2_gemm_prop_args p, const ChainTable ct, const dss2_chain_head hd) {
  constexpr int TM = SP_TM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int ncg = nthreads >> 6;
  const int tile = blockIdx.x;
  constexpr bool F16 = MS == 2;
  constexpr int NP = F16 ? 2 : 3;      // planes per stripe / per weight fragment group
  const uint64_t drop_seed = p.drop_state ? p.drop_state[0] : 0, drop_off = p.drop_state ? p.drop_state[1] : 0;
  const bool probe = ct.clock_probe != nullptr && (tile & 255) == 0 && tile < 1024;      // (uniform; diagnostic, see ChainTable)
  if (probe && tid == 0) {      // (written at once: nothing of the probe stays live across the kernel)
    unsigned long long* q = ct.clock_probe + (tile >> 8) * 4;
    q[0] = __builtin_amdgcn_s_memtime(); q[1] = __builtin_amdgcn_s_memrealtime();
  }
  __bf16* xpl = reinterpret_cast<__bf16*>(smem);               // stripe s: xpl + s * (2 * SP_REGION)
  int2* ell = reinterpret_cast<int2*>(smem + ncg * SP_REGION);
  const int D = p.ell_width;
  float* mxw = reinterpret_cast<float*>(ell + D * TM);      // (MS = 2) [ncg]: every wave's maximum over its part of the tile
  int ea = 0;                                                // (MS = 2) the planes in LDS hold 2^ea x
  auto tile_exponent = [&]() {      // after a barrier behind the waves' writes of mxw
    float m = 0.f;
    for (int w = 0; w < ncg; ++w) m = fmaxf(m, mxw[w]);
    return 14 - __builtin_amdgcn_readfirstlane(exp_of(m));
  };
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int kq = p.kpad >> 2;

  const int c32 = lane & 31, half = lane >> 5;
  const int cg = wave;
  const int nks = p.kpad >> 4;
  const __bf16* xa = xpl + c32 * SP_RS + half * 8;
  float* slot0 = smem + cg * SP_REGION;
  __bf16* own_planes = xpl + cg * (2 * SP_REGION);
  const int cq = (lane & 7) * 4, r8 = lane >> 3;
  const int col0 = cg * 32 + cq;
  const bool col_ok = col0 < p.hout;

  // ---- stage the tile's ELL slice and the first layer's input tile as split planes (zero padded to 64 x kpad)
  {
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
    for (int idx = tid; idx < D * TM; idx += nthreads) ell[idx] = src[idx];
  }
  if constexpr (HM != 2 && !F16) {
    for (int idx = tid; idx < TM * kq; idx += nthreads) {
      const int r = idx / kq, c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      sp_store_split(xpl + (c >> 5) * (2 * SP_REGION) + r * SP_RS + (c & 31), v);
    }
  } else if constexpr (HM != 2) {
    // the input tile waits in registers (TM kq / threads = kpad / (4 ncg) <= 8 row pieces per thread) while its maximum is formed
    f32x4 xin[8];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = tid + j * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      const bool on = idx < TM * kq && r < R && c < p.kreal;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + (on ? r : 0)) * p.ldx + (on ? c : 0));      // (unconditional, masked)
      xin[j] = on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      mx = absmax4(mx, xin[j]);
    }
    mx = wave_max(mx);
    if (lane == 0) mxw[wave] = mx;
    sp_barrier();
    ea = tile_exponent();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = tid + j * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      if (idx < TM * kq) sp_store_split_h(xpl + (c >> 5) * (2 * SP_REGION) + r * SP_RS + (c & 31), xin[j], ea);
    }
  } else {
    // X[row][c] = gate(row, c) * sum_{m, o} ((P^T)^m G)[row][o] W_m[o][c]: every wave builds its own 32-column stripe.
    // Gate source rows (the head's input activation) requested first; the hops of G are nout wide and run once per WAVE
    // (lane = row, wave-private scratch in the wave's own region, which the planes below overwrite at the very end).
    const int nout = hd.nout;
    f32x4 ga[8];
    if (hd.gate && col_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int row = r8 + 8 * i; ga[i] = *reinterpret_cast<const f32x4*>(hd.gate + (size_t)(ts + (row < R ? row : 0)) * hd.ld_gate + col0); }
    }
    f32x4 wl[NMAT][SP_HEAD_MAX];      // W_m[o][col0 .. col0 + 3]
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int o = 0; o < SP_HEAD_MAX; ++o)
        wl[m][o] = (o < nout && col_ok) ? *reinterpret_cast<const f32x4*>(hd.W[m] + (size_t)o * p.hout + col0) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* zt = slot0;                       // [64][NMAT * 4]: ((P^T)^m G)[row][o]
    float* hs = slot0 + TM * NMAT * 4;       // [64][4] hop scratch
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (lane < R) {
#pragma unroll
      for (int o = 0; o < SP_HEAD_MAX; ++o) if (o < nout) z[o] = hd.G[(size_t)(ts + lane) * hd.ldg + o];
    }
    *reinterpret_cast<f32x4*>(zt + lane * (NMAT * 4)) = z;
    const float zg0 = z[0], zg1 = z[1];      // (the upstream gradient of this lane's row: the head's bias sums below)
    sp_barrier();                            // the ELL slice is staged
#pragma unroll
    for (int m = 1; m < NMAT; ++m) {
      *reinterpret_cast<f32x4*>(hs + lane * 4) = z;
      wave_lds_sync();
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < D; ++k) {
        const int2 en = ell[k * TM + lane];
        t += *reinterpret_cast<const f32x4*>(hs + en.x * 4) * __int_as_float(en.y);
      }
      wave_lds_sync();
      z = t;
      *reinterpret_cast<f32x4*>(zt + lane * (NMAT * 4) + m * 4) = z;
    }
    wave_lds_sync();
    // The head's WEIGHT gradient rides here too (round 5; hd.wg_slab, nout <= 2): dW_m[o][c] = sum_rows ((P^T)^m G)[row][o] h[row][c] needs
    // exactly what this block holds -- the hop results zt and the head's input rows h (= the gate rows ga) -- so the launch that
    // re-read the [N, hid] activation for it (wgrad_narrow_stream_kernel, 15 us at C2) is gone.  Per lane: its 8 rows x 4 columns x
    // NMAT x nout partial sums, summed over the 8 lanes that share the columns through wave-private LDS in row order: one slab per
    // tile, [NMAT nout][hid] + nout bias sums, reduced with the step's other slabs (fixed order: reproducible).
    const bool wg = hd.wg_slab != nullptr && hd.gate != nullptr;
    f32x4 hw[NMAT][2];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) { hw[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hw[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = r8 + 8 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NMAT; ++m) {
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zt + row * (NMAT * 4) + m * 4);
#pragma unroll
        for (int o = 0; o < SP_HEAD_MAX; ++o) v += wl[m][o] * zz[o];
        if (wg && col_ok) {      // (rows beyond the tile: zz = 0; ga was read from a clamped row)
          hw[m][0] += ga[i] * zz[0];
          hw[m][1] += ga[i] * zz[1];
        }
      }
      if (hd.gate) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = relu_open(ga[i][q]) ? v[q] : 0.f;
      }
      if (hd.drop_id) v *= dropout_mult4(drop_seed, drop_off, (uint32_t)hd.drop_id, (uint32_t)(ts + row), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
      if (row >= R || !col_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      else *reinterpret_cast<f32x4*>(hd.Xout + (size_t)(ts + row) * hd.ldxo + col0) = v;      // (the weight gradients read it)
      xv[i] = v;
    }
    if (wg) {      // (uniform)
      float* scr = slot0 + 2048;             // [64 lanes][NMAT 2][4]: behind zt / hs inside the wave's own region, before the planes go over it
#pragma unroll
      for (int m = 0; m < NMAT; ++m) {
        *reinterpret_cast<f32x4*>(scr + (lane * (NMAT * 2) + 2 * m) * 4) = hw[m][0];
        *reinterpret_cast<f32x4*>(scr + (lane * (NMAT * 2) + 2 * m + 1) * 4) = hw[m][1];
      }
      wave_lds_sync();
      const int per = NMAT * 2 * 4;          // values per (row group, column group)
      float* slab = hd.wg_slab + (size_t)tile * (hd.pad > 0 ? (size_t)hd.pad : (size_t)NMAT * nout * p.hout + nout);      // (pad: the slabs' stride in floats)
      for (int idx = lane; idx < 8 * per; idx += 64) {
        const int cqi = idx / per, rem = idx - cqi * per, mo = rem >> 2, q = rem & 3, m = mo >> 1, o = mo & 1;
        float sum = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) sum += scr[((g8 * 8 + cqi) * (NMAT * 2) + mo) * 4 + q];      // rows r8 = 0 .. 7 in order
        const int col = cg * 32 + cqi * 4 + q;
        if (o < nout && col < p.hout) slab[(size_t)(m * nout + o) * p.hout + col] = sum;
      }
      if (wave == 0) {                       // bias sums: the upstream gradient's column sums over the tile's rows, a butterfly over the
        float sb0 = zg0, sb1 = zg1;          // 64 lanes (= rows): pairwise, fixed order
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { sb0 += __shfl_xor(sb0, d); sb1 += __shfl_xor(sb1, d); }
        if (lane == 0) {
          slab[(size_t)NMAT * nout * p.hout] = sb0;
          if (nout > 1) slab[(size_t)NMAT * nout * p.hout + 1] = sb1;
        }
      }
    }
    if constexpr (F16) {
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = absmax4(mx, xv[i]);
      mx = wave_max(mx);
      if (lane == 0) mxw[wave] = mx;
      sp_barrier();
      ea = tile_exponent();
    }
    wave_lds_sync();                         // every lane is done with zt (and the scratch behind it): the planes go over it
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (F16) sp_store_split_h(own_planes + (r8 + 8 * i) * SP_RS + cq, xv[i], ea);
      else sp_store_split(own_planes + (r8 + 8 * i) * SP_RS + cq, xv[i]);
    }
  }

  bf16x8 b0[NP][NMAT];
  auto load_b = [&](const bf16x8* __restrict__ bp16, bf16x8 (&b)[NP][NMAT], int ks) {
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) b[pl][m] = bp16[(((size_t)(m * ncg + cg) * nks + ks) * NP + pl) * 64 + lane];
  };
  // MS = 1: the fragments of k-step kb (32 k) and 16-column block nb, out of the 32x32x16 fragment order
  // (a uniform 64-bit base per matrix -- scalar registers -- plus ONE 32-bit lane offset: with per-matrix 64-bit lane pointers the
  //  16x16 form kept three pointer pairs in scratch memory and reloaded them, each behind an s_waitcnt vmcnt(0), at the top of every k-step)
  const uint32_t b16_lane = (uint32_t)(((lane >> 5) * NP * 64 + ((lane >> 4) & 1) * 32 + (lane & 15)) * 16);
  auto load_b16 = [&](const bf16x8* __restrict__ bp16, bf16x8 (&b)[NP][NMAT], int kb, int nb) {
#pragma unroll
    for (int m = 0; m < NMAT; ++m) {
      const char* mbase = reinterpret_cast<const char*>(bp16) + (size_t)(uint32_t)(((m * ncg + cg) * nks + 2 * kb) * (NP * 1024) + nb * 256);      // uniform
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) b[pl][m] = *reinterpret_cast<const bf16x8*>(mbase + (size_t)(b16_lane + (uint32_t)(pl * 1024)));
    }
  };
  if constexpr (MS == 0) load_b(reinterpret_cast<const bf16x8*>(ct.l[0].Bp), b0, 0);
  else load_b16(reinterpret_cast<const bf16x8*>(ct.l[0].Bp), b0, 0, 0);
  CSTAMP(0);
  sp_barrier();
  CSTAMP(1);
  CSTAMP_RT(62);

  for (int li = 0; li < ct.n; ++li) {
    const dss2_chain_layer& L = ct.l[li];    // uniform: scalar loads from the kernel-argument segment
    const bf16x8* __restrict__ bp16 = reinterpret_cast<const bf16x8*>(L.Bp);
    f32x16 acc[MS == 0 ? 2 : 1][NMAT];
    f32x4_acc c16[MS >= 1 ? 4 : 1][2][NMAT];
    [[maybe_unused]] float su[NMAT][2];    // (MS = 2) 2^ue
    [[maybe_unused]] int ue[NMAT][2];      // (MS = 2) what takes the scales out of matrix m's accumulators, per output column of the lane

    if constexpr (MS >= 1) {
      // ---- tile GEMM on 16x16x32 MFMAs: a k-step (32 k) is two half-steps, one per 16-column block; the four row blocks' A fragments
      // live for the whole k-step and are re-requested for the next one right after their last use; B fragments ping-pong one half-step ahead
      const int nkb = p.kpad >> 5;
      bf16x8 b1[NP][NMAT], a[4][NP];
      auto load_a16 = [&](bf16x8 (&af)[NP], int mb, int kb) {
        const __bf16* src = xpl + kb * (2 * SP_REGION) + (mb * 16 + (lane & 15)) * SP_RS + (lane >> 4) * 8;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(src + pl * SP_PLANE);
      };
      auto mma16 = [&](const bf16x8 (&af)[NP], const bf16x8 (&b)[NP][NMAT], f32x4_acc (&c)[NMAT], const bool first) {
        const f32x4_acc zero = {0.f, 0.f, 0.f, 0.f};
        if constexpr (F16) {      // lo hi + hi lo + hi hi, smallest terms first
          auto h8 = [](const bf16x8 v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(af[1]), h8(b[0][m]), first ? zero : c[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(af[0]), h8(b[1][m]), c[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(af[0]), h8(b[0][m]), c[m], 0, 0, 0);
          return;
        }
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NP - 1], b[0][m], first ? zero : c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b[NP - 1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], b[0][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], b[0][m], c[m], 0, 0, 0);
      };
      // one k-step: b0 holds (kb, column block 0); branch-free (the last step re-requests its own operands)
      auto kstep = [&](int kb, const bool first) {
        const int kn = kb + 1 < nkb ? kb + 1 : kb;
        load_b16(bp16, b1, kb, 1);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) mma16(a[mb], b0, c16[mb][0], first);
        load_b16(bp16, b0, kn, 0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { mma16(a[mb], b1, c16[mb][1], first); load_a16(a[mb], mb, kn); }
        // one memory request per few MFMAs: half-step 0 carries the NP NMAT weight fragments of half-step 1, half-step 1 those of the next
        // k-step and, behind each row block's MFMAs, that block's NP plane fragments
        constexpr int NPR = F16 ? 3 : 6;              // MFMAs per product group
        constexpr int NQ = NP * NMAT;                 // weight requests per half-step
        constexpr int RBM = NPR * NMAT;               // MFMAs per row block and half-step
        // half-step 0 (4 RBM MFMAs): its NQ weight requests, one per two MFMAs, then the rest
#pragma unroll
        for (int i = 0; i < NQ; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * RBM - 2 * NQ, 0);
        // half-step 1: row block 0's RBM MFMAs carry the next weight requests; behind them row block 0's plane registers are free and
        // row block j's MFMAs carry the NP plane reads of row block j - 1; the last NP reads close the step
#pragma unroll
        for (int i = 0; i < NQ; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, F16 ? 1 : 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        if constexpr (RBM - (F16 ? 1 : 2) * NQ > 0) __builtin_amdgcn_sched_group_barrier(0x008, RBM - (F16 ? 1 : 2) * NQ, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (F16) {
            __builtin_amdgcn_sched_group_barrier(0x008, (RBM + 1) / 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, RBM / 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, RBM / 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
      };
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) load_a16(a[mb], mb, 0);
#ifndef DSS2_SP_NOPRIO
      __builtin_amdgcn_s_setprio(2);
#endif
      kstep(0, true);
      for (int kb = 1; kb < nkb; ++kb) kstep(kb, false);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- tile GEMM, 16 k per step: B fragments (L2) ping-pong one step ahead, A fragments (LDS planes) one row block ahead
    if constexpr (MS == 0) {
      bf16x8 b1[3][NMAT], a[2][3];
      auto load_a = [&](bf16x8 (&af)[3], int rb, int ks) {
        const __bf16* src = xa + (ks >> 1) * (2 * SP_REGION) + rb * 32 * SP_RS + (ks & 1) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(src + pl * SP_PLANE);
      };
      auto mma = [&](const bf16x8 (&af)[3], const bf16x8 (&b)[3][NMAT], f32x16 (&c)[NMAT], const bool first) {
        // per accumulator the smallest terms first (same sequence as the fp32-tile form); the NMAT chains are independent
        // (first step of a layer: the accumulator operand is the inline constant 0 -- no 96 v_mov per layer)
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0][m], first ? zero : c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[2][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0][m], c[m], 0, 0, 0);
      };
      // (no branches inside the k loop: the last step re-requests its own operands.  A conditional prefetch splits the loop
      //  body into basic blocks, and the waitcnt pass then waits for the NEW requests at the join: one L2 latency per step)
      auto step = [&](const bf16x8 (&bc)[3][NMAT], bf16x8 (&bn)[3][NMAT], int ks, const bool first) {
        const int kn = ks + 1 < nks ? ks + 1 : ks;
        load_a(a[1], 1, ks);
        load_b(bp16, bn, kn);
        mma(a[0], bc, acc[0], first);
        load_a(a[0], 0, kn);
        mma(a[1], bc, acc[1], first);
        // one memory instruction per MFMA gap (a request issued in a cluster holds the SIMD's vector issue with the matrix pipe
        // idle: 12.1 K cycles per 288 MFMAs with the 15 requests of a step clustered, 32 cycles per MFMA = 9.2 K is the floor):
        // gaps 1-3 this step's second row block (LDS), 4-12 the next step's weight fragments (L2), 19-21 the next first row block
#pragma unroll
        for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int i = 0; i < 3 * NMAT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMAT - 3 - 3 * NMAT, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMAT - 3, 0);
      };
      // (b0 holds step 0's fragments: requested before the first barrier / at the start of the previous layer's epilogue)
      load_a(a[0], 0, 0);
#ifndef DSS2_SP_NOPRIO
      __builtin_amdgcn_s_setprio(2);      // the matrix-pipe phase wins the SIMD's issue arbitration over the other workgroup's Horner / epilogue
#endif
      step(b0, b1, 0, true);
      int ks = 1;
      for (; ks + 2 <= nks; ks += 2) {
        step(b1, b0, ks, false);
        step(b0, b1, ks + 1, false);
      }
      if (ks < nks) step(b1, b0, ks, false);
      __builtin_amdgcn_s_setprio(0);
    }
    CSTAMP(2 + li * 6 + 0);      // GEMM phase done
    // ---- everything the epilogue reads from HBM per row, requested before the hops
    // (rowv: this lane's first row, made opaque once per layer so that the per-row 64-bit addresses of Y / relu_src / ... are
    //  computed where they are used instead of being hoisted out of the layer loop into 60 spilled registers)
    int rowv = r8;
    asm volatile("" : "+v"(rowv));
    if constexpr (F16) {      // -(ea + ew[m][column]), column 16 nb + (lane & 15) of the wave's stripe: requested here (the fragment registers are free
                              // now; at the top of the layer the six values were spilled across the GEMM phase), used in the hand-off behind the barrier
      const int* whdr = reinterpret_cast<const int*>(reinterpret_cast<const char*>(L.Bp) + (size_t)NMAT * ncg * nks * 2048) + cg * 32 + (lane & 15);
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          ue[m][nb] = -(ea + whdr[m * ncg * 32 + nb * 16]);
          // the scale as a float, once per matrix and column: 96 v_mul_f32 in the hand-off instead of 96 v_ldexp_f32 (twice the issue cycles).
          // Beyond fp32's exponent range (an all-zero or a diverged tile: |ea + ew| > 126) the exponent is clamped -- the accumulators it meets are 0 / non-finite.
          su[m][nb] = ldexpf(1.f, ue[m][nb] < -126 ? -126 : (ue[m][nb] > 127 ? 127 : ue[m][nb]));
        }
    }
    const bool has_pre = L.prebias != nullptr, has_dm = L.dmask != nullptr, has_rs = L.relu_src != nullptr, has_add = L.add_src != nullptr;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (L.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(L.bias + col0);
    auto grow_of = [&](int i) { const int row = rowv + 8 * i; return (size_t)(ts + (row < R ? row : 0)); };      // (clamped: loads only)
    f32x4 gate[8];
    // The ReLU gate of a data-gradient layer as ONE BIT per element where the forward chain of the same tiles left the sign bits of
    // its output (L.gate_bits <- that layer's y_bits; layout [tile][column group][lane], bit 4 i + q = element q of the lane's row
    // piece i, as in dss2_gemm_chain_sp6.hip): one register instead of eight row pieces, 1/32 of the bytes, no exposed latency.
    uint32_t gate_word = 0u;
    const bool gbits = L.gate_bits != nullptr;
    if (gbits) gate_word = reinterpret_cast<const uint32_t*>(L.gate_bits)[((size_t)tile * ncg + cg) * 64 + lane];
    // every wave is done with this layer's planes: the hops below overwrite the wave's own stripe
    sp_barrier();
    CSTAMP(2 + li * 6 + 1);

    // (the gate rows are requested inside the Horner phase, after the first two accumulator hand-offs: before the barrier above they
    //  were live beside all 96 accumulator registers and, in the 16x16x32 form, went to scratch memory straight after their loads --
    //  a full memory latency in front of every layer's hops: 131 us instead of 111 for the backward chain)
    auto request_gates = [&]() {
      if (has_rs) {      // (uniform; the row addresses hang on a value defined inside the branch, or they are formed -- 80 vector
                         //  instructions per layer -- whether or not the layer has a gate: the forward chain has none)
        int rowg = rowv;
        asm volatile("" : "+v"(rowg));
        if (col_ok) {
          // one 64-bit product for the lane's first row, a uniform stride for the others; rows beyond the tile read its first row
          const char* g0 = reinterpret_cast<const char*>(L.relu_src + (size_t)ts * p.ld_relu + col0);
          const char* gp = g0 + (size_t)rowg * p.ld_relu * 4;
          const size_t gstep = (size_t)p.ld_relu * 32;
  #pragma unroll
          for (int i = 0; i < 8; ++i) {
            gate[i] = *reinterpret_cast<const f32x4*>(rowg + 8 * i < R ? gp : g0);
            gp += gstep;
          }
        }
      }
    };
    // ---- Horner on row pieces, wave-private: T in one slot, G_m in the other; U = G_m + P T replaces G_m
    f32x4 U[8];
    {
      auto put = [&](float* slot, int m) {
        if constexpr (MS == 0) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) slot[(rb * 32 + acc_row(r, half)) * 32 + c32] = acc[rb][m][r];
        } else {      // lane (n = lane & 15, row group lane >> 4) holds rows 16 mb + 4 (lane >> 4) + r of column 16 nb + n
          float* dst = slot + (4 * (lane >> 4)) * 32 + (lane & 15);
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int r = 0; r < 4; ++r) dst[(mb * 16 + r) * 32 + nb * 16] = F16 ? c16[mb][nb][m][r] * su[m][nb] : c16[mb][nb][m][r];
        }
      };
      put(slot0, NMAT - 1);
#pragma unroll
      for (int m = NMAT - 2; m >= 0; --m) {
        float* cur = slot0 + (((NMAT - 2 - m) & 1) ? SP_SLOT : 0);      // holds T
        float* oth = slot0 + (((NMAT - 2 - m) & 1) ? 0 : SP_SLOT);      // receives G_m, then U
        put(oth, m);
        if (m == NMAT - 2 && !gbits) request_gates();
        wave_lds_sync();
        int2 ea[8], eb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { ea[i] = ell[r8 + 8 * i]; U[i] = *reinterpret_cast<const f32x4*>(oth + (r8 + 8 * i) * 32 + cq); }
        // one hop entry: gather with `e`, request the entry after next into `nx`, accumulate.  Two entries per trip with the two
        // register sets swapping roles: the one-entry loop copied the next entries over the current ones (16 v_mov per entry)
        auto hop1 = [&](const int2 (&e)[8], int2 (&nx)[8], int kn) {
          f32x4 z[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) z[i] = *reinterpret_cast<const f32x4*>(cur + e[i].x * 32 + cq);
#pragma unroll
          for (int i = 0; i < 8; ++i) nx[i] = ell[kn * TM + r8 + 8 * i];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float w = __int_as_float(e[i].y);
#pragma unroll
            for (int q = 0; q < 4; ++q) U[i][q] = fmaf(w, z[i][q], U[i][q]);
          }
        };
        for (int k = 0; k < D; k += 2) {
          hop1(ea, eb, k + 1 < D ? k + 1 : k);
          if (k + 1 < D) hop1(eb, ea, k + 2 < D ? k + 2 : k + 1);      // (uniform)
        }
        if (m > 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(oth + (r8 + 8 * i) * 32 + cq) = U[i];
        }
        wave_lds_sync();      // the other lanes' gathers precede the next writes into `cur` (m > 0) / into the planes (m == 0)
      }
    }

    CSTAMP(2 + li * 6 + 2);      // Horner done
    CSTAMP(2 + li * 6 + 5);
    // ---- epilogue: bias / folded bias / masks / dropout / ReLU / gate / residual -> HBM and, split, the next layer's planes
    // (one uniform branch per feature around a loop over the lane's rows, not the other way round)
    const bool keep = li + 1 < ct.n;
    if (keep) {      // the next layer's first fragments ride under the epilogue
      if constexpr (MS == 0) load_b(reinterpret_cast<const bf16x8*>(ct.l[li + 1].Bp), b0, 0);
      else load_b16(reinterpret_cast<const bf16x8*>(ct.l[li + 1].Bp), b0, 0, 0);
    }
    // fused head: its weights (this lane's share of the wave's [m nout + o][32] block), bias and residual row, requested under the epilogue
    [[maybe_unused]] float hw_pre[3] = {0.f, 0.f, 0.f};      // (nout <= 2: all of them; wider heads fetch the rest in the head block)
    if constexpr (HM == 1) {
      if (!keep) {
        const int nout = hd.nout, kmax = p.hout - cg * 32;
        int lp = lane;
        asm volatile("" : "+v"(lp));      // (opaque: the three addresses are formed here, not hoisted out of the layer loop and spilled)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int idx = lp + 64 * i, mo = idx >> 5, k = idx & 31, m = mo / nout, o = mo - m * nout;
          const float* wp = m == 0 ? hd.W[0] : (m == 1 ? hd.W[1] : (m == 2 ? hd.W[2] : hd.W[3]));
          if (mo < NMAT * nout && k < kmax) hw_pre[i] = wp[(size_t)o * p.hout + cg * 32 + k];
        }
      }
    }
    // (the folded layer's row scales: all eight rows requested TOGETHER, before anything uses one -- inside the loop below each load
    //  sat right in front of its use, eight L2 round trips in sequence: tools/isa_waits.py)
    // (MS = 2 only: the bf16x6 forms have 40 fragment registers more and would spill these 24)
    typedef float f32x3_e __attribute__((ext_vector_type(3)));
    f32x3_e psr[F16 ? 8 : 1];
    if (F16 && has_pre && col_ok) {
      const char* pr0 = reinterpret_cast<const char*>(p.pre_rowscale + (size_t)ts * 4);
#pragma unroll
      for (int i = 0; i < (F16 ? 8 : 1); ++i) { const int row = rowv + 8 * i; psr[i] = *reinterpret_cast<const f32x3_e*>(pr0 + (size_t)(uint32_t)((row < R ? row : 0) * 16)); }
    }
    uint32_t relu_word = 0u;      // the ReLU's compare masks as the lane's sign-bit word (layers that write y_bits)
#pragma unroll
    for (int i = 0; i < 8; ++i) U[i] += bias4;
    if (col_ok) {
      if (has_pre) {
        f32x4 pb4[NMAT];
#pragma unroll
        for (int m = 0; m < NMAT; ++m) pb4[m] = *reinterpret_cast<const f32x4*>(L.prebias + (size_t)m * p.hout + col0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x3_e ps;
          if constexpr (F16) ps = psr[i];
          else ps = *reinterpret_cast<const f32x3_e*>(p.pre_rowscale + grow_of(i) * 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) U[i] += pb4[m] * ps[m];
        }
      }
      if (has_dm) {
#pragma unroll
        for (int i = 0; i < 8; ++i) U[i] *= *reinterpret_cast<const f32x4*>(L.dmask + grow_of(i) * p.ld_dmask + col0);
      }
      if (L.drop_id) {
#pragma unroll 1
        for (int i = 0; i < 8; ++i)
          U[i] *= dropout_mult4(drop_seed, drop_off, (uint32_t)L.drop_id, (uint32_t)grow_of(i), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
      }
      if (L.relu & 1) {      // (v_cmp_nge + v_cndmask per value: a NaN stays a NaN as under nn.ReLU -- v_max_f32(NaN, 0) is 0; round 5's
                             //  one-instruction `v_max_f32 0, x` turned a NaN weight into a finite loss, dss2_common.hpp relu_nan.
                             //  Four compares, then their four selects: left to itself the compiler pairs every v_cmp (VCC) with its
                             //  v_cndmask behind an `s_nop 1` -- a third issue slot per value, 32 per layer and wave)
        if (L.y_bits && !has_add) {      // (uniform) the sign-bit word rides in the compares: relu_nan4_bits
#pragma unroll
          for (int i = 7; i >= 0; --i) relu_nan4_bits(U[i], relu_word);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) relu_nan4(U[i]);
        }
      }
      if (gbits) {
        // (piece i's four gates from bits 4 (i & 7) ..: gate_bits4, dss2_common.hpp; the switch folds once the loop is unrolled)
        auto gate_piece = [&](int i) {
          const uint32_t w_ = gate_word;
          switch (i & 7) {
            case 0: gate_bits4<0>(U[i], w_); break;   case 1: gate_bits4<4>(U[i], w_); break;   case 2: gate_bits4<8>(U[i], w_); break;   case 3: gate_bits4<12>(U[i], w_); break;
            case 4: gate_bits4<16>(U[i], w_); break;  case 5: gate_bits4<20>(U[i], w_); break;  case 6: gate_bits4<24>(U[i], w_); break;  default: gate_bits4<28>(U[i], w_); break;
          }
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) gate_piece(i);
      } else if (has_rs) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) U[i][q] = relu_open(gate[i][q]) ? U[i][q] : 0.f;
      }
      if (has_add) {
#pragma unroll
        for (int i = 0; i < 8; ++i) U[i] += *reinterpret_cast<const f32x4*>(L.add_src + grow_of(i) * p.ld_add + col0);
      }
      {
        char* yp = reinterpret_cast<char*>(L.Y + (size_t)(ts + rowv) * p.ldy + col0);      // one 64-bit product per layer, then a uniform stride
        const size_t ystep = (size_t)p.ldy * 32;                                           // 8 rows
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (rowv + 8 * i < R) *reinterpret_cast<f32x4*>(yp) = U[i];
          yp += ystep;
        }
      }
    }
    if (L.y_bits) {      // (uniform) the sign bits of what went to Y, in the layout the data-gradient form reads (pad rows / columns: zero bits)
      uint32_t word = 0u;
      if ((L.relu & 1) && !has_add) {
        // behind a ReLU (and no residual) a stored value is +0, positive or NaN: "open" = its bit pattern is not zero -- v_min_u32 +
        // v_lshl_or_b32 per value instead of compare, select and or; NaN counts as open, as in relu_open (torch's threshold_backward)
        // (round 6: the bits are the ReLU's own compare masks, shifted into relu_word there -- before: v_min_u32 + v_lshl_or_b32 per stored value)
        word = relu_word;
        uint32_t valid = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) valid |= (col_ok && rowv + 8 * i < R) ? (0xFu << (4 * i)) : 0u;
        word &= valid;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool in_y = col_ok && rowv + 8 * i < R;
#pragma unroll
          for (int q = 0; q < 4; ++q) word |= ((in_y && relu_open(U[i][q])) ? 1u : 0u) << (4 * i + q);
        }
      }
      reinterpret_cast<uint32_t*>(L.y_bits)[((size_t)tile * ncg + cg) * 64 + lane] = word;
    }
    if (keep) {
      // (rows beyond the tile's R rows are NOT zeroed -- 32 selects per layer: their values are finite (bias-driven like any
      //  activation), no real row ever gathers from them (ELL neighbours are real rows; a padding row's own entries carry weight 0),
      //  nothing of them is stored, the fused head below masks its own copy.  Columns beyond hout are zero by construction:
      //  zero weight columns, no bias.)
      if constexpr (F16) {
        // the next layer's scale: the tile's maximum over all stripes (one more barrier per layer; the planes then go over the slots)
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = absmax4(mx, U[i]);
        mx = wave_max(mx);
        if (lane == 0) mxw[wave] = mx;
        sp_barrier();
        ea = tile_exponent();
#pragma unroll
        for (int i = 0; i < 8; ++i) sp_store_split_h(own_planes + (rowv + 8 * i) * SP_RS + cq, U[i], ea);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) sp_store_split(own_planes + (rowv + 8 * i) * SP_RS + cq, U[i]);
      }
    }
    if constexpr (HM == 1) {
      if (!keep) {
        // ---- fused head: Y = bias + sum_m P^m (h W_m^T) (+ residual), h = this tile's last activations (U, zero outside the tile)
        const int nout = hd.nout, nmo = NMAT * nout;
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (opaque: the per-lane addresses of this block are formed here, not hoisted -- and spilled -- across the layer loop)
        f32x4 hb_pre = {0.f, 0.f, 0.f, 0.f}, ha_pre = {0.f, 0.f, 0.f, 0.f};      // bias and residual row: requested now, used by wave 0 at the very end
        if (wave == 0) {
#pragma unroll
          for (int o = 0; o < SP_HEAD_MAX; ++o) {
            if (o < nout) {
              if (hd.bias) hb_pre[o] = hd.bias[o];
              if (hd.add_src && ln < R) ha_pre[o] = hd.add_src[(size_t)(ts + ln) * hd.ld_add + o];
            }
          }
        }
        float* rowbuf = slot0;                           // [64][36]: the wave's 32 columns, row-major
        float* part = slot0 + TM * 36;                   // [64][17]: this wave's partial products (m, o)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = rowv + 8 * i;
          *reinterpret_cast<f32x4*>(rowbuf + row * 36 + cq) = (row < R && col_ok) ? U[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wave_lds_sync();
        // W_m[o][this wave's 32 columns] -> LDS [m nout + o][32] (requested under the epilogue; uniform scalar reads of global memory
        // in the loop below compile to one waited-for load per weight: 25 us per launch), then broadcast ds_read_b128
        float* wls = part + TM * 17;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (ln + 64 * i < nmo * 32) wls[ln + 64 * i] = hw_pre[i];
        for (int idx = ln + 192; idx < nmo * 32; idx += 64) {
          const int mo = idx >> 5, k = idx & 31, m = mo / nout, o = mo - m * nout;
          const float* wp = m == 0 ? hd.W[0] : (m == 1 ? hd.W[1] : (m == 2 ? hd.W[2] : hd.W[3]));
          wls[idx] = k < p.hout - cg * 32 ? wp[(size_t)o * p.hout + cg * 32 + k] : 0.f;
        }
        f32x4 hr[8];                                     // lane = row: its 32 values
#pragma unroll
        for (int j = 0; j < 8; ++j) hr[j] = *reinterpret_cast<const f32x4*>(rowbuf + ln * 36 + 4 * j);
        wave_lds_sync();
#pragma unroll 1
        for (int mo = 0; mo < nmo; ++mo) {
          float sacc = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wls + mo * 32 + 4 * j);      // same address in every lane
#pragma unroll
            for (int q = 0; q < 4; ++q) sacc = fmaf(hr[j][q], wv[q], sacc);
          }
          part[ln * 17 + mo] = sacc;
        }
        sp_barrier();
        if (wave == 0) {
          float g[NMAT][SP_HEAD_MAX];
#pragma unroll
          for (int m = 0; m < NMAT; ++m)
#pragma unroll
            for (int o = 0; o < SP_HEAD_MAX; ++o) {
              float t = 0.f;
              if (o < nout)
                for (int w = 0; w < ncg; ++w) t += smem[w * SP_REGION + TM * 36 + ln * 17 + m * nout + o];
              g[m][o] = t;
            }
          float* hs = rowbuf;                            // [64][4] hop scratch (the row buffer is done)
          f32x4 t = {g[NMAT - 1][0], g[NMAT - 1][1], g[NMAT - 1][2], g[NMAT - 1][3]};
#pragma unroll
          for (int m = NMAT - 2; m >= 0; --m) {
            wave_lds_sync();
            *reinterpret_cast<f32x4*>(hs + ln * 4) = t;
            wave_lds_sync();
            f32x4 a = {g[m][0], g[m][1], g[m][2], g[m][3]};
            for (int k = 0; k < D; ++k) {
              const int2 en = ell[k * TM + ln];
              const f32x4 zz = *reinterpret_cast<const f32x4*>(hs + en.x * 4);
              const float w = __int_as_float(en.y);
#pragma unroll
              for (int q = 0; q < 4; ++q) a[q] = fmaf(w, zz[q], a[q]);
            }
            t = a;
          }
          if (ln < R) {
#pragma unroll
            for (int o = 0; o < SP_HEAD_MAX; ++o) {
              if (o < nout) {
                hd.Y[(size_t)(ts + ln) * hd.ldy + o] = t[o] + hb_pre[o] + ha_pre[o];
              }
            }
          }
        }
      }
    }
    CSTAMP(2 + li * 6 + 3);      // epilogue done
    if (keep) sp_barrier();   // the next layer's planes are complete
    CSTAMP(2 + li * 6 + 4);
    if (!keep) CSTAMP_RT(63);
  }
  if (probe && tid == 0) {
    unsigned long long* q = ct.clock_probe + (tile >> 8) * 4;
    q[2] = __builtin_amdgcn_s_memtime(); q[3] = __builtin_amdgcn_s_memrealtime();
  }
}

inline size_t chain_sp_lds_bytes(int ncg, int ell_width) { return (size_t)ncg * SP_REGION * 4 + (size_t)SP_TM * ell_width * 8 + 64; }      // (+ the f16x3 form's maxima)

bool chain_sp_supported(const dss2_gemm_prop_args& a) {
  static const int on = [] { const char* e = getenv("DSS2_CHAIN_SP"); return e ? atoi(e) : 1; }();
  static const int f16on = [] { const char* e = getenv("DSS2_CHAIN_SP_F16"); return e ? atoi(e) : 1; }();
  if (a.b_format == 2 && (!f16on || (a.kpad & 31) != 0)) return false;      // the f16x3 form: 16x16x32 MFMAs only
  return on && (a.b_format == 1 || a.b_format == 2) && a.nrb == 2 && a.nmat >= 2 && a.nmat <= 3 && (a.kpad & 15) == 0 && a.kpad <= 32 * a.ncg &&
         a.ncg >= 3 && a.ncg <= 8 && chain_sp_lds_bytes(a.ncg, a.ell_width) <= (size_t)kMaxLdsBytes;
}

template <int NMAT, int NW, int HM, int MS>
static int launch_sp(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head& hd, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_chain_sp_kernel<NMAT, NW, HM, MS>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_prop_chain(split planes)")) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * a.ncg), chain_sp_lds_bytes(a.ncg, a.ell_width), stream, a, ct, hd);
  return check_launch("gemm_prop_chain(split planes)");
}

template <int HM>
static int launch_sp_hm(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head& hd, hipStream_t s) {
  static const int ms16 = [] { const char* e = getenv("DSS2_CHAIN_MFMA16"); return e ? atoi(e) : 1; }();      // 0: the 32x32x16 form
  // (a layer gated by fp32 activations instead of bit words keeps eight row pieces of gate values beside the accumulators: the 16x16
  //  form has 24 registers less room, spills them on arrival and waits a memory latency per layer -- 124 us against 107)
  bool fp32_gates = false;
  for (int i = 0; i < ct.n; ++i) fp32_gates = fp32_gates || (ct.l[i].relu_src && !ct.l[i].gate_bits);
  if (a.b_format == 2) {      // weights as two fp16 planes: the f16x3 form
    if (fp32_gates) { set_error("gemm_prop_chain(f16x3): layers gated by fp32 activations need bf16x3 weights (b_format 1)"); return 2; }
    if (a.nmat == 2) return a.ncg <= 4 ? launch_sp<2, 4, HM, 2>(a, ct, hd, s) : launch_sp<2, 8, HM, 2>(a, ct, hd, s);
    return a.ncg <= 4 ? launch_sp<3, 4, HM, 2>(a, ct, hd, s) : launch_sp<3, 8, HM, 2>(a, ct, hd, s);
  }
  if (ms16 && (a.kpad & 31) == 0 && !fp32_gates) {
    if (a.nmat == 2) return a.ncg <= 4 ? launch_sp<2, 4, HM, 1>(a, ct, hd, s) : launch_sp<2, 8, HM, 1>(a, ct, hd, s);
    return a.ncg <= 4 ? launch_sp<3, 4, HM, 1>(a, ct, hd, s) : launch_sp<3, 8, HM, 1>(a, ct, hd, s);
  }
  if (a.nmat == 2) return a.ncg <= 4 ? launch_sp<2, 4, HM, 0>(a, ct, hd, s) : launch_sp<2, 8, HM, 0>(a, ct, hd, s);
  return a.ncg <= 4 ? launch_sp<3, 4, HM, 0>(a, ct, hd, s) : launch_sp<3, 8, HM, 0>(a, ct, hd, s);      // (K = 3 would spill: chain_sp_supported says no)
}

static std::atomic<unsigned long long*> g_chain_clock_probe{nullptr};

int launch_chain_sp(const dss2_gemm_prop_args& a, const ChainTable& ct_in, const dss2_chain_head* head, hipStream_t s) {
  dss2_chain_head hd = {};
  if (head) hd = *head;
  ChainTable ct = ct_in;
  ct.clock_probe = g_chain_clock_probe.load(std::memory_order_relaxed);
  if (hd.mode == 1) return launch_sp_hm<1>(a, ct, hd, s);
  if (hd.mode == 2) return launch_sp_hm<2>(a, ct, hd, s);
  return launch_sp_hm<0>(a, ct, hd, s);
}

}  // namespace dss2

// Diagnostic (bench.py's held_clock_ghz): probe != NULL -> every later split-plane chain launch of 64-row tiles writes four records of
// {s_memtime at start, s_memrealtime at start, s_memtime at end, s_memrealtime at end} (16 uint64, device memory) -- workgroups 0, 256,
// 512 and 768; NULL switches it off again.  The records of the LAST launch stay in the buffer.
extern "C" void dss2_debug_chain_clock_probe(unsigned long long* probe) { dss2::g_chain_clock_probe.store(probe, std::memory_order_relaxed); }

#ifdef DSS2_CHAIN_STAMPS
extern "C" int dss2_debug_read_cstamps_sp(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_cstamps), sizeof(unsigned long long) * n);
}
#endif
