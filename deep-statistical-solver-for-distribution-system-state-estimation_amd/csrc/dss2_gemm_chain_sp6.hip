// Split-plane layer chain for tall tiles: 192 rows (graphs of 97..192 buses, one per tile: the 179-bus feeder) and 96 rows
// (the 70-bus ober_sub graphs), NRB = 6 / 3 row blocks per wave.  (The round-3 one-workgroup-per-CU 96-row kernel it
// replaced is kept as a record: profiles/experiments/r03_gemm_chain_sp3_one_workgroup_per_cu.hip.txt.)
//
// One wave per 32-column group owns all 192 rows of its columns: 288 accumulator registers (AGPRs; one wave per SIMD, 512-register
// budget), the Horner hops wave-private, every weight fragment fetched once per column group, two barriers per layer.  What
// differs from the 96-row form is the LDS budget -- the tile alone is 144 KB as split planes:
//   * plane rows are UNPADDED (32 bf16 = 64 bytes) and XOR-swizzled by 16-byte chunk with key (row / 4) mod 4: the ds_read_b128
//     lane groups of an A fragment still fall on 16 distinct slots;
//   * ONE Horner slot per column group ([192][32] fp32 = 24 KB inside the group's 36 KB stripe): a hop gathers
//     z = sum_k w T[nbr] from the slot into registers, then G_m takes the slot's place (accumulator layout -> rows) and
//     U = G_m + z is read back from the lane's own rows -- instead of T and G_m side by side.  (The sums are therefore formed as
//     G_m + (sum), not as an fma chain starting at G_m: results differ from the per-layer kernel by rounding.)
// This configuration had no layer chain at all (gemm_prop_kernel<6,3,true,2> per layer: X tile re-read, in-register operand split,
// accumulator-layout hops with workgroup barriers: 100 K cycles per layer and tile).  LDS 4 x 36 KB + ELL, one workgroup per CU.
// Compiled without packed fp32 ops like the other bf16x6 translation units (build.sh, dss2_gemm_chain16.hip).
#include <stdlib.h>

#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

#define S6STAMP(slot) CSTAMP(slot)
#ifndef DSS2_S6_PASSES
#define DSS2_S6_PASSES 2      // gather passes per hop (NRP / passes row pieces in flight per lane).  Measured (chainbench, 192-row fwd / bwd): 1 pass 472 / 531 us, 2 passes 406 / 473, 4 passes 439 / 499
#endif
constexpr int S6_RS = 32;                      // bf16 per plane row: unpadded, chunks swizzled
constexpr int s6_plane(int nrb) { return 32 * nrb * S6_RS; }           // bf16 per plane
constexpr int s6_region(int nrb) { return 3 * s6_plane(nrb) / 2; }     // floats per column group: three planes (192 rows: 36 KB >= the Horner slot, [192][32] fp32 = 24 KB)

// bf16 offset of (row, k) inside a plane: 16-byte chunk (k / 8) XOR (row / 4) mod 4
__device__ __forceinline__ int s6_off(int row, int k) { return row * S6_RS + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); }

__device__ __forceinline__ void s6_barrier() {      // LDS-only hand-off: the Y stores of the epilogue stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef uint32_t u32x2_s6 __attribute__((ext_vector_type(2)));

template <int S6_PLANE>
__device__ __forceinline__ void s6_store_split(__bf16* dst, const f32x4 v) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(v[0], v[1], h0, m0, l0);
  split3_pair(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<u32x2_s6*>(dst) = u32x2_s6{h0, h1};
  *reinterpret_cast<u32x2_s6*>(dst + S6_PLANE) = u32x2_s6{m0, m1};
  *reinterpret_cast<u32x2_s6*>(dst + 2 * S6_PLANE) = u32x2_s6{l0, l1};
}

// f16x3 form (F16): the two fp16 pieces of 4 consecutive values scaled by 2^e -> planes 0, 1 (dss2_gemm_chain_sp.hip, MS = 2)
template <int S6_PLANE>
__device__ __forceinline__ void s6_store_split_h(__bf16* dst, const f32x4 v, int e) {
  uint32_t h0, l0, h1, l1;
  const float s = pow2f(e);      // (v_mul_f32: half the issue cycles of v_ldexp_f32, the same bits)
  split2_pair(v[0] * s, v[1] * s, h0, l0);
  split2_pair(v[2] * s, v[3] * s, h1, l1);
  *reinterpret_cast<u32x2_s6*>(dst) = u32x2_s6{h0, h1};
  *reinterpret_cast<u32x2_s6*>(dst + S6_PLANE) = u32x2_s6{l0, l1};
}

// DIR: which epilogue features the launch's layers use, so that each direction gets its own register allocation (one generic
// kernel: 77 / 130 spilled VGPRs at K = 2 and a backward chain 8-17 % slower than the forward one; specialised: 10-13 / 12).
//   1 = forward set (bias, folded bias, in-kernel dropout, ReLU, y_bits);  2 = data-gradient set (gate_bits, in-kernel dropout);
//   0 = everything (mask tensor, residual, fp32 gates: tests and callers outside networks.py)
// HM = 2 (backward launches only, DIR = 2): the chain's input tile is the narrow head's data gradient, computed in the staging
// from the nout-wide upstream gradient (dss2_gemm_chain_sp.hip, dss2_gemm_prop_chain_head) instead of by a launch of its own
// that writes [N, hid] and is re-read here (C3: 22.6 us, 179-bus: 60 us).
// F16 (round 5; DIR 1 and 2): the tile GEMM as f16x3 -- two fp16 planes per stripe scaled by 2^ea per tile and layer, the weights as two
// fp16 planes with per-matrix exponents (b_format 2), three v_mfma_f32_32x32x16_f16 per product; the accumulators are handed to the hops
// with the scales taken out.  One more barrier per layer (the tile's maximum).  See dss2_gemm_chain_sp.hip, MS = 2.
// RPA (round 6): ACTIVE row pieces per lane, <= 4 NRB.  Where every tile of the launch holds at most 8 RPA rows (args.max_tile_rows: 70-bus
// graphs in 96-row tiles -> 9 of 12 pieces) the accumulator hand-off, the hops, the epilogue and the split of the next layer's planes leave
// out the pieces that are padding in every tile -- a quarter of a layer's vector and LDS work at C3.  The planes of those rows are never
// rewritten (they hold whatever the Horner slot left there: only the accumulators of the same padding rows see it, and those are never read).
template <int NRB, int NMAT, int DIR, int HM = 0, bool F16 = false, int RPA = 4 * NRB>
__global__ void __launch_bounds__(256, NRB <= 3 ? 2 : 1) gemm_chain_sp6_kernel(const dss2_gemm_prop_args p, const ChainTable ct, const dss2_chain_head hd) {
  constexpr int TM = 32 * NRB, S6_PLANE = s6_plane(NRB), S6_REGION = s6_region(NRB);
  // 96 rows: the accumulator hand-off's scales as v_mul_f32 and the sign-bit words from the ReLU's compare masks (end of round 6, dss2_common.hpp).  192 rows keep
  // v_ldexp_f32 and the bits formed from the stored values: 288 accumulator registers leave no room -- with the products' float scales live across the hand-off the
  // data-gradient kernel spills 205 registers instead of 29 (179-bus step 1.34 -> 1.46 ms), with the compare-mask words the forward one 91 instead of 59
  constexpr bool S6_MUL = NRB <= 3, S6_RELU_BITS = NRB <= 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int ncg = nthreads >> 6;
  const int cg = wave;
  const int tile = blockIdx.x;
  const uint64_t drop_seed = p.drop_state ? p.drop_state[0] : 0, drop_off = p.drop_state ? p.drop_state[1] : 0;
  __bf16* xpl = reinterpret_cast<__bf16*>(smem);               // stripe s: xpl + s * (2 * S6_REGION)
  int2* ell = reinterpret_cast<int2*>(smem + ncg * S6_REGION);
  const int D = p.ell_width;
  constexpr int NP = F16 ? 2 : 3;
  float* mxw = reinterpret_cast<float*>(ell + D * TM);      // (F16) [ncg]: every wave's maximum over its part of the tile
  int ea = 0;                                                // (F16) the planes in LDS hold 2^ea x
  auto tile_exponent = [&]() {
    float m = 0.f;
    for (int w = 0; w < ncg; ++w) m = fmaxf(m, mxw[w]);
    return 14 - __builtin_amdgcn_readfirstlane(exp_of(m));
  };
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int kq = p.kpad >> 2;
  const int c32 = lane & 31, half = lane >> 5;
  const int nks = p.kpad >> 4;
  float* slot0 = smem + cg * S6_REGION;
  __bf16* own_planes = xpl + cg * (2 * S6_REGION);
  const int cq = (lane & 7) * 4, r8 = lane >> 3;
  const int col0 = cg * 32 + cq;
  const bool col_ok = col0 < p.hout;
  constexpr int NRP = 4 * NRB, HP = (RPA + DSS2_S6_PASSES - 1) / DSS2_S6_PASSES;      // row pieces per lane: rows r8 + 8 i (RPA of them active in the layer loop); HP of them per gather pass
  constexpr int NPASS = (RPA + HP - 1) / HP;
  static_assert(RPA >= 1 && RPA <= NRP, "active row pieces");

  // ---- stage the tile's ELL slice and the first layer's input tile as split planes (zero padded to 96 x kpad)
  {
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
    for (int idx = tid; idx < D * TM; idx += nthreads) ell[idx] = src[idx];
  }
  if constexpr (HM != 2 && !F16) {
    for (int idx = tid; idx < TM * kq; idx += nthreads) {
      const int r = idx / kq, c = (idx - r * kq) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
      s6_store_split<S6_PLANE>(xpl + (c >> 5) * (2 * S6_REGION) + s6_off(r, c & 31), v);
    }
  } else if constexpr (HM != 2) {
    // the input tile waits in registers (TM kq / threads <= TM / 8 row pieces per thread) while its maximum is formed
    f32x4 xin[NRP];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < NRP; ++j) {
      const int idx = tid + j * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      const bool on = idx < TM * kq && r < R && c < p.kreal;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + (on ? r : 0)) * p.ldx + (on ? c : 0));      // (unconditional, masked)
      xin[j] = on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      mx = absmax4(mx, xin[j]);
    }
    mx = wave_max(mx);
    if (lane == 0) mxw[wave] = mx;
    s6_barrier();
    ea = tile_exponent();
#pragma unroll
    for (int j = 0; j < NRP; ++j) {
      const int idx = tid + j * nthreads;
      const int r = idx / kq, c = (idx - r * kq) << 2;
      if (idx < TM * kq) s6_store_split_h<S6_PLANE>(xpl + (c >> 5) * (2 * S6_REGION) + s6_off(r, c & 31), xin[j], ea);
    }
  } else {
    // X[row][c] = gate(row, c) * sum_{m, o} ((P^T)^m G)[row][o] W_m[o][c]: every wave builds its own 32-column stripe.  The hops of
    // G are nout wide and run once per WAVE (a lane owns rows lane, lane + 64, ..; wave-private scratch at the start of the wave's
    // own region, which the planes overwrite at the very end).
    constexpr int RPL = (TM + 63) / 64;      // rows per lane in the hop phase
    static_assert(TM * NMAT * 4 + TM * 4 + 64 * NMAT * 2 * 4 <= S6_REGION, "the head's scratches must fit the wave's own region");
    const int nout = hd.nout;
    f32x4 ga[NRP];
    if (hd.gate && col_ok) {
#pragma unroll
      for (int i = 0; i < NRP; ++i) { const int row = r8 + 8 * i; ga[i] = *reinterpret_cast<const f32x4*>(hd.gate + (size_t)(ts + (row < R ? row : 0)) * hd.ld_gate + col0); }
    }
    f32x4 wl[NMAT][4];      // W_m[o][col0 .. col0 + 3]
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int o = 0; o < 4; ++o)
        wl[m][o] = (o < nout && col_ok) ? *reinterpret_cast<const f32x4*>(hd.W[m] + (size_t)o * p.hout + col0) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* zt = slot0;                       // [TM][NMAT * 4]: ((P^T)^m G)[row][o]
    float* hs = slot0 + TM * NMAT * 4;       // [TM][4] hop scratch
    f32x4 z[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const int row = lane + 64 * j;
      z[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < R) {
#pragma unroll
        for (int o = 0; o < 4; ++o) if (o < nout) z[j][o] = hd.G[(size_t)(ts + row) * hd.ldg + o];
      }
      if (row < TM) *reinterpret_cast<f32x4*>(zt + row * (NMAT * 4)) = z[j];
    }
    float zg0 = 0.f, zg1 = 0.f;              // (this lane's rows of the upstream gradient, summed: the head's bias sums below)
#pragma unroll
    for (int j = 0; j < RPL; ++j) { zg0 += z[j][0]; zg1 += z[j][1]; }
    s6_barrier();                            // the ELL slice is staged
#pragma unroll
    for (int m = 1; m < NMAT; ++m) {
#pragma unroll
      for (int j = 0; j < RPL; ++j) if (lane + 64 * j < TM) *reinterpret_cast<f32x4*>(hs + (lane + 64 * j) * 4) = z[j];
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
        const int row = lane + 64 * j;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (row < TM) {
          for (int k = 0; k < D; ++k) {
            const int2 en = ell[k * TM + row];
            t += *reinterpret_cast<const f32x4*>(hs + en.x * 4) * __int_as_float(en.y);
          }
        }
        z[j] = t;
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < RPL; ++j) if (lane + 64 * j < TM) *reinterpret_cast<f32x4*>(zt + (lane + 64 * j) * (NMAT * 4) + m * 4) = z[j];
    }
    wave_lds_sync();
    // The head's WEIGHT gradient rides here too (hd.wg_slab, nout <= 2; as in the 64-row chain, dss2_gemm_chain_sp.hip): dW_m[o][c] =
    // sum_rows ((P^T)^m G)[row][o] h[row][c] needs what this block holds -- the hop results zt and the head's input rows h (= the gate
    // rows ga) -- so the launch that re-read the [N, hid] activation for it (wgrad_narrow_stream_kernel: 18 us at C3) is not needed.
    // Per lane: its NRP rows x 4 columns x NMAT x nout partial sums, then the 8 lanes that share the columns meet in wave-private LDS
    // in row order: one slab per tile, [NMAT nout][hid] + nout bias sums, reduced with the step's other slabs (fixed order).
    const bool wg = hd.wg_slab != nullptr && hd.gate != nullptr;
    f32x4 hw[NMAT][2];
#pragma unroll
    for (int m = 0; m < NMAT; ++m) { hw[m][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hw[m][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 xv[NRP];
#pragma unroll
    for (int i = 0; i < NRP; ++i) {
      const int row = r8 + 8 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NMAT; ++m) {
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zt + row * (NMAT * 4) + m * 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) v += wl[m][o] * zz[o];
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
      float* scr = hs + TM * 4;              // [64 lanes][NMAT 2][4]: behind zt / hs inside the wave's own region, before the planes go over it
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
        for (int g8 = 0; g8 < 8; ++g8) sum += scr[((g8 * 8 + cqi) * (NMAT * 2) + mo) * 4 + q];      // row groups r8 = 0 .. 7 in order
        const int col = cg * 32 + cqi * 4 + q;
        if (o < nout && col < p.hout) slab[(size_t)(m * nout + o) * p.hout + col] = sum;
      }
      if (wave == 0) {                       // bias sums: the upstream gradient's column sums over the tile's rows (a lane's rows first,
        float sb0 = zg0, sb1 = zg1;          // then a butterfly over the 64 lanes: pairwise, fixed order)
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
      for (int i = 0; i < NRP; ++i) mx = absmax4(mx, xv[i]);
      mx = wave_max(mx);
      if (lane == 0) mxw[wave] = mx;
      s6_barrier();
      ea = tile_exponent();
    }
    wave_lds_sync();                         // every lane is done with zt (and the scratch behind it): the planes go over it
#pragma unroll
    for (int i = 0; i < NRP; ++i) {
      if constexpr (F16) s6_store_split_h<S6_PLANE>(own_planes + s6_off(r8 + 8 * i, cq), xv[i], ea);
      else s6_store_split<S6_PLANE>(own_planes + s6_off(r8 + 8 * i, cq), xv[i]);
    }
  }
  bf16x8 b0[NP][NMAT];
  auto load_b = [&](const bf16x8* __restrict__ bp16, bf16x8 (&bb)[NP][NMAT], int ks) {
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bb[pl][m] = bp16[(((size_t)(m * ncg + cg) * nks + ks) * NP + pl) * 64 + lane];
  };
  load_b(reinterpret_cast<const bf16x8*>(ct.l[0].Bp), b0, 0);
  S6STAMP(0);
  s6_barrier();
  S6STAMP(1);

  for (int li = 0; li < ct.n; ++li) {
    const dss2_chain_layer& L = ct.l[li];    // uniform: scalar loads from the kernel-argument segment
    const bf16x8* __restrict__ bp16 = reinterpret_cast<const bf16x8*>(L.Bp);
    f32x16 acc[NRB][NMAT];
    [[maybe_unused]] int ue[NMAT];      // (F16) what takes the scales out of matrix m's accumulators for the lane's output column: -(ea + ew[m][column])
    [[maybe_unused]] float su[NMAT];      // (F16) 2^ue

    // ---- tile GEMM, 16 k per step: B fragments (L2) ping-pong one step ahead, A fragments (LDS planes) one row block ahead;
    // one memory request per MFMA gap (dss2_gemm_chain_sp.hip)
    {
      bf16x8 b1[NP][NMAT], a[2][NP];
      auto load_a = [&](bf16x8 (&af)[NP], int rb, int ks) {
        const __bf16* src = xpl + (ks >> 1) * (2 * S6_REGION) + s6_off(rb * 32 + c32, (ks & 1) * 16 + half * 8);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(src + pl * S6_PLANE);
      };
      auto mma = [&](const bf16x8 (&af)[NP], const bf16x8 (&b)[NP][NMAT], f32x16 (&c)[NMAT], const bool first) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (F16) {      // lo hi + hi lo + hi hi, smallest terms first
          auto h8 = [](const bf16x8 v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(af[1]), h8(b[0][m]), first ? zero : c[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(af[0]), h8(b[1][m]), c[m], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(af[0]), h8(b[0][m]), c[m], 0, 0, 0);
          return;
        }
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NP - 1], b[0][m], first ? zero : c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[NP - 1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0][m], c[m], 0, 0, 0);
      };
      // a[rb & 1] holds row block rb's fragment; right after its MFMAs the register set is re-requested for row block rb + 2
      // (of this step, or of the next one; branch-free: the last step re-requests its own operands).  Two sets, not six:
      // 6 x NMAT accumulator blocks already take 288 of the wave's 512 registers at NMAT = 3.
      auto step = [&](const bf16x8 (&bc)[NP][NMAT], bf16x8 (&bn)[NP][NMAT], int ks, const bool first, const int par) {
        const int kn = ks + 1 < nks ? ks + 1 : ks;
        load_b(bp16, bn, kn);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) { mma(a[(par + rb) & 1], bc, acc[rb], first); load_a(a[(par + rb) & 1], rb + 2 < NRB ? rb + 2 : rb + 2 - NRB, rb + 2 < NRB ? ks : kn); }
        // gaps 1-9: the next step's weight fragments; after each row block's MFMAs: the re-request of its fragment
        constexpr int RBM = (F16 ? 3 : 6) * NMAT;      // MFMAs per row block
#pragma unroll
        for (int i = 0; i < NP * NMAT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        if constexpr (RBM - NP * NMAT > 0) __builtin_amdgcn_sched_group_barrier(0x008, RBM - NP * NMAT, 0);
#pragma unroll
        for (int rb = 1; rb < NRB; ++rb) {
#pragma unroll
          for (int i = 0; i < NP; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
          __builtin_amdgcn_sched_group_barrier(0x008, RBM - NP, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
      };
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) load_a(a[rb], rb, 0);
      // (par: which A-fragment set row block 0 of the step finds its operands in -- with an odd NRB the sets swap roles every step)
      step(b0, b1, 0, true, 0);
      int ks = 1;
      for (; ks + 2 <= nks; ks += 2) {
        step(b1, b0, ks, false, NRB & 1);
        step(b0, b1, ks + 1, false, 0);
      }
      if (ks < nks) step(b1, b0, ks, false, NRB & 1);
    }
    S6STAMP(2 + li * 6 + 0);      // GEMM phase done
    // ---- what the epilogue reads from HBM per row, requested before the hops (rowv opaque: see dss2_gemm_chain_sp.hip)
    int rowv = r8;
    asm volatile("" : "+v"(rowv));
    if constexpr (F16) {      // (requested here, behind the GEMM phase: its fragment registers are free; used in the hand-off behind the barrier)
      const int* whdr = reinterpret_cast<const int*>(reinterpret_cast<const char*>(L.Bp) + (size_t)NMAT * ncg * nks * 2048) + cg * 32 + c32;
#pragma unroll
      for (int m = 0; m < NMAT; ++m) {
        ue[m] = -(ea + whdr[m * ncg * 32]);
        su[m] = ldexpf(1.f, ue[m] < -126 ? -126 : (ue[m] > 127 ? 127 : ue[m]));      // (the hand-off multiplies: dss2_gemm_chain_sp.hip)
      }
    }
    const bool has_pre = DIR != 2 && L.prebias != nullptr, has_dm = DIR == 0 && L.dmask != nullptr, has_rs = DIR != 1 && L.relu_src != nullptr, has_add = DIR == 0 && L.add_src != nullptr;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (DIR != 2 && L.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(L.bias + col0);
    auto grow_of = [&](int i) { const int row = rowv + 8 * i; return (size_t)(ts + (row < R ? row : 0)); };      // (clamped: loads only)
    // The ReLU gate of the backward form, ONE BIT per element, (NRP + 7) / 8 registers per lane: bit (i & 7) * 4 + q of word i / 8
    // belongs to element q of the lane's row piece i.
    //  * L.gate_bits (written by the forward chain of the same tiles through y_bits, see below; layout [tile][column group][word]
    //    [lane]): the lane's own words, requested HERE, before the hops -- two or three registers, 1/32 of the bytes, no exposed
    //    latency.  (First form: one ballot word per row piece and element -- 4 x NRP uniform words per wave.  Writing them cost the
    //    192-row forward chain 50 us (96 ballots, selects and 24 four-lane stores per layer); reading them as uniform loads in the
    //    epilogue -- twelve s_load_dwordx16 in sequence, each waited for -- 70 us of the backward chain, through v_readlane 38 us.)
    //  * otherwise (generic kernel only) from relu_src: each gather pass of the LAST hop requests the fp32 values of its own HP row
    //    pieces at its top and folds them into bits at its end.  (Requested as one block of NRP vectors -- before the hops, as the
    //    round-3 96-row kernel (profiles/experiments/) did with 48 registers -- the register allocator has no room beside the
    //    accumulators and U and spills every pair the moment it arrives: "load, load, wait, store, store" NRP / 2 times = as many
    //    SERIALISED global round trips per layer: backward chain 606 us against 426 us forward on the 179-bus configuration.)
    constexpr int NGW = (NRP + 7) / 8;
    uint32_t gate_bits[NGW] = {};
    const bool gbits = DIR != 1 && L.gate_bits != nullptr;
    if (gbits) {
      const uint32_t* gb = reinterpret_cast<const uint32_t*>(L.gate_bits) + ((size_t)tile * ncg + cg) * (NGW * 64) + lane;
#pragma unroll
      for (int w = 0; w < NGW; ++w) gate_bits[w] = gb[w * 64];
    }
    const bool fp32_gate = DIR == 0 && has_rs && !L.gate_bits;
    s6_barrier();      // every wave is done with this layer's planes: the slots below go over the wave's own stripe
    S6STAMP(2 + li * 6 + 1);

    // ---- Horner on row pieces, wave-private: T in one slot, G_m in the other; U = G_m + P T replaces G_m
    f32x4 U[RPA];
    {
      auto put = [&](int m) {      // (accumulator registers of rows beyond the active pieces stay where they are)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rb * 32 + 8 * (r >> 2) < 8 * RPA) slot0[(rb * 32 + acc_row(r, half)) * 32 + c32] = F16 ? (S6_MUL ? acc[rb][m][r] * su[m] : ldexpf(acc[rb][m][r], ue[m])) : acc[rb][m][r];
      };
      put(NMAT - 1);
      wave_lds_sync();
#pragma unroll
      for (int m = NMAT - 2; m >= 0; --m) {
        const bool want_gate = m == 0 && fp32_gate && col_ok;
        // z = P T gathered from the slot (NRP / HP passes of HP row pieces), then G_m takes the slot and U = G_m + z
#pragma unroll
        for (int h12 = 0; h12 < NPASS; ++h12) {
          int2 en[HP];
          f32x4 gt[HP];
          auto on = [&](int i) { return HP * h12 + i < RPA; };      // (compile-time after unrolling: the last pass may be shorter)
          if (want_gate) {
#pragma unroll
            for (int i = 0; i < HP; ++i) if (on(i)) gt[i] = *reinterpret_cast<const f32x4*>(L.relu_src + grow_of(HP * h12 + i) * p.ld_relu + col0);
          }
#pragma unroll
          for (int i = 0; i < HP; ++i) if (on(i)) { en[i] = ell[r8 + 8 * (HP * h12 + i)]; U[HP * h12 + i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          for (int k = 0; k < D; ++k) {
            const int kn = k + 1 < D ? k + 1 : k;
            f32x4 z[HP];
#pragma unroll
            for (int i = 0; i < HP; ++i) if (on(i)) z[i] = *reinterpret_cast<const f32x4*>(slot0 + en[i].x * 32 + cq);
            int2 en_next[HP];
#pragma unroll
            for (int i = 0; i < HP; ++i) if (on(i)) en_next[i] = ell[kn * TM + r8 + 8 * (HP * h12 + i)];
#pragma unroll
            for (int i = 0; i < HP; ++i) {
              if (!on(i)) continue;
              const float w = __int_as_float(en[i].y);
#pragma unroll
              for (int q = 0; q < 4; ++q) U[HP * h12 + i][q] = fmaf(w, z[i][q], U[HP * h12 + i][q]);
              en[i] = en_next[i];
            }
          }
          if (want_gate) {
#pragma unroll
            for (int i = 0; i < HP; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) if (on(i)) gate_bits[(HP * h12 + i) >> 3] |= (relu_open(gt[i][q]) ? 1u : 0u) << (((HP * h12 + i) & 7) * 4 + q);
          }
        }
        wave_lds_sync();      // every lane's gathers are done: G_m goes over T
        put(m);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < RPA; ++i) U[i] += *reinterpret_cast<const f32x4*>(slot0 + (r8 + 8 * i) * 32 + cq);
        if (m > 0) {
          wave_lds_sync();
#pragma unroll
          for (int i = 0; i < RPA; ++i) *reinterpret_cast<f32x4*>(slot0 + (r8 + 8 * i) * 32 + cq) = U[i];
        }
        wave_lds_sync();      // (m > 0: the new T is complete; m == 0: the planes may go over the slot)
      }
    }

    S6STAMP(2 + li * 6 + 2);      // hops done
    S6STAMP(2 + li * 6 + 5);
    // ---- epilogue: bias / folded bias / masks / dropout / ReLU / gate / residual -> HBM and, split, the next layer's planes
    const bool keep = li + 1 < ct.n;
    [[maybe_unused]] uint32_t relu_words[NGW] = {};      // (forward chains) the ReLU's compare masks as the lane's sign-bit words
#pragma unroll
    for (int i = 0; i < RPA; ++i) U[i] += bias4;
    if (col_ok) {
      if (has_pre) {
        f32x4 pb4[NMAT];
#pragma unroll
        for (int m = 0; m < NMAT; ++m) pb4[m] = *reinterpret_cast<const f32x4*>(L.prebias + (size_t)m * p.hout + col0);
#pragma unroll
        for (int i = 0; i < RPA; ++i) {
          const f32x4 ps = *reinterpret_cast<const f32x4*>(p.pre_rowscale + grow_of(i) * 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) U[i] += pb4[m] * ps[m];
        }
      }
      if (has_dm) {
#pragma unroll
        for (int i = 0; i < RPA; ++i) U[i] *= *reinterpret_cast<const f32x4*>(L.dmask + grow_of(i) * p.ld_dmask + col0);
      }
      if (L.drop_id) {
#pragma unroll      // (fully unrolled: a rolled loop indexes U at run time and sends the whole array to scratch memory)
        for (int i = 0; i < RPA; ++i)
          U[i] *= dropout_mult4(drop_seed, drop_off, (uint32_t)L.drop_id, (uint32_t)grow_of(i), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
      }
      if (DIR != 2 && (L.relu & 1)) {
        if (S6_RELU_BITS && DIR == 1 && L.y_bits) {      // (uniform; end of round 6) the sign-bit words ride in the ReLU's compares: relu_nan4_bits, dss2_common.hpp
#pragma unroll
          for (int i = RPA - 1; i >= 0; --i) relu_nan4_bits(U[i], relu_words[i >> 3]);
        } else {
#pragma unroll
          for (int i = 0; i < RPA; ++i) relu_nan4(U[i]);
        }
      }
      if (has_rs) {
        // (piece i's four gates from bits 4 (i & 7) ..: gate_bits4, dss2_common.hpp; the switch folds once the loop is unrolled)
        auto gate_piece = [&](int i) {
          const uint32_t w_ = gate_bits[i >> 3];
          switch (i & 7) {
            case 0: gate_bits4<0>(U[i], w_); break;   case 1: gate_bits4<4>(U[i], w_); break;   case 2: gate_bits4<8>(U[i], w_); break;   case 3: gate_bits4<12>(U[i], w_); break;
            case 4: gate_bits4<16>(U[i], w_); break;  case 5: gate_bits4<20>(U[i], w_); break;  case 6: gate_bits4<24>(U[i], w_); break;  default: gate_bits4<28>(U[i], w_); break;
          }
        };
#pragma unroll
        for (int i = 0; i < RPA; ++i) gate_piece(i);
      }
      if (has_add) {
#pragma unroll
        for (int i = 0; i < RPA; ++i) U[i] += *reinterpret_cast<const f32x4*>(L.add_src + grow_of(i) * p.ld_add + col0);
      }
#pragma unroll
      for (int i = 0; i < RPA; ++i)
        if (rowv + 8 * i < R) *reinterpret_cast<f32x4*>(L.Y + (size_t)(ts + rowv + 8 * i) * p.ldy + col0) = U[i];
    }
    if (DIR != 2 && L.y_bits) {      // (uniform) the sign bits of what went to Y, in the layout the data-gradient form reads
      uint32_t* yb = reinterpret_cast<uint32_t*>(L.y_bits) + ((size_t)tile * ncg + cg) * (NGW * 64) + lane;
#pragma unroll
      for (int w = 0; w < NGW; ++w) {
        uint32_t word = 0u;
        if (S6_RELU_BITS && DIR == 1 && (L.relu & 1)) {      // (uniform) behind the forward set's ReLU the bits are its compare masks (no residual in that set)
          uint32_t valid = 0u;
#pragma unroll
          for (int i = 8 * w; i < 8 * w + 8 && i < RPA; ++i) valid |= (col_ok && rowv + 8 * i < R) ? (0xFu << ((i & 7) * 4)) : 0u;      // (pad rows and pad columns: zero bits)
          word = relu_words[w] & valid;
        } else {
#pragma unroll
          for (int i = 8 * w; i < 8 * w + 8 && i < RPA; ++i) {
            const bool in_y = col_ok && rowv + 8 * i < R;
#pragma unroll
            for (int q = 0; q < 4; ++q) word |= ((in_y && relu_open(U[i][q])) ? 1u : 0u) << ((i & 7) * 4 + q);
          }
        }
        yb[w * 64] = word;
      }
    }
    if (keep) {
      load_b(reinterpret_cast<const bf16x8*>(ct.l[li + 1].Bp), b0, 0);      // the next layer's first fragments
      if constexpr (F16) {
        // the next layer's scale: the tile's maximum over all stripes (one more barrier per layer; the planes then go over the slots)
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < RPA; ++i) {
          if (!(rowv + 8 * i < R && col_ok)) U[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          mx = absmax4(mx, U[i]);
        }
        mx = wave_max(mx);
        if (lane == 0) mxw[wave] = mx;
        s6_barrier();
        ea = tile_exponent();
#pragma unroll
        for (int i = 0; i < RPA; ++i) s6_store_split_h<S6_PLANE>(own_planes + s6_off(rowv + 8 * i, cq), U[i], ea);
      } else {
#pragma unroll
        for (int i = 0; i < RPA; ++i) {
          const int row = rowv + 8 * i;
          s6_store_split<S6_PLANE>(own_planes + s6_off(row, cq), (row < R && col_ok) ? U[i] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
      }
      S6STAMP(2 + li * 6 + 3);
      s6_barrier();   // the next layer's planes are complete
      S6STAMP(2 + li * 6 + 4);
    } else {
      S6STAMP(2 + li * 6 + 3);
      S6STAMP(2 + li * 6 + 4);
    }
  }
}

static size_t chain_sp6_lds_bytes(int nrb, int ncg, int ell_width) { return (size_t)ncg * s6_region(nrb) * 4 + (size_t)32 * nrb * ell_width * 8 + 64; }      // (+ the f16x3 form's maxima)

}  // namespace dss2
extern "C" int dss2_chain_sp6_single_group_min_tiles(void) {
  static const int v = [] { const char* e = getenv("DSS2_CHAIN_SP6_NCG1"); return (e ? atoi(e) : 1) ? 768 : -1; }();
  return v;
}
namespace dss2 {

bool chain_sp6_supported(const dss2_gemm_prop_args& a) {
  static const int on = [] { const char* e = getenv("DSS2_CHAIN_SP"); return e ? atoi(e) : 1; }();
  static const int f16on = [] { const char* e = getenv("DSS2_CHAIN_SP_F16"); return e ? atoi(e) : 1; }();
  if (a.b_format == 2 && !f16on) return false;
  // (round 6: ONE column group -- dim_hid 32, the reference driver's model on 70-bus grids -- runs this form too, as single-wave
  //  workgroups, seven of them per CU, where there are enough tiles to fill the chip that way: measured on the driver's model line,
  //  2.47 -> 2.33 ms per step at 1024 tiles against the three-waves-per-column-group bf16x6 chain, but 1.55 -> 1.63 at 512 and 1.06 -> 1.15 ms at 64 tiles, where
  //  a tile's latency is what counts.  A query (ntiles = 0) answers for the capability; the launch and the host's policy
  //  (ops.py, dss2_chain_sp6_single_group_min_tiles) apply the tile count.  DSS2_CHAIN_SP6_NCG1=0: never)
  if (a.ncg == 1 && (dss2_chain_sp6_single_group_min_tiles() < 0 || (a.ntiles > 0 && a.ntiles < dss2_chain_sp6_single_group_min_tiles()))) return false;
  const int min_ncg = 1;
  return on && (a.b_format == 1 || a.b_format == 2) && (a.nrb == 6 || a.nrb == 3) && a.nmat >= 2 && a.nmat <= 3 && (a.kpad & 15) == 0 && a.kpad <= 32 * a.ncg &&
         a.ncg >= min_ncg && a.ncg <= 4 && chain_sp6_lds_bytes(a.nrb, a.ncg, a.ell_width) <= (size_t)(a.nrb == 3 ? kMaxLdsBytes / 2 : kMaxLdsBytes);
}

template <int NRB, int NMAT, int DIR, int HM = 0, bool F16 = false, int RPA = 4 * NRB>
static int launch_sp6(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head& hd, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_chain_sp6_kernel<NRB, NMAT, DIR, HM, F16, RPA>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_prop_chain(split planes, 192 rows)")) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * a.ncg), chain_sp6_lds_bytes(NRB, a.ncg, a.ell_width), stream, a, ct, hd);
  return check_launch("gemm_prop_chain(split planes, 192 rows)");
}

int launch_chain_sp6(const dss2_gemm_prop_args& a, const ChainTable& ct, const dss2_chain_head* head, hipStream_t s) {
  bool fwd = true, bwd = true;
  for (int i = 0; i < ct.n; ++i) {
    const dss2_chain_layer& L = ct.l[i];
    if (L.relu_src || L.gate_bits || L.dmask || L.add_src) fwd = false;
    if (L.bias || L.prebias || L.dmask || L.add_src || L.y_bits || (L.relu & 1) || (L.relu_src && !L.gate_bits)) bwd = false;
  }
  dss2_chain_head hd = {};
  // 96-row tiles none of which holds more than 72 rows (70-bus graphs, one per tile): 9 of the 12 row pieces per lane are active (RPA, round 6;
  // the f16x3 K = 2 forms of the C3 path; DSS2_CHAIN_RPA=0: all 12)
  static const int rpa_on = [] { const char* e = getenv("DSS2_CHAIN_RPA"); return e ? atoi(e) : 1; }();
  const bool rpa9 = rpa_on && a.b_format == 2 && a.nrb == 3 && a.nmat == 3 && a.max_tile_rows > 0 && a.max_tile_rows <= 72;
  if (head) {      // the backward head (mode 2) rides in the staging of the data-gradient launch; nothing else is built here
    if (head->mode != 2 || !bwd) { set_error("gemm_prop_chain_head: tall tiles take the backward head (mode 2) on a data-gradient chain only"); return 2; }
    hd = *head;
    if (rpa9) return launch_sp6<3, 3, 2, 2, true, 9>(a, ct, hd, s);
    if (a.b_format == 2) {
      if (a.nrb == 3) return a.nmat == 2 ? launch_sp6<3, 2, 2, 2, true>(a, ct, hd, s) : launch_sp6<3, 3, 2, 2, true>(a, ct, hd, s);
      return a.nmat == 2 ? launch_sp6<6, 2, 2, 2, true>(a, ct, hd, s) : launch_sp6<6, 3, 2, 2, true>(a, ct, hd, s);
    }
    if (a.nrb == 3) return a.nmat == 2 ? launch_sp6<3, 2, 2, 2>(a, ct, hd, s) : launch_sp6<3, 3, 2, 2>(a, ct, hd, s);
    return a.nmat == 2 ? launch_sp6<6, 2, 2, 2>(a, ct, hd, s) : launch_sp6<6, 3, 2, 2>(a, ct, hd, s);
  }
  const int dir = fwd ? 1 : (bwd ? 2 : 0);      // (0: a layer table that mixes the feature sets runs the generic instantiation)
  if (a.b_format == 2) {      // f16x3: the direction-specialised forms only
    if (dir == 0) { set_error("gemm_prop_chain(f16x3): a layer table that mixes forward and data-gradient features needs bf16x3 weights (b_format 1)"); return 2; }
    if (rpa9) return dir == 1 ? launch_sp6<3, 3, 1, 0, true, 9>(a, ct, hd, s) : launch_sp6<3, 3, 2, 0, true, 9>(a, ct, hd, s);
#define DSS2_S6_LAUNCH_H(NRB, NMAT) (dir == 1 ? launch_sp6<NRB, NMAT, 1, 0, true>(a, ct, hd, s) : launch_sp6<NRB, NMAT, 2, 0, true>(a, ct, hd, s))
    if (a.nrb == 3) return a.nmat == 2 ? DSS2_S6_LAUNCH_H(3, 2) : DSS2_S6_LAUNCH_H(3, 3);
    return a.nmat == 2 ? DSS2_S6_LAUNCH_H(6, 2) : DSS2_S6_LAUNCH_H(6, 3);
#undef DSS2_S6_LAUNCH_H
  }
#define DSS2_S6_LAUNCH(NRB, NMAT) (dir == 1 ? launch_sp6<NRB, NMAT, 1>(a, ct, hd, s) : (dir == 2 ? launch_sp6<NRB, NMAT, 2>(a, ct, hd, s) : launch_sp6<NRB, NMAT, 0>(a, ct, hd, s)))
  if (a.nrb == 3) return a.nmat == 2 ? DSS2_S6_LAUNCH(3, 2) : DSS2_S6_LAUNCH(3, 3);
  return a.nmat == 2 ? DSS2_S6_LAUNCH(6, 2) : DSS2_S6_LAUNCH(6, 3);
#undef DSS2_S6_LAUNCH
}

}  // namespace dss2

#ifdef DSS2_CHAIN_STAMPS
extern "C" int dss2_debug_read_cstamps_sp6(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_cstamps), sizeof(unsigned long long) * n);
}
#endif
