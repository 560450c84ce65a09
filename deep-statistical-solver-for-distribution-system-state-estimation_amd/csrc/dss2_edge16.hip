// Edge MLP, first Linear on the bf16 matrix pipe as bf16x6 (fp32-accurate, dss2_common.hpp: split3): forward, and the
// recomputation of the pre-activation in the backward.  gfx950.  (/root/reference/networks.py:176-181: the per-edge
// Linear(22 -> hid) + ReLU, summed per target.)
//
// dss2_edge.hip's matrix-pipe kernels run  Z_k = A_k W1^T,  A_k = [x_i | x_j | edge_attr],  as twelve v_mfma_f32_32x32x2_f32 per
// ELL slot and row block (64 cycles each), between two workgroup barriers per slot (A_k is rebuilt in LDS for every slot), and
// read the slot's validity word per accumulator element.  Here
//   * the x_i term does not depend on the slot: C_i = X_i W1a^T once per tile and row block (K = 8, one k-step of 16);
//   * a slot adds [x_j | edge_attr | 1 | invalid] W1bc^T: K = 8 + 6 + 2 = 16, ONE k-step -- six v_mfma_f32_32x32x16_bf16
//     (32 cycles each) per slot and row block, started on the accumulator C_i;
//   * the bias rides in the k = 14 column (a = 1 on valid slots, b = b1: exact, 1 is a bf16 number and b1 = h + m + l), and
//     the k = 15 column carries a = 1 on EMPTY slots against b = -1e30: an empty slot's pre-activation is hugely negative, the
//     ReLU removes it, and the epilogue is two vector instructions per element (max, add) with no validity lookups;
//   * the split planes of every slot are built in ONE phase after the staging (each input element is split once per workgroup,
//     16-byte row pieces: conflict-free b128 fragment reads without padding), so the slot loop of the forward has no barrier;
//   * 192-row tiles (end of round 5; round 6: 128-row tiles as two parts of 64 rows) are walked as two PARTS of 96 rows by the 96-row instantiations: the whole tile's x rows are staged
//     for each part (x_j may be any of them), everything else is the part's own (EdgeTileArgs::parts / xtm, edge_stage_part).
// The backward recomputes the pre-activation with the very same plane values, fragments and MFMA order -- its gates are bit for
// bit the forward's -- and forms dW1 += dZ_k^T A_k, a contraction over the tile's rows, as bf16x6 too (see edge16_bwd_kernel).
// Built without packed fp32 VALU ops like every translation unit that runs bf16 MFMAs beside other workgroups (csrc/build.sh).
#include <stdlib.h>

#include "dss2_edge_tile.hpp"

namespace dss2 {

constexpr float E16_KILL = -1e30f;

// 8 consecutive k of one operand row -> the three bf16 planes (16 bytes each)
__device__ __forceinline__ void e16_split8(const f32x4 v0, const f32x4 v1, uint4& h, uint4& m, uint4& l) {
  split3_pair(v0[0], v0[1], h.x, m.x, l.x);
  split3_pair(v0[2], v0[3], h.y, m.y, l.y);
  split3_pair(v1[0], v1[1], h.z, m.z, l.z);
  split3_pair(v1[2], v1[3], h.w, m.w, l.w);
}
__device__ __forceinline__ bf16x8 e16_frag(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }

struct E16Lds {
  EdgeStage s;
  const float* xi;      // the x rows of this part's own rows (s.xs + r0 * FN)
  char* PI;      // [3 planes][TM][16 B]: x_i
  char* PK;      // [D][3 planes][2 halves][TM][16 B]: half 0 = x_j, half 1 = edge_attr (6) | valid | empty
};

template <int NRB>
__device__ __forceinline__ E16Lds e16_ptrs(float* esm, int D, int XT) {
  constexpr int TM = NRB * 32;
  E16Lds L;
  L.s.xs = esm;
  L.xi = esm;
  L.s.eaL = L.s.xs + XT * FN;
  L.s.other = reinterpret_cast<int*>(L.s.eaL + D * TM * 8);
  L.PI = reinterpret_cast<char*>(L.s.other + D * TM);
  L.PK = L.PI + 3 * TM * 16;
  L.s.Ak = nullptr; L.s.st = nullptr;
  return L;
}

static size_t e16_lds_bytes(int TM, int D, int XT) {      // forward: all slots' planes at once (XT: rows of the whole tile, whose x rows are staged)
  return ((size_t)XT * FN + (size_t)D * TM * 8) * 4 + (size_t)D * TM * 4 + 3 * (size_t)TM * 16 + (size_t)D * 6 * TM * 16;
}

// all slots' planes in one phase: unit = (image, row) of 8 elements; images: x_i, then (slot k, half) for k < D
template <int NRB>
__device__ __forceinline__ void e16_build(const E16Lds& L, int D, int tid, int nthreads) {
  constexpr int TM = NRB * 32;
  const EdgeStage& s = L.s;
  for (int u = tid; u < TM * (1 + 2 * D); u += nthreads) {
    const int img = u / TM, row = u - img * TM;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    char* dst;
    int pstride;
    if (img == 0) {
      v0 = *reinterpret_cast<const f32x4*>(L.xi + row * FN);
      v1 = *reinterpret_cast<const f32x4*>(L.xi + row * FN + 4);
      dst = L.PI + row * 16;
      pstride = TM * 16;
    } else {
      const int k = (img - 1) >> 1, hh = (img - 1) & 1;
      const int o = s.other[k * TM + row];
      if (hh == 0) {
        if (o >= 0) { v0 = *reinterpret_cast<const f32x4*>(s.xs + o * FN); v1 = *reinterpret_cast<const f32x4*>(s.xs + o * FN + 4); }
      } else if (o >= 0) {
        v0 = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8);
        const f32x4 t = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8 + 4);
        v1 = f32x4{t[0], t[1], 1.f, 0.f};
      } else {
        v1 = f32x4{0.f, 0.f, 0.f, 1.f};
      }
      dst = L.PK + ((size_t)(k * 3) * 2 + hh) * TM * 16 + row * 16;
      pstride = 2 * TM * 16;
    }
    uint4 h, m, l;
    e16_split8(v0, v1, h, m, l);
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + pstride) = m;
    *reinterpret_cast<uint4*>(dst + 2 * pstride) = l;
  }
}

// the wave's weight fragments: lane (c32 = hidden column of the wave's group, half) holds k = 8 half .. 8 half + 7
struct E16W { bf16x8 ah, am, al, bh, bm, bl; };      // a*: W1a (x_i columns), b*: W1bc (x_j | edge_attr | b1 | kill)
__device__ __forceinline__ E16W e16_weights(const float* __restrict__ W1, const float* __restrict__ b1, int j, int half) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0, b1v;
  const float* w = W1 + (size_t)j * FC;
  if (half == 0) {
    a0 = f32x4{w[0], w[1], w[2], w[3]}; a1 = f32x4{w[4], w[5], w[6], w[7]};
    b0 = f32x4{w[8], w[9], w[10], w[11]}; b1v = f32x4{w[12], w[13], w[14], w[15]};
  } else {
    b0 = f32x4{w[16], w[17], w[18], w[19]}; b1v = f32x4{w[20], w[21], b1[j], E16_KILL};
  }
  uint4 h, m, l;
  E16W r;
  e16_split8(a0, a1, h, m, l);
  r.ah = e16_frag(h); r.am = e16_frag(m); r.al = e16_frag(l);
  e16_split8(b0, b1v, h, m, l);
  r.bh = e16_frag(h); r.bm = e16_frag(m); r.bl = e16_frag(l);
  return r;
}

// six MFMAs, smallest terms first (the order of the other bf16x6 kernels)
__device__ __forceinline__ f32x16 e16_mma6(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm,
                                           const bf16x8 bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  return c;
}

// C_i of row block rb: x_i W1a^T (the k = 8 .. 15 half of the step is zero on both sides)
template <int NRB>
__device__ __forceinline__ f32x16 e16_xterm(const E16Lds& L, const E16W& w, int rb, int c32, int half) {
  constexpr int TM = NRB * 32;
  const uint4 z = {0u, 0u, 0u, 0u};
  const char* pi = L.PI + (rb * 32 + c32) * 16;
  const uint4 h = half ? z : *reinterpret_cast<const uint4*>(pi);
  const uint4 m = half ? z : *reinterpret_cast<const uint4*>(pi + TM * 16);
  const uint4 l = half ? z : *reinterpret_cast<const uint4*>(pi + 2 * TM * 16);
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  return e16_mma6(e16_frag(h), e16_frag(m), e16_frag(l), w.ah, w.am, w.al, c);
}

// pre-activation of slot k, row block rb (bias included; hugely negative on empty slots): THE definition, forward and backward
template <int NRB>
__device__ __forceinline__ f32x16 e16_z(const E16Lds& L, const E16W& w, int k, int rb, int c32, int half, const f32x16 ci) {
  constexpr int TM = NRB * 32;
  const char* pk = L.PK + ((size_t)(k * 3) * 2 + half) * TM * 16 + (rb * 32 + c32) * 16;
  const uint4 h = *reinterpret_cast<const uint4*>(pk);
  const uint4 m = *reinterpret_cast<const uint4*>(pk + 2 * TM * 16);
  const uint4 l = *reinterpret_cast<const uint4*>(pk + 4 * TM * 16);
  return e16_mma6(e16_frag(h), e16_frag(m), e16_frag(l), w.bh, w.bm, w.bl, ci);
}

// (four waves per SIMD: the kernel waits on its staging chain tile_start -> ELL entry -> edge_attr row, and a fourth resident
//  workgroup covers more of it than the 3-8 spilled registers cost: 18.5 -> 15.6 us at C2; the backward, whose vector and matrix
//  work are balanced, lost 3 us when squeezed from two to three waves)
template <int NRB>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8))) edge16_fwd_kernel(const EdgeTileArgs p) {
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);      // one 32-column group of the hidden layer per wave
  const int c32 = lane & 31, half = lane >> 5;
  const int D = p.D;
  const int parts = p.parts > 1 ? p.parts : 1, XT = p.xtm > 0 ? p.xtm : NRB * 32;      // (192-row tiles: two parts of 96 rows)
  E16Lds L = e16_ptrs<NRB>(esm, D, XT);
  const int j = cg * 32 + c32;
  const E16W w = e16_weights(p.W1, p.b1, j, half);
  for (int vt = blockIdx.x; vt < p.ntiles * parts; vt += gridDim.x) {
    const int tile = vt / parts, r0 = (vt - tile * parts) * (NRB * 32);
    const int ts_full = p.tile_start[tile], R_full = p.tile_start[tile + 1] - ts_full;
    const int ts = ts_full + r0;
    const int R = R_full - r0 < NRB * 32 ? R_full - r0 : NRB * 32;
    if (R <= 0) continue;      // (uniform: a part beyond the tile's rows)
    L.xi = L.s.xs + r0 * FN;
    edge_stage_part<NRB>(p, L.s, tile, XT, r0, ts_full, R_full, tid, nthreads);
    __syncthreads();
    e16_build<NRB>(L, D, tid, nthreads);
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const f32x16 ci = e16_xterm<NRB>(L, w, rb, c32, half);
      f32x16 Sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) Sacc[r] = 0.f;
      for (int k = 0; k < D; ++k) {
        const f32x16 c = e16_z<NRB>(L, w, k, rb, c32, half, ci);
#pragma unroll
        for (int r = 0; r < 16; ++r) Sacc[r] += relu_nan(c[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row(r, half);
        if (row < R) p.S[(int64_t)(ts + row) * p.h + j] = Sacc[r];
      }
    }
    __syncthreads();      // (a workgroup that walks several tiles restages over the images)
  }
}

// ---- backward by target: dW1, db1 (slab per workgroup) and optionally U[row] = sum over incoming edges of dZ.
// dW1[o][i] += sum_rows dZ_k[row][o] A_k[row][i] is a contraction over the tile's rows, and it is bf16x6 as well (round 4; the
// fp32 form -- sixteen v_mfma_f32_32x32x2_f32 per slot and row block, dZ through a wave-private LDS tile -- took five sixths of
// the kernel's matrix-pipe time: profiles/experiments/r04_edge16_bwd_fp32_dw.hip.txt):
//   * the A operand is dZ^T: lane (o, half) of the recomputation's accumulator already holds dZ[.][o] for the sixteen rows
//     acc_row(r, half) of a row block.  Any order of the contraction index is as good as another, so k-step s takes the lane's
//     registers r = 8 s .. 8 s + 7 as they are: dZ is split in registers and never touches LDS;
//   * the B operand then needs A_k[rho][i] for the same eight rows rho = acc_row(8 s + p, half): A_k is kept as TRANSPOSED
//     planes [plane][column][row block][32 rows in that order] (a row's position is its index with bits 2 and 3 swapped), one
//     ds_read_b128 per plane.  Columns: x_i (8, once per tile -- rows of an empty slot have dZ = 0, the values need no mask),
//     then x_j | edge_attr | 1 | 0 (16, per slot); the 1 column yields db1.
// The recomputation's planes are built per slot too (the forward builds all slots at once and runs its slot loop without a
// barrier; here every slot has its barriers anyway), so a 96-row tile at D = 5 takes 52 KB instead of 130 KB and two workgroups
// share a CU.
constexpr int E16_ATN = 16;                                   // per-slot transposed columns: x_j (8) | edge_attr (6) | 1 | 0
__host__ __device__ constexpr int e16_cs(int TM) { return TM * 2 + 16; }      // bytes per transposed column and plane (+16: the b128 reads of 32 columns spread over the banks)
__device__ __forceinline__ int e16_qpos(int row) {            // position of a row inside its transposed row block: bits 2 and 3 swapped
  return (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
}

struct E16Bwd {
  EdgeStage s;      // xs, eaL, other (Ak / st unused)
  const float* xi;  // the x rows of this part's own rows
  char* PI;         // [3 planes][TM][16 B]: x_i, the recomputation's row operand
  char* PK;         // [3 planes][2 halves][TM][16 B]: ONE slot
  char* ATI;        // [3 planes][8 columns][CS]: x_i transposed
  char* ATK;        // [3 planes][16 columns][CS]: the slot's x_j | edge_attr | 1 | 0 transposed
};

template <int NRB>
__device__ __forceinline__ E16Bwd e16_bwd_ptrs(float* esm, int D, int XT) {
  constexpr int TM = NRB * 32, CS = e16_cs(TM);
  E16Bwd L;
  L.s.xs = esm;
  L.xi = esm;
  L.s.eaL = L.s.xs + XT * FN;
  L.s.other = reinterpret_cast<int*>(L.s.eaL + D * TM * 8);
  L.s.Ak = nullptr; L.s.st = nullptr;
  L.PI = reinterpret_cast<char*>(L.s.other + D * TM);
  L.PK = L.PI + 3 * TM * 16;
  L.ATI = L.PK + 6 * TM * 16;
  L.ATK = L.ATI + 3 * FN * CS;
  return L;
}

static size_t e16_bwd_lds_bytes(int TM, int D, int XT) {
  return ((size_t)XT * FN + (size_t)D * TM * 8) * 4 + (size_t)D * TM * 4 + 3 * (size_t)TM * 16 + 6 * (size_t)TM * 16 +
         3 * (size_t)(FN + E16_ATN) * e16_cs(TM);
}

// rows (2 rp, 2 rp + 1) x four columns c0 .. c0 + 3 of a transposed image with NC columns
template <int NRB>
__device__ __forceinline__ void e16_store_t(char* img, int NC, int c0, int rp, const f32x4 v0, const f32x4 v1) {
  constexpr int CS = e16_cs(NRB * 32);
  const int row = 2 * rp;
  char* dst = img + c0 * CS + (row >> 5) * 64 + 2 * e16_qpos(row & 31);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, m, l;
    split3_pair(v0[q], v1[q], h, m, l);
    *reinterpret_cast<uint32_t*>(dst + q * CS) = h;
    *reinterpret_cast<uint32_t*>(dst + (NC + q) * CS) = m;
    *reinterpret_cast<uint32_t*>(dst + (2 * NC + q) * CS) = l;
  }
}

// once per tile: x_i as the recomputation's row planes and as transposed columns
template <int NRB>
__device__ __forceinline__ void e16_bwd_build_tile(const E16Bwd& L, int tid, int nthreads) {
  constexpr int TM = NRB * 32;
  for (int row = tid; row < TM; row += nthreads) {
    uint4 h, m, l;
    e16_split8(*reinterpret_cast<const f32x4*>(L.xi + row * FN), *reinterpret_cast<const f32x4*>(L.xi + row * FN + 4), h, m, l);
    char* dst = L.PI + row * 16;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + TM * 16) = m;
    *reinterpret_cast<uint4*>(dst + 2 * TM * 16) = l;
  }
  for (int u = tid; u < (TM / 2) * 2; u += nthreads) {
    const int rp = u >> 1, q4 = u & 1;
    const float* src = L.xi + (2 * rp) * FN + 4 * q4;
    e16_store_t<NRB>(L.ATI, FN, 4 * q4, rp, *reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + FN));
  }
}

// per slot: the recomputation's planes (exactly the forward's values: same inputs, same split) and the transposed columns
template <int NRB>
__device__ __forceinline__ void e16_bwd_build_slot(const E16Bwd& L, int k, int tid, int nthreads) {
  constexpr int TM = NRB * 32;
  const EdgeStage& s = L.s;
  for (int u = tid; u < 2 * TM; u += nthreads) {
    const int hh = u / TM, row = u - hh * TM;
    const int o = s.other[k * TM + row];
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (hh == 0) {
      if (o >= 0) { v0 = *reinterpret_cast<const f32x4*>(s.xs + o * FN); v1 = *reinterpret_cast<const f32x4*>(s.xs + o * FN + 4); }
    } else if (o >= 0) {
      v0 = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8);
      const f32x4 t = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8 + 4);
      v1 = f32x4{t[0], t[1], 1.f, 0.f};
    } else {
      v1 = f32x4{0.f, 0.f, 0.f, 1.f};
    }
    uint4 h, m, l;
    e16_split8(v0, v1, h, m, l);
    char* dst = L.PK + (size_t)hh * TM * 16 + row * 16;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + 2 * TM * 16) = m;
    *reinterpret_cast<uint4*>(dst + 4 * TM * 16) = l;
  }
  for (int u = tid; u < (TM / 2) * 4; u += nthreads) {
    const int rp = u >> 2, q4 = u & 3;
    f32x4 v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = 2 * rp + e;
      const int o = s.other[k * TM + row];
      v[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (o >= 0) {
        if (q4 < 2) v[e] = *reinterpret_cast<const f32x4*>(s.xs + o * FN + 4 * q4);
        else if (q4 == 2) v[e] = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8);
        else { const f32x4 t = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + row) * 8 + 4); v[e] = f32x4{t[0], t[1], 1.f, 0.f}; }
      }
    }
    e16_store_t<NRB>(L.ATK, E16_ATN, 4 * q4, rp, v[0], v[1]);
  }
}

template <int NRB, bool WITH_U>
__global__ void __launch_bounds__(512) edge16_bwd_kernel(const EdgeTileArgs p) {
  constexpr int TM = NRB * 32, CS = e16_cs(TM);
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, half = lane >> 5;
  const int D = p.D;
  const int parts = p.parts > 1 ? p.parts : 1, XT = p.xtm > 0 ? p.xtm : NRB * 32;      // (192-row tiles: two parts of 96 rows)
  E16Bwd B = e16_bwd_ptrs<NRB>(esm, D, XT);
  E16Lds L;                           // the view e16_xterm / e16_z read: the one-slot planes are "slot 0"
  L.s = B.s; L.xi = B.xi; L.PI = B.PI; L.PK = B.PK;
  const int j = cg * 32 + c32;
  const E16W w = e16_weights(p.W1, p.b1, j, half);
  // this lane's B-operand column of the weight-gradient product: input column c32 -> x_i (0..7), the slot's columns (8..23), none
  const char* bcol = c32 < FN ? B.ATI + c32 * CS : B.ATK + ((c32 < FN + E16_ATN ? c32 : FN) - FN) * CS;
  const int bps = c32 < FN ? FN * CS : E16_ATN * CS;      // plane stride of that image
  const bool bcol_ok = c32 < FN + E16_ATN;
  f32x16 dWacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dWacc[r] = 0.f;
  for (int vt = blockIdx.x; vt < p.ntiles * parts; vt += gridDim.x) {
    const int tile = vt / parts, r0 = (vt - tile * parts) * (NRB * 32);
    const int ts_full = p.tile_start[tile], R_full = p.tile_start[tile + 1] - ts_full;
    const int ts = ts_full + r0;
    const int R = R_full - r0 < NRB * 32 ? R_full - r0 : NRB * 32;
    if (R <= 0) continue;      // (uniform: a part beyond the tile's rows)
    B.xi = B.s.xs + r0 * FN;
    edge_stage_part<NRB>(p, B.s, tile, XT, r0, ts_full, R_full, tid, nthreads);
    f32x16 gS[NRB], Uacc[WITH_U ? NRB : 1];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row(r, half);
        gS[rb][r] = row < R ? p.dS[(int64_t)(ts + row) * p.h + j] : 0.f;
        if (WITH_U) Uacc[rb][r] = 0.f;
      }
    // dS is split into its three bf16 planes ONCE per tile (pairs of consecutive registers, the pairing of the weight-gradient product's
    // A fragments below); a slot then gates the PLANES with 16-bit masks -- 4 vector instructions per element and slot instead of the
    // 7.5 of selecting the gated value and splitting it again.  split3(gate ? g : 0) = gate ? split3(g) : 0 plane by plane: same bits.
    // (where the planes fit beside everything else: not with U -- the fp32 values stay live for it -- and not at 96 rows: 26 / 8 spilled
    //  registers otherwise)
    constexpr bool PRE = !WITH_U && NRB <= 2;
    uint32_t gh[PRE ? NRB : 1][8], gm[PRE ? NRB : 1][8], gl[PRE ? NRB : 1][8];
    if constexpr (PRE) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j) split3_pair(gS[rb][2 * j], gS[rb][2 * j + 1], gh[rb][j], gm[rb][j], gl[rb][j]);
    }
    __syncthreads();
    e16_bwd_build_tile<NRB>(B, tid, nthreads);
    f32x16 ci[NRB];
    for (int k = 0; k < D; ++k) {
      e16_bwd_build_slot<NRB>(B, k, tid, nthreads);
      __syncthreads();
      if (k == 0) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) ci[rb] = e16_xterm<NRB>(L, w, rb, c32, half);
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const f32x16 c = e16_z<NRB>(L, w, 0, rb, c32, half, ci[rb]);
        [[maybe_unused]] uint32_t mk[8];      // (PRE) the gates of registers (2 j, 2 j + 1) as a mask over the two bf16 halves of a plane word
        [[maybe_unused]] f32x16 dz;
        if constexpr (PRE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) mk[j] = (relu_open(c[2 * j]) ? 0x0000ffffu : 0u) | (relu_open(c[2 * j + 1]) ? 0xffff0000u : 0u);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            dz[r] = relu_open(c[r]) ? gS[rb][r] : 0.f;
            if (WITH_U) Uacc[rb][r] += dz[r];
          }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          uint4 ah, am, al;
          if constexpr (PRE) {
            ah = uint4{gh[rb][4 * st] & mk[4 * st], gh[rb][4 * st + 1] & mk[4 * st + 1], gh[rb][4 * st + 2] & mk[4 * st + 2], gh[rb][4 * st + 3] & mk[4 * st + 3]};
            am = uint4{gm[rb][4 * st] & mk[4 * st], gm[rb][4 * st + 1] & mk[4 * st + 1], gm[rb][4 * st + 2] & mk[4 * st + 2], gm[rb][4 * st + 3] & mk[4 * st + 3]};
            al = uint4{gl[rb][4 * st] & mk[4 * st], gl[rb][4 * st + 1] & mk[4 * st + 1], gl[rb][4 * st + 2] & mk[4 * st + 2], gl[rb][4 * st + 3] & mk[4 * st + 3]};
          } else {
            split3_pair(dz[8 * st + 0], dz[8 * st + 1], ah.x, am.x, al.x);
            split3_pair(dz[8 * st + 2], dz[8 * st + 3], ah.y, am.y, al.y);
            split3_pair(dz[8 * st + 4], dz[8 * st + 5], ah.z, am.z, al.z);
            split3_pair(dz[8 * st + 6], dz[8 * st + 7], ah.w, am.w, al.w);
          }
          const char* bp = bcol + rb * 64 + 32 * st + 16 * half;
          const uint4 z4 = {0u, 0u, 0u, 0u};
          const uint4 bh = bcol_ok ? *reinterpret_cast<const uint4*>(bp) : z4;
          const uint4 bm = bcol_ok ? *reinterpret_cast<const uint4*>(bp + bps) : z4;
          const uint4 bl = bcol_ok ? *reinterpret_cast<const uint4*>(bp + 2 * bps) : z4;
          dWacc = e16_mma6(e16_frag(ah), e16_frag(am), e16_frag(al), e16_frag(bh), e16_frag(bm), e16_frag(bl), dWacc);
        }
      }
      __syncthreads();   // everyone is done with the slot's planes
    }
    if (WITH_U) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb * 32 + acc_row(r, half);
          if (row < R) p.U[(int64_t)(ts + row) * p.ldu + j] = Uacc[rb][r];
        }
    }
  }
  if (!p.slab) return;
  float* out = p.slab + (size_t)blockIdx.x * ((size_t)p.h * FC + p.h);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = cg * 32 + acc_row(r, half);
    if (c32 < FC) out[(size_t)o * FC + c32] = dWacc[r];
    else if (c32 == FC) out[(size_t)p.h * FC + o] = dWacc[r];
  }
}

bool edge16_ok(int h, int nrb, int D, bool bwd, bool with_u) {
  // (read per call, like DSS2_EDGE_MFMA: lets a test switch forms inside one process; both passes must switch together,
  //  the backward recomputes the forward's gates)
  const char* env = getenv("DSS2_EDGE_BF16");
  if (env && atoi(env) == 0) return false;
  if ((h & 31) || h > 256 || !(nrb == 1 || nrb == 2 || nrb == 3 || nrb == 4 || nrb == 6) || D < 1 || D > 32) return false;      // (no kernel of the library tiles at 160 rows)
  if (bwd && with_u && nrb >= 3) return false;      // (that instantiation misses its register budget, as in dss2_edge.hip)
  // (128- / 192-row tiles: two parts of 64 / 96 rows, the whole tile's x rows staged for each)
  const int tm = nrb == 4 ? 64 : (nrb == 6 ? 96 : nrb * 32);
  return (bwd ? e16_bwd_lds_bytes(tm, D, nrb * 32) : e16_lds_bytes(tm, D, nrb * 32)) <= (size_t)kMaxLdsBytes;
}

template <int NRB>
static int launch16(const EdgeTileArgs& a, int grid, bool bwd, hipStream_t s) {
  const int nw = a.h >> 5;
  const int xt = a.xtm > 0 ? a.xtm : NRB * 32;
  const size_t lds = bwd ? e16_bwd_lds_bytes(NRB * 32, a.D, xt) : e16_lds_bytes(NRB * 32, a.D, xt);
  if (bwd && a.U) {
    if constexpr (NRB <= 2) {
      static std::atomic<uint32_t> lds_done{0};
      auto kern = edge16_bwd_kernel<NRB, true>;
      if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge16_bwd")) return 1;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
    } else {
      set_error("edge16_bwd with U: tiles above 64 rows are served by the VALU kernel"); return 2;
    }
  } else if (bwd) {
    static std::atomic<uint32_t> lds_done{0};
    auto kern = edge16_bwd_kernel<NRB, false>;
    if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge16_bwd")) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
  } else {
    static std::atomic<uint32_t> lds_done{0};
    auto kern = edge16_fwd_kernel<NRB>;
    if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge16_fwd")) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
  }
  return check_launch(bwd ? "edge16_bwd" : "edge16_fwd");
}

int launch_edge16(const EdgeTileArgs& a, int nrb, int grid, bool bwd, hipStream_t s) {
  switch (nrb) {
    case 1: return launch16<1>(a, grid, bwd, s);
    case 2: return launch16<2>(a, grid, bwd, s);
    case 4: {      // 128-row tiles as two parts of 64 rows (round 6; the VALU tile kernels before)
      EdgeTileArgs b = a;
      b.xtm = 128; b.parts = 2;
      return launch16<2>(b, bwd ? grid : 2 * grid, bwd, s);
    }
    case 6: {      // 192-row tiles as two parts of 96 rows each (the forward: one workgroup per part)
      EdgeTileArgs b = a;
      b.xtm = 32 * nrb; b.parts = 2;
      return launch16<3>(b, bwd ? grid : 2 * grid, bwd, s);
    }
    default: return launch16<3>(a, grid, bwd, s);
  }
}

}  // namespace dss2
