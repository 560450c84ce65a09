#pragma once
// Shared by dss2_edge.hip (fp32 MFMA / VALU tile kernels of the edge MLP) and dss2_edge16.hip (its first Linear as bf16x6).
#include "dss2_common.hpp"

namespace dss2 {

constexpr int FN = 8, FE = 6, FC = 2 * FN + FE;  // feature dims of the reference's data (networks.py:170)

struct EdgeTileArgs {
  const float* x; int64_t ldx; const float* ea; int64_t ldea;
  const float* W1; const float* b1; const float* dS;
  const int32_t* tile_start; const int2* ell_ent;   // [ntiles][D][TM] {local other node, eid | flip<<31 ; -1 = empty}
  float* S; float* slab; float* U; int64_t ldu;
  int h, D, TM, by_source, ntiles;
  // dss2_edge16.hip on 192-row tiles: a tile is walked as `parts` parts of 96 rows; xtm = rows of the whole tile (the stride of the ELL
  // table, and what x_j may point into).  0 / 1: whole tiles.
  int xtm = 0, parts = 1;
};

constexpr int EM_LDA = 36;   // A_k row stride: 24 inputs + 8 zero columns (B operand of the dW MFMA) + 4 pad

struct EdgeStage {
  float* xs; float* eaL; int* other; float* Ak; float* st;
};

template <int NRB>
__device__ __forceinline__ EdgeStage edge_stage_ptrs(float* esm, int D, int nw, bool with_st) {
  constexpr int TM = NRB * 32;
  EdgeStage s;
  s.xs = esm;
  s.eaL = s.xs + TM * FN;
  s.other = reinterpret_cast<int*>(s.eaL + D * TM * 8);
  s.Ak = reinterpret_cast<float*>(s.other + D * TM);
  s.st = with_st ? s.Ak + TM * EM_LDA : nullptr;
  (void)nw;
  return s;
}

template <int NRB>
__device__ __forceinline__ void edge_stage_tile(const EdgeTileArgs& p, const EdgeStage& s, int tile, int ts, int R, int tid, int nthreads) {
  constexpr int TM = NRB * 32;
  const int D = p.D;
  for (int idx = tid; idx < TM * FN; idx += nthreads) {
    const int r = idx / FN, k = idx - r * FN;
    s.xs[idx] = r < R ? p.x[(int64_t)(ts + r) * p.ldx + k] : 0.f;
  }
  for (int idx = tid; idx < D * TM; idx += nthreads) {
    const int2 en = p.ell_ent[(size_t)tile * D * TM + idx];
    const bool ok = en.y != -1;
    s.other[idx] = ok ? en.x : -1;
    float* d = s.eaL + idx * 8;
    if (ok) {
      const int eid = en.y & 0x7fffffff;
      const float sgn = en.y < 0 ? -1.f : 1.f;
      const float* e = p.ea + (int64_t)eid * p.ldea;
      d[0] = e[0] * sgn; d[1] = e[1]; d[2] = e[2] * sgn; d[3] = e[3]; d[4] = e[4]; d[5] = e[5];
    } else {
      d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f; d[4] = 0.f; d[5] = 0.f;
    }
  }
}

// One PART of a tile (rows r0 .. r0 + TM - 1 of a tile of XT rows): the x rows of the WHOLE tile (a neighbour may be any of them), the
// part's slice of the ELL table and its edge features.  XT = TM, r0 = 0: edge_stage_tile.
template <int NRB>
__device__ __forceinline__ void edge_stage_part(const EdgeTileArgs& p, const EdgeStage& s, int tile, int XT, int r0, int ts_full, int R_full,
                                                int tid, int nthreads) {
  constexpr int TM = NRB * 32;
  const int D = p.D;
  for (int idx = tid; idx < XT * FN; idx += nthreads) {
    const int r = idx / FN, k = idx - r * FN;
    s.xs[idx] = r < R_full ? p.x[(int64_t)(ts_full + r) * p.ldx + k] : 0.f;
  }
  for (int idx = tid; idx < D * TM; idx += nthreads) {
    const int k = idx / TM, r = idx - k * TM;
    int2 en = make_int2(0, -1);
    if (r0 + r < XT) en = p.ell_ent[((size_t)tile * D + k) * XT + r0 + r];
    const bool ok = en.y != -1;
    s.other[idx] = ok ? en.x : -1;
    float* d = s.eaL + idx * 8;
    if (ok) {
      const int eid = en.y & 0x7fffffff;
      const float sgn = en.y < 0 ? -1.f : 1.f;
      const float* e = p.ea + (int64_t)eid * p.ldea;
      d[0] = e[0] * sgn; d[1] = e[1]; d[2] = e[2] * sgn; d[3] = e[3]; d[4] = e[4]; d[5] = e[5];
    } else {
      d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f; d[4] = 0.f; d[5] = 0.f;
    }
  }
}

// A_k for slot k: [x_i (8) | x_j (8) | ea (6) | one (1: valid slot, bias-gradient column) | 0 ...]
template <int NRB>
__device__ __forceinline__ void edge_build_ak(const EdgeStage& s, int k, int tid, int nthreads, float ones) {
  constexpr int TM = NRB * 32;
  for (int idx = tid; idx < TM * 4; idx += nthreads) {
    const int r = idx >> 2, q = idx & 3;
    const int o = s.other[k * TM + r];
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (o >= 0) {
      if (q == 0) { v0 = *reinterpret_cast<const f32x4*>(s.xs + r * FN); v1 = *reinterpret_cast<const f32x4*>(s.xs + r * FN + 4); }
      else if (q == 1) { v0 = *reinterpret_cast<const f32x4*>(s.xs + o * FN); v1 = *reinterpret_cast<const f32x4*>(s.xs + o * FN + 4); }
      else if (q == 2) {
        v0 = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + r) * 8);
        const f32x4 t = *reinterpret_cast<const f32x4*>(s.eaL + (k * TM + r) * 8 + 4);
        v1 = f32x4{t[0], t[1], ones, 0.f};
      }
    }
    *reinterpret_cast<f32x4*>(s.Ak + r * EM_LDA + q * 8) = v0;
    *reinterpret_cast<f32x4*>(s.Ak + r * EM_LDA + q * 8 + 4) = v1;
  }
}


// dss2_edge16.hip: the edge MLP's first Linear on the bf16 matrix pipe (bf16x6, fp32-accurate), forward and the backward's
// recomputation.  edge16_ok: the shape is covered (h % 32 == 0, h <= 256, 32 / 64 / 96-row tiles, LDS).
bool edge16_ok(int h, int nrb, int D, bool bwd, bool with_u);
int launch_edge16(const EdgeTileArgs& a, int nrb, int grid, bool bwd, hipStream_t s);

}  // namespace dss2
