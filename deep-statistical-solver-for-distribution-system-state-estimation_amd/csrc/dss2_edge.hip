// EdgeAggregation hidden layer (first Linear + ReLU of the edge MLP, aggregated per target
// node), its backward, and the standalone CSR segmented sum (K6).  gfx950.
//
// Work mapping: one wavefront per node row; lanes across the hidden features (feature j on lane
// j % 64, FPL = ceil(h/64) features per lane), the 22 first-layer weights of each owned feature
// live in registers.  Node / edge inputs of a row are wave-uniform, so they are fetched with
// scalar loads and broadcast for free; the only vector memory traffic is the coalesced S row.
#include "dss2_common.hpp"
#include "dss2_edge_tile.hpp"
#include "dss2_weightspace.hpp"

namespace dss2 {


template <int FPL>
__global__ void __launch_bounds__(256) edge_hidden_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ ea, int64_t ldea,
    const float* __restrict__ W1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ ent, float* __restrict__ S,
    int64_t n_nodes, int h) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  float w[FPL][FC], b[FPL];
#pragma unroll
  for (int f = 0; f < FPL; ++f) {
    const int j = lane + 64 * f;
    const bool ok = j < h;
    b[f] = ok ? b1[j] : 0.f;
#pragma unroll
    for (int k = 0; k < FC; ++k) w[f][k] = ok ? W1[(size_t)j * FC + k] : 0.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * wpb + wave; i < n_nodes; i += (int64_t)gridDim.x * wpb) {
    float xi[FN];
#pragma unroll
    for (int k = 0; k < FN; ++k) xi[k] = x[i * ldx + k];
    float P[FPL], acc[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      float s = b[f];
#pragma unroll
      for (int k = 0; k < FN; ++k) s = fmaf(w[f][k], xi[k], s);
      P[f] = s;
      acc[f] = 0.f;
    }
    const int e1 = rowptr[i + 1];
    for (int e = rowptr[i]; e < e1; ++e) {
      const int src = col[e];
      const int en = ent[e];
      const int eid = en & 0x7fffffff;
      const float sgn = en < 0 ? -1.f : 1.f;
      float xs[FN], a[FE];
#pragma unroll
      for (int k = 0; k < FN; ++k) xs[k] = x[(int64_t)src * ldx + k];
#pragma unroll
      for (int k = 0; k < FE; ++k) a[k] = ea[(int64_t)eid * ldea + k];
      a[0] *= sgn;  // reverse edges carry (-c0, c1, -c2, c3, ...)  (networks.py:250-254)
      a[2] *= sgn;
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        float z = P[f];
#pragma unroll
        for (int k = 0; k < FN; ++k) z = fmaf(w[f][FN + k], xs[k], z);
#pragma unroll
        for (int k = 0; k < FE; ++k) z = fmaf(w[f][2 * FN + k], a[k], z);
        acc[f] += relu_nan(z);
      }
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      const int j = lane + 64 * f;
      if (j < h) S[i * h + j] = acc[f];
    }
  }
}

// Backward.  by_source == 0: rows are targets (CSR by target), accumulates dW1 / db1 partials and
// optionally U = sum of dz over incoming edges.  by_source == 1: rows are sources (transposed
// CSR), writes U = sum of dz over outgoing edges only.
template <int FPL>
__global__ void __launch_bounds__(256) edge_hidden_bwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ ea, int64_t ldea,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ dS,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ ent,
    float* __restrict__ slab, float* __restrict__ U, int64_t ldu, int64_t n_nodes, int h, int by_source) {
  __shared__ float red[256 * (FC + 1)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  float w[FPL][FC], b[FPL], dw[FPL][FC], db[FPL];
#pragma unroll
  for (int f = 0; f < FPL; ++f) {
    const int j = lane + 64 * f;
    const bool ok = j < h;
    b[f] = ok ? b1[j] : 0.f;
    db[f] = 0.f;
#pragma unroll
    for (int k = 0; k < FC; ++k) {
      w[f][k] = ok ? W1[(size_t)j * FC + k] : 0.f;
      dw[f][k] = 0.f;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * wpb + wave; i < n_nodes; i += (int64_t)gridDim.x * wpb) {
    float u[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) u[f] = 0.f;
    const int e1 = rowptr[i + 1];
    for (int e = rowptr[i]; e < e1; ++e) {
      const int other = col[e];
      const int en = ent[e];
      const int eid = en & 0x7fffffff;
      const float sgn = en < 0 ? -1.f : 1.f;
      const int64_t tgt = by_source ? (int64_t)other : i;
      const int64_t src = by_source ? i : (int64_t)other;
      float c[FC];
#pragma unroll
      for (int k = 0; k < FN; ++k) c[k] = x[tgt * ldx + k];
#pragma unroll
      for (int k = 0; k < FN; ++k) c[FN + k] = x[src * ldx + k];
#pragma unroll
      for (int k = 0; k < FE; ++k) c[2 * FN + k] = ea[(int64_t)eid * ldea + k];
      c[2 * FN + 0] *= sgn;
      c[2 * FN + 2] *= sgn;
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        float z = b[f];
#pragma unroll
        for (int k = 0; k < FC; ++k) z = fmaf(w[f][k], c[k], z);
        const float g = (j < h) ? dS[tgt * h + j] : 0.f;
        const float dz = relu_open(z) ? g : 0.f;
        u[f] += dz;
        if (!by_source) {
          db[f] += dz;
#pragma unroll
          for (int k = 0; k < FC; ++k) dw[f][k] = fmaf(dz, c[k], dw[f][k]);
        }
      }
    }
    if (U) {
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        if (j < h) U[i * ldu + j] = u[f];
      }
    }
  }
  if (by_source || !slab) return;
  // fixed-order reduction over the workgroup's waves through LDS, then one slab per workgroup
  float* out = slab + (size_t)blockIdx.x * ((size_t)h * FC + h);
  for (int wv = 0; wv < wpb; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        if (j < h) {
#pragma unroll
          for (int k = 0; k < FC; ++k) {
            const float prev = wv ? red[j * (FC + 1) + k] : 0.f;
            red[j * (FC + 1) + k] = prev + dw[f][k];
          }
          const float prevb = wv ? red[j * (FC + 1) + FC] : 0.f;
          red[j * (FC + 1) + FC] = prevb + db[f];
        }
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < h * FC; idx += blockDim.x) {
    const int j = idx / FC, k = idx - j * FC;
    out[idx] = red[j * (FC + 1) + k];
  }
  for (int j = threadIdx.x; j < h; j += blockDim.x) out[(size_t)h * FC + j] = red[j * (FC + 1) + FC];
}


// ------------------------------------------------------------------------------------------
// EdgeAggregation with node features of ANY width (MultiMPN / MaskEmbdMultiMPN interleave it with TAGConv layers, so
// its input is the hidden activation, networks.py:486-498).  The first Linear acts on [x_i | x_j | ea]:
//     z_e = W1a x_i + W1b x_j + W1c ea_e + b1,
// so the two node products are ONE tile GEMM per node, AB = X [W1a; W1b]^T  ([N, 2 hid], dss2_gemm_prop), and only the
// per-edge combination lives here: S[i] = sum_{e -> i} relu(A[i] + B[src(e)] + W1c ea'_e + b1)   (row per wave, lanes over
// the hidden features, W1c and b1 in registers; fe <= 8 with up to 256 hidden units per launch, <= 16 with 128, <= 32 with 64).  Backward: dz_e = dS[i] (z_e > 0); by target it writes
// dAB[i, :hid] = sum dz and the slabs of dW1c / db1, by source dAB[j, hid:] = sum over the edges leaving j.
// ------------------------------------------------------------------------------------------
constexpr int EC_MAXFE = 32;      // widest edge feature vector (template MAXFE = 8 / 16 / 32 holds a unit's edge-feature weights in registers)

struct EdgeCombineArgs {
  const float* AB; int64_t ldab; const float* ea; int64_t ldea; const float* W1c; int64_t ldw; const float* b1;
  const float* dS; const int32_t* rowptr; const int32_t* col; const int32_t* ent;
  float* S; float* dAB; float* slab; int64_t n_nodes; int h, fe, by_source;
  int hfull, c0;      // the launch covers hidden units c0 .. c0 + h - 1 of hfull (wider layers run as several launches of <= 256 units)
};

template <int FPL, bool BWD, int MAXFE>
__global__ void __launch_bounds__(256) edge_combine_kernel(const EdgeCombineArgs p) {
  constexpr int EC_MAXFE = MAXFE;      // (shadows the file-scope bound: this instantiation's register / LDS width)
  __shared__ float red[BWD ? 64 * FPL * (EC_MAXFE + 1) : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  const int h = p.h, fe = p.fe, hf = p.hfull, c0 = p.c0;
  float w[FPL][EC_MAXFE], b[FPL], dw[BWD ? FPL : 1][EC_MAXFE], db[BWD ? FPL : 1];
#pragma unroll
  for (int f = 0; f < FPL; ++f) {
    const int j = lane + 64 * f;
    b[f] = j < h ? p.b1[c0 + j] : 0.f;
    if (BWD) db[f] = 0.f;
#pragma unroll
    for (int k = 0; k < EC_MAXFE; ++k) {
      w[f][k] = (j < h && k < fe) ? p.W1c[(size_t)(c0 + j) * p.ldw + k] : 0.f;
      if (BWD) dw[f][k] = 0.f;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * wpb + wave; i < p.n_nodes; i += (int64_t)gridDim.x * wpb) {
    float own[FPL], g[FPL], acc[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      const int j = lane + 64 * f;
      // rows are targets (own = A part) or, in the by-source pass, sources (own = B part)
      own[f] = j < h ? p.AB[i * p.ldab + (BWD && p.by_source ? hf : 0) + c0 + j] : 0.f;
      g[f] = (BWD && !p.by_source && j < h) ? p.dS[i * hf + c0 + j] : 0.f;
      acc[f] = 0.f;
    }
    const int e1 = p.rowptr[i + 1];
    for (int e = p.rowptr[i]; e < e1; ++e) {
      const int64_t other = p.col[e];
      const int en = p.ent[e];
      const int eid = en & 0x7fffffff;
      const float sgn = en < 0 ? -1.f : 1.f;
      float a[EC_MAXFE];
#pragma unroll
      for (int k = 0; k < EC_MAXFE; ++k) a[k] = k < fe ? p.ea[(int64_t)eid * p.ldea + k] : 0.f;
      a[0] *= sgn;          // reverse edges of the reference's MPN doubling carry (-c0, c1, -c2, ...); the Multi* variants
      a[2] *= sgn;          // double WITHOUT sign flips (networks.py:519-523): their structure sets no flip flag
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        const float oth = j < h ? p.AB[other * p.ldab + (BWD && p.by_source ? 0 : hf) + c0 + j] : 0.f;
        float z = (own[f] + oth) + b[f];
#pragma unroll
        for (int k = 0; k < EC_MAXFE; ++k) z = fmaf(w[f][k], a[k], z);
        if (!BWD) {
          acc[f] += relu_nan(z);
        } else {
          const float gg = p.by_source ? ((j < h) ? p.dS[other * hf + c0 + j] : 0.f) : g[f];
          const float dz = relu_open(z) ? gg : 0.f;
          acc[f] += dz;
          if (!p.by_source) {
            db[f] += dz;
#pragma unroll
            for (int k = 0; k < EC_MAXFE; ++k) dw[f][k] = fmaf(dz, a[k], dw[f][k]);
          }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      const int j = lane + 64 * f;
      if (j < h) {
        if (!BWD) p.S[i * hf + c0 + j] = acc[f];
        else p.dAB[i * p.ldab + (p.by_source ? hf : 0) + c0 + j] = acc[f];
      }
    }
  }
  if (!BWD) return;
  if (p.by_source || !p.slab) return;
  float* out = p.slab + (size_t)blockIdx.x * ((size_t)hf * fe + hf) + (size_t)c0 * fe;      // [hfull][fe] dW1c, then [hfull] db1
  for (int wv = 0; wv < wpb; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        if (j < h) {
#pragma unroll
          for (int k = 0; k < EC_MAXFE; ++k) red[j * (EC_MAXFE + 1) + k] = (wv ? red[j * (EC_MAXFE + 1) + k] : 0.f) + dw[f][k];
          red[j * (EC_MAXFE + 1) + EC_MAXFE] = (wv ? red[j * (EC_MAXFE + 1) + EC_MAXFE] : 0.f) + db[f];
        }
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < h * fe; idx += blockDim.x) {
    const int j = idx / fe, k = idx - j * fe;
    out[idx] = red[j * (EC_MAXFE + 1) + k];
  }
  for (int j = threadIdx.x; j < h; j += blockDim.x) out[(size_t)(hf - c0) * fe + c0 + j] = red[j * (EC_MAXFE + 1) + EC_MAXFE];
}

// ------------------------------------------------------------------------------------------
// Tile-based variants (used when the topology provides per-tile ELL slices with edge ids).
// One workgroup = one tile of whole graphs (<= TMAX rows).  Staged in LDS: the tile's x rows, an ELL
// slice {local other-node, stored edge id | flip} per (slot, row), and the gathered, sign-corrected
// edge_attr row of every slot.  All per-row / per-edge operands are then LDS broadcasts instead of
// chains of dependent scalar global loads (the row-per-wave kernels above are latency bound on
// rowptr -> col/ent -> x/ea).
// ------------------------------------------------------------------------------------------

// HALF (round 6; FPL = 1, h <= 32: the reference driver's dim_hid on 70-bus grids, where this kernel serves the inner blocks' backward with U):
// a lane is a hidden unit, so at h <= 32 half of every wave idled -- here the two halves of a wave take two ROWS (lane & 31 = hidden unit,
// lane >> 5 = which row of the pair); a slot that is empty in one row of the pair is computed on row 0's operands and masked.  Forward and
// backward form z with the same fma chain as before: the recomputed gates are the forward's.
template <int FPL, bool BWD, bool HALF = false>
__global__ void __launch_bounds__(256) edge_tile_kernel(const EdgeTileArgs p) {
  static_assert(!HALF || FPL == 1, "HALF: one hidden unit per lane");
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int TM = p.TM, D = p.D;
  float* xs = esm;                                   // [TM][8]
  float* eaL = xs + TM * FN;                         // [D*TM][8]  (6 used; 8 keeps 16-byte rows)
  int* other = reinterpret_cast<int*>(eaL + D * TM * 8);   // [D*TM] local other node, -1 = empty slot
  __shared__ float red[BWD ? 256 * (FC + 1) : 1];
  const int tid = threadIdx.x, lane = HALF ? (tid & 31) : (tid & 63);      // (HALF: the hidden unit)
  const int sub = HALF ? ((tid >> 5) & 1) : 0;                              // (HALF: which row of the wave's pair)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float w[FPL][FC], b[FPL];
#pragma unroll
  for (int f = 0; f < FPL; ++f) {
    const int j = lane + 64 * f;
    const bool ok = j < p.h;
    b[f] = ok ? p.b1[j] : 0.f;
#pragma unroll
    for (int k = 0; k < FC; ++k) w[f][k] = ok ? p.W1[(size_t)j * FC + k] : 0.f;
  }
  float dw[BWD ? FPL : 1][FC], db[BWD ? FPL : 1];
  if (BWD) {
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      db[f] = 0.f;
#pragma unroll
      for (int k = 0; k < FC; ++k) dw[f][k] = 0.f;
    }
  }
  // persistent over tiles: the weight-gradient partials stay in registers, one slab per workgroup
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;

  for (int idx = tid; idx < TM * FN; idx += 256) {
    const int r = idx / FN, k = idx - r * FN;
    xs[idx] = r < R ? p.x[(int64_t)(ts + r) * p.ldx + k] : 0.f;
  }
  for (int idx = tid; idx < D * TM; idx += 256) {
    const int2 en = p.ell_ent[(size_t)tile * D * TM + idx];
    const bool ok = en.y != -1;
    other[idx] = ok ? en.x : -1;
    if (ok) {
      const int eid = en.y & 0x7fffffff;
      const float sgn = en.y < 0 ? -1.f : 1.f;
      const float* e = p.ea + (int64_t)eid * p.ldea;
      float* d = eaL + idx * 8;
      d[0] = e[0] * sgn; d[1] = e[1]; d[2] = e[2] * sgn; d[3] = e[3]; d[4] = e[4]; d[5] = e[5];
    }
  }
  __syncthreads();

  for (int r0 = HALF ? 2 * wave : wave; r0 < R; r0 += HALF ? 8 : 4) {
    const bool rv = !HALF || r0 + sub < R;            // (HALF: the pair's second row may lie beyond the tile)
    const int r = rv ? r0 + sub : r0;
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + r * FN), xb = *reinterpret_cast<const f32x4*>(xs + r * FN + 4);
    float accv[FPL], grow[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      accv[f] = 0.f;
      const int j = lane + 64 * f;
      grow[f] = (BWD && !p.by_source && j < p.h && rv) ? p.dS[(int64_t)(ts + r) * p.h + j] : 0.f;   // one load per row
    }
    for (int k = 0; k < D; ++k) {
      const int o_ = other[k * TM + r];
      const bool valid = rv && o_ >= 0;
      if (HALF ? __builtin_amdgcn_ballot_w64(valid) == 0 : o_ < 0) continue;      // wave-uniform
      const int o = (HALF && !valid) ? 0 : o_;        // (an empty slot of one row of the pair: any staged row; masked below)
      const f32x4 oa = *reinterpret_cast<const f32x4*>(xs + o * FN), ob = *reinterpret_cast<const f32x4*>(xs + o * FN + 4);
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(eaL + (k * TM + r) * 8), e1 = *reinterpret_cast<const f32x4*>(eaL + (k * TM + r) * 8 + 4);
      // c = [x_tgt | x_src | ea]; rows are targets (by_source == 0) or sources (by_source == 1)
      float c[FC];
      const bool bs = BWD && p.by_source;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = bs ? oa[q] : xa[q];
        c[4 + q] = bs ? ob[q] : xb[q];
        c[FN + q] = bs ? xa[q] : oa[q];
        c[FN + 4 + q] = bs ? xb[q] : ob[q];
      }
      // (HALF: an empty slot's edge features were never staged -- whatever LDS held, possibly not finite: 0 * that would poison dW1)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[2 * FN + q] = (!HALF || valid) ? e0[q] : 0.f;
      c[2 * FN + 4] = (!HALF || valid) ? e1[0] : 0.f;
      c[2 * FN + 5] = (!HALF || valid) ? e1[1] : 0.f;
      const int64_t tg = bs ? (int64_t)(ts + o) : (int64_t)(ts + r);
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        float z = b[f];
#pragma unroll
        for (int q = 0; q < FC; ++q) z = fmaf(w[f][q], c[q], z);
        if (!BWD) {
          accv[f] += (!HALF || valid) ? relu_nan(z) : 0.f;
        } else {
          const float g = p.by_source ? ((j < p.h && (!HALF || valid)) ? p.dS[tg * p.h + j] : 0.f) : grow[f];
          const float dz = (relu_open(z) && (!HALF || valid)) ? g : 0.f;
          accv[f] += dz;
          if (!p.by_source) {
            db[f] += dz;
#pragma unroll
            for (int q = 0; q < FC; ++q) dw[f][q] = fmaf(dz, c[q], dw[f][q]);
          }
        }
      }
    }
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      const int j = lane + 64 * f;
      if (j < p.h && rv) {
        if (!BWD) p.S[(int64_t)(ts + r) * p.h + j] = accv[f];
        else if (p.U) p.U[(int64_t)(ts + r) * p.ldu + j] = accv[f];
      }
    }
  }
  __syncthreads();   // all waves are done with the staged tile before the next one is written
  }  // persistent tile loop
  if (!BWD) return;
  if (p.by_source || !p.slab) return;
  float* out = p.slab + (size_t)blockIdx.x * ((size_t)p.h * FC + p.h);
  for (int wv = 0; wv < (HALF ? 8 : 4); ++wv) {      // fixed order: wave 0 .. 3 (HALF: wave 0's first rows, its second rows, wave 1's ...)
    if (HALF ? (wave == (wv >> 1) && sub == (wv & 1)) : wave == wv) {
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        if (j < p.h) {
#pragma unroll
          for (int k = 0; k < FC; ++k) red[j * (FC + 1) + k] = (wv ? red[j * (FC + 1) + k] : 0.f) + dw[f][k];
          red[j * (FC + 1) + FC] = (wv ? red[j * (FC + 1) + FC] : 0.f) + db[f];
        }
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < p.h * FC; idx += 256) {
    const int j = idx / FC, k = idx - j * FC;
    out[idx] = red[j * (FC + 1) + k];
  }
  for (int j = tid; j < p.h; j += 256) out[(size_t)p.h * FC + j] = red[j * (FC + 1) + FC];
}

// K6: out[i, :] = sum_{e in [rowptr[i], rowptr[i+1])} msg[ent[e], :].  A group of h/4 lanes owns
// one row and moves float4; several rows per wave when h < 256.  HBM-bound by construction:
// every message row is read exactly once with 16-B lanes, every output row written once.
__global__ void __launch_bounds__(256) segment_sum_kernel(
    const float* __restrict__ msg, int64_t ldm, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ ent, float* __restrict__ out, int64_t ldo, int64_t n_rows, int h) {
  const int nv = h >> 2;                       // float4 per row (8, 16, 32 or 64)
  const int gpb = blockDim.x / nv;             // row groups per block
  const int g = threadIdx.x / nv;
  const int cv = threadIdx.x - g * nv;
  for (int64_t i = (int64_t)blockIdx.x * gpb + g; i < n_rows; i += (int64_t)gridDim.x * gpb) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    int e = rowptr[i];
    const int e1 = rowptr[i + 1];
    for (; e + 1 < e1; e += 2) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(msg + (int64_t)ent[e] * ldm + 4 * cv);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(msg + (int64_t)ent[e + 1] * ldm + 4 * cv);
      s0 += v0;
      s1 += v1;
    }
    if (e < e1) s0 += *reinterpret_cast<const f32x4*>(msg + (int64_t)ent[e] * ldm + 4 * cv);
    *reinterpret_cast<f32x4*>(out + i * ldo + 4 * cv) = s0 + s1;
  }
}

// Any feature width (MessagePassing.propagate with a user-defined message): one thread per output element, the
// entries of a row are summed in CSR order (ascending directed edge id, the order index_add_ visits them).
__global__ void __launch_bounds__(256) segment_sum_generic_kernel(
    const float* __restrict__ msg, int64_t ldm, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ent,
    float* __restrict__ out, int64_t ldo, int64_t n_rows, int h) {
  const int64_t total = n_rows * h;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / h;
    const int c = (int)(idx - i * h);
    float s = 0.f;
    const int e1 = rowptr[i + 1];
    for (int e = rowptr[i]; e < e1; ++e) s += msg[(int64_t)ent[e] * ldm + c];
    out[i * ldo + c] = s;
  }
}

// out[r, :] = src[idx[r], :]: the gather half of MessagePassing.propagate (x_j = x[edge_index[0]], x_i = x[edge_index[1]])
// and the backward of the segmented sum.  VEC: 16-byte lanes (h % 4 == 0, aligned operands).
template <bool VEC>
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t ldo, int64_t n_rows, int h) {
  const int per = VEC ? h >> 2 : h;
  const int64_t total = n_rows * per;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / per;
    const int c = (int)(t - r * per);
    const int64_t s = idx[r];
    if (VEC) *reinterpret_cast<f32x4*>(out + r * ldo + 4 * c) = *reinterpret_cast<const f32x4*>(src + s * lds + 4 * c);
    else out[r * ldo + c] = src[s * lds + c];
  }
}

// ------------------------------------------------------------------------------------------
// One hop of the TAGConv propagation in GLOBAL memory, for graphs whose connected components exceed the LDS-resident
// tiles (> 192 buses): out[i,:] = epi( add[i,:] + sum_{e in row i} w[e] * T[col[e],:] ), rows sorted as in the CSR.
// A group of h/4 lanes owns a row and moves float4 (h % 4 == 0); HBM / L2-bound gather.  Epilogue (last hop of a layer
// only): + bias, * in-kernel dropout mask, ReLU, * (relu_src > 0), + add_src -- the order of dss2_gemm_prop's epilogue.
// ------------------------------------------------------------------------------------------
struct CsrAxpyArgs {
  const int32_t* rowptr; const int32_t* col; const float* w;
  const float* T; int64_t ldt; const float* add; int64_t ld_add; float* out; int64_t ldo;
  const float* bias; const float* relu_src; int64_t ld_relu; const float* add_src; int64_t ld_src;
  const uint64_t* drop_state; uint32_t drop_thr; float drop_scale; int32_t drop_id; int32_t relu;
  int64_t n_rows; int h;
};
template <int VEC>
__global__ void __launch_bounds__(256) csr_axpy_kernel(const CsrAxpyArgs p) {
  using V = float __attribute__((ext_vector_type(VEC)));
  const int nv = (p.h + VEC - 1) / VEC;
  const int64_t total = p.n_rows * nv;
  const uint64_t seed = p.drop_id ? p.drop_state[0] : 0, off = p.drop_id ? p.drop_state[1] : 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / nv;
    const int c = (int)(t - i * nv) * VEC;
    V s;
    if (p.add) s = *reinterpret_cast<const V*>(p.add + i * p.ld_add + c);
    else s = V(0.f);
    const int e1 = p.rowptr[i + 1];
    for (int e = p.rowptr[i]; e < e1; ++e) {
      const V v = *reinterpret_cast<const V*>(p.T + (int64_t)p.col[e] * p.ldt + c);
      const float wv = p.w[e];
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] = fmaf(wv, v[q], s[q]);
    }
    if (p.bias) s += *reinterpret_cast<const V*>(p.bias + c);
    if (p.drop_id) {     // the mask is defined per group of four columns (dropout_mult4)
      const f32x4 m = dropout_mult4(seed, off, (uint32_t)p.drop_id, (uint32_t)i, (uint32_t)(c >> 2), p.drop_thr, p.drop_scale);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] *= m[(c + q) & 3];
    }
    if (p.relu) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] = relu_nan(s[q]);
    }
    if (p.relu_src) {
      const V r = *reinterpret_cast<const V*>(p.relu_src + i * p.ld_relu + c);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] = relu_open(r[q]) ? s[q] : 0.f;
    }
    if (p.add_src) s += *reinterpret_cast<const V*>(p.add_src + i * p.ld_src + c);
    *reinterpret_cast<V*>(p.out + i * p.ldo + c) = s;
  }
}

// out[j] = sum_k slab[k][j].  blockDim = (64, 4): threadIdx.y owns a contiguous quarter of the slabs
// and keeps 8 independent loads in flight; partial sums are combined in a fixed order, so the
// result is bitwise reproducible (no atomics).
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ slab, int n_slabs,
                                                           int64_t stride, float* __restrict__ out, int64_t len) {
  __shared__ float part[4][64];
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.y;
  const int per = (n_slabs + 3) >> 2;
  const int k0 = q * per, k1 = min(n_slabs, k0 + per);
  float s = 0.f;
  if (j < len) {
    const float* p = slab + j;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; k < k1; ++k) s += p[(size_t)k * stride];
  }
  part[q][threadIdx.x] = s;
  __syncthreads();
  if (q == 0 && j < len) out[j] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// (ReduceTable, reduce_slabs_body: dss2_weightspace.hpp)
__global__ void __launch_bounds__(256) reduce_slabs_multi_v4_kernel(const ReduceTable tab) {
  __shared__ f32x4 part[16][16];
  reduce_slabs_body(tab.d[blockIdx.y], ((tab.scalar_mask >> blockIdx.y) & 1u) != 0, (int)blockIdx.x, part);
}

}  // namespace dss2

using namespace dss2;

static int dss2_edge_hidden_fwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fn, int fe, void* stream);
extern "C" int dss2_edge_hidden_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fn, int fe, void* stream) {
  DSS2_RECORD([x, ldx, ea, ldea, W1, b1, rowptr, col, ent, S, n_nodes, h, fn, fe](void* s_) { return dss2_edge_hidden_fwd_launch(x, ldx, ea, ldea, W1, b1, rowptr, col, ent, S, n_nodes, h, fn, fe, s_); });
  return dss2_edge_hidden_fwd_launch(x, ldx, ea, ldea, W1, b1, rowptr, col, ent, S, n_nodes, h, fn, fe, stream);
}
static int dss2_edge_hidden_fwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fn, int fe, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_hidden_fwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256) { set_error("edge_hidden_fwd: h=%d unsupported (1..256)", h); return 2; }
  if (n_nodes <= 0) return 0;
  const int wpb = 4;
  int64_t blocks = (n_nodes + wpb - 1) / wpb;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipStream_t s = as_stream(stream);
  const int fpl = (h + 63) / 64;
#define L(FPL) hipLaunchKernelGGL(edge_hidden_fwd_kernel<FPL>, dim3((unsigned)blocks), dim3(64 * wpb), 0, s, x, ldx, ea, ldea, W1, b1, rowptr, col, ent, S, n_nodes, h)
  if (fpl == 1) L(1); else if (fpl == 2) L(2); else if (fpl == 3) L(3); else L(4);
#undef L
  return check_launch("edge_hidden_fwd");
}

static int dss2_edge_hidden_bwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* slab, int n_slabs, float* U, int64_t ldu, int64_t n_nodes, int h, int fn, int fe, int by_source, void* stream);
extern "C" int dss2_edge_hidden_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* slab, int n_slabs, float* U, int64_t ldu, int64_t n_nodes, int h, int fn, int fe, int by_source, void* stream) {
  DSS2_RECORD([x, ldx, ea, ldea, W1, b1, dS, rowptr, col, ent, slab, n_slabs, U, ldu, n_nodes, h, fn, fe, by_source](void* s_) { return dss2_edge_hidden_bwd_launch(x, ldx, ea, ldea, W1, b1, dS, rowptr, col, ent, slab, n_slabs, U, ldu, n_nodes, h, fn, fe, by_source, s_); });
  return dss2_edge_hidden_bwd_launch(x, ldx, ea, ldea, W1, b1, dS, rowptr, col, ent, slab, n_slabs, U, ldu, n_nodes, h, fn, fe, by_source, stream);
}
static int dss2_edge_hidden_bwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* slab, int n_slabs, float* U, int64_t ldu, int64_t n_nodes, int h, int fn, int fe, int by_source, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_hidden_bwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256) { set_error("edge_hidden_bwd: h=%d unsupported (1..256)", h); return 2; }
  if (n_slabs <= 0) { set_error("edge_hidden_bwd: n_slabs must be > 0"); return 2; }
  if (!by_source && !slab) { set_error("edge_hidden_bwd: slab is NULL"); return 2; }
  if (by_source && !U) { set_error("edge_hidden_bwd: by_source needs U"); return 2; }
  hipStream_t s = as_stream(stream);
  const int fpl = (h + 63) / 64;
#define L(FPL) hipLaunchKernelGGL(edge_hidden_bwd_kernel<FPL>, dim3(n_slabs), dim3(256), 0, s, x, ldx, ea, ldea, W1, b1, dS, rowptr, col, ent, slab, U, ldu, n_nodes, h, by_source)
  if (fpl == 1) L(1); else if (fpl == 2) L(2); else if (fpl == 3) L(3); else L(4);
#undef L
  return check_launch("edge_hidden_bwd");
}

// ------------------------------------------------------------------------------------------
// Edge MLP on the matrix pipe.  For ELL slot k the inputs of the k-th incoming edge of every row of the tile
// form a dense [TM x 24] matrix A_k = [x_i | x_j | edge_attr | 0]; Z_k = A_k W1^T is three k-steps of the
// 32x32x2 MFMA per row block, and because row r of A_k IS target row r, the aggregation is a masked
// accumulation of relu(Z_k + b1) in the accumulator layout -- no gather, no scatter.  Backward recomputes
// Z_k the same way, gates dS with it, and forms dW1 += dZ_k^T A_k as a second MFMA whose extra A_k column of
// ones yields db1.  (The VALU kernels above need ~90 / ~200 vector instructions per edge and lane; this
// path needs ~25 MFMAs per slot and row block.)
// ------------------------------------------------------------------------------------------
template <int NRB>
__global__ void __launch_bounds__(512) edge_mfma_fwd_kernel(const EdgeTileArgs p) {
  constexpr int TM = NRB * 32;
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);      // one 32-column group of the hidden layer per wave
  const int c32 = lane & 31, half = lane >> 5;
  const int D = p.D;
  const EdgeStage s = edge_stage_ptrs<NRB>(esm, D, nthreads >> 6, false);
  // W1 fragments and bias of this wave's columns stay in registers for the whole kernel
  const int j = cg * 32 + c32;
  const float b1v = p.b1[j];
  f32x4 bf[3];
#pragma unroll
  for (int kc = 0; kc < 3; ++kc)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kc * 8 + half * 4 + q;
      bf[kc][q] = k < FC ? p.W1[(size_t)j * FC + k] : 0.f;
    }
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    edge_stage_tile<NRB>(p, s, tile, ts, R, tid, nthreads);
    __syncthreads();
    f32x16 Sacc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) Sacc[rb][r] = 0.f;
    for (int k = 0; k < D; ++k) {
      edge_build_ak<NRB>(s, k, tid, nthreads, 0.f);
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(s.Ak + (rb * 32 + c32) * EM_LDA + kc * 8 + half * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bf[kc][q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool valid = s.other[k * TM + rb * 32 + acc_row(r, half)] >= 0;
          Sacc[rb][r] += valid ? relu_nan(acc[r] + b1v) : 0.f;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row(r, half);
        if (row < R) p.S[(int64_t)(ts + row) * p.h + j] = Sacc[rb][r];
      }
  }
}

// backward by target: dW1, db1 (slab per workgroup) and optionally U[row] = sum over incoming edges of dZ
template <int NRB, bool WITH_U>
__global__ void __launch_bounds__(512) edge_mfma_bwd_kernel(const EdgeTileArgs p) {
  constexpr int TM = NRB * 32;
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c32 = lane & 31, half = lane >> 5;
  const int D = p.D;
  const EdgeStage s = edge_stage_ptrs<NRB>(esm, D, nthreads >> 6, true);
  float* st = s.st + cg * (TM * 32);        // wave-private dZ tile [TM][32]
  const int j = cg * 32 + c32;
  const float b1v = p.b1[j];
  f32x4 bf[3];
#pragma unroll
  for (int kc = 0; kc < 3; ++kc)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kc * 8 + half * 4 + q;
      bf[kc][q] = k < FC ? p.W1[(size_t)j * FC + k] : 0.f;
    }
  f32x16 dWacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dWacc[r] = 0.f;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int ts = p.tile_start[tile];
    const int R = p.tile_start[tile + 1] - ts;
    edge_stage_tile<NRB>(p, s, tile, ts, R, tid, nthreads);
    f32x16 gS[NRB], Uacc[WITH_U ? NRB : 1];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + acc_row(r, half);
        gS[rb][r] = row < R ? p.dS[(int64_t)(ts + row) * p.h + j] : 0.f;
        if (WITH_U) Uacc[rb][r] = 0.f;
      }
    __syncthreads();
    for (int k = 0; k < D; ++k) {
      edge_build_ak<NRB>(s, k, tid, nthreads, 1.f);
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(s.Ak + (rb * 32 + c32) * EM_LDA + kc * 8 + half * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bf[kc][q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb * 32 + acc_row(r, half);
          const bool on = s.other[k * TM + row] >= 0 && relu_open(acc[r] + b1v);
          const float dz = on ? gS[rb][r] : 0.f;
          if (WITH_U) Uacc[rb][r] += dz;
          st[row * 32 + c32] = dz;
        }
      }
      wave_lds_sync();
      // dW1[o][i] += sum_rows dZ[row][o] * A_k[row][i]   (column 22 of A_k is 1 on valid slots: db1)
      const float* ap = st + half * 32 + c32;
      const float* bp2 = s.Ak + half * EM_LDA + c32;
      // a fresh accumulator per slot, added to the running total afterwards: keeps the fp32 accumulation chains as
      // short as the VALU kernel's (one long chain over all tiles of a workgroup lost ~3e-5 of the largest entry)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        f32x16 part;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
        for (int n2 = rb * 16; n2 < rb * 16 + 16; ++n2)
          part = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[n2 * 64], bp2[n2 * 2 * EM_LDA], part, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) dWacc[r] += part[r];
      }
      __syncthreads();   // everyone is done with A_k (and this wave with its dZ tile)
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

static size_t edge_mfma_lds(int TM, int D, int nw, bool bwd) {
  size_t b = ((size_t)TM * FN + (size_t)D * TM * 8) * 4 + (size_t)D * TM * 4 + (size_t)TM * EM_LDA * 4;
  if (bwd) b += (size_t)nw * TM * 32 * 4;
  return b;
}

static bool edge_mfma_ok(int h, int nrb, int D, bool bwd = false) {
  // Default: both passes on the matrix pipe (forward 26 -> 24 us, backward 55 -> 40 us at C2), so that the forward's
  // ReLU gates and the gates the backward recomputes come from the same arithmetic.  DSS2_EDGE_MFMA=0 (or _FWD / _BWD)
  // selects the VALU tile kernels, which also serve shapes the MFMA kernels do not cover (h % 32 != 0, tall tiles).
  // (read per call: lets a test switch paths inside one process)
  const char* env = getenv("DSS2_EDGE_MFMA");
  const char* env2 = getenv(bwd ? "DSS2_EDGE_MFMA_BWD" : "DSS2_EDGE_MFMA_FWD");
  const int enabled = env2 ? atoi(env2) : (env ? atoi(env) : 1);
  if (!enabled || (h & 31) || h > 256 || !(nrb == 1 || nrb == 2 || nrb == 3)) return false;
  return edge_mfma_lds(nrb * 32, D, h >> 5, true) <= (size_t)kMaxLdsBytes;
}

template <int NRB>
static int launch_edge_mfma(const EdgeTileArgs& a, int grid, bool bwd, hipStream_t s) {
  const int nw = a.h >> 5;                  // one wave per 32-column group of the hidden layer (<= 8)
  const size_t lds = edge_mfma_lds(NRB * 32, a.D, nw, bwd);
  if (bwd && a.U) {
    if constexpr (NRB <= 2) {
      static std::atomic<uint32_t> lds_done{0};
      auto kern = edge_mfma_bwd_kernel<NRB, true>;
      if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge_mfma_bwd")) return 1;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
    } else {
      set_error("edge_mfma_bwd with U: tiles above 64 rows are served by the VALU kernel"); return 2;
    }
  } else if (bwd) {
    static std::atomic<uint32_t> lds_done{0};
    auto kern = edge_mfma_bwd_kernel<NRB, false>;
    if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge_mfma_bwd")) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
  } else {
    static std::atomic<uint32_t> lds_done{0};
    auto kern = edge_mfma_fwd_kernel<NRB>;
    if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "edge_mfma_fwd")) return 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, s, a);
  }
  return check_launch(bwd ? "edge_mfma_bwd" : "edge_mfma_fwd");
}

static int dispatch_edge_mfma(const EdgeTileArgs& a, int nrb, int grid, bool bwd, hipStream_t s) {
  switch (nrb) {
    case 1: return launch_edge_mfma<1>(a, grid, bwd, s);
    case 2: return launch_edge_mfma<2>(a, grid, bwd, s);
    default: return launch_edge_mfma<3>(a, grid, bwd, s);
  }
}

template <bool BWD>
static int launch_edge_tile(const EdgeTileArgs& a, int grid, hipStream_t s) {
  const size_t lds = ((size_t)a.TM * FN + (size_t)a.D * a.TM * 8) * 4 + (size_t)a.D * a.TM * 4;
  const int fpl = (a.h + 63) / 64;
#define L(FPL) hipLaunchKernelGGL((edge_tile_kernel<FPL, BWD>), dim3(grid), dim3(256), lds, s, a)
  static const int half_on = [] { const char* e = getenv("DSS2_EDGE_TILE_HALF"); return e ? atoi(e) : 1; }();      // 0: a wave per row at every width
  if (a.h <= 32 && half_on) hipLaunchKernelGGL((edge_tile_kernel<1, BWD, true>), dim3(grid), dim3(256), lds, s, a);
  else if (fpl == 1) L(1); else if (fpl == 2) L(2); else if (fpl == 3) L(3); else L(4);
#undef L
  return check_launch(BWD ? "edge_tile_bwd" : "edge_tile_fwd");
}

extern "C" int dss2_edge_tile_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                                  const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width,
                                  int nrb, int ntiles, float* S, int h, int fn, int fe, void* stream) {
  return dss2_edge_tile_fwd_paired(x, ldx, ea, ldea, W1, b1, tile_start, ell_ent, ell_width, nrb, ntiles, S, h, fn, fe, 0, stream);
}

static int dss2_edge_tile_fwd_paired_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* S, int h, int fn, int fe, int bwd_with_u, void* stream);
extern "C" int dss2_edge_tile_fwd_paired(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* S, int h, int fn, int fe, int bwd_with_u, void* stream) {
  DSS2_RECORD([x, ldx, ea, ldea, W1, b1, tile_start, ell_ent, ell_width, nrb, ntiles, S, h, fn, fe, bwd_with_u](void* s_) { return dss2_edge_tile_fwd_paired_launch(x, ldx, ea, ldea, W1, b1, tile_start, ell_ent, ell_width, nrb, ntiles, S, h, fn, fe, bwd_with_u, s_); });
  return dss2_edge_tile_fwd_paired_launch(x, ldx, ea, ldea, W1, b1, tile_start, ell_ent, ell_width, nrb, ntiles, S, h, fn, fe, bwd_with_u, stream);
}
static int dss2_edge_tile_fwd_paired_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* S, int h, int fn, int fe, int bwd_with_u, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_tile_fwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256 || ell_width <= 0 || ell_width > 32) { set_error("edge_tile_fwd: bad h=%d or ell_width=%d", h, ell_width); return 2; }
  if (ntiles <= 0) return 0;
  EdgeTileArgs a{x, ldx, ea, ldea, W1, b1, nullptr, tile_start, reinterpret_cast<const int2*>(ell_ent), S, nullptr, nullptr, 0,
                 h, ell_width, nrb * 32, 0, ntiles};
  // The backward recomputes the ReLU gates, so forward and backward must run the same arithmetic.  dss2_edge_tile_bwd with U on 96-row
  // tiles runs the VALU tile kernel (neither matrix-pipe backward is built for it): a caller that announces such a backward gets the
  // VALU tile forward.
  const bool valu_pair = bwd_with_u && nrb >= 3;
  const int part_nrb = nrb == 4 ? 2 : 3;      // 128- / 192-row tiles: the bf16x6 kernels on two parts of 64 / 96 rows (both passes or neither)
  if ((nrb == 4 || nrb == 6) && !valu_pair && edge_mfma_ok(h, part_nrb, ell_width) && edge16_ok(h, nrb, ell_width, false, false) && edge16_ok(h, nrb, ell_width, true, false))
    return launch_edge16(a, nrb, ntiles, false, as_stream(stream));
  if (edge_mfma_ok(h, nrb, ell_width) && !valu_pair) {
    // first Linear as bf16x6 on the bf16 matrix pipe (dss2_edge16.hip; DSS2_EDGE_BF16=0: the fp32 MFMA form below)
    if (edge16_ok(h, nrb, ell_width, false, false)) return launch_edge16(a, nrb, ntiles, false, as_stream(stream));
    return dispatch_edge_mfma(a, nrb, ntiles, false, as_stream(stream));
  }
  return launch_edge_tile<false>(a, ntiles, as_stream(stream));
}

static int dss2_edge_tile_bwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* slab, int n_slabs, float* U, int64_t ldu, int h, int fn, int fe, int by_source, void* stream);
extern "C" int dss2_edge_tile_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* slab, int n_slabs, float* U, int64_t ldu, int h, int fn, int fe, int by_source, void* stream) {
  DSS2_RECORD([x, ldx, ea, ldea, W1, b1, dS, tile_start, ell_ent, ell_width, nrb, ntiles, slab, n_slabs, U, ldu, h, fn, fe, by_source](void* s_) { return dss2_edge_tile_bwd_launch(x, ldx, ea, ldea, W1, b1, dS, tile_start, ell_ent, ell_width, nrb, ntiles, slab, n_slabs, U, ldu, h, fn, fe, by_source, s_); });
  return dss2_edge_tile_bwd_launch(x, ldx, ea, ldea, W1, b1, dS, tile_start, ell_ent, ell_width, nrb, ntiles, slab, n_slabs, U, ldu, h, fn, fe, by_source, stream);
}
static int dss2_edge_tile_bwd_launch(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1, const float* b1, const float* dS, const int32_t* tile_start, const void* ell_ent, int ell_width, int nrb, int ntiles, float* slab, int n_slabs, float* U, int64_t ldu, int h, int fn, int fe, int by_source, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_tile_bwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256 || ell_width <= 0 || ell_width > 32) { set_error("edge_tile_bwd: bad h=%d or ell_width=%d", h, ell_width); return 2; }
  if (!by_source && !slab) { set_error("edge_tile_bwd: slab is NULL"); return 2; }
  if (by_source && !U) { set_error("edge_tile_bwd: by_source needs U"); return 2; }
  if (n_slabs <= 0) { set_error("edge_tile_bwd: n_slabs must be > 0"); return 2; }
  if (ntiles <= 0) return 0;
  EdgeTileArgs a{x, ldx, ea, ldea, W1, b1, dS, tile_start, reinterpret_cast<const int2*>(ell_ent), nullptr, slab, U, ldu,
                 h, ell_width, nrb * 32, by_source, ntiles};
  // n_slabs workgroups walk the tiles (the slab buffer holds one partial per workgroup)
  // (96-row tiles WITH the per-row sums U, the PFN inner-block case: that instantiation misses its register budget, so it is
  //  not compiled -- the VALU tile kernel below serves it)
  const int part_nrb = nrb == 4 ? 2 : 3;
  if (!by_source && (nrb == 4 || nrb == 6) && !U && edge_mfma_ok(h, part_nrb, ell_width, true) && edge_mfma_ok(h, part_nrb, ell_width) && edge16_ok(h, nrb, ell_width, false, false) && edge16_ok(h, nrb, ell_width, true, false))
    return launch_edge16(a, nrb, n_slabs < ntiles ? n_slabs : ntiles, true, as_stream(stream));
  if (!by_source && edge_mfma_ok(h, nrb, ell_width, true) && !(nrb == 3 && U)) {
    if (edge16_ok(h, nrb, ell_width, true, U != nullptr))      // (the recomputed gates come from the arithmetic of the bf16x6 forward)
      return launch_edge16(a, nrb, n_slabs < ntiles ? n_slabs : ntiles, true, as_stream(stream));
    return dispatch_edge_mfma(a, nrb, n_slabs < ntiles ? n_slabs : ntiles, true, as_stream(stream));
  }
  return launch_edge_tile<true>(a, n_slabs < ntiles ? n_slabs : ntiles, as_stream(stream));
}

static int dss2_edge_combine_fwd_launch(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fe, void* stream);
extern "C" int dss2_edge_combine_fwd(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fe, void* stream) {
  DSS2_RECORD([AB, ldab, ea, ldea, W1c, ldw, b1, rowptr, col, ent, S, n_nodes, h, fe](void* s_) { return dss2_edge_combine_fwd_launch(AB, ldab, ea, ldea, W1c, ldw, b1, rowptr, col, ent, S, n_nodes, h, fe, s_); });
  return dss2_edge_combine_fwd_launch(AB, ldab, ea, ldea, W1c, ldw, b1, rowptr, col, ent, S, n_nodes, h, fe, stream);
}
static int dss2_edge_combine_fwd_launch(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* S, int64_t n_nodes, int h, int fe, void* stream) {
  if (h <= 0 || fe < 0 || fe > EC_MAXFE) { set_error("edge_combine_fwd: h=%d / fe=%d (0..32) unsupported", h, fe); return 2; }
  if (n_nodes <= 0) return 0;
  int64_t blocks = (n_nodes + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const int cw = fe <= 8 ? 256 : (fe <= 16 ? 128 : 64);      // hidden units per launch: 64 FPL lanes x (FPL x MAXFE <= 32 weight registers)
  for (int c0 = 0; c0 < h; c0 += cw) {      // hidden units are independent: wider layers run as several launches
    const int hc = h - c0 < cw ? h - c0 : cw;
    EdgeCombineArgs a{AB, ldab, ea, ldea, W1c, ldw, b1, nullptr, rowptr, col, ent, S, nullptr, nullptr, n_nodes, hc, fe, 0, h, c0};
    const int fpl = (hc + 63) / 64;
#define L(FPL, MFE) hipLaunchKernelGGL((edge_combine_kernel<FPL, false, MFE>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a)
    if (fe <= 8) { if (fpl == 1) L(1, 8); else if (fpl == 2) L(2, 8); else if (fpl == 3) L(3, 8); else L(4, 8); }
    else if (fe <= 16) { if (fpl == 1) L(1, 16); else L(2, 16); }
    else L(1, 32);
#undef L
  }
  return check_launch("edge_combine_fwd");
}

static int dss2_edge_combine_bwd_launch(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* dAB, float* slab, int n_slabs, int64_t n_nodes, int h, int fe, int by_source, void* stream);
extern "C" int dss2_edge_combine_bwd(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* dAB, float* slab, int n_slabs, int64_t n_nodes, int h, int fe, int by_source, void* stream) {
  DSS2_RECORD([AB, ldab, ea, ldea, W1c, ldw, b1, dS, rowptr, col, ent, dAB, slab, n_slabs, n_nodes, h, fe, by_source](void* s_) { return dss2_edge_combine_bwd_launch(AB, ldab, ea, ldea, W1c, ldw, b1, dS, rowptr, col, ent, dAB, slab, n_slabs, n_nodes, h, fe, by_source, s_); });
  return dss2_edge_combine_bwd_launch(AB, ldab, ea, ldea, W1c, ldw, b1, dS, rowptr, col, ent, dAB, slab, n_slabs, n_nodes, h, fe, by_source, stream);
}
static int dss2_edge_combine_bwd_launch(const float* AB, int64_t ldab, const float* ea, int64_t ldea, const float* W1c, int64_t ldw, const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col, const int32_t* ent, float* dAB, float* slab, int n_slabs, int64_t n_nodes, int h, int fe, int by_source, void* stream) {
  if (h <= 0 || fe < 0 || fe > EC_MAXFE) { set_error("edge_combine_bwd: h=%d / fe=%d (0..32) unsupported", h, fe); return 2; }
  if (!by_source && (!slab || n_slabs <= 0)) { set_error("edge_combine_bwd: slab missing"); return 2; }
  if (n_nodes <= 0) return 0;
  int64_t blocks = by_source ? (n_nodes + 3) / 4 : n_slabs;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const int cw = fe <= 8 ? 256 : (fe <= 16 ? 128 : 64);
  for (int c0 = 0; c0 < h; c0 += cw) {
    const int hc = h - c0 < cw ? h - c0 : cw;
    EdgeCombineArgs a{AB, ldab, ea, ldea, W1c, ldw, b1, dS, rowptr, col, ent, nullptr, dAB, slab, n_nodes, hc, fe, by_source, h, c0};
    const int fpl = (hc + 63) / 64;
#define L(FPL, MFE) hipLaunchKernelGGL((edge_combine_kernel<FPL, true, MFE>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a)
    if (fe <= 8) { if (fpl == 1) L(1, 8); else if (fpl == 2) L(2, 8); else if (fpl == 3) L(3, 8); else L(4, 8); }
    else if (fe <= 16) { if (fpl == 1) L(1, 16); else L(2, 16); }
    else L(1, 32);
#undef L
  }
  return check_launch("edge_combine_bwd");
}

static int dss2_csr_axpy_launch(const dss2_csr_axpy_args* ap, void* stream);
extern "C" int dss2_csr_axpy(const dss2_csr_axpy_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_csr_axpy: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_csr_axpy_launch(&a, s_); });
  return dss2_csr_axpy_launch(ap, stream);
}
static int dss2_csr_axpy_launch(const dss2_csr_axpy_args* ap, void* stream) {
  const dss2_csr_axpy_args& a = *ap;
  if (a.n_rows <= 0) return 0;
  if (!a.rowptr || !a.col || !a.w || !a.T || !a.out || a.h <= 0) { set_error("csr_axpy: bad arguments"); return 2; }
  if (a.drop_id && !a.drop_state) { set_error("csr_axpy: in-kernel dropout needs drop_state"); return 2; }
  auto al = [](const void* q, int64_t ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
  const bool vec = (a.h & 3) == 0 && al(a.T, a.ldt) && al(a.out, a.ldo) && (!a.add || al(a.add, a.ld_add)) &&
                   (!a.relu_src || al(a.relu_src, a.ld_relu)) && (!a.add_src || al(a.add_src, a.ld_src)) &&
                   (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
  CsrAxpyArgs k{a.rowptr, a.col, a.w, a.T, a.ldt, a.add, a.ld_add, a.out, a.ldo, a.bias, a.relu_src, a.ld_relu, a.add_src, a.ld_src,
                a.drop_state, a.drop_thr, a.drop_scale, a.drop_id, a.relu, a.n_rows, a.h};
  int64_t blocks = (a.n_rows * (vec ? a.h / 4 : a.h) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (vec) hipLaunchKernelGGL(csr_axpy_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k);
  else hipLaunchKernelGGL(csr_axpy_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k);
  return check_launch("csr_axpy");
}

static int dss2_segment_sum_launch(const float* msg, int64_t ldm, const int32_t* rowptr, const int32_t* ent, float* out, int64_t ldo, int64_t n_rows, int h, void* stream);
extern "C" int dss2_segment_sum(const float* msg, int64_t ldm, const int32_t* rowptr, const int32_t* ent, float* out, int64_t ldo, int64_t n_rows, int h, void* stream) {
  DSS2_RECORD([msg, ldm, rowptr, ent, out, ldo, n_rows, h](void* s_) { return dss2_segment_sum_launch(msg, ldm, rowptr, ent, out, ldo, n_rows, h, s_); });
  return dss2_segment_sum_launch(msg, ldm, rowptr, ent, out, ldo, n_rows, h, stream);
}
static int dss2_segment_sum_launch(const float* msg, int64_t ldm, const int32_t* rowptr, const int32_t* ent, float* out, int64_t ldo, int64_t n_rows, int h, void* stream) {
  if (h <= 0) { set_error("segment_sum: h=%d", h); return 2; }
  if (n_rows <= 0) return 0;
  const bool fast = (h == 32 || h == 64 || h == 128 || h == 256) && !(ldm & 3) && !(ldo & 3) &&
                    !(reinterpret_cast<uintptr_t>(msg) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15);
  if (!fast) {      // any width / alignment: one thread per output element
    int64_t blocks = (n_rows * h + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(segment_sum_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), msg, ldm, rowptr,
                       ent, out, ldo, n_rows, h);
    return check_launch("segment_sum");
  }
  const int gpb = 256 / (h / 4);
  int64_t blocks = (n_rows + gpb - 1) / gpb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), msg, ldm, rowptr, ent,
                     out, ldo, n_rows, h);
  return check_launch("segment_sum");
}

static int dss2_gather_rows_launch(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo, int64_t n_rows, int h, void* stream);
extern "C" int dss2_gather_rows(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo, int64_t n_rows, int h, void* stream) {
  DSS2_RECORD([src, lds, idx, out, ldo, n_rows, h](void* s_) { return dss2_gather_rows_launch(src, lds, idx, out, ldo, n_rows, h, s_); });
  return dss2_gather_rows_launch(src, lds, idx, out, ldo, n_rows, h, stream);
}
static int dss2_gather_rows_launch(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo, int64_t n_rows, int h, void* stream) {
  if (h <= 0) { set_error("gather_rows: h=%d", h); return 2; }
  if (n_rows <= 0) return 0;
  const bool vec = !(h & 3) && !(lds & 3) && !(ldo & 3) && !(reinterpret_cast<uintptr_t>(src) & 15) &&
                   !(reinterpret_cast<uintptr_t>(out) & 15);
  int64_t blocks = (n_rows * (vec ? h / 4 : h) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (vec) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, lds, idx, out, ldo, n_rows, h);
  else hipLaunchKernelGGL(gather_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, lds, idx, out, ldo, n_rows, h);
  return check_launch("gather_rows");
}

static int dss2_reduce_slabs_launch(const float* slab, int n_slabs, int64_t stride, float* out, int64_t len, void* stream);
extern "C" int dss2_reduce_slabs(const float* slab, int n_slabs, int64_t stride, float* out, int64_t len, void* stream) {
  DSS2_RECORD([slab, n_slabs, stride, out, len](void* s_) { return dss2_reduce_slabs_launch(slab, n_slabs, stride, out, len, s_); });
  return dss2_reduce_slabs_launch(slab, n_slabs, stride, out, len, stream);
}
static int dss2_reduce_slabs_launch(const float* slab, int n_slabs, int64_t stride, float* out, int64_t len, void* stream) {
  if (len <= 0) return 0;
  if (stride % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && n_slabs > 0) {
    // the same summation order as the batched form (dss2_reduce_slabs_multi): a reduction gives the same bits either way
    ReduceTable tab = {};
    tab.d[0].slab = slab; tab.d[0].out = out; tab.d[0].stride = stride; tab.d[0].len = len; tab.d[0].n_slabs = n_slabs;
    hipLaunchKernelGGL(reduce_slabs_multi_v4_kernel, dim3((unsigned)reduce_blocks(tab.d[0], false), 1), dim3(256), 0, as_stream(stream), tab);      // (many short slabs: 16 floats per workgroup)
    return check_launch("reduce_slabs");
  }
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((len + 63) / 64)), dim3(64, 4), 0, as_stream(stream), slab,
                     n_slabs, stride, out, len);
  return check_launch("reduce_slabs");
}

static int dss2_reduce_slabs_multi_launch(const dss2_reduce_desc* descs_host, int n_desc, void* stream);
extern "C" int dss2_reduce_slabs_multi(const dss2_reduce_desc* descs_host, int n_desc, void* stream) {
  DSS2_RECORD([d = dss2::plan_keep(descs_host, (size_t)(n_desc > 0 ? n_desc : 0)), n_desc](void* s_) { return dss2_reduce_slabs_multi_launch(dss2::plan_ptr(d), n_desc, s_); });
  return dss2_reduce_slabs_multi_launch(descs_host, n_desc, stream);
}
static int dss2_reduce_slabs_multi_launch(const dss2_reduce_desc* descs_host, int n_desc, void* stream) {
  if (n_desc <= 0) return 0;
  if (!descs_host || n_desc > REDUCE_MAX_DESC) { set_error("reduce_slabs_multi: 1..%d descriptors, got %d", REDUCE_MAX_DESC, n_desc); return 2; }
  ReduceTable tab = {};
  int64_t max_len = 0;
  for (int i = 0; i < n_desc; ++i) {
    tab.d[i] = descs_host[i];
    if (!tab.d[i].slab || !tab.d[i].out || tab.d[i].n_slabs <= 0 || tab.d[i].len < 0) { set_error("reduce_slabs_multi: descriptor %d is incomplete", i); return 2; }
    if (tab.d[i].len > max_len) max_len = tab.d[i].len;
  }
  if (max_len == 0) return 0;
  int64_t blocks = 0;
  for (int i = 0; i < n_desc; ++i) {     // 16-byte lanes per reduction that allows them
    if ((tab.d[i].stride % 4 != 0) || (reinterpret_cast<uintptr_t>(tab.d[i].slab) & 15) || (reinterpret_cast<uintptr_t>(tab.d[i].out) & 15))
      tab.scalar_mask |= 1u << i;
    const int64_t b = reduce_blocks(tab.d[i], ((tab.scalar_mask >> i) & 1u) != 0);      // (many short slabs: 16 floats per workgroup)
    if (b > blocks) blocks = b;
  }
  hipLaunchKernelGGL(reduce_slabs_multi_v4_kernel, dim3((unsigned)blocks, (unsigned)n_desc), dim3(256), 0,
                     as_stream(stream), tab);
  return check_launch("reduce_slabs_multi");
}
