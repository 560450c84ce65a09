// EdgeAggregation hidden layer (first Linear + ReLU of the edge MLP, aggregated per target
// node), its backward, and the standalone CSR segmented sum (K6).  gfx950.
//
// Work mapping: one wavefront per node row; lanes across the hidden features (feature j on lane
// j % 64, FPL = ceil(h/64) features per lane), the 22 first-layer weights of each owned feature
// live in registers.  Node / edge inputs of a row are wave-uniform, so they are fetched with
// scalar loads and broadcast for free; the only vector memory traffic is the coalesced S row.
#include "dss2_common.hpp"

namespace dss2 {

constexpr int FN = 8, FE = 6, FC = 2 * FN + FE;  // feature dims of the reference's data (networks.py:170)

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
        acc[f] += fmaxf(z, 0.f);
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
        const float dz = z > 0.f ? g : 0.f;
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
// Tile-based variants (used when the topology provides per-tile ELL slices with edge ids).
// One workgroup = one tile of whole graphs (<= TMAX rows).  Staged in LDS: the tile's x rows, an ELL
// slice {local other-node, stored edge id | flip} per (slot, row), and the gathered, sign-corrected
// edge_attr row of every slot.  All per-row / per-edge operands are then LDS broadcasts instead of
// chains of dependent scalar global loads (the row-per-wave kernels above are latency bound on
// rowptr -> col/ent -> x/ea).
// ------------------------------------------------------------------------------------------
struct EdgeTileArgs {
  const float* x; int64_t ldx; const float* ea; int64_t ldea;
  const float* W1; const float* b1; const float* dS;
  const int32_t* tile_start; const int2* ell_ent;   // [ntiles][D][TM] {local other node, eid | flip<<31 ; -1 = empty}
  float* S; float* slab; float* U; int64_t ldu;
  int h, D, TM, by_source, ntiles;
};

template <int FPL, bool BWD>
__global__ void __launch_bounds__(256) edge_tile_kernel(const EdgeTileArgs p) {
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int TM = p.TM, D = p.D;
  float* xs = esm;                                   // [TM][8]
  float* eaL = xs + TM * FN;                         // [D*TM][8]  (6 used; 8 keeps 16-byte rows)
  int* other = reinterpret_cast<int*>(eaL + D * TM * 8);   // [D*TM] local other node, -1 = empty slot
  __shared__ float red[BWD ? 256 * (FC + 1) : 1];
  const int tid = threadIdx.x, lane = tid & 63;
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

  for (int r = wave; r < R; r += 4) {
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + r * FN), xb = *reinterpret_cast<const f32x4*>(xs + r * FN + 4);
    float accv[FPL], grow[FPL];
#pragma unroll
    for (int f = 0; f < FPL; ++f) {
      accv[f] = 0.f;
      const int j = lane + 64 * f;
      grow[f] = (BWD && !p.by_source && j < p.h) ? p.dS[(int64_t)(ts + r) * p.h + j] : 0.f;   // one load per row
    }
    for (int k = 0; k < D; ++k) {
      const int o = other[k * TM + r];
      if (o < 0) continue;                            // wave-uniform
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
#pragma unroll
      for (int q = 0; q < 4; ++q) c[2 * FN + q] = e0[q];
      c[2 * FN + 4] = e1[0];
      c[2 * FN + 5] = e1[1];
      const int64_t tg = bs ? (int64_t)(ts + o) : (int64_t)(ts + r);
#pragma unroll
      for (int f = 0; f < FPL; ++f) {
        const int j = lane + 64 * f;
        float z = b[f];
#pragma unroll
        for (int q = 0; q < FC; ++q) z = fmaf(w[f][q], c[q], z);
        if (!BWD) {
          accv[f] += fmaxf(z, 0.f);
        } else {
          const float g = p.by_source ? ((j < p.h) ? p.dS[tg * p.h + j] : 0.f) : grow[f];
          const float dz = z > 0.f ? g : 0.f;
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
      if (j < p.h) {
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
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
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

}  // namespace dss2

using namespace dss2;

extern "C" int dss2_edge_hidden_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                                    const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* ent,
                                    float* S, int64_t n_nodes, int h, int fn, int fe, void* stream) {
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

extern "C" int dss2_edge_hidden_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                                    const float* b1, const float* dS, const int32_t* rowptr, const int32_t* col,
                                    const int32_t* ent, float* slab, int n_slabs, float* U, int64_t ldu,
                                    int64_t n_nodes, int h, int fn, int fe, int by_source, void* stream) {
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

template <bool BWD>
static int launch_edge_tile(const EdgeTileArgs& a, int grid, hipStream_t s) {
  const size_t lds = ((size_t)a.TM * FN + (size_t)a.D * a.TM * 8) * 4 + (size_t)a.D * a.TM * 4;
  const int fpl = (a.h + 63) / 64;
#define L(FPL) hipLaunchKernelGGL((edge_tile_kernel<FPL, BWD>), dim3(grid), dim3(256), lds, s, a)
  if (fpl == 1) L(1); else if (fpl == 2) L(2); else if (fpl == 3) L(3); else L(4);
#undef L
  return check_launch(BWD ? "edge_tile_bwd" : "edge_tile_fwd");
}

extern "C" int dss2_edge_tile_fwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                                  const float* b1, const int32_t* tile_start, const void* ell_ent, int ell_width,
                                  int nrb, int ntiles, float* S, int h, int fn, int fe, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_tile_fwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256 || ell_width <= 0 || ell_width > 32) { set_error("edge_tile_fwd: bad h=%d or ell_width=%d", h, ell_width); return 2; }
  if (ntiles <= 0) return 0;
  EdgeTileArgs a{x, ldx, ea, ldea, W1, b1, nullptr, tile_start, reinterpret_cast<const int2*>(ell_ent), S, nullptr, nullptr, 0,
                 h, ell_width, nrb * 32, 0, ntiles};
  return launch_edge_tile<false>(a, ntiles, as_stream(stream));
}

extern "C" int dss2_edge_tile_bwd(const float* x, int64_t ldx, const float* ea, int64_t ldea, const float* W1,
                                  const float* b1, const float* dS, const int32_t* tile_start, const void* ell_ent,
                                  int ell_width, int nrb, int ntiles, float* slab, int n_slabs, float* U, int64_t ldu,
                                  int h, int fn, int fe, int by_source, void* stream) {
  if (fn != FN || fe != FE) { set_error("edge_tile_bwd: only dim_featn=8, dim_feate=6 are built (got %d, %d)", fn, fe); return 2; }
  if (h <= 0 || h > 256 || ell_width <= 0 || ell_width > 32) { set_error("edge_tile_bwd: bad h=%d or ell_width=%d", h, ell_width); return 2; }
  if (!by_source && !slab) { set_error("edge_tile_bwd: slab is NULL"); return 2; }
  if (by_source && !U) { set_error("edge_tile_bwd: by_source needs U"); return 2; }
  if (n_slabs <= 0) { set_error("edge_tile_bwd: n_slabs must be > 0"); return 2; }
  if (ntiles <= 0) return 0;
  EdgeTileArgs a{x, ldx, ea, ldea, W1, b1, dS, tile_start, reinterpret_cast<const int2*>(ell_ent), nullptr, slab, U, ldu,
                 h, ell_width, nrb * 32, by_source, ntiles};
  // n_slabs workgroups walk the tiles (the slab buffer holds one partial per workgroup)
  return launch_edge_tile<true>(a, n_slabs < ntiles ? n_slabs : ntiles, as_stream(stream));
}

extern "C" int dss2_segment_sum(const float* msg, int64_t ldm, const int32_t* rowptr, const int32_t* ent, float* out,
                                int64_t ldo, int64_t n_rows, int h, void* stream) {
  if (h != 32 && h != 64 && h != 128 && h != 256) { set_error("segment_sum: h=%d unsupported (32/64/128/256)", h); return 2; }
  if ((ldm & 3) || (ldo & 3)) { set_error("segment_sum: leading dimensions must be multiples of 4"); return 2; }
  if (n_rows <= 0) return 0;
  const int gpb = 256 / (h / 4);
  int64_t blocks = (n_rows + gpb - 1) / gpb;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), msg, ldm, rowptr, ent,
                     out, ldo, n_rows, h);
  return check_launch("segment_sum");
}

extern "C" int dss2_reduce_slabs(const float* slab, int n_slabs, int64_t stride, float* out, int64_t len, void* stream) {
  if (len <= 0) return 0;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((len + 63) / 64)), dim3(64, 4), 0, as_stream(stream), slab,
                     n_slabs, stride, out, len);
  return check_launch("reduce_slabs");
}
