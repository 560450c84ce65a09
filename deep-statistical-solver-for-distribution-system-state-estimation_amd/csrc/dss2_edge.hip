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
