// Dataset side of the path: measurement model, masked z-score, batch collation.
// Replaces, on the device, /root/reference/data.py:119-190 (data_from_pickles' arithmetic) and the
// torch_geometric DataLoader collation of /root/reference/dss2_run.py:68-69,134.  All of it is
// element-wise / gather work on small rows: HBM-bound, no LDS tiling needed; the point is that a
// training epoch never leaves the device (no host collate, no per-batch H2D copy, no host sync).
#include "dss2_common.hpp"

namespace dss2 {

// ---- measurement model -----------------------------------------------------------------------
// numpy evaluates data.py:128-136 in float64 and rounds to float32 once; the same here, with
// explicit round-to-nearest intrinsics so that no multiply-add is contracted into an fma.
__device__ __forceinline__ float inv_var(double std, float floor_, float cap) {
  const float s = fabsf((float)std);
  const float m = fmaxf(s, floor_);
  const float c = __fdiv_rn(1.0f, __fmul_rn(m, m));
  return c < cap ? c : 0.f;
}

__global__ void __launch_bounds__(256) measure_nodes_kernel(const double* __restrict__ nodes, const uint8_t* __restrict__ meas_v,
                                                            int n_per_sample, const double* __restrict__ z, double v_noise,
                                                            double pm_noise, double p_noise, double zero_inj_coef,
                                                            float* __restrict__ x, long long R) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const double* t = nodes + r * 7;          // vm_pu, va_rad, p_mw, q_mvar, vn_kv, bool_slack, bool_zero_inj
  const double slack = t[5], zinj = t[6];
  const double mv = meas_v[r % n_per_sample] ? 1.0 : 0.0;
  const double nodes_noise[4] = {v_noise, v_noise, pm_noise, pm_noise};
  const double slack_noise[4] = {v_noise, zero_inj_coef, p_noise, p_noise};
  float* o = x + r * 11;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double mask = c == 0 ? mv : (c == 1 ? 0.0 : 1.0);
    const double mean = __dmul_rn(t[c], mask);
    const double coef = __dadd_rn(__dmul_rn(slack_noise[c], slack), __dmul_rn(nodes_noise[c], __dsub_rn(1.0, slack)));
    double std = __dmul_rn(mean, coef);
    o[2 * c] = (float)__dadd_rn(mean, __dmul_rn(z[r * 4 + c], fabs(std)));
    if (c >= 2) std = __dadd_rn(std, __dmul_rn(zero_inj_coef, zinj));
    if (c == 1) std = __dadd_rn(std, __dmul_rn(zero_inj_coef, slack));
    o[2 * c + 1] = inv_var(std, 1e-6f, 1e12f);
  }
  o[8] = (float)t[4];
  o[9] = (float)slack;
  o[10] = (float)zinj;
}

__global__ void __launch_bounds__(256) measure_edges_kernel(const double* __restrict__ edges, const uint8_t* __restrict__ meas_pf,
                                                            int e_per_sample, const double* __restrict__ z, double p_noise,
                                                            float* __restrict__ ea, long long R) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  // from_bus, to_bus, p_from_mw, q_from_mvar, G, B, Gs, Bs, closed line, phase shift, imax or sn
  const double* t = edges + r * 11;
  const double mask = meas_pf[r % e_per_sample] ? 1.0 : 0.0;
  float* o = ea + r * 13;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const double mean = __dmul_rn(t[2 + c], mask);
    const double std = __dmul_rn(mean, p_noise);
    o[2 * c] = (float)__dadd_rn(mean, __dmul_rn(z[r * 2 + c], fabs(std)));
    o[2 * c + 1] = inv_var(std, 1e-5f, 1e10f);
  }
  o[4] = (float)t[4];
  o[5] = (float)t[5];
#pragma unroll
  for (int c = 0; c < 7; ++c) o[6 + c] = (float)t[4 + c];
}

// ---- masked z-score (data.py:179-190) ----------------------------------------------------------
// pass 0: per column sum and count of the non-zero entries; pass 1: sum of squared deviations from the
// (fp32) mean.  Per-block partials in double, reduced in a fixed order: run-to-run deterministic.
constexpr int ZS_MAXC = 16;
__global__ void __launch_bounds__(256) masked_stats_kernel(const float* __restrict__ t, long long rows, int ld, int ncols,
                                                           const float* __restrict__ mean, int pass, double* __restrict__ partials) {
  __shared__ double red[4][2 * ZS_MAXC];
  double s[ZS_MAXC], cnt[ZS_MAXC];
#pragma unroll
  for (int c = 0; c < ZS_MAXC; ++c) { s[c] = 0.0; cnt[c] = 0.0; }
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < ZS_MAXC; ++c) {
      if (c < ncols) {
        const float v = t[r * ld + c];
        if (v != 0.f) {
          if (pass == 0) { s[c] += (double)v; cnt[c] += 1.0; }
          else { const float d = __fsub_rn(v, mean[c]); s[c] += (double)__fmul_rn(d, d); }
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < ZS_MAXC; ++c) {
    double a = s[c], b = cnt[c];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if (lane == 0) { red[wave][c] = a; red[wave][ZS_MAXC + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * ZS_MAXC) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partials[(size_t)blockIdx.x * 2 * ZS_MAXC + threadIdx.x] = v;
  }
}

// pass 0 -> mean[c], count kept in cnt[c]; pass 1 -> std[c]
__global__ void masked_stats_finish_kernel(const double* __restrict__ partials, int nblocks, int ncols, int pass,
                                           float* __restrict__ mean, float* __restrict__ stdv, double* __restrict__ cnt) {
  const int c = threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0, n = 0.0;
  for (int b = 0; b < nblocks; ++b) { s += partials[(size_t)b * 2 * ZS_MAXC + c]; n += partials[(size_t)b * 2 * ZS_MAXC + ZS_MAXC + c]; }
  if (pass == 0) {
    cnt[c] = n;
    const float m = (float)s / (float)n;                      // 0/0 = nan -> nan_to_num -> 0
    mean[c] = (m != m) ? 0.f : m;
  } else {
    const float v = sqrtf((float)s / (float)cnt[c]);
    stdv[c] = (v != v) ? 0.f : v;
  }
}

__device__ __forceinline__ float nan_to_num_f(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.4028234663852886e38f;
  if (v == -INFINITY) return -3.4028234663852886e38f;
  return v;
}

__global__ void __launch_bounds__(256) masked_normalize_kernel(const float* __restrict__ t, long long rows, int ld, int num_feat,
                                                               const float* __restrict__ mean, const float* __restrict__ stdv,
                                                               float* __restrict__ out, int ld_out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ld) return;
  const long long r = idx / ld;
  const int c = (int)(idx - r * ld);
  const float v = t[idx];
  float o = v;
  if (c < num_feat) {
    const float m = v != 0.f ? 1.f : 0.f;
    o = nan_to_num_f(__fdiv_rn(__fmul_rn(__fsub_rn(v, mean[c]), m), stdv[c]));
  }
  out[r * ld_out + c] = o;
}

// ---- collation -----------------------------------------------------------------------------------
// One workgroup per (selected sample, array): a contiguous chunk copy; edge_index chunks get the node
// offset of the sample's slot in the batch added (PyG Batch semantics).
struct CollateTable { dss2_collate_desc d[4]; };   // by value in the kernel arguments: no descriptor copy to the device
__global__ void __launch_bounds__(256) collate_kernel(const CollateTable tab, const long long* __restrict__ ids, long long B,
                                                      const long long* __restrict__ cursor) {
  const dss2_collate_desc& d = tab.d[blockIdx.y];
  const long long b = blockIdx.x;
  // cursor (dss2_collate_cursor): the batch's ids start at element cursor[0] of `ids`, wrapped into [0, cursor[1]) -- the position
  // lives on the device, so the launch is the same every step and can be replayed (hipGraph / launch plan)
  long long pos = b;
  if (cursor) { pos += cursor[0]; const long long n = cursor[1]; if (n > 0) pos %= n; }
  const long long s = ids ? ids[pos] : pos;
  if (d.kind == 0) {
    const float* src = static_cast<const float*>(d.src) + s * d.chunk;
    float* dst = static_cast<float*>(d.dst) + b * d.chunk;
    for (int k = threadIdx.x; k < d.chunk; k += blockDim.x) dst[k] = src[k];
  } else {   // edge_index: src [S][2][e] (or one shared [2][e] when d.shared), dst [2][B*e]
    const int e = d.chunk;
    const long long* src = static_cast<const long long*>(d.src) + (d.shared ? 0 : s * 2 * e);
    long long* dst = static_cast<long long*>(d.dst);
    const long long off = b * d.nodes_per_sample;
    for (int k = threadIdx.x; k < 2 * e; k += blockDim.x) {
      const int row = k / e, j = k - row * e;
      dst[(long long)row * B * e + b * e + j] = src[k] + off;
    }
  }
}

// Ragged variant for batches that mix cases with different closed-branch counts (BASELINE config C5: cigre14 and
// cigre14_reswitched in one batch): one launch per case, item j = the j-th sample of that case in the batch, copied to
// the row offsets of ITS slot (the host decides the batch composition and uploads the three small offset arrays; it
// never reads anything back).  kind 0: dst[(off[j]) * width ...] <- src[samp[j]][0..chunk) with off = node_off
// (shared == 0) or edge_off (shared != 0 reuses the flag as "edge-row operand");  kind 1: edge_index,
// dst[r][edge_off[j] + k] = src[samp[j]][r][k] + node_off[j], dst row stride = e_total.
struct RaggedArgs { dss2_collate_desc d[4]; const long long* samp; const long long* node_off; const long long* edge_off; long long e_total; };
__global__ void __launch_bounds__(256) collate_ragged_kernel(const RaggedArgs a) {
  const dss2_collate_desc& d = a.d[blockIdx.y];
  const long long j = blockIdx.x;
  const long long s = a.samp[j];
  if (d.kind == 0) {
    const long long row = d.shared ? a.edge_off[j] : a.node_off[j];
    const float* src = static_cast<const float*>(d.src) + s * d.chunk;
    float* dst = static_cast<float*>(d.dst) + row * d.nodes_per_sample;       // nodes_per_sample = floats per row here
    for (int k = threadIdx.x; k < d.chunk; k += blockDim.x) dst[k] = src[k];
  } else {
    const int e = d.chunk;
    const long long* src = static_cast<const long long*>(d.src) + (d.shared ? 0 : s * 2 * e);
    long long* dst = static_cast<long long*>(d.dst);
    const long long noff = a.node_off[j], eoff = a.edge_off[j];
    for (int k = threadIdx.x; k < 2 * e; k += blockDim.x) {
      const int row = k / e, i = k - row * e;
      dst[(long long)row * a.e_total + eoff + i] = src[k] + noff;
    }
  }
}

// advances a collation cursor by one batch (stream-ordered behind the collation that read it) and wraps it at the epoch's end
__global__ void cursor_advance_kernel(long long* cursor, long long B) {
  long long p = cursor[0] + B;
  const long long n = cursor[1];
  if (n > 0 && p >= n) p -= n;
  cursor[0] = p;
}

// acc[0] += *value, acc[1] += 1: the per-epoch mean of the step losses (dss2_run.py:146-147) without a torch kernel in the step
__global__ void accum_scalar_kernel(double* acc, const float* value) {
  acc[0] += (double)value[0];
  acc[1] += 1.0;
}

// several cases in ONE launch (blockIdx.z = case): a mixed batch is assembled by a single launch from a single uploaded table
struct RaggedMulti { RaggedArgs a[4]; long long count[4]; };
__global__ void __launch_bounds__(256) collate_ragged_multi_kernel(const RaggedMulti m) {
  const RaggedArgs& a = m.a[blockIdx.z];
  if ((long long)blockIdx.x >= m.count[blockIdx.z]) return;
  const dss2_collate_desc& d = a.d[blockIdx.y];
  const long long j = blockIdx.x;
  const long long s = a.samp[j];
  if (d.kind == 0) {
    const long long row = d.shared ? a.edge_off[j] : a.node_off[j];
    const float* src = static_cast<const float*>(d.src) + s * d.chunk;
    float* dst = static_cast<float*>(d.dst) + row * d.nodes_per_sample;
    for (int k = threadIdx.x; k < d.chunk; k += blockDim.x) dst[k] = src[k];
  } else {
    const int e = d.chunk;
    const long long* src = static_cast<const long long*>(d.src) + (d.shared ? 0 : s * 2 * e);
    long long* dst = static_cast<long long*>(d.dst);
    const long long noff = a.node_off[j], eoff = a.edge_off[j];
    for (int k = threadIdx.x; k < 2 * e; k += blockDim.x) {
      const int row = k / e, i = k - row * e;
      dst[(long long)row * a.e_total + eoff + i] = src[k] + noff;
    }
  }
}

}  // namespace dss2

using namespace dss2;

extern "C" int dss2_measure_nodes(const double* nodes, const uint8_t* meas_v_mask, int32_t n_per_sample, const double* z,
                                  double v_noise, double pm_noise, double p_noise, double zero_inj_coef, float* x,
                                  int64_t rows, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_measure_nodes");
  if (rows <= 0) return 0;
  if (!nodes || !meas_v_mask || !z || !x || n_per_sample <= 0) { set_error("measure_nodes: null argument"); return 2; }
  hipLaunchKernelGGL(measure_nodes_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), nodes,
                     meas_v_mask, n_per_sample, z, v_noise, pm_noise, p_noise, zero_inj_coef, x, (long long)rows);
  return check_launch("measure_nodes");
}

extern "C" int dss2_measure_edges(const double* edges, const uint8_t* meas_pflow_mask, int32_t e_per_sample, const double* z,
                                  double p_noise, float* edge_attr, int64_t rows, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_measure_edges");
  if (rows <= 0) return 0;
  if (!edges || !meas_pflow_mask || !z || !edge_attr || e_per_sample <= 0) { set_error("measure_edges: null argument"); return 2; }
  hipLaunchKernelGGL(measure_edges_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), edges,
                     meas_pflow_mask, e_per_sample, z, p_noise, edge_attr, (long long)rows);
  return check_launch("measure_edges");
}

extern "C" int dss2_masked_zscore(const float* t, int64_t rows, int32_t ld, int32_t num_feat, float* out, int32_t ld_out,
                                  float* mean, float* stdv, double* scratch, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_masked_zscore");
  if (rows <= 0) return 0;
  if (num_feat <= 0 || num_feat > ZS_MAXC || num_feat > ld) { set_error("masked_zscore: num_feat %d out of range 1..%d", num_feat, ZS_MAXC); return 2; }
  if (!t || !out || !mean || !stdv || !scratch) { set_error("masked_zscore: null argument"); return 2; }
  hipStream_t s = as_stream(stream);
  int nb = (int)((rows + 255) / 256);
  if (nb > 256) nb = 256;
  double* cnt = scratch;                       // [ZS_MAXC]
  double* partials = scratch + ZS_MAXC;        // [nb][2 * ZS_MAXC]
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(masked_stats_kernel, dim3(nb), dim3(256), 0, s, t, (long long)rows, ld, num_feat, mean, pass, partials);
    hipLaunchKernelGGL(masked_stats_finish_kernel, dim3(1), dim3(64), 0, s, partials, nb, num_feat, pass, mean, stdv, cnt);
  }
  hipLaunchKernelGGL(masked_normalize_kernel, dim3((unsigned)((rows * ld + 255) / 256)), dim3(256), 0, s, t, (long long)rows, ld,
                     num_feat, mean, stdv, out, ld_out);
  return check_launch("masked_zscore");
}

extern "C" int64_t dss2_masked_zscore_scratch_doubles(int64_t rows) {
  int64_t nb = (rows + 255) / 256;
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  return ZS_MAXC + nb * 2 * ZS_MAXC;
}

static int collate_launch(const CollateTable& tab, int32_t n_desc, const int64_t* sample_ids, int64_t batch, int64_t* cursor, int advance,
                          void* stream) {
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(collate_kernel, dim3((unsigned)batch, (unsigned)n_desc), dim3(256), 0, s, tab,
                     reinterpret_cast<const long long*>(sample_ids), (long long)batch, reinterpret_cast<const long long*>(cursor));
  if (cursor && advance)
    hipLaunchKernelGGL(cursor_advance_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(cursor), (long long)batch);
  return check_launch("collate");
}

static int collate_table(const dss2_collate_desc* descs_host, int32_t n_desc, CollateTable* tab) {
  if (!descs_host || n_desc < 1 || n_desc > 4) { set_error("collate: 1..4 descriptors expected, got %d", n_desc); return 2; }
  *tab = CollateTable{};
  for (int i = 0; i < n_desc; ++i) {
    tab->d[i] = descs_host[i];
    if (!tab->d[i].src || !tab->d[i].dst || tab->d[i].chunk <= 0) { set_error("collate: descriptor %d is incomplete", i); return 2; }
  }
  return 0;
}

extern "C" int dss2_collate(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* sample_ids, int64_t batch,
                            void* stream) {
  if (batch <= 0 || n_desc <= 0) return 0;
  CollateTable tab;
  if (int rc = collate_table(descs_host, n_desc, &tab)) return rc;
  DSS2_RECORD([tab, n_desc, sample_ids, batch](void* s_) { return collate_launch(tab, n_desc, sample_ids, batch, nullptr, 0, s_); });
  return collate_launch(tab, n_desc, sample_ids, batch, nullptr, 0, stream);
}

extern "C" int dss2_collate_cursor(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* sample_ids, int64_t* cursor,
                                   int64_t batch, int advance, void* stream) {
  if (batch <= 0 || n_desc <= 0) return 0;
  if (!sample_ids || !cursor) { set_error("collate_cursor: null argument"); return 2; }
  CollateTable tab;
  if (int rc = collate_table(descs_host, n_desc, &tab)) return rc;
  DSS2_RECORD([tab, n_desc, sample_ids, batch, cursor, advance](void* s_) { return collate_launch(tab, n_desc, sample_ids, batch, cursor, advance, s_); });
  return collate_launch(tab, n_desc, sample_ids, batch, cursor, advance, stream);
}

extern "C" int dss2_collate_ragged_multi(const dss2_collate_desc* descs_host, int32_t n_cases, int32_t n_desc, const int64_t* const* samp,
                                         const int64_t* const* node_off, const int64_t* const* edge_off, const int64_t* count, int64_t e_total,
                                         void* stream) {
  using namespace dss2;
  DSS2_NOT_IN_PLAN("dss2_collate_ragged_multi");
  if (!descs_host || n_cases < 1 || n_cases > 4 || n_desc < 1 || n_desc > 4 || !samp || !node_off || !edge_off || !count) { set_error("collate_ragged_multi: bad arguments"); return 2; }
  RaggedMulti m = {};
  long long cmax = 0;
  for (int c = 0; c < n_cases; ++c) {
    for (int i = 0; i < n_desc; ++i) {
      m.a[c].d[i] = descs_host[c * n_desc + i];
      if (count[c] > 0 && (!m.a[c].d[i].src || !m.a[c].d[i].dst || m.a[c].d[i].chunk <= 0)) { set_error("collate_ragged_multi: descriptor %d of case %d is incomplete", i, c); return 2; }
    }
    if (count[c] > 0 && (!samp[c] || !node_off[c] || !edge_off[c])) { set_error("collate_ragged_multi: case %d has no tables", c); return 2; }
    m.a[c].samp = reinterpret_cast<const long long*>(samp[c]);
    m.a[c].node_off = reinterpret_cast<const long long*>(node_off[c]);
    m.a[c].edge_off = reinterpret_cast<const long long*>(edge_off[c]);
    m.a[c].e_total = e_total;
    m.count[c] = count[c];
    if (count[c] > cmax) cmax = count[c];
  }
  if (cmax <= 0) return 0;
  hipLaunchKernelGGL(collate_ragged_multi_kernel, dim3((unsigned)cmax, (unsigned)n_desc, (unsigned)n_cases), dim3(256), 0, as_stream(stream), m);
  return check_launch("collate_ragged_multi");
}

static int accum_scalar_launch(double* acc, const float* value, void* stream) {
  hipLaunchKernelGGL(accum_scalar_kernel, dim3(1), dim3(1), 0, as_stream(stream), acc, value);
  return check_launch("accum_scalar");
}
extern "C" int dss2_accum_scalar(double* acc, const float* value, void* stream) {
  if (!acc || !value) { set_error("accum_scalar: null argument"); return 2; }
  DSS2_RECORD([acc, value](void* s_) { return accum_scalar_launch(acc, value, s_); });
  return accum_scalar_launch(acc, value, stream);
}

extern "C" int dss2_collate_ragged(const dss2_collate_desc* descs_host, int32_t n_desc, const int64_t* samp,
                                   const int64_t* node_off, const int64_t* edge_off, int64_t count, int64_t e_total,
                                   void* stream) {
  using namespace dss2;
  DSS2_NOT_IN_PLAN("dss2_collate_ragged");      // (a mixed-topology batch changes the graph structure: such a step is not replayable)
  if (count <= 0) return 0;
  if (!descs_host || n_desc < 1 || n_desc > 4 || !samp || !node_off || !edge_off) { set_error("collate_ragged: bad arguments"); return 2; }
  RaggedArgs a = {};
  for (int i = 0; i < n_desc; ++i) {
    a.d[i] = descs_host[i];
    if (!a.d[i].src || !a.d[i].dst || a.d[i].chunk <= 0) { set_error("collate_ragged: descriptor %d is incomplete", i); return 2; }
  }
  a.samp = reinterpret_cast<const long long*>(samp);
  a.node_off = reinterpret_cast<const long long*>(node_off);
  a.edge_off = reinterpret_cast<const long long*>(edge_off);
  a.e_total = e_total;
  hipLaunchKernelGGL(collate_ragged_kernel, dim3((unsigned)count, (unsigned)n_desc), dim3(256), 0, as_stream(stream), a);
  return check_launch("collate_ragged");
}
