// gsp_wls_edge + get_pflow (reference data.py:328-459), forward and analytic backward, gfx950.
//
// Node-centric, atomics-free: thread i owns bus i and walks its incident stored branches through
// the incidence CSR, recomputing the branch physics for each (each branch is evaluated from both
// of its ends; the work is ~60 flops + one sincos per visit and the whole batch is a few MB).
//   phase 0  vminmax_kernel      batch-global V_lv / V_hv           (data.py:335-336)
//   phase 1  wls_partials_kernel in-place slack masking, flows, bus injections, per-bus
//                                residual coefficients, 5 partial sums per workgroup (double)
//            wls_finish_kernel   fixed-order sum of the workgroup partials -> sums[0..6]
//   [data-parallel callers all-reduce sums here]
//   phase 2  wls_grad_kernel     loss scalar + d loss / d output (chain rule through get_pflow)
// The dead dense Laplacian of data.py:422-423 is not reproduced.
#include "dss2_common.hpp"

namespace dss2 {

struct EdgeP { float G, B, Gs, Bs, tp, imax; };

struct Flow {
  float pf, qf, pt, qt, i_f, i_t, load_line, load_trafo, d, s, c;
};

__device__ __forceinline__ EdgeP load_edge_param(const float* ep) {
  EdgeP e;
  e.G = ep[0]; e.B = ep[1]; e.Gs = ep[2]; e.Bs = ep[3];
  e.tp = ceilf(ep[5]);  // trafo flag = ceil(phase shift): 1 for CIGRE, 3 for ober_sub (data.py:367)
  e.imax = ep[6];
  return e;
}

// data.py:370-388.  shift: 0 with the reference's default phase_shift=True (data.py:362-363, the training path);
// edge_param[:, 5] with phase_shift=False (data.py:364-365)
__device__ __forceinline__ Flow branch_flow(float vf, float vt, float thf, float tht, const EdgeP& e, float vlv,
                                            float vhv, float shift = 0.f) {
  Flow f;
  f.d = thf - tht - shift;
  sincosf(f.d, &f.s, &f.c);
  const float kk = vlv * vlv;
  const float gg = e.G + e.Gs / 2.f, bb = e.B + e.Bs / 2.f;
  const float vv = vf * vt;
  f.pf = (-vv * (e.G * f.c + e.B * f.s) + gg * (vf * vf)) * kk;
  f.qf = (vv * (-e.G * f.s + e.B * f.c) - bb * (vf * vf)) * kk;
  f.pt = (-vv * (e.G * f.c - e.B * f.s) + gg * (vt * vt)) * kk;
  f.qt = (vv * (e.G * f.s + e.B * f.c) - bb * (vt * vt)) * kk;
  const float sqrt3 = 1.7320508075688772f;
  const float ratio = vhv / vlv;
  f.i_f = hypotf(f.pf, f.qf) / (vf * vlv * sqrt3);
  f.i_f = f.i_f / (1.f - (e.tp * (1.f - ratio)));
  f.i_t = hypotf(f.pt, f.qt) / (vt * vlv * sqrt3);
  f.load_line = ((1.f - e.tp) * fmaxf(f.i_f, f.i_t)) / e.imax;
  f.load_trafo = (e.tp * fmaxf(f.i_f * vhv, f.i_t * vlv)) / e.imax;
  return f;
}

// Stage 1 of the batch-global min/max of vn_kv: VMM_BLOCKS workgroups write (min, max) pairs to
// vmm[2 + 2*b]; stage 2 (vminmax_fold, in every consumer wave) folds them.  min/max are
// order independent, so the result is exact and reproducible.
constexpr int VMM_BLOCKS = 64;   // == wavefront size (vminmax_fold)

__global__ void __launch_bounds__(256) vminmax_kernel(const float* __restrict__ np, int64_t ld, int64_t n,
                                                       float* __restrict__ vmm) {
  __shared__ float smin[4], smax[4];
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = np[i * ld];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smin[w] = lo; smax[w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { lo = fminf(lo, smin[k]); hi = fmaxf(hi, smax[k]); }
    vmm[2 + 2 * blockIdx.x] = lo;
    vmm[3 + 2 * blockIdx.x] = hi;
  }
}

// Stage 2 is done by the consumers themselves: every wave folds the VMM_BLOCKS (= 64, one per lane) partial pairs
// with a butterfly -- one coalesced load and twelve cross-lane ops instead of a kernel launch of its own.
__device__ __forceinline__ void vminmax_fold(const float* __restrict__ vmm, float& vlv, float& vhv) {
  const int lane = threadIdx.x & 63;
  float lo = vmm[2 + 2 * lane], hi = vmm[3 + 2 * lane];
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  vlv = lo;
  vhv = hi;
}

struct NodeMeas { float Z[4], R[4]; };

__device__ __forceinline__ NodeMeas node_meas(const float* in, const float* xm, const float* xs) {
  NodeMeas m;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = in[c];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float z = v[2 * c], r = v[2 * c + 1];
    m.Z[c] = (z != 0.f) ? (z * xs[2 * c] + xm[2 * c]) : 0.f;
    m.R[c] = (r != 0.f) ? (r * xs[2 * c + 1] + xm[2 * c + 1]) : 0.f;
  }
  return m;
}

constexpr int LB = 256;  // loss kernels' block size

// Everything a bus needs of up to four of its incident branches, requested in three ROUNDS (entries -> other ends -> values) instead of
// three dependent round trips PER branch: the loss kernels are one 4-wave workgroup per CU at C2, pure latency (round 5; the per-branch
// loop cost wls_partials 11 us and wls_grad 9 us for 61 K buses).  Entries beyond the bus's list repeat its last one (loaded, not used).
constexpr int WLS_GB = 4;
struct BranchIn {
  int en;                      // incidence entry: stored edge id, sign bit = this bus is the edge's to-end
  int other;                   // the other end's bus
  float o0, o1, slack_o;       // the other end's outputs and slack flag
  float ep[7];                 // edge_param row (G, B, Gs, Bs, closed, shift, imax)
  float ei[4];                 // edge_input row (z_p, r_p, z_q, r_q)
  float apq_o[2];              // (gradient kernel) the other end's dL/dP, dL/dQ
};
template <bool GRAD>
__device__ __forceinline__ void gather_branches(const dss2_wls_args& p, int k0, int k1, BranchIn (&b)[WLS_GB]) {
#pragma unroll
  for (int j = 0; j < WLS_GB; ++j) b[j].en = p.inc_ent[k0 + j < k1 ? k0 + j : k1 - 1];
#pragma unroll
  for (int j = 0; j < WLS_GB; ++j) {
    const int e = b[j].en & 0x7fffffff;
    const int32_t* ends = b[j].en < 0 ? p.efrom : p.eto;
    b[j].other = ends[e];
  }
#pragma unroll
  for (int j = 0; j < WLS_GB; ++j) {
    const int e = b[j].en & 0x7fffffff;
    const float* oo = p.output + (int64_t)b[j].other * p.ld_out;
    b[j].o0 = oo[0]; b[j].o1 = oo[1];
    b[j].slack_o = p.node_param[(int64_t)b[j].other * p.ld_np + 1];
    const float* ep = p.edge_param + (int64_t)e * p.ld_ep;
#pragma unroll
    for (int c = 0; c < 7; ++c) b[j].ep[c] = (c == 4) ? 0.f : ep[c];
    const float* ei = p.edge_input + (int64_t)e * p.ld_ein;
#pragma unroll
    for (int c = 0; c < 4; ++c) b[j].ei[c] = ei[c];
    if constexpr (GRAD) { b[j].apq_o[0] = p.apq[2 * (int64_t)b[j].other]; b[j].apq_o[1] = p.apq[2 * (int64_t)b[j].other + 1]; }
  }
}
__device__ __forceinline__ EdgeP edge_param_of(const BranchIn& b) {
  EdgeP e;
  e.G = b.ep[0]; e.B = b.ep[1]; e.Gs = b.ep[2]; e.Bs = b.ep[3];
  e.tp = ceilf(b.ep[5]);
  e.imax = b.ep[6];
  return e;
}

__global__ void __launch_bounds__(LB) wls_partials_kernel(const dss2_wls_args p) {
  __shared__ double red[LB / 64][5];
  const int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x;
  float vlv, vhv;
  vminmax_fold(p.vminmax, vlv, vhv);
  // (the 8 + 8 + 4 + 4 normalisation constants: read ONCE, unconditionally -- inside `z != 0 ? z * std[c] + mean[c] : 0` every one of
  //  them was a load behind a branch behind the load of z, a memory round trip each)
  float xmv[8], xsv[8], emv[4], esv[4];
#pragma unroll
  for (int c = 0; c < 8; ++c) { xmv[c] = p.x_mean[c]; xsv[c] = p.x_std[c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) { emv[c] = p.edge_mean[c]; esv[c] = p.edge_std[c]; }
  const float xs0 = xsv[0], xm0 = xmv[0];
  double acc[5] = {0, 0, 0, 0, 0};
  if (i < p.n_nodes) {
    float* o = p.output + i * p.ld_out;
    const float mi = 1.f - p.node_param[i * p.ld_np + 1];
    const float th_i = o[1] * mi;
    o[1] = th_i;  // theta_i *= (1 - slack), in place on the model output (data.py:413)
    const float v_i = o[0] * xs0 + xm0;
    const NodeMeas nm = node_meas(p.input + i * p.ld_in, xmv, xsv);
    float p_i = 0.f, q_i = 0.f, s_edge = 0.f, s_t = 0.f, s_l = 0.f;
    const int k1 = p.inc_rowptr[i + 1];
    for (int k0 = p.inc_rowptr[i]; k0 < k1; k0 += WLS_GB) {
     BranchIn bin[WLS_GB];
     gather_branches<false>(p, k0, k1, bin);
#pragma unroll
     for (int j = 0; j < WLS_GB; ++j) {
      if (k0 + j >= k1) break;
      const BranchIn& bi = bin[j];
      const int e = bi.en & 0x7fffffff;
      const bool to_end = bi.en < 0;
      const float v_o = bi.o0 * xs0 + xm0;
      // mask is 0/1, so re-applying it to a possibly already-masked value is idempotent
      const float th_o = bi.o1 * (1.f - bi.slack_o);
      const EdgeP ep = edge_param_of(bi);
      const Flow f = to_end ? branch_flow(v_o, v_i, th_o, th_i, ep, vlv, vhv)
                            : branch_flow(v_i, v_o, th_i, th_o, ep, vlv, vhv);
      if (to_end) {
        p_i -= f.pt;
        q_i -= f.qt;
      } else {
        p_i -= f.pf;
        q_i -= f.qf;
        // each stored edge is accounted once, at its from-end
        const float zp = bi.ei[0], rp = bi.ei[1], zq = bi.ei[2], rq = bi.ei[3];
        const float Zp = (zp != 0.f) ? (zp * esv[0] + emv[0]) : 0.f;
        const float Rp = (rp != 0.f) ? (rp * esv[1] + emv[1]) : 0.f;
        const float Zq = (zq != 0.f) ? (zq * esv[2] + emv[2]) : 0.f;
        const float Rq = (rq != 0.f) ? (rq * esv[3] + emv[3]) : 0.f;
        const float dp = Zp - f.pf, dq = Zq - f.qf;
        s_edge += dp * dp * Rp * p.lam_pf + dq * dq * Rq * p.lam_pf;
        s_t += relu_nan(fabsf(f.d) - 0.5f);
        s_l += relu_nan(f.load_line + f.load_trafo - 1.5f);
        if (p.pflow) {
          float* pf = p.pflow + (int64_t)e * 8;
          pf[0] = f.load_line; pf[1] = f.load_trafo; pf[2] = f.pf; pf[3] = f.qf;
          pf[4] = f.pt; pf[5] = f.qt; pf[6] = f.i_f; pf[7] = f.i_t;
        }
      }
     }
    }
    const float h[4] = {v_i, th_i, p_i, q_i};
    const float lam[4] = {p.lam_v, p.lam_v, p.lam_p, p.lam_p};
    float s_node = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = nm.Z[c] - h[c];
      s_node += d * d * nm.R[c] * lam[c];
    }
    p.apq[2 * i + 0] = -2.f * (nm.Z[2] - p_i) * nm.R[2] * p.lam_p;
    p.apq[2 * i + 1] = -2.f * (nm.Z[3] - q_i) * nm.R[3] * p.lam_p;
    acc[0] = s_node;
    acc[1] = s_edge;
    acc[2] = relu_nan(v_i - 1.1f) + relu_nan(0.9f - v_i);
    acc[3] = s_t;
    acc[4] = s_l;
  }
#pragma unroll
  for (int c = 0; c < 5; ++c)
    for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int c = 0; c < 5; ++c) red[w][c] = acc[c];
  __syncthreads();
  const bool fused = (p.flags & DSS2_WLS_FUSED_FINISH) != 0;
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  typedef __attribute__((address_space(1))) unsigned gu32;
  if (threadIdx.x < 5) {
    double s = 0;
    for (int k = 0; k < LB / 64; ++k) s += red[k][threadIdx.x];
    // fused finish: the partial is PUBLISHED -- stored write-through at agent scope, so that it is in memory (not in this XCD's L2,
    // which the other XCDs do not see) when the arrival below is counted
    if (fused) __hip_atomic_store((gu64*)(p.partials + (size_t)blockIdx.x * 5 + threadIdx.x), (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p.partials[(size_t)blockIdx.x * 5 + threadIdx.x] = s;
  }
  if (!fused) return;
  // ---- the last workgroup to arrive finishes: fixed-order sum over the workgroups (independent of WHICH one is last).
  // No release fence: a device-scope fence per workgroup writes back its XCD's whole L2 on MI355X (the first form of this path:
  // 28 us at N = 61 440 against 10 + 4.7 us for two launches).  Here nothing but the five partials is handed over, and those go
  // through memory on both sides: write-through stores, drained (s_waitcnt vmcnt(0)) before the workgroup's arrival is counted by
  // a RELAXED agent-scope atomic; the last workgroup reads them with agent-scope atomic loads (no stale line of an earlier step in
  // its own L2 / L1 can answer).  The hand-off scheme of round 5's merged weight-space launches (HISTORY), without anybody polling.
  // That form rests on gfx950's sc1 write-through behaviour, not on the HIP memory model: it is kept for the opt-in
  // large grids only (DSS2_WLS_FUSED_FINISH=1 above 16 workgroups).  Up to 16 workgroups -- the DEFAULT use of this path, small
  // batches, where an XCD's L2 holds next to nothing to write back -- the arrival is a RELEASE fetch_add at agent scope and the last
  // workgroup ACQUIRES before it reads: a happens-before edge from every partial store to the loads below, by the book.
  __shared__ unsigned last_flag;
  const bool by_the_book = gridDim.x <= 16;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = by_the_book ? __hip_atomic_fetch_add((gu32*)p.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_fetch_add((gu32*)p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = prev == gridDim.x - 1 ? 1u : 0u;
    if (by_the_book && last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (pairs with the other workgroups' release arrivals)
  }
  __syncthreads();
  if (!last_flag) return;
  if (by_the_book) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __shared__ double tot[5];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;      // LB = 256: waves 0..3 take columns 0..3, wave 0 also column 4
  for (int col = c; col < 5; col += LB / 64) {
    double s = 0;
    for (int b = lane; b < (int)gridDim.x; b += 64)
      s += __longlong_as_double((long long)__hip_atomic_load((const gu64*)(p.partials + (size_t)b * 5 + col), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { p.sums[col] = s; tot[col] = s; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double Nn = (double)p.n_nodes, Ee = (double)p.n_edges, lr = (double)p.lam_reg;
    p.sums[5] = Nn; p.sums[6] = Ee; p.sums[7] = 0;
    const double mv = tot[2] / Nn, mt = tot[3] / Ee, ml = tot[4] / Ee;
    p.loss[0] = (float)(tot[0] / Nn + tot[1] / Ee + lr * mv * mv + lr * mt * mt + lr * ml * ml);
    __hip_atomic_store((gu32*)p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (a hipGraph replay included)
  }
}

__global__ void wls_value_kernel(const double* __restrict__ sums, float lam_reg, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double Nn = sums[5], Ee = sums[6], lr = (double)lam_reg;
  const double mv = sums[2] / Nn, mt = sums[3] / Ee, ml = sums[4] / Ee;
  loss[0] = (float)(sums[0] / Nn + sums[1] / Ee + lr * mv * mv + lr * mt * mt + lr * ml * ml);
}

// 5 waves, wave c sums column c of the workgroup partials: lanes stride over the workgroups, then
// a fixed-order butterfly.  Deterministic.
__global__ void __launch_bounds__(320) wls_finish_kernel(const double* __restrict__ partials, int n_blocks,
                                                         double* __restrict__ sums, double n_nodes, double n_edges,
                                                         float lam_reg, float* __restrict__ loss) {
  __shared__ double tot[5];
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0;
  for (int b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * 5 + c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) { sums[c] = s; tot[c] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[5] = n_nodes;
    sums[6] = n_edges;
    sums[7] = 0;
    if (loss) {      // the (local-batch) loss value, so that the gradient kernel can wait for the backward pass
      const double lr = (double)lam_reg, mv = tot[2] / n_nodes, mt = tot[3] / n_edges, ml = tot[4] / n_edges;
      loss[0] = (float)(tot[0] / n_nodes + tot[1] / n_edges + lr * mv * mv + lr * mt * mt + lr * ml * ml);
    }
  }
}

__global__ void __launch_bounds__(LB) wls_grad_kernel(const dss2_wls_args p) {
  const int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x;
  float vlv, vhv;
  vminmax_fold(p.vminmax, vlv, vhv);   // all lanes (before any early return)
  const double Nn = p.sums[5], Ee = p.sums[6];
  const double mean_v = p.sums[2] / Nn, mean_t = p.sums[3] / Ee, mean_l = p.sums[4] / Ee;
  if (i == 0 && !(p.flags & DSS2_WLS_NO_LOSS_WRITE)) {
    const double lr = (double)p.lam_reg;
    p.loss[0] = (float)(p.sums[0] / Nn + p.sums[1] / Ee + lr * mean_v * mean_v + lr * mean_t * mean_t +
                        lr * mean_l * mean_l);
  }
  if (i >= p.n_nodes) return;
  const float invN = (float)(1.0 / Nn), invE = (float)(1.0 / Ee);
  const float m_v = (float)(2.0 * p.lam_reg * mean_v / Nn);
  const float m_t = (float)(2.0 * p.lam_reg * mean_t / Ee);
  const float m_l = (float)(2.0 * p.lam_reg * mean_l / Ee);
  
  float xmv[8], xsv[8], emv[4], esv[4];      // (read once, unconditionally: see wls_partials_kernel)
#pragma unroll
  for (int c = 0; c < 8; ++c) { xmv[c] = p.x_mean[c]; xsv[c] = p.x_std[c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) { emv[c] = p.edge_mean[c]; esv[c] = p.edge_std[c]; }
  const float xs0 = xsv[0], xm0 = xmv[0];
  const float kk = vlv * vlv;
  const float sqrt3 = 1.7320508075688772f;
  const float ratio = vhv / vlv;

  const float* o = p.output + i * p.ld_out;
  const float mi = 1.f - p.node_param[i * p.ld_np + 1];
  const float th_i = o[1] * mi;
  const float v_i = o[0] * xs0 + xm0;
  const NodeMeas nm = node_meas(p.input + i * p.ld_in, xmv, xsv);
  float gv = invN * (-2.f * (nm.Z[0] - v_i) * nm.R[0] * p.lam_v);
  gv += m_v * ((v_i > 1.1f ? 1.f : 0.f) - (v_i < 0.9f ? 1.f : 0.f));
  float gth = invN * (-2.f * (nm.Z[1] - th_i) * nm.R[1] * p.lam_v);
  const float ap_i = p.apq[2 * i] * invN, aq_i = p.apq[2 * i + 1] * invN;

  const int k1 = p.inc_rowptr[i + 1];
  for (int k0 = p.inc_rowptr[i]; k0 < k1; k0 += WLS_GB) {
   BranchIn bin[WLS_GB];
   gather_branches<true>(p, k0, k1, bin);
#pragma unroll
   for (int j = 0; j < WLS_GB; ++j) {
    if (k0 + j >= k1) break;
    const BranchIn& bi = bin[j];
    const bool to_end = bi.en < 0;
    const float v_o = bi.o0 * xs0 + xm0;
    const float th_o = bi.o1 * (1.f - bi.slack_o);
    const float ap_o = bi.apq_o[0] * invN, aq_o = bi.apq_o[1] * invN;
    const EdgeP ep = edge_param_of(bi);
    const float vf = to_end ? v_o : v_i, vt = to_end ? v_i : v_o;
    const float thf = to_end ? th_o : th_i, tht = to_end ? th_i : th_o;
    const float ap_f = to_end ? ap_o : ap_i, aq_f = to_end ? aq_o : aq_i;
    const float ap_t = to_end ? ap_i : ap_o, aq_t = to_end ? aq_i : aq_o;
    const Flow f = branch_flow(vf, vt, thf, tht, ep, vlv, vhv);
    // upstream gradients w.r.t. the four flows
    const float zp = bi.ei[0], rp = bi.ei[1], zq = bi.ei[2], rq = bi.ei[3];
    const float Zp = (zp != 0.f) ? (zp * esv[0] + emv[0]) : 0.f;
    const float Rp = (rp != 0.f) ? (rp * esv[1] + emv[1]) : 0.f;
    const float Zq = (zq != 0.f) ? (zq * esv[2] + emv[2]) : 0.f;
    const float Rq = (rq != 0.f) ? (rq * esv[3] + emv[3]) : 0.f;
    float uPf = invE * (-2.f * (Zp - f.pf) * Rp * p.lam_pf) - ap_f;
    float uQf = invE * (-2.f * (Zq - f.qf) * Rq * p.lam_pf) - aq_f;
    float uPt = -ap_t, uQt = -aq_t;
    float dvf_direct = 0.f, dvt_direct = 0.f;
    // loading penalty -> currents -> flows and voltages
    const float loading = f.load_line + f.load_trafo;
    if (loading > 1.5f && m_l != 0.f) {
      const float dL_dIf = ((1.f - ep.tp) * (f.i_f >= f.i_t ? 1.f : 0.f) +
                            ep.tp * vhv * (f.i_f * vhv >= f.i_t * vlv ? 1.f : 0.f)) / ep.imax;
      const float dL_dIt = ((1.f - ep.tp) * (f.i_f >= f.i_t ? 0.f : 1.f) +
                            ep.tp * vlv * (f.i_f * vhv >= f.i_t * vlv ? 0.f : 1.f)) / ep.imax;
      const float gIf = m_l * dL_dIf, gIt = m_l * dL_dIt;
      const float cf = vlv * sqrt3 * (1.f - (ep.tp * (1.f - ratio)));
      const float ct = vlv * sqrt3;
      const float Af = hypotf(f.pf, f.qf), At = hypotf(f.pt, f.qt);
      if (Af > 0.f) {
        uPf += gIf * f.pf / (Af * vf * cf);
        uQf += gIf * f.qf / (Af * vf * cf);
      }
      if (At > 0.f) {
        uPt += gIt * f.pt / (At * vt * ct);
        uQt += gIt * f.qt / (At * vt * ct);
      }
      dvf_direct = -gIf * f.i_f / vf;
      dvt_direct = -gIt * f.i_t / vt;
    }
    // |theta_i - theta_j| penalty
    float gd = 0.f;
    if (fabsf(f.d) > 0.5f) gd = m_t * (f.d > 0.f ? 1.f : -1.f);
    // partial derivatives of the flows (data.py:370-376) w.r.t. vf, vt, d = thf - tht
    const float G = ep.G, B = ep.B, gg = ep.G + ep.Gs / 2.f, bb = ep.B + ep.Bs / 2.f;
    const float c = f.c, s = f.s, vv = vf * vt;
    const float a1 = G * c + B * s;    // in P_from
    const float a2 = -G * s + B * c;   // in Q_from (= d a1 / dd)
    const float a3 = G * c - B * s;    // in P_to
    const float a4 = G * s + B * c;    // in Q_to
    if (!to_end) {
      const float dPf = (-vt * a1 + 2.f * gg * vf) * kk, dQf = (vt * a2 - 2.f * bb * vf) * kk;
      const float dPt = (-vt * a3) * kk, dQt = (vt * a4) * kk;
      gv += uPf * dPf + uQf * dQf + uPt * dPt + uQt * dQt + dvf_direct;
    } else {
      const float dPf = (-vf * a1) * kk, dQf = (vf * a2) * kk;
      const float dPt = (-vf * a3 + 2.f * gg * vt) * kk, dQt = (vf * a4 - 2.f * bb * vt) * kk;
      gv += uPf * dPf + uQf * dQf + uPt * dPt + uQt * dQt + dvt_direct;
    }
    // d/dd: d a1 = a2, d a2 = -a1, d a3 = -a4, d a4 = a3
    const float dd = (uPf * (-vv * a2) + uQf * (vv * (-a1)) + uPt * (-vv * (-a4)) + uQt * (vv * a3)) * kk + gd;
    gth += to_end ? -dd : dd;
   }
  }
  const float gs = p.gscale ? p.gscale[0] : 1.f;      // upstream gradient of the loss (1 for loss.backward())
  p.grad_output[2 * i + 0] = gv * xs0 * gs;
  p.grad_output[2 * i + 1] = gth * mi * gs;
}

__global__ void __launch_bounds__(LB) pflow_kernel(const float* __restrict__ y, int64_t ldy,
                                                   const float* __restrict__ edge_param, int64_t ld_ep,
                                                   const int32_t* __restrict__ efrom, const int32_t* __restrict__ eto,
                                                   int64_t n_edges, const float* __restrict__ vminmax,
                                                   float* __restrict__ pflow, int apply_shift) {
  const int64_t e = (int64_t)blockIdx.x * LB + threadIdx.x;
  float vlv, vhv;
  vminmax_fold(vminmax, vlv, vhv);     // all lanes (before the early return)
  if (e >= n_edges) return;
  const int64_t a = efrom[e], b = eto[e];
  const EdgeP ep = load_edge_param(edge_param + e * ld_ep);
  const float shift = apply_shift ? edge_param[e * ld_ep + 5] : 0.f;
  const Flow f = branch_flow(y[a * ldy], y[b * ldy], y[a * ldy + 1], y[b * ldy + 1], ep, vlv, vhv, shift);
  float* pf = pflow + e * 8;
  pf[0] = f.load_line; pf[1] = f.load_trafo; pf[2] = f.pf; pf[3] = f.qf;
  pf[4] = f.pt; pf[5] = f.qt; pf[6] = f.i_f; pf[7] = f.i_t;
}


// ---- evaluation metrics of one test batch (dss2_run.py:178-206), accumulated on the device ------------
// yhat = (v de-normalised, theta masked at the slack) (:183-184); then squared / absolute errors of v and
// theta, of the line and trafo loadings where the true loading is non-zero (:196-205), and the sums for
// the per-column standard deviations (:207-208).  18 double sums per block, fixed-order finish.
constexpr int EV_SUMS = 18;
__global__ void __launch_bounds__(LB) eval_denorm_kernel(const float* __restrict__ out, int64_t ldo,
                                                         const float* __restrict__ node_param, int64_t ld_np,
                                                         const float* __restrict__ x_mean, const float* __restrict__ x_std,
                                                         float* __restrict__ yhat, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x;
  if (i >= n) return;
  yhat[2 * i + 0] = out[i * ldo] * x_std[0] + x_mean[0];
  yhat[2 * i + 1] = out[i * ldo + 1] * (1.f - node_param[i * ld_np + 1]);
}

__global__ void __launch_bounds__(256) eval_partials_kernel(const float* __restrict__ yhat, const float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ pf_true, const float* __restrict__ pf_out,
                                                            int64_t n, int64_t ne, double* __restrict__ partials) {
  __shared__ double red[4][EV_SUMS];
  double s[EV_SUMS];
#pragma unroll
  for (int k = 0; k < EV_SUMS; ++k) s[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float a0 = yhat[2 * i], a1 = yhat[2 * i + 1], b0 = y[i * ldy], b1 = y[i * ldy + 1];
    const float dv = a0 - b0, dt = a1 - b1;
    s[0] += (double)(dv * dv); s[1] += (double)fabsf(dv);
    s[2] += (double)(dt * dt); s[3] += (double)fabsf(dt);
    s[4] += a0; s[5] += (double)a0 * a0; s[6] += a1; s[7] += (double)a1 * a1;
    s[8] += b0; s[9] += (double)b0 * b0; s[10] += b1; s[11] += (double)b1 * b1;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {       // 0: line loading, 1: trafo loading
      const float t = pf_true[e * 8 + c];
      if (t != 0.f) {
        const float d = pf_out[e * 8 + c] - t;
        s[12 + 3 * c] += (double)(d * d); s[13 + 3 * c] += (double)fabsf(d); s[14 + 3 * c] += 1.0;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < EV_SUMS; ++k) {
    double a = s[k];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) red[wave][k] = a;
  }
  __syncthreads();
  if (threadIdx.x < EV_SUMS)
    partials[(size_t)blockIdx.x * EV_SUMS + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// acc[0..9] += rmse_v, mae_v, rmse_th, mae_th, rmse_loading, mae_loading, rmse_loading_trafos, mae_loading_trafos,
//              prop_std_v, prop_std_th   (the per-batch quantities the reference sums over the test loader)
__global__ void eval_finish_kernel(const double* __restrict__ partials, int nb, double n, double* __restrict__ acc) {
  __shared__ double t[EV_SUMS];
  if (threadIdx.x < EV_SUMS) {
    double v = 0.0;
    for (int b = 0; b < nb; ++b) v += partials[(size_t)b * EV_SUMS + threadIdx.x];
    t[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  acc[0] += sqrt(t[0] / n); acc[1] += t[1] / n;
  acc[2] += sqrt(t[2] / n); acc[3] += t[3] / n;
  acc[4] += sqrt(t[12] / t[14]); acc[5] += t[13] / t[14];       // empty selection: 0/0 = nan, like mse_loss of nothing
  acc[6] += sqrt(t[15] / t[17]); acc[7] += t[16] / t[17];
  auto sd = [&](double sx, double sxx) { return sqrt((sxx - sx * sx / n) / (n - 1.0)); };   // unbiased, torch.std
  acc[8] += sd(t[4], t[5]) / sd(t[8], t[9]) * 100.0;
  acc[9] += sd(t[6], t[7]) / sd(t[10], t[11]) * 100.0;
}

}  // namespace dss2

using namespace dss2;

static int dss2_wls_loss_partials_launch(const dss2_wls_args* ap, void* stream);
extern "C" int dss2_wls_loss_partials(const dss2_wls_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_wls_loss_partials: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_wls_loss_partials_launch(&a, s_); });
  return dss2_wls_loss_partials_launch(ap, stream);
}
static int dss2_wls_loss_partials_launch(const dss2_wls_args* ap, void* stream) {
  const dss2_wls_args& a = *ap;
  if (a.n_nodes <= 0 || a.n_edges <= 0) { set_error("wls_loss: empty batch"); return 2; }
  const int64_t nb = (a.n_nodes + LB - 1) / LB;
  hipStream_t s = as_stream(stream);
  if ((a.flags & DSS2_WLS_FUSED_FINISH) && !a.counter) { set_error("wls_loss: DSS2_WLS_FUSED_FINISH needs a counter word"); return 2; }
  if (!(a.flags & DSS2_WLS_VMM_CACHED))
    hipLaunchKernelGGL(vminmax_kernel, dim3(VMM_BLOCKS), dim3(256), 0, s, a.node_param, a.ld_np, a.n_nodes, a.vminmax);
  hipLaunchKernelGGL(wls_partials_kernel, dim3((unsigned)nb), dim3(LB), 0, s, a);
  if (!(a.flags & DSS2_WLS_FUSED_FINISH))
    hipLaunchKernelGGL(wls_finish_kernel, dim3(1), dim3(320), 0, s, a.partials, (int)nb, a.sums, (double)a.n_nodes,
                       (double)a.n_edges, a.lam_reg, a.loss);
  return check_launch("wls_loss_partials");
}

static int dss2_wls_loss_value_launch(const dss2_wls_args* ap, void* stream);
extern "C" int dss2_wls_loss_value(const dss2_wls_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_wls_loss_value: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_wls_loss_value_launch(&a, s_); });
  return dss2_wls_loss_value_launch(ap, stream);
}
static int dss2_wls_loss_value_launch(const dss2_wls_args* ap, void* stream) {
  if (!ap->sums || !ap->loss) { set_error("wls_loss_value: null argument"); return 2; }
  hipLaunchKernelGGL(wls_value_kernel, dim3(1), dim3(64), 0, as_stream(stream), ap->sums, ap->lam_reg, ap->loss);
  return check_launch("wls_loss_value");
}

static int dss2_wls_loss_grad_launch(const dss2_wls_args* ap, void* stream);
extern "C" int dss2_wls_loss_grad(const dss2_wls_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_wls_loss_grad: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_wls_loss_grad_launch(&a, s_); });
  return dss2_wls_loss_grad_launch(ap, stream);
}
static int dss2_wls_loss_grad_launch(const dss2_wls_args* ap, void* stream) {
  const dss2_wls_args& a = *ap;
  if (a.n_nodes <= 0) { set_error("wls_loss: empty batch"); return 2; }
  const int64_t nb = (a.n_nodes + LB - 1) / LB;
  hipLaunchKernelGGL(wls_grad_kernel, dim3((unsigned)nb), dim3(LB), 0, as_stream(stream), a);
  return check_launch("wls_loss_grad");
}

static int dss2_get_pflow_launch(const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                                 const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                                 int64_t n_nodes, int64_t n_edges, float* vminmax, float* pflow, int apply_shift, void* stream);
extern "C" int dss2_get_pflow(const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                              const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                              int64_t n_nodes, int64_t n_edges, float* vminmax, float* pflow, int apply_shift,
                              void* stream) {
  DSS2_RECORD([=](void* s_) { return dss2_get_pflow_launch(y, ldy, node_param, ld_np, edge_param, ld_ep, efrom, eto, n_nodes, n_edges, vminmax, pflow, apply_shift, s_); });
  return dss2_get_pflow_launch(y, ldy, node_param, ld_np, edge_param, ld_ep, efrom, eto, n_nodes, n_edges, vminmax, pflow, apply_shift, stream);
}
static int dss2_get_pflow_launch(const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                                 const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                                 int64_t n_nodes, int64_t n_edges, float* vminmax, float* pflow, int apply_shift, void* stream) {
  if (n_nodes <= 0 || n_edges <= 0) { set_error("get_pflow: empty batch"); return 2; }
  if (!y || !node_param || !edge_param || !efrom || !eto || !vminmax || !pflow) { set_error("get_pflow: null argument"); return 2; }
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(vminmax_kernel, dim3(VMM_BLOCKS), dim3(256), 0, s, node_param, ld_np, n_nodes, vminmax);
  hipLaunchKernelGGL(pflow_kernel, dim3((unsigned)((n_edges + LB - 1) / LB)), dim3(LB), 0, s, y, ldy, edge_param, ld_ep,
                     efrom, eto, n_edges, vminmax, pflow, apply_shift);
  return check_launch("get_pflow");
}

extern "C" int64_t dss2_eval_scratch_doubles(void) { return 256 * dss2::EV_SUMS; }

static int dss2_eval_batch_launch(const float* out, int64_t ldo, const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                                  const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                                  int64_t n_nodes, int64_t n_edges, const float* x_mean, const float* x_std, float* yhat,
                                  float* pf_true, float* pf_out, float* vminmax, double* scratch, double* acc, void* stream);
extern "C" int dss2_eval_batch(const float* out, int64_t ldo, const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                               const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                               int64_t n_nodes, int64_t n_edges, const float* x_mean, const float* x_std, float* yhat,
                               float* pf_true, float* pf_out, float* vminmax, double* scratch, double* acc, void* stream) {
  DSS2_RECORD([=](void* s_) { return dss2_eval_batch_launch(out, ldo, y, ldy, node_param, ld_np, edge_param, ld_ep, efrom, eto, n_nodes, n_edges,
                                                            x_mean, x_std, yhat, pf_true, pf_out, vminmax, scratch, acc, s_); });
  return dss2_eval_batch_launch(out, ldo, y, ldy, node_param, ld_np, edge_param, ld_ep, efrom, eto, n_nodes, n_edges, x_mean, x_std, yhat,
                                pf_true, pf_out, vminmax, scratch, acc, stream);
}
static int dss2_eval_batch_launch(const float* out, int64_t ldo, const float* y, int64_t ldy, const float* node_param, int64_t ld_np,
                                  const float* edge_param, int64_t ld_ep, const int32_t* efrom, const int32_t* eto,
                                  int64_t n_nodes, int64_t n_edges, const float* x_mean, const float* x_std, float* yhat,
                                  float* pf_true, float* pf_out, float* vminmax, double* scratch, double* acc, void* stream) {
  if (n_nodes <= 1 || n_edges <= 0) { set_error("eval_batch: needs at least 2 nodes and 1 edge"); return 2; }
  if (!out || !y || !node_param || !edge_param || !efrom || !eto || !x_mean || !x_std || !yhat || !pf_true || !pf_out ||
      !vminmax || !scratch || !acc) { set_error("eval_batch: null argument"); return 2; }
  hipStream_t s = as_stream(stream);
  const unsigned nbn = (unsigned)((n_nodes + LB - 1) / LB), nbe = (unsigned)((n_edges + LB - 1) / LB);
  hipLaunchKernelGGL(eval_denorm_kernel, dim3(nbn), dim3(LB), 0, s, out, ldo, node_param, ld_np, x_mean, x_std, yhat, n_nodes);
  hipLaunchKernelGGL(vminmax_kernel, dim3(VMM_BLOCKS), dim3(256), 0, s, node_param, ld_np, n_nodes, vminmax);
  hipLaunchKernelGGL(pflow_kernel, dim3(nbe), dim3(LB), 0, s, y, ldy, edge_param, ld_ep, efrom, eto, n_edges, vminmax, pf_true, 0);
  hipLaunchKernelGGL(pflow_kernel, dim3(nbe), dim3(LB), 0, s, yhat, (int64_t)2, edge_param, ld_ep, efrom, eto, n_edges, vminmax, pf_out, 0);
  int64_t m = n_nodes > n_edges ? n_nodes : n_edges;
  int nb = (int)((m + 255) / 256);
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(eval_partials_kernel, dim3(nb), dim3(256), 0, s, yhat, y, ldy, pf_true, pf_out, n_nodes, n_edges, scratch);
  hipLaunchKernelGGL(eval_finish_kernel, dim3(1), dim3(64), 0, s, scratch, nb, (double)n_nodes, acc);
  return check_launch("eval_batch");
}
