// Whole-stack kernels for the reference driver's own model line (include/dss2_hip.h "whole-stack kernels").
//
//   /root/reference/networks.py:340-388  PFN / SkipPFN: PFN-L chained MPN / SkipMPN blocks on the same edge inputs
//   /root/reference/networks.py:212-338  one block: EdgeAggregation -> TAGConv x L (dropout, ReLU between) [-> + input]
//   /root/reference/dss2_run.py:72-88    SkipPFN(dim_hid 32, 8 layers, K 2, dropout 0.3, L 5), batch_size 64
//
// At dim_hid = 32 a layer is 0.2 MFLOP per graph: nothing is MFMA- or HBM-bound, every per-layer launch is a chain of
// launch / L2 / LDS latencies (round 2: 67 launches per training step, 0.93 ms at B = 64).  Here ONE forward launch walks a
// tile of whole graphs (<= 64 rows) through every block of the stack and ONE backward launch walks it back:
//
//   forward, per block:  edge MLP (first Linear + ReLU, summed per target; VALU, weights in registers)            -> S
//                        conv 0 on S with the edge MLP's second Linear folded in (Wf_m = W_m W2, rank-3 bias term)
//                        conv 1 .. n_hh-1 (H -> H), each: tile GEMM [64 x 32] x [32 x 96] as bf16x6 on the bf16 matrix
//                        pipe (six waves: row block x matrix), two Horner hops on 16-byte row pieces in LDS, bias /
//                        in-kernel dropout (Philox, dss2_common.hpp) / ReLU, output to HBM once (the backward needs it)
//                        and in place into the LDS tile
//                        head H -> dout (<= 8): one 32-column GEMM, two hops on dout-wide rows, bias, residual; the
//                        8-wide block output is the next block's input and never leaves LDS on its way there
//   backward, per block (reverse), per tile of the workgroup:
//                        head / conv l: Z = [g | A^T g | (A^T)^2 g] by two hops; then ONE MFMA phase in which three waves
//                        accumulate dW_k += Z_k^T a_l (fp32 MFMA, persistent accumulators: block q = 3 l + k lives on wave
//                        q mod 8, slot q / 8), two waves compute the data gradient Z [W_0; W_1; W_2] (bf16x6, K = 96) and
//                        the spare waves the bias sums; gate by the saved activation + regenerated dropout mask
//                        edge MLP: gates recomputed with the forward's own arithmetic (edge_z), dW1 / db1 in registers,
//                        dx = U0 W1[:, :8] + U1 W1[:, 8:16] (+ residual) handed to the block below through HBM (8 floats / row)
//                        one slab of weight gradients per workgroup and block; dss2_stack_reduce sums the slabs in a fixed
//                        order and applies the chain rule of the fold.
//
// No float atomics: results are bitwise reproducible.  Shapes outside (dim_hid 32, K 2, 8 / 6 input features, tiles <= 64
// rows with ELL slices, <= 8 layers per block) run the per-block kernels (dss2_stack_supported).
#include "dss2_common.hpp"

namespace dss2 {

#ifdef DSS2_STACK_STAMPS
// Diagnostic build only (csrc/build.sh -DDSS2_STACK_STAMPS, tools/sstamps.py): s_memtime stamps of every wave of the first 64
// workgroups at the phase boundaries of the first block a workgroup processes (forward: block 0; backward: the last block,
// first tile).  They go to a buffer no kernel reads.
__device__ unsigned long long g_sstamps[2][64 * 8 * 128];
#define SSTAMP(which, on)                                                                                     \
  do {                                                                                                        \
    if (on) {                                                                                                 \
      unsigned long long t_;                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      if (lane == 0 && blockIdx.x < 64 && sidx < 128) g_sstamps[which][((size_t)blockIdx.x * 8 + wave) * 128 + sidx] = t_; \
      ++sidx;                                                                                                 \
    }                                                                                                         \
  } while (0)
#else
#define SSTAMP(which, on) do {} while (0)
#endif

constexpr int SH = 32, SNM = 3, SFN = 8, SFC = 22, SW1LD = 24, STM = 64, SXLD = 36;
constexpr int S_MAX_HH = 7, S_MAX_ELL = 4;      // (ELL width <= 4: the backward keeps a row's entries in registers and its LDS is full)

// ---- wpack layout (32-bit words), per block -------------------------------------------------------------------------------
constexpr int WP_W1 = 0, WP_B1 = SH * SW1LD, WP_CONV0 = WP_B1 + SH;
constexpr int WP_FRAG = 3 * 64 * 4;                          // one k-group of 16: 3 planes x 64 lanes x 8 bf16
constexpr int WP_CONV_FWD = 0;                               // [m][kg 0..1][plane][lane][4 words]
constexpr int WP_CONV_BWD = SNM * 2 * WP_FRAG;               // [ib 0..1][k-step 0..2][plane][lane][4]: 16x16x32 A fragments (k = 32 m + j)
constexpr int WP_CONV_BIAS = WP_CONV_BWD + 6 * WP_FRAG;      // [32]
constexpr int WP_CONV_PRE = WP_CONV_BIAS + SH;               // [3][32]  (conv 0: bf_m = W_m b2)
constexpr int WP_CONV_STRIDE = WP_CONV_PRE + SNM * SH;
constexpr int WP_HEAD_FWD = 0, WP_HEAD_BWD = 2 * WP_FRAG, WP_HEAD_BIAS = 4 * WP_FRAG, WP_HEAD_WORDS = WP_HEAD_BIAS + 8;
__host__ __device__ inline int wp_head(int n_hh) { return WP_CONV0 + n_hh * WP_CONV_STRIDE; }
__host__ __device__ inline int wp_block_words(int n_hh) { return wp_head(n_hh) + WP_HEAD_WORDS; }

// ---- flat gradient layout (floats), per block = networks.MPN._flat_offsets -------------------------------------------------
constexpr int FL_W1 = 0, FL_B1 = SH * SFC, FL_W2 = FL_B1 + SH, FL_B2 = FL_W2 + SH * SH, FL_CONV0 = FL_B2 + SH;
constexpr int FL_CONV_STRIDE = SNM * SH * SH + SH;
__host__ __device__ inline int fl_head(int n_hh) { return FL_CONV0 + n_hh * FL_CONV_STRIDE; }
__host__ __device__ inline int fl_block(int n_hh, int dout) { return fl_head(n_hh) + SNM * dout * SH + dout; }
__host__ __device__ inline int params_per_block(int n_hh) { return 4 + 4 * (n_hh + 1); }

__device__ __forceinline__ uint32_t bf16_bits(__bf16 v) { return (uint32_t)__builtin_bit_cast(uint16_t, v); }

// eight fp32 values -> the three bf16x8 planes (4 words each)
__device__ __forceinline__ void split8_store(const float (&v)[8], uint32_t* dst /* plane h */, int plane_stride) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    uint32_t h, m, l;
    split3_pair(v[2 * w], v[2 * w + 1], h, m, l);
    dst[w] = h; dst[plane_stride + w] = m; dst[2 * plane_stride + w] = l;
  }
}

// =====================================================================================================================
// pack: fold + bf16x3 fragments + fp32 copies, one launch per step
// =====================================================================================================================
struct EaPrep {      // per-tile edge-feature cache: [ntiles][D + DT][64][8] floats, rows as stage_edges lays them out in LDS
  const float* ea; int64_t ldea; const int32_t* ell_e; const int32_t* ellT_e; int D, DT, tm, ntiles; float* cache;
};

__global__ void __launch_bounds__(256) stack_pack_kernel(const dss2_stack_dims d, const float* const* __restrict__ params,
                                                         uint32_t* __restrict__ wpack, unsigned long long* rng_state,
                                                         unsigned long long* rng_snap, unsigned long long host_seed,
                                                         int use_host_seed, float* tick, const EaPrep ep) {
  __shared__ float src[SH][SH + 1], wm[SH][SH + 1], w2s[SH][SH + 1];
  const int tid = threadIdx.x;
  const int per_block = d.n_hh * SNM + 2;
  if ((int)blockIdx.x >= d.n_blocks * per_block) {
    // ---- edge-feature cache of one tile: the gathered, sign-corrected edge_attr row of every (slot, row) of its two ELL slices,
    // so that the forward / backward launches stage them with plain 16-byte copies instead of dependent gathers
    const int tile = blockIdx.x - d.n_blocks * per_block;
    float* out = ep.cache + (size_t)tile * (ep.D + ep.DT) * STM * 8;
    for (int idx = tid; idx < (ep.D + ep.DT) * STM; idx += 256) {
      const bool tr = idx >= ep.D * STM;
      const int li = tr ? idx - ep.D * STM : idx, k = li >> 6, r = li & 63;
      const int W = tr ? ep.DT : ep.D;
      int2 en = make_int2(0, -1);
      if (r < ep.tm) en = reinterpret_cast<const int2*>(tr ? ep.ellT_e : ep.ell_e)[((size_t)tile * W + k) * ep.tm + r];
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (en.y != -1) {
        const float* e = ep.ea + (int64_t)(en.y & 0x7fffffff) * ep.ldea;
        const float sgn = en.y < 0 ? -1.f : 1.f;
        a = f32x4{e[0] * sgn, e[1], e[2] * sgn, e[3]};
        c = f32x4{e[4], e[5], 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(out + idx * 8) = a;
      *reinterpret_cast<f32x4*>(out + idx * 8 + 4) = c;
    }
    return;
  }
  const int b = blockIdx.x / per_block, u = blockIdx.x - b * per_block;
  const float* const* P = params + (size_t)b * params_per_block(d.n_hh);
  uint32_t* wb = wpack + (size_t)b * wp_block_words(d.n_hh);
  if (blockIdx.x == 0 && tid == 0) {
    if (rng_snap) {
      if (use_host_seed) { rng_snap[0] = host_seed; rng_snap[1] = 0; }
      else { rng_snap[0] = rng_state[0]; rng_snap[1] = rng_state[1]; rng_state[1] = rng_state[1] + 1; }
    }
    if (tick) tick[0] += 1.f;
  }
  const int dout = (b == d.n_blocks - 1) ? d.dout_last : d.dout_inner;
  if (u < d.n_hh * SNM) {
    const int l = u / SNM, m = u - l * SNM;
    const float* W = P[4 + 4 * l + 1 + m];
    uint32_t* cw = wb + WP_CONV0 + l * WP_CONV_STRIDE;
    if (l == 0) {
      const float* W2 = P[2];
      const float* b2 = P[3];
      for (int idx = tid; idx < SH * SH; idx += 256) { wm[idx >> 5][idx & 31] = W[idx]; w2s[idx >> 5][idx & 31] = W2[idx]; }
      __syncthreads();
      for (int idx = tid; idx < SH * SH; idx += 256) {
        const int j = idx >> 5, c = idx & 31;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SH; ++i) s = fmaf(wm[j][i], w2s[i][c], s);
        src[j][c] = s;
      }
      if (tid < SH) {
        float s = 0.f;
        for (int i = 0; i < SH; ++i) s = fmaf(wm[tid][i], b2[i], s);
        reinterpret_cast<float*>(cw + WP_CONV_PRE)[m * SH + tid] = s;
      }
    } else {
      for (int idx = tid; idx < SH * SH; idx += 256) src[idx >> 5][idx & 31] = W[idx];
      if (tid < SH) reinterpret_cast<float*>(cw + WP_CONV_PRE)[m * SH + tid] = 0.f;
    }
    __syncthreads();
    const int t = tid & 127, kg = t >> 6, lane = t & 63, c32 = lane & 31, half = lane >> 5;
    float v[8];
    if (tid < 128) {      // forward operand: B[k = in][j = out] = src[j][k]
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[c32][16 * kg + 8 * half + q];
      split8_store(v, cw + WP_CONV_FWD + (m * 2 + kg) * WP_FRAG + lane * 4, 256);
    } else {              // data-gradient operand (v_mfma_f32_16x16x32_bf16, A side): A[i = 16 ib + lane % 16][k = 32 m + 8 (lane / 16) + q]
      const int ib = kg;  //   = W_m[j = 8 (lane / 16) + q][i]; fragment [ib][k-step m][plane][lane]
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[8 * (lane >> 4) + q][16 * ib + (lane & 15)];
      split8_store(v, cw + WP_CONV_BWD + (ib * SNM + m) * WP_FRAG + lane * 4, 256);
    }
  } else if (u == d.n_hh * SNM) {
    // head: three [dout][32] matrices side by side in one 32-column group (forward) / stacked along k (data-gradient)
    float* hs = &src[0][0];      // [3 * dout][33]
    for (int idx = tid; idx < SNM * dout * SH; idx += 256) {
      const int jj = idx >> 5, i = idx & 31, m = jj / dout, o = jj - m * dout;
      hs[jj * (SH + 1) + i] = P[4 + 4 * d.n_hh + 1 + m][o * SH + i];
    }
    __syncthreads();
    uint32_t* hw = wb + wp_head(d.n_hh);
    const int t = tid & 127, kg = t >> 6, lane = t & 63, c32 = lane & 31, half = lane >> 5;
    float v[8];
    if (tid < 128) {      // B[k = i][jj = c32] = W_m[o][i]
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = c32 < SNM * dout ? hs[c32 * (SH + 1) + 16 * kg + 8 * half + q] : 0.f;
      split8_store(v, hw + WP_HEAD_FWD + kg * WP_FRAG + lane * 4, 256);
    } else {              // A[i = 16 ib + lane % 16][kk = m dout + o] = W_m[o][i], kk = 8 (lane / 16) + q < 32 (one k-step, zero padded)
      const int ib = kg;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int kk = 8 * (lane >> 4) + q; v[q] = kk < SNM * dout ? hs[kk * (SH + 1) + 16 * ib + (lane & 15)] : 0.f; }
      split8_store(v, hw + WP_HEAD_BWD + ib * WP_FRAG + lane * 4, 256);
    }
    if (tid < 8) reinterpret_cast<float*>(hw + WP_HEAD_BIAS)[tid] = tid < dout ? P[4 + 4 * d.n_hh][tid] : 0.f;
  } else {
    float* f = reinterpret_cast<float*>(wb);
    for (int idx = tid; idx < SH * SW1LD; idx += 256) {
      const int j = idx / SW1LD, q = idx - j * SW1LD;
      f[WP_W1 + idx] = q < SFC ? P[0][j * SFC + q] : 0.f;
    }
    if (tid < SH) f[WP_B1 + tid] = P[1][tid];
    for (int idx = tid; idx < d.n_hh * SH; idx += 256) {
      const int l = idx >> 5, c = idx & 31;
      f[WP_CONV0 + l * WP_CONV_STRIDE + WP_CONV_BIAS + c] = P[4 + 4 * l][c];
    }
  }
}

// =====================================================================================================================
// shared device pieces
// =====================================================================================================================
// z = b1[j] + W1[j, :] . [x_target | x_source | edge_attr]  -- networks.py:181 concat order; ONE definition, so the gates
// the backward recomputes are bit for bit the forward's
__device__ __forceinline__ float edge_z(const float (&w)[SW1LD], float b, f32x4 ta, f32x4 tb, f32x4 sa, f32x4 sb, f32x4 e0, f32x4 e1) {
  // three independent chains (target, source, edge features), then two adds: a third of the dependent-FMA latency
  float zt = b, zs = 0.f, ze = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { zt = fmaf(w[q], ta[q], zt); zs = fmaf(w[8 + q], sa[q], zs); ze = fmaf(w[16 + q], e0[q], ze); }
#pragma unroll
  for (int q = 0; q < 4; ++q) { zt = fmaf(w[4 + q], tb[q], zt); zs = fmaf(w[12 + q], sb[q], zs); }
  ze = fmaf(w[20], e1[0], ze);
  ze = fmaf(w[21], e1[1], ze);
  return (zt + zs) + ze;
}

// the tile's slice of an {other node, ent} ELL table -> other[D][64] (-1 = empty) and its edge_attr rows [D][64][8] from the
// per-tile cache the pack launch has gathered (sign-corrected; independent 16-byte copies, no dependent gather)
__device__ __forceinline__ void stage_edges(const int32_t* __restrict__ ell_e, int tile, int D, int tm, const float* __restrict__ cache,
                                            int* other, float* eaL, int tid, int nthreads) {
  const int2* src = reinterpret_cast<const int2*>(ell_e) + (size_t)tile * D * tm;
  for (int idx = tid; idx < D * STM; idx += nthreads) {
    const int k = idx >> 6, r = idx & 63;
    int2 en = make_int2(0, -1);
    if (r < tm) en = src[k * tm + r];
    other[idx] = en.y != -1 ? en.x : -1;
    *reinterpret_cast<f32x4*>(eaL + idx * 8) = *reinterpret_cast<const f32x4*>(cache + idx * 8);
    *reinterpret_cast<f32x4*>(eaL + idx * 8 + 4) = *reinterpret_cast<const f32x4*>(cache + idx * 8 + 4);
  }
}

__device__ __forceinline__ void stage_ell_w(const int32_t* __restrict__ ell_w, int tile, int D, int tm, int2* dst, int tid, int nthreads) {
  const int2* src = reinterpret_cast<const int2*>(ell_w) + (size_t)tile * D * tm;
  for (int idx = tid; idx < D * STM; idx += nthreads) {
    const int k = idx >> 6, r = idx & 63;
    dst[idx] = r < tm ? src[k * tm + r] : make_int2(r, 0);
  }
}

// Workgroup barrier that drains LDS traffic only.  __syncthreads() is a full workgroup fence: it also waits for every
// outstanding GLOBAL access of the wave (s_waitcnt vmcnt(0)) -- the activation stores of the forward epilogues and the
// operand prefetches these kernels keep in flight across phases -- which put an L2 / HBM round trip (~1-2 K cycles) on
// the critical path of every phase.  All cross-wave hand-offs inside a tile go through LDS, so draining lgkmcnt is enough;
// the one hand-off through global memory (dx between the blocks of the backward) keeps __threadfence + __syncthreads.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// acc += A[32 rows of this lane's row block][16 k] x B (bf16x6): arow -> 8 consecutive fp32 of this lane's row and k-half
__device__ __forceinline__ f32x16 mma6(f32x16 c, const float* arow, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl) {
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(arow), v1 = *reinterpret_cast<const f32x4*>(arow + 4);
  uint32_t sh[4], sm[4], sl[4];
  split3_pair(v0[0], v0[1], sh[0], sm[0], sl[0]);
  split3_pair(v0[2], v0[3], sh[1], sm[1], sl[1]);
  split3_pair(v1[0], v1[1], sh[2], sm[2], sl[2]);
  split3_pair(v1[2], v1[3], sh[3], sm[3], sl[3]);
  const bf16x8 ah = __builtin_bit_cast(bf16x8, u32x4{sh[0], sh[1], sh[2], sh[3]});
  const bf16x8 am = __builtin_bit_cast(bf16x8, u32x4{sm[0], sm[1], sm[2], sm[3]});
  const bf16x8 al = __builtin_bit_cast(bf16x8, u32x4{sl[0], sl[1], sl[2], sl[3]});
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);      // smallest terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// =====================================================================================================================
// forward
// =====================================================================================================================
__global__ void __launch_bounds__(512, 2) stack_fwd_kernel(const dss2_stack_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = p.ell_width;
  float* Xs = smem;                                   // [64][36]  the activation tile
  float* G = Xs + STM * SXLD;                         // [3][64][36]  G_m = X W_m^T, then the Horner partials
  float* x8 = G + SNM * STM * SXLD;                   // [64][8]   block input
  float* dps = x8 + STM * SFN;                        // [64][4]   A^m deg
  float* eaL = dps + STM * 4;                         // [D][64][8]
  int2* ellw = reinterpret_cast<int2*>(eaL + D * STM * 8);      // [D][64]
  int* other = reinterpret_cast<int*>(ellw + D * STM);          // [D][64]
  const int tid = threadIdx.x, lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int ts = p.tile_start[tile], R = p.tile_start[tile + 1] - ts;
  const int n_hh = p.dims.n_hh, NB = p.dims.n_blocks;
  const int64_t N = p.n_nodes;
  const uint64_t dseed = p.drop_state ? p.drop_state[0] : 0, doff = p.drop_state ? p.drop_state[1] : 0;
  const int blk_words = wp_block_words(n_hh);

  // ---- stage what every block of the stack reads: input rows, ELL slices, edge features, bias row scales
  for (int idx = tid; idx < STM * SFN; idx += 512) {
    const int r = idx >> 3, c = idx & 7;
    x8[idx] = r < R ? p.x[(int64_t)(ts + r) * p.ldx + c] : 0.f;
  }
  if (tid < STM) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tid < R) v = *reinterpret_cast<const f32x4*>(p.deg_pows + (int64_t)(ts + tid) * 4);
    *reinterpret_cast<f32x4*>(dps + tid * 4) = v;
  }
  stage_ell_w(p.ell_w, tile, D, p.tm, ellw, tid, 512);
  stage_edges(p.ell_e, tile, D, p.tm, p.eacache + (size_t)tile * (D + p.ellT_width) * STM * 8, other, eaL, tid, 512);
  if (p.tm <= 32)      // rows 32..63 of the tile buffers are never produced: keep them zero (finite) for the pad-row hops
    for (int idx = tid; idx < 32 * SXLD * 4; idx += 512) { const int bsel = idx / (32 * SXLD); smem[bsel * STM * SXLD + 32 * SXLD + idx % (32 * SXLD)] = 0.f; }

  // MFMA roles: waves 0..5 = (row block, matrix); the head runs on the two waves with m == 0
  const int mm = wave % SNM, rb = wave / SNM;
  const bool short_t = p.tm <= 32;                     // 32-row tiles (small batches: twice the workgroups): row block 1 is empty
  const bool mma_wave = wave < (short_t ? SNM : 2 * SNM);
  const int n_et = short_t ? 2 : STM / 16;             // row quads of the edge phase
  const int prow = tid >> 3, pcq = (tid & 7) * 4;      // row piece of the hops / epilogue: 4 columns of one row
  const int hrow = tid >> 3, ho = tid & 7;             // head item: one output of one row
  bf16x8 nb[2][3];                                     // the NEXT unit's weight fragments of this wave (requested one unit ahead)
  auto load_conv = [&](int b, int l) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(p.wpack + (size_t)b * blk_words + WP_CONV0 + l * WP_CONV_STRIDE + WP_CONV_FWD);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) nb[kg][pl] = src[((mm * 2 + kg) * 3 + pl) * 64 + lane];
  };
  auto load_head = [&](int b) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(p.wpack + (size_t)b * blk_words + wp_head(n_hh) + WP_HEAD_FWD);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) nb[kg][pl] = src[(kg * 3 + pl) * 64 + lane];
  };
  if (mma_wave) load_conv(0, 0);
  [[maybe_unused]] int sidx = 0;
  SSTAMP(0, true);                 // 0: staging issued
  wg_barrier();
  SSTAMP(0, true);                 // 1: staged

  for (int b = 0; b < NB; ++b) {
    const float* wf = reinterpret_cast<const float*>(p.wpack + (size_t)b * blk_words);
    const int dout = (b == NB - 1) ? p.dims.dout_last : p.dims.dout_inner;
    const bool skip = (b == NB - 1) ? p.dims.skip_last != 0 : p.dims.skip_inner != 0;
    float* act_b = p.acts + (size_t)b * (n_hh + 1) * N * SH;
    // ---- edge MLP: S[r][j] = sum over incoming edges of relu(b1 + W1 [x_r | x_src | ea])
    {
      float w[SW1LD];
#pragma unroll
      for (int q4 = 0; q4 < SW1LD / 4; ++q4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wf + WP_W1 + c32 * SW1LD + q4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q4 * 4 + q] = v[q];
      }
      const float bb = wf[WP_B1 + c32];
      const int stream = wave * 2 + half;
#pragma unroll 1
      for (int t = 0; t < n_et; ++t) {
        const int r = stream + 16 * t;
        const f32x4 ta = *reinterpret_cast<const f32x4*>(x8 + r * SFN), tb = *reinterpret_cast<const f32x4*>(x8 + r * SFN + 4);
        float acc = 0.f;
#pragma unroll 2
        for (int k = 0; k < D; ++k) {      // branch-free: empty slots read row 0 and contribute nothing
          const int o = other[k * STM + r];
          const int oc = o < 0 ? 0 : o;
          const f32x4 sa = *reinterpret_cast<const f32x4*>(x8 + oc * SFN), sb = *reinterpret_cast<const f32x4*>(x8 + oc * SFN + 4);
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(eaL + (k * STM + r) * 8), e1 = *reinterpret_cast<const f32x4*>(eaL + (k * STM + r) * 8 + 4);
          const float z = edge_z(w, bb, ta, tb, sa, sb, e0, e1);
          acc += o >= 0 ? relu_nan(z) : 0.f;
        }
        Xs[r * SXLD + c32] = acc;
        if (r < R) act_b[(size_t)(ts + r) * SH + c32] = acc;
      }
    }
    SSTAMP(0, b == 0);             // 2: edge phase done
    wg_barrier();
    SSTAMP(0, b == 0);             // 3

    // ---- the H -> H layers
    for (int l = 0; l < n_hh; ++l) {
      const float* cw = wf + WP_CONV0 + l * WP_CONV_STRIDE;
      const uint32_t did = p.drop_state ? (uint32_t)(b * p.drop_stride + l + 1) : 0u;
      if (mma_wave) {
        bf16x8 bq[2][3];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bq[kg][pl] = nb[kg][pl];
        const float* arow = Xs + (rb * 32 + c32) * SXLD + half * 8;
        const f32x16 acc = mma6(zero16(), arow, bq[0][0], bq[0][1], bq[0][2]);            // two independent chains
        const f32x16 acc1 = mma6(zero16(), arow + 16, bq[1][0], bq[1][1], bq[1][2]);
        // request the next unit's fragments now: they arrive during the hops
        if (l + 1 < n_hh) load_conv(b, l + 1);
        else if (mm == 0) load_head(b);
        float* g = G + mm * (STM * SXLD) + (rb * 32) * SXLD + c32;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[acc_row(r, half) * SXLD] = acc[r] + acc1[r];
      }
      // (the dropout multipliers depend on (row, column, layer) only: computed here, off the critical path)
      f32x4 mult = {1.f, 1.f, 1.f, 1.f};
      if (did) mult = dropout_mult4(dseed, doff, did, (uint32_t)(ts + prow), (uint32_t)(pcq >> 2), p.drop_thr, p.drop_scale);
      f32x4 v = *reinterpret_cast<const f32x4*>(cw + WP_CONV_BIAS + pcq);      // bias (+ folded rank-3 term of conv 0)
      if (l == 0) {
        const f32x4 ps = *reinterpret_cast<const f32x4*>(dps + prow * 4);
#pragma unroll
        for (int m = 0; m < SNM; ++m) v += *reinterpret_cast<const f32x4*>(cw + WP_CONV_PRE + m * SH + pcq) * ps[m];
      }
      SSTAMP(0, b == 0);           // 4 + 6 l: MFMA + mask done
      wg_barrier();
      SSTAMP(0, b == 0);           // 5 + 6 l
      {   // hop 1: G_1 <- G_1 + A G_2
        float* own = G + 1 * (STM * SXLD) + prow * SXLD + pcq;
        const float* src = G + 2 * (STM * SXLD) + pcq;
        f32x4 U = *reinterpret_cast<const f32x4*>(own);
        for (int k = 0; k < D; ++k) {
          const int2 en = ellw[k * STM + prow];
          U += *reinterpret_cast<const f32x4*>(src + en.x * SXLD) * __int_as_float(en.y);
        }
        *reinterpret_cast<f32x4*>(own) = U;
      }
      SSTAMP(0, b == 0);           // 6 + 6 l: hop 1 done
      wg_barrier();
      SSTAMP(0, b == 0);           // 7 + 6 l
      {   // hop 2 + epilogue: out = G_0 + A G_1 + bias terms -> dropout -> ReLU -> HBM and the tile
        const float* src = G + 1 * (STM * SXLD) + pcq;
        f32x4 U = *reinterpret_cast<const f32x4*>(G + prow * SXLD + pcq);
        for (int k = 0; k < D; ++k) {
          const int2 en = ellw[k * STM + prow];
          U += *reinterpret_cast<const f32x4*>(src + en.x * SXLD) * __int_as_float(en.y);
        }
        v += U;
        v *= mult;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = prow < R ? relu_nan(v[q]) : 0.f;
        if (prow < R) *reinterpret_cast<f32x4*>(act_b + ((size_t)(l + 1) * N + ts + prow) * SH + pcq) = v;
        *reinterpret_cast<f32x4*>(Xs + prow * SXLD + pcq) = v;
      }
      SSTAMP(0, b == 0);           // 8 + 6 l: hop 2 + epilogue done
      wg_barrier();
      SSTAMP(0, b == 0);           // 9 + 6 l
    }

    // ---- head: H -> dout, three matrices side by side in one 32-column group
    if (mma_wave && mm == 0) {
      bf16x8 bq[2][3];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[kg][pl] = nb[kg][pl];
      f32x16 acc = zero16();
      const float* arow = Xs + (rb * 32 + c32) * SXLD + half * 8;
      acc = mma6(acc, arow, bq[0][0], bq[0][1], bq[0][2]);
      acc = mma6(acc, arow + 16, bq[1][0], bq[1][1], bq[1][2]);
      float* g = G + (rb * 32) * SXLD + c32;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[acc_row(r, half) * SXLD] = acc[r];
    }
    if (mma_wave && b + 1 < NB) load_conv(b + 1, 0);
    const float hbias = ho < dout ? wf[wp_head(n_hh) + WP_HEAD_BIAS + ho] : 0.f;
    wg_barrier();
    if (ho < dout) {      // T1 = G_1 + A G_2  (dout-wide)
      float t1 = G[hrow * SXLD + dout + ho];
      for (int k = 0; k < D; ++k) {
        const int2 en = ellw[k * STM + hrow];
        t1 = fmaf(__int_as_float(en.y), G[en.x * SXLD + 2 * dout + ho], t1);
      }
      G[STM * SXLD + hrow * SXLD + ho] = t1;
    }
    wg_barrier();
    if (ho < dout) {
      float o = G[hrow * SXLD + ho];
      for (int k = 0; k < D; ++k) {
        const int2 en = ellw[k * STM + hrow];
        o = fmaf(__int_as_float(en.y), G[STM * SXLD + en.x * SXLD + ho], o);
      }
      o += hbias;
      if (skip) o += x8[hrow * SFN + ho];
      if (b + 1 < NB) {
        x8[hrow * SFN + ho] = hrow < R ? o : 0.f;      // the next block's input stays in LDS
        if (hrow < R) p.xs[((size_t)(b + 1) * N + ts + hrow) * SFN + ho] = o;
      } else if (hrow < R) {
        p.out[(int64_t)(ts + hrow) * p.ldo + ho] = o;
      }
    }
    wg_barrier();
    SSTAMP(0, b == 0);             // 4 + 6 n_hh: head done
  }
  SSTAMP(0, true);                 // last: all blocks done
}

static size_t stack_fwd_lds(int D) { return (size_t)(STM * SXLD * 4 + STM * SFN + STM * 4 + D * STM * 8) * 4 + (size_t)D * STM * 8 + (size_t)D * STM * 4; }

// =====================================================================================================================
// backward
// =====================================================================================================================
constexpr int SZ2 = 68;       // row stride of [Z_1 | Z_2] (64 columns + 4: 16-byte aligned rows)
constexpr int STLD = 68;      // row stride of the transposed fp32 copies [column][64 rows]: 16-byte rows of 4 consecutive tile
                              // rows, conflict-free ds_read_b128 for 32 lanes on 32 consecutive columns (68 c mod 64 = 4 c)

// acc[j][i] += sum over the tile's rows of Z[row][j] a[row][i] as bf16x6 on the bf16 matrix pipe (fp32-accurate, dss2_common.hpp).
// Both operands come from the TRANSPOSED fp32 copies, where the contraction index (the tile row) is the fast one: lane
// (c32, half) reads rows 16 kg + 8 half .. + 7 of its column as two ds_read_b128 -- exactly an A / B fragment of
// v_mfma_f32_32x32x16_bf16 -- and splits them in registers.  4 k-groups x 6 MFMAs = 768 cycles of matrix pipe per block (the
// fp32 form, 32 x v_mfma_f32_32x32x2_f32, held the pipe for 2048 and starved the data-gradient waves of the same SIMD).
__device__ __forceinline__ void split8v(const f32x4 v0, const f32x4 v1, bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t sh[4], sm[4], sl[4];
  split3_pair(v0[0], v0[1], sh[0], sm[0], sl[0]);
  split3_pair(v0[2], v0[3], sh[1], sm[1], sl[1]);
  split3_pair(v1[0], v1[1], sh[2], sm[2], sl[2]);
  split3_pair(v1[2], v1[3], sh[3], sm[3], sl[3]);
  h = __builtin_bit_cast(bf16x8, u32x4{sh[0], sh[1], sh[2], sh[3]});
  m = __builtin_bit_cast(bf16x8, u32x4{sm[0], sm[1], sm[2], sm[3]});
  l = __builtin_bit_cast(bf16x8, u32x4{sl[0], sl[1], sl[2], sl[3]});
}
__device__ __forceinline__ void split8(const float* src, bf16x8& h, bf16x8& m, bf16x8& l) {
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  uint32_t sh[4], sm[4], sl[4];
  split3_pair(v0[0], v0[1], sh[0], sm[0], sl[0]);
  split3_pair(v0[2], v0[3], sh[1], sm[1], sl[1]);
  split3_pair(v1[0], v1[1], sh[2], sm[2], sl[2]);
  split3_pair(v1[2], v1[3], sh[3], sm[3], sl[3]);
  h = __builtin_bit_cast(bf16x8, u32x4{sh[0], sh[1], sh[2], sh[3]});
  m = __builtin_bit_cast(bf16x8, u32x4{sm[0], sm[1], sm[2], sm[3]});
  l = __builtin_bit_cast(bf16x8, u32x4{sl[0], sl[1], sl[2], sl[3]});
}

__device__ __forceinline__ f32x16 wgrad_block(f32x16 acc, const float* zt /* ZT + (col block + c32) * STLD + 8 half */,
                                              const float* at /* AT + c32 * STLD + 8 half */, bool short_t) {
#pragma unroll 1
  for (int kg = 0; kg < (short_t ? 2 : 4); ++kg) {
    bf16x8 ah, am, al, bh, bm, bl;
    split8(zt + 16 * kg, ah, am, al);
    split8(at + 16 * kg, bh, bm, bl);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
  return acc;
}

// column sums of a [64][ld] fp32 tile, 32 columns, ONE wave: lane (r8 = lane >> 3, cq = 4 (lane & 7)) adds rows r8 + 8 i; the
// eight row lanes meet by one shuffle and a wave-private LDS scratch [4][32] (one round trip: cross-lane shuffles are ds_bpermute, a
// dependent chain of three LDS latencies per value); lanes 0..31 return the sum of their column.  `scale`: optional per-row
// factor [64][4], column m.
__device__ __forceinline__ float colsum32(const float* tile, int ld, int lane, const float* scale, int m, float* scratch) {
  const int r8 = lane >> 3, cq = (lane & 7) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = r8 + 8 * i;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * ld + cq);
    if (scale) s += v * scale[row * 4 + m]; else s += v;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], 32);      // rows r8 and r8 + 4 (one cross-lane step, four values in flight)
  wave_lds_sync();      // (the previous call's readers are done)
  if (lane < 32) *reinterpret_cast<f32x4*>(scratch + r8 * SH + cq) = s;
  wave_lds_sync();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) t += scratch[k * SH + (lane & 31)];
  return t;
}

__global__ void __launch_bounds__(512, 1) stack_bwd_kernel(const dss2_stack_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = p.ell_width, DT = p.ellT_width;
  float* Z12 = smem;                                  // [64][68]: Z_1 = A^T g | Z_2 = (A^T)^2 g (columns 0..31 | 32..63); U0 | U1 in the edge phase
  float* A0 = Z12 + STM * SZ2;                        // [64][36] x 2: the unit's saved input activation (ping-pong)
  float* A1 = A0 + STM * SXLD;
  float* Z0a = A1 + STM * SXLD;                       // [64][36] x 2: Z_0 = g of the unit (ping-pong: the data-gradient waves write the
  float* Z0b = Z0a + STM * SXLD;                      //   gated gradient of the unit below while the others still read this unit's)
  float* ZT12 = Z0b + STM * SXLD;                     // [64][68]: Z_1, Z_2 transposed (weight-gradient operands)
  float* ZT0a = ZT12 + 2 * SH * STLD;                 // [32][68] x 2: Z_0 transposed (ping-pong like Z_0)
  float* ZT0b = ZT0a + SH * STLD;
  float* AT0 = ZT0b + SH * STLD;                      // [32][68] x 2: the activation transposed
  float* AT1 = AT0 + SH * STLD;
  float* x8 = AT1 + SH * STLD;                        // [64][8] block input
  float* gx = x8 + STM * SFN;                         // [64][8] gradient of the block output
  float* dps = gx + STM * SFN;                        // [64][4]
  float* w1L = dps + STM * 4;                         // [32][24]
  float* w1T = w1L + SH * SW1LD;                      // [16][36]: W1[:, 0:16] transposed (dx)
  float* accS = w1T + 16 * SXLD;                      // bias sums: db[l][32] (l < 7) | dbf[3][32] | dbh[8] | dW1 [32][24] (col 22 = db1)
  constexpr int ACC_DBF = S_MAX_HH * SH, ACC_DBH = ACC_DBF + SNM * SH, ACC_W1 = ACC_DBH + 8, ACC_WORDS = ACC_W1 + SH * SW1LD;
  float* csum = accS + ACC_WORDS;                     // [4][4][32] wave-private scratches of the bias sums
  bf16x8* wfrag = reinterpret_cast<bf16x8*>(csum + 4 * 4 * SH);         // [6 k-groups][3 planes][64 lanes]: the unit's data-gradient operand
  float* eaL = reinterpret_cast<float*>(wfrag + 6 * 3 * 64);        // [D][64][8]
  float* eaT = eaL + D * STM * 8;                     // [DT][64][8]
  int2* ellTw = reinterpret_cast<int2*>(eaT + DT * STM * 8);    // [DT][64]
  int* other = reinterpret_cast<int*>(ellTw + DT * STM);        // [D][64]
  int* otherT = other + D * STM;                                // [DT][64]
  const int tid = threadIdx.x, lane = tid & 63, c32 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_hh = p.dims.n_hh, NB = p.dims.n_blocks;
  const int64_t N = p.n_nodes;
  const uint64_t dseed = p.drop_state ? p.drop_state[0] : 0, doff = p.drop_state ? p.drop_state[1] : 0;
  const int blk_words = wp_block_words(n_hh);
  const int prow = tid >> 3, pcq = (tid & 7) * 4, ho = tid & 7;
  const int bsz_inner = fl_block(n_hh, p.dims.dout_inner);
  // (ping-pong buffers are addressed as base + parity * size: an ARRAY of LDS pointers indexed at run time lives in private memory
  //  as generic pointers, and every access through one becomes a flat_load / flat_store that waits for vmcnt(0) AND lgkmcnt(0))
  auto Abuf = [&](int par) { return A0 + (par & 1) * (STM * SXLD); };
  auto ATbuf = [&](int par) { return AT0 + (par & 1) * (SH * STLD); };
  auto Z0buf = [&](int par) { return Z0a + (par & 1) * (STM * SXLD); };
  auto ZT0buf = [&](int par) { return ZT0a + (par & 1) * (SH * STLD); };
  // (pad rows of the Z_0 buffers: 32-row tiles never write rows 32..63, the hops and column sums read them)
  for (int idx = tid; idx < 2 * STM * SXLD; idx += 512) Z0a[idx] = 0.f;
  const bool short_t = p.tm <= 32;      // 32-row tiles: rows 32..63 are zero padding everywhere
  const int n_et = short_t ? 2 : STM / 16;
  const bool one_tile = (int)gridDim.x >= p.ntiles;      // small batches: the dx hand-over between blocks stays in LDS (gx)
  int staged_tile = -1;      // the tile whose block-independent operands (ELL slices, edge features, row scales) are in LDS

  for (int b = NB - 1; b >= 0; --b) {
    const float* wf = reinterpret_cast<const float*>(p.wpack + (size_t)b * blk_words);
    const int dout = (b == NB - 1) ? p.dims.dout_last : p.dims.dout_inner;
    const bool skip = (b == NB - 1) ? p.dims.skip_last != 0 : p.dims.skip_inner != 0;
    const bool need_dx = b > 0 || p.dx_out != nullptr;
    const float* act_b = p.acts + (size_t)b * (n_hh + 1) * N * SH;
    const float* xin = b == 0 ? p.x : p.xs + (size_t)b * N * SFN;
    const int64_t ldxin = b == 0 ? p.ldx : SFN;
    // persistent over this workgroup's tiles: weight-gradient accumulators
    f32x16 acc0 = zero16(), acc1 = zero16(), acc2 = zero16();
    const float bb = wf[WP_B1 + c32];
    for (int idx = tid; idx < SH * SW1LD; idx += 512) {
      const float v = wf[WP_W1 + idx];
      w1L[idx] = v;
      const int j = idx / SW1LD, q = idx - j * SW1LD;
      if (q < 2 * SFN) w1T[q * SXLD + j] = v;
    }
    for (int idx = tid; idx < ACC_WORDS; idx += 512) accS[idx] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
      const int ts = p.tile_start[tile], R = p.tile_start[tile + 1] - ts;
      [[maybe_unused]] int sidx = 0;
      #if defined(DSS2_SSTAMP_BLOCK_BACK) && defined(DSS2_SSTAMP_TILE)      // (steady state instead of the kernel's cold first tile)
      [[maybe_unused]] const bool st_on = b == NB - 1 - DSS2_SSTAMP_BLOCK_BACK && tile == (int)(blockIdx.x + DSS2_SSTAMP_TILE * gridDim.x);
#else
      [[maybe_unused]] const bool st_on = b == NB - 1 && tile == (int)blockIdx.x;
#endif
      SSTAMP(1, st_on);            // 0
      // ---- stage the tile
      for (int idx = tid; idx < STM * SFN; idx += 512) {
        const int r = idx >> 3, c = idx & 7;
        x8[idx] = r < R ? xin[(int64_t)(ts + r) * ldxin + c] : 0.f;
        if (b == NB - 1 || !one_tile) {      // (one tile per workgroup: the block above has left its dx in gx)
          float g = 0.f;
          if (r < R) {
            if (b == NB - 1) { if (c < dout) g = p.gout[(int64_t)(ts + r) * p.ldg + c]; }
            else g = __hip_atomic_load(p.dxbuf + (int64_t)(ts + r) * SFN + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (from L2: see the block end)
          }
          gx[idx] = g;
        }
      }
      if (tile != staged_tile) {      // (a workgroup with ONE tile -- small batches -- stages these once for all blocks)
        if (tid < STM) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (tid < R) v = *reinterpret_cast<const f32x4*>(p.deg_pows + (int64_t)(ts + tid) * 4);
          *reinterpret_cast<f32x4*>(dps + tid * 4) = v;
        }
        stage_ell_w(p.ellT_w, tile, DT, p.tm, ellTw, tid, 512);
      }
      {   // the head's data-gradient operand fragments (2 k-groups)
        const bf16x8* bsrc = reinterpret_cast<const bf16x8*>(p.wpack + (size_t)b * blk_words + wp_head(n_hh) + WP_HEAD_BWD);
        if (tid < 2 * 3 * 64) wfrag[tid] = bsrc[tid];
      }
      {   // the head's input activation h_{n_hh}, row-major (gate) and transposed (weight-gradient operand)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (prow < R) v = *reinterpret_cast<const f32x4*>(act_b + ((size_t)n_hh * N + ts + prow) * SH + pcq);
        *reinterpret_cast<f32x4*>(Abuf(n_hh & 1) + prow * SXLD + pcq) = v;
        float* at = ATbuf(n_hh & 1) + pcq * STLD + prow;
#pragma unroll
        for (int q = 0; q < 4; ++q) at[q * STLD] = v[q];
      }
      // The tile's edge tables and edge features (20 KB, only the edge passes at the END of the tile read them): requested here,
      // AFTER every load the staging barrier waits for (vector loads return in order),
      // into registers, stored to LDS at the top of the second unit -- a global round trip off the staging barrier's critical path
      // (threads 0..255: the by-target table, 256..511: the by-source table; D, DT <= 4: one item per thread).
      asm volatile("" ::: "memory");      // (keeps the requests below behind the stores above)
      const bool e_pend = tile != staged_tile;
      int2 e_en = make_int2(0, -1);
      f32x4 e_a = {0.f, 0.f, 0.f, 0.f}, e_c = {0.f, 0.f, 0.f, 0.f};
      const int e_tr = tid >> 8, e_idx = tid & 255, e_W = e_tr ? DT : D;
      if (e_pend && e_idx < e_W * STM) {
        const int k = e_idx >> 6, r = e_idx & 63;
        if (r < p.tm) e_en = (reinterpret_cast<const int2*>(e_tr ? p.ellT_e : p.ell_e) + (size_t)tile * e_W * p.tm)[k * p.tm + r];
        const float* ec = p.eacache + ((size_t)tile * (D + DT) + (e_tr ? D : 0)) * STM * 8 + e_idx * 8;
        e_a = *reinterpret_cast<const f32x4*>(ec);
        e_c = *reinterpret_cast<const f32x4*>(ec + 4);
      }
      staged_tile = tile;
      SSTAMP(1, st_on);            // 1: staging issued
      wg_barrier();
      SSTAMP(1, st_on);            // 2: staged
      // this thread's row of the transposed ELL slice: the same for every hop of every unit of the tile (DT <= 4)
      int2 en4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) en4[k] = k < DT ? ellTw[k * STM + prow] : make_int2(prow, 0);

      // ---- units u = n_hh (head), n_hh - 1 .. 0 (H -> H layers; conv 0 folded)
      // Per unit: two hops (Z_1, Z_2 and their transposed copies), then ONE phase in which three waves accumulate the weight
      // gradients, one wave the bias sums, and four waves compute the data gradient AND gate it: D^T = Wcat^T Z^T on
      // v_mfma_f32_16x16x32_bf16 (bf16x6), whose accumulator gives a lane 4 consecutive input features of ONE tile row -- the
      // 16-byte row piece the gate (saved activation > 0, regenerated dropout mask) and the next unit's Z_0 want, so the
      // gradient never visits LDS between the GEMM and the gate (round 3, first form: 32x32 blocks split over k halves,
      // exchanged through LDS and gated in a phase of its own: 5.7 K of a unit's 7.6 K cycles).  Z_0 is double-buffered.
      f32x4 apf = {0.f, 0.f, 0.f, 0.f};
      bf16x8 wpf[3];
      for (int u = n_hh; u >= 0; --u) {
        const bool head = u == n_hh;
        float* Z0 = Z0buf(u & 1);
        float* ZT0 = ZT0buf(u & 1);
        float* Acur = Abuf(u & 1);
        if (!head) {
          // this unit's operands, requested during the unit above: data-gradient fragments, input activation (both forms)
#pragma unroll
          for (int i = 0; i < 3; ++i) { const int f = tid + 512 * i; if (f < 6 * 3 * 64) wfrag[f] = wpf[i]; }
          *reinterpret_cast<f32x4*>(Acur + prow * SXLD + pcq) = apf;
          float* at = ATbuf(u & 1) + pcq * STLD + prow;
#pragma unroll
          for (int q = 0; q < 4; ++q) at[q * STLD] = apf[q];
          if (u == n_hh - 1 && e_pend && e_idx < e_W * STM) {      // the deferred edge staging (see above)
            (e_tr ? otherT : other)[e_idx] = e_en.y != -1 ? e_en.x : -1;
            float* dst = (e_tr ? eaT : eaL) + e_idx * 8;
            *reinterpret_cast<f32x4*>(dst) = e_a;
            *reinterpret_cast<f32x4*>(dst + 4) = e_c;
          }
        }
        // the operands of the unit below: L2 -> registers now, -> LDS at the top of that unit (a whole unit of latency cover)
        apf = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u > 0 && prow < R) apf = *reinterpret_cast<const f32x4*>(act_b + ((size_t)(u - 1) * N + ts + prow) * SH + pcq);
        if (u > 0) {
          const bf16x8* bsrc = reinterpret_cast<const bf16x8*>(p.wpack + (size_t)b * blk_words + WP_CONV0 + (u - 1) * WP_CONV_STRIDE + WP_CONV_BWD);
#pragma unroll
          for (int i = 0; i < 3; ++i) wpf[i] = bsrc[tid + 512 * i < 6 * 3 * 64 ? tid + 512 * i : 0];
        }
        // wave roles of this unit: weight gradient blocks q = 3 u + k on wave q mod 8 (slot q / 8); data gradient + gate on the
        // next four waves; bias sums on the last one
        const int q0 = 3 * u;
        const int di = (wave - (q0 + 3)) & 7;            // 0..3: data-gradient wave
        const bool sum_wave = wave == ((q0 + 7) & 7);
        if (head) {
          // Z_0[:, 0:dout] = g, then two dout-wide hops inside the same 32 columns; columns 3 dout .. 31 zero (every value also
          // into the transposed copy)
          if (ho < dout) { const float g = gx[prow * SFN + ho]; Z0[prow * SXLD + ho] = g; ZT0[ho * STLD + prow] = g; }
          for (int c = 3 * dout + ho; c < 32; c += 8) { Z0[prow * SXLD + c] = 0.f; ZT0[c * STLD + prow] = 0.f; }
          wg_barrier();
          if (ho < dout) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t = fmaf(__int_as_float(en4[k].y), Z0[en4[k].x * SXLD + ho], t);
            Z0[prow * SXLD + dout + ho] = t; ZT0[(dout + ho) * STLD + prow] = t;
          }
          wg_barrier();
          if (ho < dout) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) t = fmaf(__int_as_float(en4[k].y), Z0[en4[k].x * SXLD + dout + ho], t);
            Z0[prow * SXLD + 2 * dout + ho] = t; ZT0[(2 * dout + ho) * STLD + prow] = t;
          }
          wg_barrier();
        } else {
#pragma unroll
          for (int m = 1; m < SNM; ++m) {
            const float* src = (m == 1 ? Z0 : Z12) + pcq;
            const int sld = m == 1 ? SXLD : SZ2;
            f32x4 zz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) zz[k] = *reinterpret_cast<const f32x4*>(src + en4[k].x * sld);      // all four gathers in flight
            f32x4 U = zz[0] * __int_as_float(en4[0].y);
#pragma unroll
            for (int k = 1; k < 4; ++k) U += zz[k] * __int_as_float(en4[k].y);
            *reinterpret_cast<f32x4*>(Z12 + prow * SZ2 + (m - 1) * SH + pcq) = U;
            float* zt = ZT12 + ((m - 1) * SH + pcq) * STLD + prow;
#pragma unroll
            for (int q = 0; q < 4; ++q) zt[q * STLD] = U[q];
            wg_barrier();
          }
        }
        SSTAMP(1, st_on);          // 3 + 5 i: hops done (i = n_hh - u)
        // ---- ONE phase: weight gradients (3 waves), data gradient + gate (4 waves), bias sums (1 wave)
        {
          const int nblk = head ? 1 : SNM;
          for (int k = 0; k < nblk; ++k) {
            const int q = q0 + k;
            if (wave == (q & 7)) {
              const float* zt = (k == 0 ? ZT0 : ZT12 + (k - 1) * SH * STLD) + c32 * STLD + 8 * half;      // (32-row tiles: k runs over rows 0..31 only)
              const float* at = ATbuf(u & 1) + c32 * STLD + 8 * half;
              const int slot = q >> 3;
              if (slot == 0) acc0 = wgrad_block(acc0, zt, at, short_t);
              else if (slot == 1) acc1 = wgrad_block(acc1, zt, at, short_t);
              else acc2 = wgrad_block(acc2, zt, at, short_t);
              if (u == 0) {      // folded conv 0: dbf_k[c] += sum_rows (A^k deg)[row] g[row][c]  (this wave's share of the bias sums)
                const float sf = colsum32(Z0, SXLD, lane, dps, k, csum + k * (4 * SH));
                if (lane < 32) accS[ACC_DBF + k * SH + lane] += sf;
              }
            }
          }
          if (di < 4) {
            // 64-row tiles: wave di owns tile rows 16 di .. + 15, both 16-feature blocks (the Z fragment is split once for the
            // two); 32-row tiles: (row block di & 1, feature block di >> 1)
            const int l16 = lane & 15, kg = lane >> 4;
            const int row = (short_t ? (di & 1) : di) * 16 + l16;
            const int ib0 = short_t ? (di >> 1) : 0, nib = short_t ? 1 : 2;
            const int nks = head ? 1 : SNM;
            f32x4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            // (requesting every Z fragment and the gate's inputs -- saved activation, dropout multipliers -- ahead of the k-steps
            //  was tried: 30 more live registers spill, 0.309 -> 0.316 ms at B = 64)
#pragma unroll
            for (int ks = 0; ks < SNM; ++ks) {
              if (ks < nks) {
                bf16x8 zh, zm, zl;
                split8((ks == 0 ? Z0 + row * SXLD : Z12 + row * SZ2 + (ks - 1) * SH) + 8 * kg, zh, zm, zl);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  if (j < nib) {
                    const bf16x8* wq = wfrag + ((ib0 + j) * nks + ks) * 3 * 64 + lane;      // [feature block][k-step][plane][lane]
                    const bf16x8 wh = wq[0], wm = wq[64], wl = wq[128];
                    f32x4 c = d[j];      // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, zh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, zm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, zh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, zh, c, 0, 0, 0);
                    d[j] = c;
                  }
                }
              }
            }
            // gate: gradient w.r.t. the pre-activation output of the unit below (u > 0); u == 0: dS, no gate.  Lane: row `row`,
            // features 16 ib + 4 kg .. + 3
            float* Zn = Z0buf((u + 1) & 1);
            float* ZTn = ZT0buf((u + 1) & 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j < nib) {
                const int col = 16 * (ib0 + j) + 4 * kg;
                f32x4 v = d[j];
                if (u > 0) {
                  const f32x4 a = *reinterpret_cast<const f32x4*>(Acur + row * SXLD + col);
                  f32x4 mult = {1.f, 1.f, 1.f, 1.f};
                  if (p.drop_state) mult = dropout_mult4(dseed, doff, (uint32_t)(b * p.drop_stride + u), (uint32_t)(ts + row), (uint32_t)(col >> 2), p.drop_thr, p.drop_scale);
#pragma unroll
                  for (int q = 0; q < 4; ++q) v[q] = relu_open(a[q]) ? v[q] * mult[q] : 0.f;
                }
                if (row >= R) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(Zn + row * SXLD + col) = v;
                if (u > 0) {
                  float* zt = ZTn + col * STLD + row;
#pragma unroll
                  for (int q = 0; q < 4; ++q) zt[q * STLD] = v[q];
                }
              }
            }
          }
          if (sum_wave) {
            if (head) {      // db_head[o] += sum_rows g[row][o]
              float s = 0.f;
              const int o = lane & 7, part = lane >> 3;
#pragma unroll
              for (int i = 0; i < 8; ++i) s += gx[(part * 8 + i) * SFN + o];
              s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
              if (lane < 8 && lane < dout) accS[ACC_DBH + lane] += s;
            } else {         // db_u[c] += sum_rows g[row][c]
              const float s = colsum32(Z0, SXLD, lane, nullptr, 0, csum + 3 * (4 * SH));
              if (lane < 32) accS[u * SH + lane] += s;
            }
          }
        }
        SSTAMP(1, st_on);          // 4 + 5 i: this wave's work of the phase done
        wg_barrier();
        SSTAMP(1, st_on);          // 5 + 5 i
        SSTAMP(1, st_on);          // 6 + 5 i  (the separate gate phase is gone; the stamp layout of tools/sstamps.py is kept)
        SSTAMP(1, st_on);          // 7 + 5 i
      }
      // the gradient w.r.t. the aggregated edge hidden (unit 0's output) sits in the Z_0 buffer of parity 1
      const float* dS = Z0buf(1);
      float* U0 = Z12;             // [64][SZ2]: columns 0..31 (free since the last unit's phase)
      float* U1 = Z12 + SH;

      // ---- edge MLP backward: dS = Z[:, 0:32]
      {
        const int stream = wave * 2 + half;
        float w[SW1LD];      // this lane's row of W1 (from the LDS copy: not kept live across the units above)
#pragma unroll
        for (int q4 = 0; q4 < SW1LD / 4; ++q4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(w1L + c32 * SW1LD + q4 * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q4 * 4 + q] = v[q];
        }
        float dw[SFC], db1 = 0.f;      // this stream's share of dW1 / db1 for THIS tile (summed over the streams below)
#pragma unroll
        for (int q = 0; q < SFC; ++q) dw[q] = 0.f;
#pragma unroll 1
        for (int t = 0; t < n_et; ++t) {      // by target: dW1, db1, U0
          const int r = stream + 16 * t;
          const f32x4 ta = *reinterpret_cast<const f32x4*>(x8 + r * SFN), tb = *reinterpret_cast<const f32x4*>(x8 + r * SFN + 4);
          const float gS = dS[r * SXLD + c32];
          float u0 = 0.f;
          for (int k = 0; k < D; ++k) {
            const int o = other[k * STM + r];
            if (o < 0) continue;
            const f32x4 sa = *reinterpret_cast<const f32x4*>(x8 + o * SFN), sb = *reinterpret_cast<const f32x4*>(x8 + o * SFN + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(eaL + (k * STM + r) * 8), e1 = *reinterpret_cast<const f32x4*>(eaL + (k * STM + r) * 8 + 4);
            const float dz = relu_open(edge_z(w, bb, ta, tb, sa, sb, e0, e1)) ? gS : 0.f;
            u0 += dz;
            db1 += dz;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              dw[q] = fmaf(dz, ta[q], dw[q]); dw[4 + q] = fmaf(dz, tb[q], dw[4 + q]);
              dw[8 + q] = fmaf(dz, sa[q], dw[8 + q]); dw[12 + q] = fmaf(dz, sb[q], dw[12 + q]);
              dw[16 + q] = fmaf(dz, e0[q], dw[16 + q]);
            }
            dw[20] = fmaf(dz, e1[0], dw[20]); dw[21] = fmaf(dz, e1[1], dw[21]);
          }
          U0[r * SZ2 + c32] = u0;
        }
        if (need_dx) {
#pragma unroll 1
          for (int t = 0; t < n_et; ++t) {    // by source: U1
            const int r = stream + 16 * t;
            const f32x4 sa = *reinterpret_cast<const f32x4*>(x8 + r * SFN), sb = *reinterpret_cast<const f32x4*>(x8 + r * SFN + 4);
            float u1 = 0.f;
            for (int k = 0; k < DT; ++k) {
              const int o = otherT[k * STM + r];
              if (o < 0) continue;
              const f32x4 ta = *reinterpret_cast<const f32x4*>(x8 + o * SFN), tb = *reinterpret_cast<const f32x4*>(x8 + o * SFN + 4);
              const f32x4 e0 = *reinterpret_cast<const f32x4*>(eaT + (k * STM + r) * 8), e1 = *reinterpret_cast<const f32x4*>(eaT + (k * STM + r) * 8 + 4);
              u1 += relu_open(edge_z(w, bb, ta, tb, sa, sb, e0, e1)) ? dS[o * SXLD + c32] : 0.f;
            }
            U1[r * SZ2 + c32] = u1;
          }
        }
        // the streams' shares of dW1 / db1: the two halves of a wave meet by shuffle, the eight waves through scratch
        // (ZT: free since the last unit's MFMA phase), in a fixed order
#pragma unroll
        for (int q = 0; q < SFC; ++q) dw[q] += __shfl_xor(dw[q], 32);
        db1 += __shfl_xor(db1, 32);
        if (half == 0) {
          float* red = ZT12 + (wave * SH + c32) * SW1LD;    // [8 waves][32][24] = 6144 floats <= ZT12 | ZT0a | ZT0b (8704, contiguous)
#pragma unroll
          for (int q = 0; q < SFC; ++q) red[q] = dw[q];
          red[SFC] = db1;
        }
      }
      SSTAMP(1, st_on);            // 3 + 5 (n_hh + 1): edge passes done
      wg_barrier();
      for (int idx = tid; idx < SH * SW1LD; idx += 512) {      // fixed-order sum over the 8 waves, accumulated over the tiles
        const int q = idx % SW1LD;
        if (q > SFC) continue;
        float s = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) s += ZT12[st * (SH * SW1LD) + idx];
        accS[ACC_W1 + idx] += s;
      }
      if (need_dx) {      // dx[row][c] = U0[row] . W1[:, c] + U1[row] . W1[:, 8 + c] (+ residual): four partial sums, fixed order
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j4 = 0; j4 < SH / 4; ++j4) {
          const f32x4 u0 = *reinterpret_cast<const f32x4*>(U0 + prow * SZ2 + 4 * j4), u1 = *reinterpret_cast<const f32x4*>(U1 + prow * SZ2 + 4 * j4);
          const f32x4 wa = *reinterpret_cast<const f32x4*>(w1T + ho * SXLD + 4 * j4), wb = *reinterpret_cast<const f32x4*>(w1T + (SFN + ho) * SXLD + 4 * j4);
          s4 += u0 * wa;
          s4 += u1 * wb;
        }
        const float s = (skip ? gx[prow * SFN + ho] : 0.f) + ((s4[0] + s4[1]) + (s4[2] + s4[3]));
        if (b > 0 && one_tile) gx[prow * SFN + ho] = prow < R ? s : 0.f;      // (read above by this very thread only)
        else if (prow < R) {
          if (b > 0) p.dxbuf[(int64_t)(ts + prow) * SFN + ho] = s;
          else p.dx_out[(int64_t)(ts + prow) * SFN + ho] = s;
        }
      }
      wg_barrier();
      SSTAMP(1, st_on);            // 4 + 5 (n_hh + 1): tile done
    }

    // ---- block end: this workgroup's slab of the block's weight gradients
    {
      float* sl = p.slab + (size_t)blockIdx.x * p.slab_stride + (size_t)b * bsz_inner;
      // (the lane offset is made opaque here: otherwise the compiler forms the sixteen 64-bit store addresses of every slot
      //  at the top of the block loop and carries them -- spilled -- through the whole tile loop)
      int lane_off = 4 * half * SH + c32, h4 = 4 * half;
      asm volatile("" : "+v"(lane_off), "+v"(h4));
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int q = 8 * s + wave;
        const f32x16& a = s == 0 ? acc0 : (s == 1 ? acc1 : acc2);
        if (q < 3 * n_hh) {
          float* dst = sl + FL_CONV0 + (q / 3) * FL_CONV_STRIDE + (q % 3) * (SH * SH) + lane_off;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[acc_row(r, 0) * SH] = a[r];
        } else if (q == 3 * n_hh) {
          float* dst = sl + fl_head(n_hh) + lane_off;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const int jj = acc_row(r, 0) + h4; if (jj < SNM * dout) dst[acc_row(r, 0) * SH] = a[r]; }
        }
      }
      int t2 = tid;
      asm volatile("" : "+v"(t2));      // (same reason as lane_off above)
      for (int idx = t2; idx < SH * SW1LD; idx += 512) {
        const int j = idx / SW1LD, q = idx - j * SW1LD;
        if (q < SFC) sl[FL_W1 + j * SFC + q] = accS[ACC_W1 + idx];
        else if (q == SFC) sl[FL_B1 + j] = accS[ACC_W1 + idx];
      }
      for (int idx = t2; idx < n_hh * SH; idx += 512) sl[FL_CONV0 + (idx >> 5) * FL_CONV_STRIDE + SNM * SH * SH + (idx & 31)] = accS[idx];
      if (t2 < SNM * SH) sl[FL_W2 + t2] = accS[ACC_DBF + t2];      // dbf_m travels in the W2 section (the reduce kernel knows)
      if (t2 < dout) sl[fl_head(n_hh) + SNM * dout * SH + t2] = accS[ACC_DBH + t2];
    }
    // The block below reads the dx rows written above through global memory -- but always THIS workgroup's own rows (a workgroup
    // walks the same tiles in every block), so a workgroup-scope hand-off is enough: the waves of a workgroup share their CU's L1.
    // (The device-scope __threadfence() that stood here is an L2 write-back per block on MI355X: 350 us of a 1.44 ms step at
    //  B = 4096.)  The stores are complete in L2 after __syncthreads()'s vmcnt(0); the loads of the next block's staging are
    //  device-scope relaxed atomic loads, served by that L2, so nothing depends on what the CU's L1 holds.
    __syncthreads();
  }
}

static size_t stack_bwd_lds(int D, int DT) {
  const size_t f = (size_t)STM * SZ2 + 4 * STM * SXLD + 4 * SH * STLD + 2 * SH * STLD + 2 * STM * SFN + STM * 4 + SH * SW1LD + 16 * SXLD +
                   (S_MAX_HH * SH + SNM * SH + 8 + SH * SW1LD + 4 * 4 * SH) + (size_t)(D + DT) * STM * 8;
  return f * 4 + (size_t)6 * 3 * 64 * 16 + (size_t)DT * STM * 8 + (size_t)(D + DT) * STM * 4;
}

// =====================================================================================================================
// reduce: fixed-order sum of the slabs + chain rule of the fold
// =====================================================================================================================
// Stage 1: flat[i] = sum over the slabs, for every i (the folded sections hold dWf | db0 | dbf afterwards).  A workgroup owns
// 64 consecutive floats; its 256 threads are 16 slab lanes x 16 float4 columns: lane s sums slabs s, s + 16, .. (independent
// loads, all in flight), the 16 partial sums meet in LDS and are added in lane order -- fixed order, bitwise reproducible.
// The folded sections [W2 | b2 | conv 0] of every block (they hold dbf | .. | dWf | db0 here) are ALSO written to `gsrc`,
// the read-only source of stage 2, which overwrites them in `flat` with the chain rule's results.
__global__ void __launch_bounds__(256) stack_reduce_kernel(const float* __restrict__ slab, int n_slabs, int64_t stride,
                                                           float* __restrict__ flat, int64_t total, float* __restrict__ gsrc,
                                                           int bsz_inner, int n_blocks) {
  __shared__ f32x4 part[16][16];
  const int tid = threadIdx.x, cl = tid & 15, sl = tid >> 4;
  const int64_t i4 = ((int64_t)blockIdx.x * 16 + cl) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 + 4 <= total) {
    int k = sl;
#pragma unroll 1
    for (; k + 48 < n_slabs; k += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (size_t)k * stride + i4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 16) * stride + i4);
      const f32x4 c = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 32) * stride + i4);
      const f32x4 d = *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 48) * stride + i4);
      s += a; s += b; s += c; s += d;
    }
    for (; k < n_slabs; k += 16) s += *reinterpret_cast<const f32x4*>(slab + (size_t)k * stride + i4);
  } else if (i4 < total) {
    for (int k = sl; k < n_slabs; k += 16)
      for (int q = 0; i4 + q < total; ++q) s[q] += slab[(size_t)k * stride + i4 + q];
  }
  part[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && i4 < total) {
    f32x4 t = part[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][cl];
    if (i4 + 4 <= total) *reinterpret_cast<f32x4*>(flat + i4) = t;
    else for (int q = 0; i4 + q < total; ++q) flat[i4 + q] = t[q];
    int b = (int)(i4 / bsz_inner);
    if (b > n_blocks - 1) b = n_blocks - 1;
    const int rel = (int)(i4 - (int64_t)b * bsz_inner);      // (section bounds and block sizes are multiples of 4)
    constexpr int span = FL_CONV0 + FL_CONV_STRIDE - FL_W2;
    if (rel >= FL_W2 && rel < FL_W2 + span) *reinterpret_cast<f32x4*>(gsrc + (size_t)b * span + (rel - FL_W2)) = t;
  }
}

// Stage 2: the chain rule of the fold.  Workgroup (b, r): r < 3 writes dW_r of conv 0, r == 3 writes dW2 and db2, into
// `flat`; all of them read the reduced folded gradients from `gsrc` (stage 1's copy), so no workgroup reads what another
// overwrites.  conv 0's bias gradient (db0) is already in place.
__global__ void __launch_bounds__(256) stack_fold_bwd_kernel(const dss2_stack_dims d, const float* const* __restrict__ params,
                                                             const float* __restrict__ gsrc, float* __restrict__ flat) {
  __shared__ float A[SNM][SH][SH + 1], Bm[SNM][SH][SH + 1], dbf[SNM][SH], b2s[SH];
  const int tid = threadIdx.x;
  const int b = blockIdx.x >> 2, r = blockIdx.x & 3;
  const int bsz_inner = fl_block(d.n_hh, d.dout_inner);
  const float* const* P = params + (size_t)b * params_per_block(d.n_hh);
  const float* gs = gsrc + (size_t)b * (FL_CONV0 + FL_CONV_STRIDE - FL_W2);      // [dbf 96 | pad .. | dWf 3072 | db0 32] = the W2|b2|conv0 span
  const float* dWf = gs + (FL_CONV0 - FL_W2);
  float* fb = flat + (size_t)b * bsz_inner;
  if (tid < SNM * SH) dbf[tid >> 5][tid & 31] = gs[tid];
  if (tid < SH) b2s[tid] = P[3][tid];
  if (r < SNM) {
    // dW_r[j][i] = sum_c dWf_r[j][c] W2[i][c] + dbf_r[j] b2[i]
    for (int idx = tid; idx < SH * SH; idx += 256) { A[0][idx >> 5][idx & 31] = dWf[r * SH * SH + idx]; Bm[0][idx >> 5][idx & 31] = P[2][idx]; }
    __syncthreads();
    for (int idx = tid; idx < SH * SH; idx += 256) {
      const int j = idx >> 5, i = idx & 31;
      float s = dbf[r][j] * b2s[i];
#pragma unroll
      for (int c = 0; c < SH; ++c) s = fmaf(A[0][j][c], Bm[0][i][c], s);
      fb[FL_CONV0 + r * SH * SH + idx] = s;
    }
    return;
  }
  // dW2[i][c] = sum_m sum_j W_m[j][i] dWf_m[j][c];  db2[i] = sum_m sum_j W_m[j][i] dbf_m[j]
  for (int idx = tid; idx < SNM * SH * SH; idx += 256) {
    const int m = idx >> 10, j = (idx >> 5) & 31, c = idx & 31;
    A[m][j][c] = dWf[idx];
    Bm[m][j][c] = P[4 + 1 + m][j * SH + c];
  }
  __syncthreads();
  for (int idx = tid; idx < SH * SH; idx += 256) {
    const int i = idx >> 5, c = idx & 31;
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < SNM; ++m)
      for (int j = 0; j < SH; ++j) s = fmaf(Bm[m][j][i], A[m][j][c], s);
    fb[FL_W2 + idx] = s;
  }
  if (tid < SH) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < SNM; ++m)
      for (int j = 0; j < SH; ++j) s = fmaf(Bm[m][j][tid], dbf[m][j], s);
    fb[FL_B2 + tid] = s;
  }
}

static bool dims_ok(const dss2_stack_dims& d) {
  return d.n_blocks >= 1 && d.n_blocks <= 64 && d.n_hh >= 1 && d.n_hh <= S_MAX_HH && d.dout_last >= 1 && d.dout_last <= 8 &&
         (d.n_blocks == 1 || d.dout_inner == SFN) && (!d.skip_last || d.dout_last == SFN);
}

}  // namespace dss2

using namespace dss2;

#ifdef DSS2_STACK_STAMPS
extern "C" int dss2_debug_read_sstamps(unsigned long long* host_out, int which) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_sstamps), sizeof(unsigned long long) * 64 * 8 * 128,
                                  sizeof(unsigned long long) * 64 * 8 * 128 * (size_t)which);
}
#endif

extern "C" int64_t dss2_stack_wpack_words(const dss2_stack_dims* d) { return (int64_t)d->n_blocks * wp_block_words(d->n_hh); }

extern "C" int64_t dss2_stack_fold_scratch_floats(const dss2_stack_dims* d) { return (int64_t)d->n_blocks * (FL_CONV0 + FL_CONV_STRIDE - FL_W2); }

extern "C" int64_t dss2_stack_flat_floats(const dss2_stack_dims* d) {
  return (int64_t)(d->n_blocks - 1) * fl_block(d->n_hh, d->dout_inner) + fl_block(d->n_hh, d->dout_last);
}

extern "C" int dss2_stack_supported(const dss2_stack_dims* d, int hid, int nmat, int fn, int fe, int nrb, int ell_width, int ellT_width) {
  return (dims_ok(*d) && hid == SH && nmat == SNM && fn == SFN && fe == 6 && (nrb == 1 || nrb == 2) && ell_width >= 1 &&
          ell_width <= S_MAX_ELL && ellT_width >= 1 && ellT_width <= S_MAX_ELL) ? 1 : 0;
}

static int dss2_stack_pack_launch(const dss2_stack_dims* d, const float* const* params, uint32_t* wpack, uint64_t* rng_state, uint64_t* rng_snapshot, uint64_t host_seed, int use_host_seed, float* tick, const dss2_stack_args* tiles, void* stream);
extern "C" int dss2_stack_pack(const dss2_stack_dims* d, const float* const* params, uint32_t* wpack, uint64_t* rng_state, uint64_t* rng_snapshot, uint64_t host_seed, int use_host_seed, float* tick, const dss2_stack_args* tiles, void* stream) {
  if (!d) { dss2::set_error("dss2_stack_pack: null argument"); return 2; }
  DSS2_RECORD([dd = *d, params, wpack, rng_state, rng_snapshot, host_seed, use_host_seed, tick, has_t = tiles != nullptr, tt = tiles ? *tiles : dss2_stack_args{}](void* s_) { return dss2_stack_pack_launch(&dd, params, wpack,      /* (params: a DEVICE table of pointers) */ rng_state, rng_snapshot, host_seed, use_host_seed, tick, has_t ? &tt : nullptr, s_); });
  return dss2_stack_pack_launch(d, params, wpack, rng_state, rng_snapshot, host_seed, use_host_seed, tick, tiles, stream);
}
static int dss2_stack_pack_launch(const dss2_stack_dims* d, const float* const* params, uint32_t* wpack, uint64_t* rng_state, uint64_t* rng_snapshot, uint64_t host_seed, int use_host_seed, float* tick, const dss2_stack_args* tiles, void* stream) {
  if (!dims_ok(*d) || !params || !wpack) { set_error("stack_pack: bad arguments"); return 2; }
  if (rng_snapshot && !use_host_seed && !rng_state) { set_error("stack_pack: rng_state missing"); return 2; }
  EaPrep ep = {};
  if (tiles) {      // also gather the per-tile edge-feature cache of this forward call (args: ea, ELL slices, tiles, eacache)
    const dss2_stack_args& a = *tiles;
    if (!a.ea || !a.ell_e || !a.ellT_e || !a.eacache || a.ntiles < 1 || (a.tm != 32 && a.tm != 64) || a.ell_width < 1 ||
        a.ell_width > S_MAX_ELL || a.ellT_width < 1 || a.ellT_width > S_MAX_ELL) { set_error("stack_pack: bad tile arguments"); return 2; }
    ep = EaPrep{a.ea, a.ldea, a.ell_e, a.ellT_e, a.ell_width, a.ellT_width, a.tm, a.ntiles, a.eacache};
  }
  hipLaunchKernelGGL(stack_pack_kernel, dim3(d->n_blocks * (d->n_hh * SNM + 2) + ep.ntiles), dim3(256), 0, as_stream(stream), *d, params,
                     wpack, reinterpret_cast<unsigned long long*>(rng_state), reinterpret_cast<unsigned long long*>(rng_snapshot),
                     (unsigned long long)host_seed, use_host_seed, tick, ep);
  return check_launch("stack_pack");
}

static int stack_args_ok(const dss2_stack_args& a, const char* what) {
  if (!dims_ok(a.dims)) { set_error("%s: unsupported dimensions", what); return 2; }
  if (a.tm != 32 && a.tm != 64) { set_error("%s: tiles of %d rows (needs 32 or 64)", what, a.tm); return 2; }
  if (a.ell_width < 1 || a.ell_width > S_MAX_ELL) { set_error("%s: ELL width %d", what, a.ell_width); return 2; }
  if (!a.x || !a.ea || !a.wpack || !a.tile_start || !a.ell_w || !a.ell_e || !a.deg_pows || !a.acts || !a.eacache || (a.dims.n_blocks > 1 && !a.xs)) {
    set_error("%s: null argument", what); return 2;
  }
  return 0;
}

static int dss2_stack_forward_launch(const dss2_stack_args* ap, void* stream);
extern "C" int dss2_stack_forward(const dss2_stack_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_stack_forward: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_stack_forward_launch(&a, s_); });
  return dss2_stack_forward_launch(ap, stream);
}
static int dss2_stack_forward_launch(const dss2_stack_args* ap, void* stream) {
  const dss2_stack_args& a = *ap;
  if (a.ntiles <= 0) return 0;
  if (int rc = stack_args_ok(a, "stack_forward")) return rc;
  if (!a.out) { set_error("stack_forward: null output"); return 2; }
  if (a.ellT_width < 1 || a.ellT_width > S_MAX_ELL) { set_error("stack_forward: ellT_width (the edge-feature cache's stride) missing"); return 2; }
  static std::atomic<uint32_t> done{0};
  if (ensure_max_lds(reinterpret_cast<const void*>(stack_fwd_kernel), done, "stack_forward")) return 1;
  hipLaunchKernelGGL(stack_fwd_kernel, dim3(a.ntiles), dim3(512), stack_fwd_lds(a.ell_width), as_stream(stream), a);
  return check_launch("stack_forward");
}

static int dss2_stack_backward_launch(const dss2_stack_args* ap, void* stream);
extern "C" int dss2_stack_backward(const dss2_stack_args* ap, void* stream) {
  if (!ap) { dss2::set_error("dss2_stack_backward: null argument"); return 2; }
  DSS2_RECORD([a = *ap](void* s_) { return dss2_stack_backward_launch(&a, s_); });
  return dss2_stack_backward_launch(ap, stream);
}
static int dss2_stack_backward_launch(const dss2_stack_args* ap, void* stream) {
  const dss2_stack_args& a = *ap;
  if (a.ntiles <= 0) return 0;
  if (int rc = stack_args_ok(a, "stack_backward")) return rc;
  if (a.ellT_width < 1 || a.ellT_width > S_MAX_ELL || !a.ellT_w || !a.ellT_e) { set_error("stack_backward: transposed ELL slices missing"); return 2; }
  if (!a.gout || !a.slab || a.n_wg < 1 || a.n_wg > a.ntiles || (a.dims.n_blocks > 1 && !a.dxbuf)) { set_error("stack_backward: bad arguments"); return 2; }
  if (a.slab_stride < dss2_stack_flat_floats(&a.dims)) { set_error("stack_backward: slab stride too small"); return 2; }
  static std::atomic<uint32_t> done{0};
  if (ensure_max_lds(reinterpret_cast<const void*>(stack_bwd_kernel), done, "stack_backward")) return 1;
  hipLaunchKernelGGL(stack_bwd_kernel, dim3(a.n_wg), dim3(512), stack_bwd_lds(a.ell_width, a.ellT_width), as_stream(stream), a);
  return check_launch("stack_backward");
}

static int dss2_stack_reduce_launch(const dss2_stack_dims* d, const float* slab, int32_t n_slabs, int64_t slab_stride, const float* const* params, float* flat, float* fold_scratch, void* stream);
extern "C" int dss2_stack_reduce(const dss2_stack_dims* d, const float* slab, int32_t n_slabs, int64_t slab_stride, const float* const* params, float* flat, float* fold_scratch, void* stream) {
  if (!d) { dss2::set_error("dss2_stack_reduce: null argument"); return 2; }
  DSS2_RECORD([dd = *d, slab, n_slabs, slab_stride, params, flat, fold_scratch](void* s_) { return dss2_stack_reduce_launch(&dd, slab, n_slabs, slab_stride, params, flat, fold_scratch, s_); });
  return dss2_stack_reduce_launch(d, slab, n_slabs, slab_stride, params, flat, fold_scratch, stream);
}
static int dss2_stack_reduce_launch(const dss2_stack_dims* d, const float* slab, int32_t n_slabs, int64_t slab_stride, const float* const* params, float* flat, float* fold_scratch, void* stream) {
  if (!dims_ok(*d) || !slab || !params || !flat || !fold_scratch || n_slabs < 1 || (slab_stride & 3)) { set_error("stack_reduce: bad arguments"); return 2; }
  const int64_t total = dss2_stack_flat_floats(d);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(stack_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, slab, (int)n_slabs, slab_stride, flat, total,
                     fold_scratch, fl_block(d->n_hh, d->dout_inner), d->n_blocks);
  hipLaunchKernelGGL(stack_fold_bwd_kernel, dim3(d->n_blocks * 4), dim3(256), 0, s, *d, params, fold_scratch, flat);
  return check_launch("stack_reduce");
}
