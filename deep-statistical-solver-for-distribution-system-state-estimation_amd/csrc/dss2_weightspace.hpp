// Device-side bodies of the weight-space kernels (small GEMMs, weight packing, slab reductions), shared by their own launches
// (dss2_optim.hip, dss2_gemm_prop.hip, dss2_edge.hip).  Round 5's merged step-start / step-end launches, which also instantiated
// these bodies with COH = true, measured slower and left the library (record: profiles/experiments/r05_weights_merged_launches.hip.txt);
// the COH / NBUF template arguments remain as the hook that form would need and are instantiated with their defaults only.
#pragma once
#include "dss2_common.hpp"

namespace dss2 {

// Stores of results that ANOTHER workgroup of the same launch reads (COH = true; no launch of the library does today): write-through (sc1) global stores, so that the
// producer needs no L2 write-back before it signals (MI355X: per-XCD L2s are not coherent with each other).  COH = false: plain stores.
template <bool COH>
__device__ __forceinline__ void ws_store(float* p, float v) {
  if (COH) __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH>
__device__ __forceinline__ void ws_store4(float* p, f32x4 v) {
  if (COH) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<f32x4*>(p) = v;
}

// Batched small dense products in weight space (a few 128^3 products per step; one launch).
// One 32 x 32 tile of C per workgroup.  The operands are tiny (<= a few hundred KB, L2-resident) and the
// kernel is latency-bound: a whole K-chunk of 128 is requested at once (32 loads in flight per thread), up to three
// chunks ahead, and each chunk is multiplied out of LDS on the matrix pipe, a k quarter per wave.
constexpr int SG_T = 32, SG_KC = 128, SG_LDA = SG_KC + 4;
template <bool tA, bool tB, bool COH = false, int NBUF = 3>
__device__ __forceinline__ void small_gemm_body(const dss2_sgemm_desc* __restrict__ dp, float* base_out,
                                                float (&As)[SG_T][SG_LDA], float (&Bs)[SG_KC][SG_T + 1], const int tile) {
  // every field once, into scalars (the descriptor itself stays in global memory)
  const int M = dp->M, N = dp->N, K = dp->K, lda = dp->lda, ldb = dp->ldb, ldc = dp->ldc;
  const bool accum = dp->accumulate != 0;
  const int nbatch = dp->nbatch;
  const float* u = dp->u;
  const float* v = dp->v;
  float* C = dp->c_off >= 0 ? base_out + dp->c_off : dp->C;
  const int tn = (N + SG_T - 1) / SG_T, tm = (M + SG_T - 1) / SG_T;
  if (tile >= tm * tn) return;
  const int i0 = (tile / tn) * SG_T, j0 = (tile % tn) * SG_T;
  const int t = threadIdx.x, lo = t & 31, hi = t >> 5;   // lo runs along the contiguous memory direction
  const int kchunks = (K + SG_KC - 1) / SG_KC;
  const int nchunks = nbatch * kchunks;
  // THREE chunks of operands in flight (96 registers): the chain rule's dW2 = sum_m W_m^T dWf_m walks three K-chunks per tile,
  // and with one chunk requested at a time every chunk paid its own round trip to L2 / HBM (17.9 us for that launch at C2)
  // (NBUF: chunks in flight; a merged launch would trade them for registers -- same summation order, same results)
  float rab[NBUF][16], rbb[NBUF][16];
  // unconditional loads from clamped addresses (one batch of 32 in flight per chunk), masked afterwards
  auto issue = [&](int c, float (&ra)[16], float (&rb)[16]) {
    const int bidx = c / kchunks, k0 = (c - bidx * kchunks) * SG_KC;
    // (pointers read from a descriptor in memory are generic: through them every access is a flat_load that waits for vmcnt AND
    //  lgkmcnt; they are global by contract -- say so)
    typedef const __attribute__((address_space(1))) float* gptr;
    const gptr A = (gptr)dp->A[bidx];   // uniform scalar loads from the descriptor
    const gptr B = (gptr)dp->B[bidx];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // A(i,k): row-major [M,K] (k contiguous) or, transposed, stored [K,M] (i contiguous)
      const int i = tA ? lo : hi + 8 * (q >> 2), k = tA ? hi + 8 * q : lo + 32 * (q & 3);
      const int gi = min(i0 + i, M - 1), gk = min(k0 + k, K - 1);
      const float av = tA ? A[(size_t)gk * lda + gi] : A[(size_t)gi * lda + gk];
      ra[q] = av * (((i0 + i) < M && (k0 + k) < K) ? 1.f : 0.f);   // (a select would let the compiler branch around the load)
      // B(k,j): row-major [K,N] (j contiguous) or, transposed, stored [N,K] (k contiguous)
      const int kb = tB ? lo + 32 * (q & 3) : hi + 8 * q, j = tB ? hi + 8 * (q >> 2) : lo;
      const int gj = min(j0 + j, N - 1), gkb = min(k0 + kb, K - 1);
      const float bv = tB ? B[(size_t)gj * ldb + gkb] : B[(size_t)gkb * ldb + gj];
      rb[q] = bv * (((j0 + j) < N && (k0 + kb) < K) ? 1.f : 0.f);
    }
  };
  auto stage = [&](const float (&ra)[16], const float (&rb)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (tA) As[lo][hi + 8 * q] = ra[q]; else As[hi + 8 * (q >> 2)][lo + 32 * (q & 3)] = ra[q];
      if (tB) Bs[lo + 32 * (q & 3)][hi + 8 * (q >> 2)] = rb[q]; else Bs[hi + 8 * q][lo] = rb[q];
    }
  };
  // The product itself on the matrix pipe (round 4): wave w multiplies the k quarter [32 w, 32 w + 32) of every chunk for the
  // whole 32 x 32 tile -- 16 v_mfma_f32_32x32x2_f32 per chunk and wave (exact fp32 fma chains), operands from the LDS images:
  // A as one ds_read_b128 per four k (row stride 132 floats: a 16-lane group covers all 64 banks), B as one ds_read_b32 per k
  // -- and the four partial tiles meet once, after the last chunk, in LDS, summed in wave order.  The VALU form (one column
  // and four rows per thread, 16 LDS reads per 32 fma) spent ~6 us per chunk; the chain rule's launch was 18-19 us at C2.
  const int wave = t >> 6, lane = t & 63, c32 = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u3 = 0; u3 < NBUF; ++u3)
    if (u3 < nchunks) issue(u3, rab[u3], rbb[u3]);
  for (int c0 = 0; c0 < nchunks; c0 += NBUF) {
#pragma unroll
    for (int u3 = 0; u3 < NBUF; ++u3) {
      const int c = c0 + u3;
      if (c >= nchunks) break;          // (uniform)
      __syncthreads();
      stage(rab[u3], rbb[u3]);
      __syncthreads();
      if (c + NBUF < nchunks) issue(c + NBUF, rab[u3], rbb[u3]);
      const int kw = wave * 32;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&As[c32][kw + 4 * g]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const float a = half ? av[2 * tt + 1] : av[2 * tt];                    // A[i = c32][k = kw + 4 g + 2 tt + half]
          const float bvv = Bs[kw + 4 * g + 2 * tt + half][c32];               // B[k][j = c32]
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc, 0, 0, 0);
        }
      }
    }
  }
  // ---- the four k quarters meet in LDS (the operand images are free after the last barrier below), fixed order
  __syncthreads();
  float* red = &As[0][0];                     // [4 waves][16 registers][64 lanes] = 16 KB <= sizeof(As)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  // thread t finishes rows acc_row(r, half) for r = 4 wave .. 4 wave + 3 of column c32
  const int j = j0 + c32;
  if (j < N) {
    const float vj = u ? v[j] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i >= M) continue;
      float sum = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
      if (u) sum = fmaf(u[i], vj, sum);
      float* cp = C + (size_t)i * ldc + j;
      ws_store<COH>(cp, accum ? *cp + sum : sum);
    }
  }
}

// one descriptor's tile `tile` (transposition flags dispatched here)
template <bool COH = false, int NBUF = 3>
__device__ __forceinline__ void small_gemm_tile(const dss2_sgemm_desc* __restrict__ dp, float* base_out,
                                                float (&As)[SG_T][SG_LDA], float (&Bs)[SG_KC][SG_T + 1], const int tile) {
  const bool ta = dp->transA != 0, tb = dp->transB != 0;
  if (ta) { if (tb) small_gemm_body<true, true, COH, NBUF>(dp, base_out, As, Bs, tile); else small_gemm_body<true, false, COH, NBUF>(dp, base_out, As, Bs, tile); }
  else    { if (tb) small_gemm_body<false, true, COH, NBUF>(dp, base_out, As, Bs, tile); else small_gemm_body<false, false, COH, NBUF>(dp, base_out, As, Bs, tile); }
}

// ---- weight packing (fp32 MFMA fragments / bf16x3 fragments): see dss2_hip.h, dss2_pack_desc
// one workgroup (256 threads) of the packing of descriptor d: its elements bx * 256 .. bx * 256 + 255
__device__ __forceinline__ void pack_weights_body(const dss2_pack_desc d, const int bx) {
  const bool tr = d.transpose & 1;
  const int K = tr ? d.cols : d.rows;
  const int J = tr ? d.rows : d.cols;
  if (d.transpose & 8) {
    // f16x2 layout (f16x3 products, dss2_common.hpp: split2_pair): the GROUP's buffer d.dst = [matrix][col group][kpad/16][plane 0..1]
    // [64 lanes][8 fp16] followed by one int32 per matrix and packed COLUMN j (ncg * 32 each): the exponent s of the power-of-two scale
    // that column's elements were multiplied with before the split, 2^s max_k |B[k][j]| in [2^14, 2^15) (a scale must be uniform along
    // k only; per column it is local to a workgroup -- a per-matrix maximum had every workgroup read the whole matrix again: 144 x 64
    // KB at C2, pack_weights_kernel 5.7 -> 9.8 us).  d.koff = matrices in the group, d.joff = this matrix (no offsets in this layout).
    // A workgroup owns 256 / U columns, U = the units of 8 k per column rounded up to a power of two.
    const int nkk = d.kpad >> 4, ncg = d.ncg;
    const int units = 2 * nkk;                         // per column: (k-step, half)
    int U = 2; while (U < units) U <<= 1;
    if (U > 256) return;                               // (kpad <= 2048)
    const int cpb = 256 / U;
    const int jpad = ncg * 32;
    if (bx * cpb >= jpad) return;                      // (uniform)
    __shared__ float cmx[256];
    typedef const __attribute__((address_space(1))) float* gsrc_t;
    const gsrc_t src = (gsrc_t)d.src;
    const int t = threadIdx.x, u = t & (U - 1), cj = t / U;
    const int j = bx * cpb + cj, kg = u >> 1, hf = u & 1;
    const bool active = u < units && j < jpad;
    float v[8];
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = kg * 16 + 8 * hf + q;
      const int kc = k >= K ? K - 1 : k, jc = j >= J ? J - 1 : j;
      const float x = tr ? src[(size_t)jc * d.ld + kc] : src[(size_t)kc * d.ld + jc];
      v[q] = (active && k < K && j < J) ? x : 0.f;      // (padding: zero pieces -- this layout is written whole)
      mx = fmaxf(mx, fabsf(v[q]));
    }
    cmx[t] = mx;
    __syncthreads();
    float cm = 0.f;
    for (int w = 0; w < U; ++w) cm = fmaxf(cm, cmx[cj * U + w]);      // (the column's U partial maxima: broadcast reads)
    const int s = cm > 0.f ? 14 - exp_of(cm) : 0;
    const size_t mat_bytes = (size_t)ncg * nkk * 2048;
    char* gb = reinterpret_cast<char*>(d.dst);
    if (u == 0 && j < jpad) reinterpret_cast<int*>(gb + (size_t)d.koff * mat_bytes)[(size_t)d.joff * jpad + j] = s;
    if (!active) return;
    const int cg = j >> 5, lane = hf * 32 + (j & 31);
    typedef __attribute__((address_space(1))) uint32_t* gdst_t;
    gdst_t dst = (gdst_t)(gb + (size_t)d.joff * mat_bytes + (((size_t)cg * nkk + kg) * 2 * 64 + lane) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t h, l;
      split2_pair(ldexpf(v[2 * q], s), ldexpf(v[2 * q + 1], s), h, l);
      dst[q] = h; dst[64 * 4 + q] = l;
    }
    return;
  }
  if (d.transpose & 2) {
    // bf16x3 layout (dss2_common.hpp: split3): [col group][kpad/16][plane 0..2][64 lanes][8 bf16]; lane (half, c32) holds
    // B[k = 16 kg + 8 half + 0..7][j = 32 cg + c32] -- the B operand of v_mfma_f32_32x32x16_bf16, one plane per term
    const int kg0 = d.koff >> 4, kg1 = (d.koff + K + 15) >> 4;
    const int cg0 = d.joff >> 5, cg1 = (d.joff + J + 31) >> 5;
    const int nkl = kg1 - kg0, nkk = d.kpad >> 4;
    const int total = (cg1 - cg0) * nkl * 64;
    const int idx = bx * 256 + (int)threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int kg = kg0 + (idx >> 6) % nkl;
    const int cg = cg0 + (idx >> 6) / nkl;
    const int j = cg * 32 + (lane & 31) - d.joff;
    if (j < 0 || j >= J) return;
    // (pointers out of a descriptor are generic -- flat_load / flat_store -- unless told otherwise: global by contract; the eight
    //  source values are requested together, then split)
    typedef const __attribute__((address_space(1))) float* gsrc_t;
    typedef __attribute__((address_space(1))) __bf16* gdst_t;
    const gsrc_t src = (gsrc_t)d.src;
    gdst_t dst = (gdst_t)(reinterpret_cast<__bf16*>(d.dst) + (((size_t)cg * nkk + kg) * 3 * 64 + lane) * 8);
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = kg * 16 + 8 * (lane >> 5) + q - d.koff;
      const int kc = k < 0 ? 0 : (k >= K ? K - 1 : k);
      v[q] = tr ? src[(size_t)j * d.ld + kc] : src[(size_t)kc * d.ld + j];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = kg * 16 + 8 * (lane >> 5) + q - d.koff;
      if (k < 0 || k >= K) continue;
      __bf16 h, m, l;
      split3(v[q], h, m, l);
      dst[q] = h; dst[64 * 8 + q] = m; dst[2 * 64 * 8 + q] = l;
    }
    return;
  }
  // destination k-groups / column groups this block touches
  const int kg0 = d.koff >> 3, kg1 = (d.koff + K + 7) >> 3;
  const int cg0 = d.joff >> 5, cg1 = (d.joff + J + 31) >> 5;
  const int nkl = kg1 - kg0;
  const int nkk = d.kpad >> 3;
  const int total = (cg1 - cg0) * nkl * 64;
  const int idx = bx * 256 + (int)threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int kg = kg0 + (idx >> 6) % nkl;
  const int cg = cg0 + (idx >> 6) / nkl;
  const int j = cg * 32 + (lane & 31) - d.joff;     // source column
  if (j < 0 || j >= J) return;
  float* dst = d.dst + (((size_t)cg * nkk + kg) * 64 + lane) * 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = kg * 8 + 4 * (lane >> 5) + s - d.koff;   // source k
    if (k >= 0 && k < K) dst[s] = tr ? d.src[(size_t)j * d.ld + k] : d.src[(size_t)k * d.ld + j];
  }
}


// ---- fixed-order slab reductions
// several reductions in one launch (blockIdx.y = reduction): the small ones are launch/latency-bound on their own
constexpr int REDUCE_MAX_DESC = 32;      // by-value table: 32 x 40 B of the 4 KiB kernel-argument space
struct ReduceTable { dss2_reduce_desc d[REDUCE_MAX_DESC]; uint32_t scalar_mask; };      // bit i: reduction i lacks the alignment of the 16-byte form
// The same reductions with 16-byte lanes: a workgroup owns 64 consecutive floats of one reduction, its 256 threads are 16 slab
// lanes x 16 float4 columns; lane s sums slabs s, s + 16, .. (four independent loads in flight per pass), the sixteen partial
// sums meet in LDS and are added in lane order: fixed order, bitwise reproducible.  ~2x the bytes per second of the scalar form
// (measured on the whole-stack slabs: 124 MB in 30 us).  Needs 16-byte aligned slabs / outputs and strides divisible by 4.
// Reductions whose slabs / outputs are not 16-byte aligned (bit in scalar_mask: e.g. what follows a 2-wide bias in a flat
// gradient) run the scalar form inside the same launch -- one misaligned descriptor no longer sends all of them there.
// one workgroup (256 threads) of reduction d: its 64 output floats bx * 64 .. bx * 64 + 63 (scalar: the form without 16-byte lanes)
// MANY short slabs (the per-tile slabs of the narrow head's weight gradient: 1024 x 770 floats at C2): 16 slab lanes leave every lane
// 64 slabs deep behind one another; the "tall" form gives a workgroup 16 output floats and 64 slab lanes instead (round 5).
__host__ __device__ __forceinline__ bool reduce_tall(const dss2_reduce_desc& d) { return d.n_slabs >= 256 && d.len <= 8192; }
__host__ __device__ __forceinline__ int64_t reduce_blocks(const dss2_reduce_desc& d, bool scalar) {
  return (!scalar && reduce_tall(d)) ? (d.len + 15) / 16 : (d.len + 63) / 64;
}
template <bool COH = false>
__device__ __forceinline__ void reduce_slabs_body(const dss2_reduce_desc& d, const bool scalar, const int bx, f32x4 (&part)[16][16]) {
  if (!scalar && reduce_tall(d)) {                // (uniform per descriptor)
    if ((int64_t)bx * 16 >= d.len) return;
    const int tid = threadIdx.x, cl = tid & 3, sl = tid >> 2;      // 4 float4 columns x 64 slab lanes
    const int64_t i4 = (int64_t)bx * 16 + cl * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4* pt = &part[0][0];
    if (i4 + 4 <= d.len) {
#pragma unroll 1
      for (int k0 = sl; k0 < d.n_slabs; k0 += 256) {
        f32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = k0 + 64 * j;
          r[j] = *reinterpret_cast<const f32x4*>(d.slab + (size_t)(kk < d.n_slabs ? kk : k0) * d.stride + i4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + 64 * j < d.n_slabs) s += r[j];
      }
    } else if (i4 < d.len) {
      for (int k = sl; k < d.n_slabs; k += 64)
        for (int q = 0; i4 + q < d.len; ++q) s[q] += d.slab[(size_t)k * d.stride + i4 + q];
    }
    pt[sl * 4 + cl] = s;
    __syncthreads();
    if (sl == 0 && i4 < d.len) {
      f32x4 t4[4];      // four interleaved chains, then pairwise: fixed order
#pragma unroll
      for (int c = 0; c < 4; ++c) t4[c] = pt[c * 4 + cl];
      for (int k = 4; k < 64; k += 4)
#pragma unroll
        for (int c = 0; c < 4; ++c) t4[c] += pt[(k + c) * 4 + cl];
      const f32x4 t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
      if (i4 + 4 <= d.len) ws_store4<COH>(d.out + i4, t);
      else for (int q = 0; i4 + q < d.len; ++q) ws_store<COH>(d.out + i4 + q, t[q]);
    }
    return;
  }
  if ((int64_t)bx * 64 >= d.len) return;          // uniform per workgroup
  const int tid = threadIdx.x, cl = tid & 15, sl = tid >> 4;
  if (scalar) {              // (uniform) 64 columns x 4 slab quarters, as reduce_slabs_multi_kernel
    float* sp = reinterpret_cast<float*>(part);
    const int x = tid & 63, q = tid >> 6;
    const int64_t j = (int64_t)bx * 64 + x;
    const int per = (d.n_slabs + 3) >> 2;
    const int k0 = q * per, k1 = min(d.n_slabs, k0 + per);
    float s = 0.f;
    if (j < d.len) {      // (the summation order of reduce_slabs_kernel)
      const float* p = d.slab + j;
      int k = k0;
      for (; k + 8 <= k1; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * d.stride];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; k < k1; ++k) s += p[(size_t)k * d.stride];
    }
    sp[q * 64 + x] = s;
    __syncthreads();
    if (q == 0 && j < d.len) ws_store<COH>(d.out + j, (sp[x] + sp[64 + x]) + (sp[128 + x] + sp[192 + x]));
    return;
  }
  const int64_t i4 = (int64_t)bx * 64 + cl * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 + 4 <= d.len) {
    int k = sl;
#pragma unroll 1
    for (; k + 48 < d.n_slabs; k += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d.slab + (size_t)k * d.stride + i4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(d.slab + (size_t)(k + 16) * d.stride + i4);
      const f32x4 c = *reinterpret_cast<const f32x4*>(d.slab + (size_t)(k + 32) * d.stride + i4);
      const f32x4 e = *reinterpret_cast<const f32x4*>(d.slab + (size_t)(k + 48) * d.stride + i4);
      s += a; s += b; s += c; s += e;
    }
    for (; k < d.n_slabs; k += 16) s += *reinterpret_cast<const f32x4*>(d.slab + (size_t)k * d.stride + i4);
  } else if (i4 < d.len) {
    for (int k = sl; k < d.n_slabs; k += 16)
      for (int q = 0; i4 + q < d.len; ++q) s[q] += d.slab[(size_t)k * d.stride + i4 + q];
  }
  part[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && i4 < d.len) {
    f32x4 t = part[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][cl];
    if (i4 + 4 <= d.len) ws_store4<COH>(d.out + i4, t);
    else for (int q = 0; i4 + q < d.len; ++q) ws_store<COH>(d.out + i4 + q, t[q]);
  }
}


}  // namespace dss2
