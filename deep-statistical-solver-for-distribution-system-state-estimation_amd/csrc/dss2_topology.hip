// Device-side build of the per-topology graph structure (SURVEY 8b `dss2_csr_build`, 8f rank 1), gfx950.
//
// Replaces, per distinct edge_index: the reference's MPN.is_directed / undirect_graph (networks.py:236-258: one host
// sync + three concatenations per forward), PyG's gcn_norm (degree, pow, masked_fill, two gathers per TAGConv call)
// and PyG's per-call scatter index handling -- and round 1's ~40 torch index ops + 3 device-to-host copies.
//
// Pure integer / index work, HBM- and launch-bound (a C2 batch is 0.9 MB of edge_index): counting sort by target, by
// source and by incident bus with integer atomics for the counts and the slot claims, then a per-row sort of the
// claimed slots by directed edge id, so that the final layout does not depend on the order the atomics landed in
// (rows list their entries in ascending directed edge id = the order index_add_ visits them on the CPU).  No float
// atomics anywhere; gcn_norm weights are two correctly rounded fp32 operations (1 / sqrt(deg)) and one multiply,
// bit for bit what torch's CPU `deg.pow(-0.5)[src] * deg.pow(-0.5)[tgt]` gives.
//
//   probe        64 + 64 bit content hash and the reference's first-edge-only is_directed rule, one pass
//   count        in-degree / out-degree / incidence counts, int32 endpoints, the "edge spans this cut" cover array
//   scan x2      exclusive sums (row pointers, cover) and the running maximum of legal cut positions
//   fill, sort   slot claims, per-row order, col / ent / perm / weights; segment statistics
//   tiles        whole-graph tiles for a row budget: closed form for uniform graphs, a sequential walk otherwise
//   ell          per-tile ELL slices of both CSRs (what the tile kernels stage in LDS with one coalesced copy)
//   deg_pows     [deg, A deg, A^2 deg, A^3 deg] in float64 (row scales of a bias folded through propagations)
#include "dss2_common.hpp"

namespace dss2 {

constexpr int kScanChunk = 2048;      // elements per workgroup in the scans (256 threads x 8)
constexpr uint32_t kFlip = 0x80000000u;

__device__ __forceinline__ uint64_t tmix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// out[0], out[1]: two independent position-dependent hashes combined by wrapping sums (order independent);
// out[2]: number of edges (v0 -> u0) that reverse the batch's first edge (u0 -> v0): is_directed <=> out[2] == 0
__global__ void __launch_bounds__(256) topo_probe_kernel(const int64_t* __restrict__ ei, int64_t E,
                                                         unsigned long long* __restrict__ out) {
  const int64_t u0 = ei[0], v0 = ei[E];
  uint64_t s1 = 0, s2 = 0, rev = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * E; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t v = (uint64_t)ei[i];
    s1 += tmix64(v * 0x100000001b3ull + tmix64((uint64_t)i));
    s2 += tmix64((v + 0x632be59bd9b4e019ull) * 0xff51afd7ed558ccdull ^ tmix64((uint64_t)i * 0xc4ceb9fe1a85ec53ull + 1));
    if (i < E && ei[i] == v0 && ei[E + i] == u0) ++rev;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor((unsigned long long)s1, o);
    s2 += __shfl_xor((unsigned long long)s2, o);
    rev += __shfl_xor((unsigned long long)rev, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, (unsigned long long)s1);
    atomicAdd(out + 1, (unsigned long long)s2);
    if (rev) atomicAdd(out + 2, (unsigned long long)rev);
  }
}

struct BuildPtrs {
  const int64_t* ei; int64_t E, N, E2; int doubled; int no_flip;
  int32_t *cnt, *cntT, *cntI, *cover;           // work: [N+1] each (cover: [N+2])
  int32_t *keys, *keysT, *keysI;                // work: [E2], [E2], [2E]
  int32_t *bsum;                                // work: [4][nblk]
  int32_t* meta;
};

// meta slots (int32, device): 0 max in-degree, 1 max out-degree, 2 longest segment, 3 number of segments,
// 4 shortest segment, 5 error flag (an endpoint outside [0, N)), 6 max CSR entries per tile, 7 (transposed),
// 8.. ntiles of the candidate row budgets
__global__ void __launch_bounds__(256) topo_count_kernel(const BuildPtrs p, int32_t* __restrict__ efrom,
                                                         int32_t* __restrict__ eto) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < p.E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = p.ei[e], b = p.ei[p.E + e];
    if (a < 0 || a >= p.N || b < 0 || b >= p.N) { p.meta[5] = 1; continue; }
    efrom[e] = (int32_t)a;
    eto[e] = (int32_t)b;
    atomicAdd(p.cnt + b, 1);                   // stored edge a -> b: target b, source a
    atomicAdd(p.cntT + a, 1);
    if (p.doubled) {                           // reverse edge b -> a
      atomicAdd(p.cnt + a, 1);
      atomicAdd(p.cntT + b, 1);
    }
    atomicAdd(p.cntI + a, 1);
    atomicAdd(p.cntI + b, 1);
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    atomicAdd(p.cover + lo + 1, 1);            // a cut before row q is legal iff no edge has lo < q <= hi
    atomicAdd(p.cover + hi + 1, -1);
  }
}

__global__ void meta_init_kernel(int32_t* __restrict__ meta) {
  if (threadIdx.x < 16) meta[threadIdx.x] = threadIdx.x == 4 ? 0x7fffffff : 0;     // slot 4 is a running minimum
}

// ---- scans: in[n] -> out[n + 1] exclusive sums (out[n] = total), three passes.  OP 0: int sum; OP 1: running max.
template <int OP>
__device__ __forceinline__ int scan_op(int a, int b) { return OP == 0 ? a + b : (a > b ? a : b); }

struct ScanJob { const int32_t* in; int32_t* out; int64_t n; };
struct ScanJobs { ScanJob j[4]; int32_t* bsum; int nblk; };

template <int OP>
__global__ void __launch_bounds__(256) scan_partial_kernel(const ScanJobs s) {
  const ScanJob j = s.j[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  if (base >= j.n) { if (threadIdx.x == 0) s.bsum[blockIdx.y * s.nblk + blockIdx.x] = OP == 0 ? 0 : -1; return; }
  int v = OP == 0 ? 0 : -1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t idx = base + threadIdx.x * 8 + i;
    if (idx < j.n) v = scan_op<OP>(v, j.in[idx]);
  }
  __shared__ int red[4];
  for (int o = 32; o > 0; o >>= 1) v = scan_op<OP>(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) s.bsum[blockIdx.y * s.nblk + blockIdx.x] = scan_op<OP>(scan_op<OP>(red[0], red[1]), scan_op<OP>(red[2], red[3]));
}

template <int OP>
__global__ void __launch_bounds__(64) scan_bsum_kernel(const ScanJobs s) {   // one wave per job: exclusive scan of the block sums
  int32_t* b = s.bsum + blockIdx.x * s.nblk;
  int carry = OP == 0 ? 0 : -1;
  for (int base = 0; base < s.nblk; base += 64) {
    const int i = base + threadIdx.x;
    const int v = i < s.nblk ? b[i] : (OP == 0 ? 0 : -1);
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if ((int)threadIdx.x >= o) inc = scan_op<OP>(inc, t);
    }
    int exc = __shfl_up(inc, 1);
    if (threadIdx.x == 0) exc = OP == 0 ? 0 : -1;
    if (i < s.nblk) b[i] = scan_op<OP>(carry, exc);
    carry = scan_op<OP>(carry, __shfl(inc, 63));
  }
}

// OP 0: out[i] = exclusive sum (i <= n).  OP 1: out[i] = INCLUSIVE running max (i < n)
template <int OP>
__global__ void __launch_bounds__(256) scan_apply_kernel(const ScanJobs s) {
  const ScanJob j = s.j[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  if (base > j.n) return;
  int v[8];
  int t = OP == 0 ? 0 : -1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t idx = base + threadIdx.x * 8 + i;
    v[i] = idx < j.n ? j.in[idx] : (OP == 0 ? 0 : -1);
    t = scan_op<OP>(t, v[i]);
  }
  // exclusive scan of the per-thread totals across the workgroup
  __shared__ int wsum[4];
  int inc = t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o);
    if (lane >= o) inc = scan_op<OP>(inc, u);
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int pre = s.bsum[blockIdx.y * s.nblk + blockIdx.x];
  for (int w = 0; w < wave; ++w) pre = scan_op<OP>(pre, wsum[w]);
  int exc = __shfl_up(inc, 1);
  if (lane == 0) exc = OP == 0 ? 0 : -1;
  int run = scan_op<OP>(pre, exc);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t idx = base + threadIdx.x * 8 + i;
    if (OP == 0) {
      if (idx <= j.n) j.out[idx] = run;
      run += v[i];
    } else {
      run = scan_op<OP>(run, v[i]);
      if (idx < j.n) j.out[idx] = run;
    }
  }
}

// cutpos[q] = q if a cut before row q is legal (no edge spans it; q = 0 and q = N always are), else -1
__global__ void __launch_bounds__(256) topo_cutpos_kernel(const int32_t* __restrict__ cover_excl, int64_t N,
                                                          int32_t* __restrict__ cutpos) {
  // cover_excl[q + 1] = sum of cover[0..q] = number of edges with lo < q <= hi
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= N; q += (int64_t)gridDim.x * blockDim.x)
    cutpos[q] = (q == 0 || q == N || cover_excl[q + 1] == 0) ? (int32_t)q : -1;
}

__global__ void __launch_bounds__(256) topo_fill_kernel(const BuildPtrs p, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ rowptrT,
                                                        const int32_t* __restrict__ inc_rowptr, int32_t* __restrict__ cur,
                                                        int32_t* __restrict__ curT, int32_t* __restrict__ curI) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < p.E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = p.ei[e], b = p.ei[p.E + e];
    if (a < 0 || a >= p.N || b < 0 || b >= p.N) continue;
    p.keys[rowptr[b] + atomicAdd(cur + b, 1)] = (int32_t)e;                       // d = e: a -> b
    p.keysT[rowptrT[a] + atomicAdd(curT + a, 1)] = (int32_t)e;
    if (p.doubled) {                                                               // d = E + e: b -> a
      p.keys[rowptr[a] + atomicAdd(cur + a, 1)] = (int32_t)(p.E + e);
      p.keysT[rowptrT[b] + atomicAdd(curT + b, 1)] = (int32_t)(p.E + e);
    }
    p.keysI[inc_rowptr[a] + atomicAdd(curI + a, 1)] = (int32_t)e;                  // from-end: e
    p.keysI[inc_rowptr[b] + atomicAdd(curI + b, 1)] = (int32_t)((uint32_t)e | kFlip);   // to-end: e | flag
  }
}

// ascending (as unsigned) insertion sort of one row's claimed slots; rows are a handful of entries long
__device__ __forceinline__ void sort_row(int32_t* k, int n) {
  for (int i = 1; i < n; ++i) {
    const uint32_t v = (uint32_t)k[i];
    int j = i - 1;
    while (j >= 0 && (uint32_t)k[j] > v) { k[j + 1] = k[j]; --j; }
    k[j + 1] = (int32_t)v;
  }
}

struct FinalPtrs {
  int32_t *col, *ent, *perm; float* w;
  int32_t *colT, *entT, *permT; float* wT;
  int32_t* inc_ent; float* deg;
  const int32_t *rowptr, *rowptrT, *inc_rowptr, *efrom, *eto;
  const int32_t* lastcut;
};

__device__ __forceinline__ float inv_sqrt_deg(int d) {   // torch CPU deg.pow(-0.5) with inf -> 0: two correctly rounded ops
  return d > 0 ? __fdiv_rn(1.0f, __fsqrt_rn((float)d)) : 0.f;
}

// blockIdx.y: 0 = CSR by target, 1 = CSR by source, 2 = incidence + degrees + segment statistics
__global__ void __launch_bounds__(256) topo_finalize_kernel(const BuildPtrs p, const FinalPtrs f) {
  const int which = blockIdx.y;
  if (which == 2) {
    // statistics: per-thread partials over the grid-stride loop, one wave reduction, ONE atomic per wave and slot (every
    // row hammering the same five words would serialise the whole kernel in the L2 atomic unit)
    int mdeg = 0, mdegT = 0, smax = 0, smin = 0x7fffffff, scnt = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.N; r += (int64_t)gridDim.x * blockDim.x) {
      const int i0 = f.inc_rowptr[r], ni = f.inc_rowptr[r + 1] - i0;
      sort_row(p.keysI + i0, ni);
      for (int k = 0; k < ni; ++k) f.inc_ent[i0 + k] = p.keysI[i0 + k];
      const int d = f.rowptr[r + 1] - f.rowptr[r];
      f.deg[r] = (float)d;
      mdeg = max(mdeg, d);
      mdegT = max(mdegT, f.rowptrT[r + 1] - f.rowptrT[r]);
      const int q = (int)r + 1;                    // a legal cut at q closes the segment [lastcut[q - 1], q)
      if (f.lastcut[q] == q) {
        const int len = q - f.lastcut[q - 1];
        smax = max(smax, len);
        smin = min(smin, len);
        ++scnt;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      mdeg = max(mdeg, __shfl_xor(mdeg, o));
      mdegT = max(mdegT, __shfl_xor(mdegT, o));
      smax = max(smax, __shfl_xor(smax, o));
      smin = min(smin, __shfl_xor(smin, o));
      scnt += __shfl_xor(scnt, o);
    }
    if ((threadIdx.x & 63) == 0) {      // (running extrema: read first, post an atomic only when it would change the word)
      auto cur = [&](int i) { return __hip_atomic_load(p.meta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      if (mdeg > cur(0)) atomicMax(p.meta + 0, mdeg);
      if (mdegT > cur(1)) atomicMax(p.meta + 1, mdegT);
      if (scnt) {
        if (smax > cur(2)) atomicMax(p.meta + 2, smax);
        atomicAdd(p.meta + 3, scnt);
        if (smin < cur(4)) atomicMin(p.meta + 4, smin);
      }
    }
    return;
  }
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < p.N; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t* rp = which == 0 ? f.rowptr : f.rowptrT;
    int32_t* keys = which == 0 ? p.keys : p.keysT;
    const int e0 = rp[r], n = rp[r + 1] - e0;
    auto emit = [&](int k, int d) {
      const bool flip = d >= p.E;
      const int e = flip ? d - (int)p.E : d;
      const int a = f.efrom[e], b = f.eto[e];
      const int src = flip ? b : a, tgt = flip ? a : b;
      // gcn_norm(add_self_loops=False): in-degree on the (doubled) directed list, both CSRs carry the same weight
      const float wv = __fmul_rn(inv_sqrt_deg(f.rowptr[src + 1] - f.rowptr[src]), inv_sqrt_deg(f.rowptr[tgt + 1] - f.rowptr[tgt]));
      const int en = (int)((uint32_t)e | ((flip && !p.no_flip) ? kFlip : 0u));
      if (which == 0) { f.col[e0 + k] = src; f.ent[e0 + k] = en; f.perm[e0 + k] = d; f.w[e0 + k] = wv; }
      else { f.colT[e0 + k] = tgt; f.entT[e0 + k] = en; f.permT[e0 + k] = d; f.wT[e0 + k] = wv; }
    };
    constexpr int RMAX = 8;
    if (n <= RMAX) {
      // the usual case (grid buses have a handful of branches): the row's slots are sorted in REGISTERS and the dependent
      // lookups of its entries (endpoints, degrees) are then independent of each other, so their latencies overlap
      int kreg[RMAX];
#pragma unroll
      for (int k = 0; k < RMAX; ++k) kreg[k] = k < n ? keys[e0 + k] : 0x7fffffff;
#pragma unroll
      for (int i = 1; i < RMAX; ++i)
#pragma unroll
        for (int j = i; j > 0; --j) {
          const int lo = min(kreg[j - 1], kreg[j]), hi = max(kreg[j - 1], kreg[j]);
          kreg[j - 1] = lo; kreg[j] = hi;
        }
#pragma unroll
      for (int k = 0; k < RMAX; ++k)
        if (k < n) emit(k, kreg[k]);
    } else {
      sort_row(keys + e0, n);
      for (int k = 0; k < n; ++k) emit(k, keys[e0 + k]);
    }
  }
}

// ---- tiles
__global__ void __launch_bounds__(256) tiles_uniform_kernel(int32_t* __restrict__ ts, int ntiles, int rows_per_tile, int N) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= ntiles) {
    const int64_t v = (int64_t)t * rows_per_tile;
    ts[t] = v < N ? (int)v : N;
  }
}

// Greedy packing of whole segments into tiles of <= tm rows, one thread per candidate row budget: the next tile start
// is the last legal cut within the budget.  Sequential by nature (each start depends on the previous one); only used
// for batches whose graphs differ in size -- uniform batches take the closed form above.  ntiles = -1: a segment
// exceeds the budget.
struct WalkArgs { const int32_t* lastcut; int N; int ncand; int tm[8]; int32_t* ts[8]; int32_t* ntiles; int cap; };
__global__ void tiles_walk_kernel(const WalkArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.ncand) return;
  int32_t* ts = a.ts[c];
  int p = 0, t = 0;
  ts[0] = 0;
  while (p < a.N) {
    const int lim = p + a.tm[c] < a.N ? p + a.tm[c] : a.N;
    const int q = a.lastcut[lim];
    if (q <= p || t >= a.cap) { a.ntiles[c] = -1; return; }
    p = q;
    ts[++t] = p;
  }
  a.ntiles[c] = t;
}

// ---- per-tile ELL slices.  blockIdx.y: 0 = CSR by target, 1 = CSR by source.
// ell_w  [ntiles][D][TM] int2 {local other node, weight bits}, empty slot {own row, 0}
// ell_e  [ntiles][D][TM] int2 {local other node, stored edge id | flip}, empty slot {0, -1}
struct EllArgs {
  const int32_t *rowptr[2], *col[2], *ent[2]; const float* w[2];
  int2 *ell_w[2], *ell_e[2]; int D[2];
  const int32_t* tile_start; int ntiles, TM; int32_t* meta;
};
__global__ void __launch_bounds__(256) ell_tiles_kernel(const EllArgs a) {
  const int which = blockIdx.y, tile = blockIdx.x;
  const int D = a.D[which];
  const int ts = a.tile_start[tile], R = a.tile_start[tile + 1] - ts;
  const int32_t* rp = a.rowptr[which];
  if (threadIdx.x == 0) {      // (a running maximum: read first -- thousands of tiles hammering one word with atomics cost this launch 38 us at C5;
                               //  a stale read only costs a redundant atomic)
    const int nnz = rp[ts + R] - rp[ts];
    if (nnz > __hip_atomic_load(a.meta + 6 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.meta + 6 + which, nnz);
  }
  if (D <= 0) return;
  int2* ow = a.ell_w[which] + (size_t)tile * D * a.TM;
  int2* oe = a.ell_e[which] + (size_t)tile * D * a.TM;
  for (int idx = threadIdx.x; idx < D * a.TM; idx += blockDim.x) {
    const int k = idx / a.TM, r = idx - k * a.TM;
    int2 vw = make_int2(r, 0), ve = make_int2(0, -1);
    if (r < R) {
      const int e0 = rp[ts + r], deg = rp[ts + r + 1] - e0;
      if (deg > D) a.meta[5] = 2;                  // the ELL width is smaller than a row: caller's hint was wrong
      if (k < deg) {
        const int c = a.col[which][e0 + k] - ts;
        vw = make_int2(c, __float_as_int(a.w[which][e0 + k]));
        ve = make_int2(c, a.ent[which][e0 + k]);
      }
    }
    ow[idx] = vw;
    oe[idx] = ve;
  }
}

// v_out[r] = sum_{e in row r, CSR order} w[e] * v_in[col[e]] in float64 (product and sum rounded separately, like
// torch's index_add_ of w * v[col]); column m of out[N, 4] <- (float) v
__global__ void __launch_bounds__(256) deg_pows_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ w, const double* __restrict__ vin,
                                                       double* __restrict__ vout, float* __restrict__ out, int m, int64_t N) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    const int e1 = rowptr[r + 1];
    for (int e = rowptr[r]; e < e1; ++e) s = __dadd_rn(s, __dmul_rn((double)w[e], vin[col[e]]));
    vout[r] = s;
    out[r * 4 + m] = (float)s;
  }
}
__global__ void __launch_bounds__(256) deg_pows_init_kernel(const float* __restrict__ deg, double* __restrict__ v,
                                                            float* __restrict__ out, int64_t N) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    v[r] = (double)deg[r];
    out[r * 4] = deg[r];
  }
}

static inline unsigned grid_for(int64_t n, int per_block = 256, int cap = 2048) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

}  // namespace dss2

using namespace dss2;

extern "C" int dss2_topology_probe(const int64_t* edge_index, int64_t n_edges, uint64_t* out3, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_topology_probe");
  if (n_edges <= 0) { set_error("topology_probe: empty edge list"); return 2; }
  hipLaunchKernelGGL(topo_probe_kernel, dim3(grid_for(2 * n_edges, 256, 1024)), dim3(256), 0, as_stream(stream), edge_index,
                     n_edges, reinterpret_cast<unsigned long long*>(out3));
  return check_launch("topology_probe");
}

extern "C" int64_t dss2_csr_build_work_ints(int64_t n_nodes, int64_t n_edges, int doubled) {
  const int64_t E2 = doubled ? 2 * n_edges : n_edges;
  const int64_t nblk = (n_nodes + 2 + kScanChunk - 1) / kScanChunk + 1;
  // cnt, cntT, cntI [N+1], cover [N+2] (zeroed by the build) | cur, curT, curI [N] (zeroed) | keys [E2], keysT [E2],
  // keysI [2E] | cover_excl [N+3] | cutpos [N+1] | bsum [4][nblk]
  return 3 * (n_nodes + 1) + (n_nodes + 2) + 3 * n_nodes + 2 * E2 + 2 * n_edges + (n_nodes + 3) + (n_nodes + 1) + 4 * nblk + 64;
}

extern "C" int dss2_csr_build(const dss2_csr_build_args* ap, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_csr_build");
  const dss2_csr_build_args& a = *ap;
  if (a.n_edges <= 0 || a.n_nodes <= 0) { set_error("csr_build: empty graph batch"); return 2; }
  if (a.n_nodes >= (1ll << 31) - 4 || 2 * a.n_edges >= (1ll << 31) - 4) { set_error("csr_build: graph too large for the int32 CSR"); return 2; }
  if (!a.edge_index || !a.rowptr || !a.col || !a.ent || !a.perm || !a.w || !a.rowptrT || !a.colT || !a.entT || !a.permT || !a.wT ||
      !a.inc_rowptr || !a.inc_ent || !a.efrom || !a.eto || !a.deg || !a.lastcut || !a.meta || !a.work) {
    set_error("csr_build: null argument"); return 2;
  }
  hipStream_t s = as_stream(stream);
  const int64_t N = a.n_nodes, E = a.n_edges, E2 = a.doubled ? 2 * E : E;
  const int nblk = (int)((N + 2 + kScanChunk - 1) / kScanChunk + 1);
  int32_t* wk = a.work;
  BuildPtrs p;
  p.ei = a.edge_index; p.E = E; p.N = N; p.E2 = E2; p.doubled = a.doubled ? 1 : 0; p.no_flip = a.no_flip ? 1 : 0; p.meta = a.meta;
  p.cnt = wk; wk += N + 1;
  p.cntT = wk; wk += N + 1;
  p.cntI = wk; wk += N + 1;
  p.cover = wk; wk += N + 2;
  int32_t* cur = wk; wk += N;
  int32_t* curT = wk; wk += N;
  int32_t* curI = wk; wk += N;
  const size_t zero_ints = (size_t)(wk - a.work);
  p.keys = wk; wk += E2;
  p.keysT = wk; wk += E2;
  p.keysI = wk; wk += 2 * E;
  int32_t* cover_excl = wk; wk += N + 3;
  int32_t* cutpos = wk; wk += N + 1;
  p.bsum = wk; wk += 4 * nblk;
  hipError_t e = hipMemsetAsync(a.work, 0, zero_ints * sizeof(int32_t), s);
  if (e != hipSuccess) { set_error("csr_build: hipMemsetAsync: %s", hipGetErrorString(e)); return 1; }
  hipLaunchKernelGGL(meta_init_kernel, dim3(1), dim3(64), 0, s, a.meta);
  hipLaunchKernelGGL(topo_count_kernel, dim3(grid_for(E)), dim3(256), 0, s, p, a.efrom, a.eto);
  // exclusive sums: rowptr, rowptrT, inc_rowptr, cover
  ScanJobs sj = {};
  sj.j[0] = {p.cnt, a.rowptr, N};
  sj.j[1] = {p.cntT, a.rowptrT, N};
  sj.j[2] = {p.cntI, a.inc_rowptr, N};
  sj.j[3] = {p.cover, cover_excl, N + 2};
  sj.bsum = p.bsum; sj.nblk = nblk;
  hipLaunchKernelGGL(scan_partial_kernel<0>, dim3(nblk, 4), dim3(256), 0, s, sj);
  hipLaunchKernelGGL(scan_bsum_kernel<0>, dim3(4), dim3(64), 0, s, sj);
  hipLaunchKernelGGL(scan_apply_kernel<0>, dim3(nblk, 4), dim3(256), 0, s, sj);
  // legal cut positions and their running maximum (lastcut[q] = largest legal cut <= q)
  hipLaunchKernelGGL(topo_cutpos_kernel, dim3(grid_for(N + 1)), dim3(256), 0, s, cover_excl, N, cutpos);
  ScanJobs mj = {};
  mj.j[0] = {cutpos, a.lastcut, N + 1};
  mj.bsum = p.bsum; mj.nblk = nblk;
  hipLaunchKernelGGL(scan_partial_kernel<1>, dim3(nblk, 1), dim3(256), 0, s, mj);
  hipLaunchKernelGGL(scan_bsum_kernel<1>, dim3(1), dim3(64), 0, s, mj);
  hipLaunchKernelGGL(scan_apply_kernel<1>, dim3(nblk, 1), dim3(256), 0, s, mj);
  hipLaunchKernelGGL(topo_fill_kernel, dim3(grid_for(E)), dim3(256), 0, s, p, a.rowptr, a.rowptrT, a.inc_rowptr, cur, curT, curI);
  FinalPtrs f;
  f.col = a.col; f.ent = a.ent; f.perm = a.perm; f.w = a.w;
  f.colT = a.colT; f.entT = a.entT; f.permT = a.permT; f.wT = a.wT;
  f.inc_ent = a.inc_ent; f.deg = a.deg;
  f.rowptr = a.rowptr; f.rowptrT = a.rowptrT; f.inc_rowptr = a.inc_rowptr; f.efrom = a.efrom; f.eto = a.eto;
  f.lastcut = a.lastcut;
  hipLaunchKernelGGL(topo_finalize_kernel, dim3(grid_for(N), 3), dim3(256), 0, s, p, f);
  return check_launch("csr_build");
}

extern "C" int dss2_tiles_uniform(int32_t* tile_start, int32_t ntiles, int32_t rows_per_tile, int64_t n_nodes, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_tiles_uniform");
  if (ntiles <= 0 || rows_per_tile <= 0) { set_error("tiles_uniform: bad arguments"); return 2; }
  hipLaunchKernelGGL(tiles_uniform_kernel, dim3(grid_for(ntiles + 1)), dim3(256), 0, as_stream(stream), tile_start, ntiles,
                     rows_per_tile, (int)n_nodes);
  return check_launch("tiles_uniform");
}

extern "C" int dss2_tiles_walk(const int32_t* lastcut, int64_t n_nodes, const int32_t* tm_host, int32_t n_cand,
                               int32_t* const* tile_starts_host, int32_t cap, int32_t* ntiles_dev, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_tiles_walk");
  if (n_cand <= 0 || n_cand > 8) { set_error("tiles_walk: 1..8 candidates"); return 2; }
  WalkArgs w = {};
  w.lastcut = lastcut; w.N = (int)n_nodes; w.ncand = n_cand; w.ntiles = ntiles_dev; w.cap = cap;
  for (int i = 0; i < n_cand; ++i) { w.tm[i] = tm_host[i]; w.ts[i] = tile_starts_host[i]; }
  hipLaunchKernelGGL(tiles_walk_kernel, dim3(1), dim3(64), 0, as_stream(stream), w);
  return check_launch("tiles_walk");
}

extern "C" int dss2_ell_tiles_build(const dss2_ell_build_args* ap, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_ell_tiles_build");
  const dss2_ell_build_args& b = *ap;
  if (b.ntiles <= 0 || b.tm <= 0) { set_error("ell_tiles_build: bad arguments"); return 2; }
  EllArgs a = {};
  a.rowptr[0] = b.rowptr; a.col[0] = b.col; a.ent[0] = b.ent; a.w[0] = b.w;
  a.rowptr[1] = b.rowptrT; a.col[1] = b.colT; a.ent[1] = b.entT; a.w[1] = b.wT;
  a.ell_w[0] = reinterpret_cast<int2*>(b.ell_tiles); a.ell_e[0] = reinterpret_cast<int2*>(b.ell_ent_tiles);
  a.ell_w[1] = reinterpret_cast<int2*>(b.ellT_tiles); a.ell_e[1] = reinterpret_cast<int2*>(b.ellT_ent_tiles);
  a.D[0] = b.ell_width; a.D[1] = b.ellT_width;
  a.tile_start = b.tile_start; a.ntiles = b.ntiles; a.TM = b.tm; a.meta = b.meta;
  if ((a.D[0] > 0 && (!a.ell_w[0] || !a.ell_e[0])) || (a.D[1] > 0 && (!a.ell_w[1] || !a.ell_e[1]))) { set_error("ell_tiles_build: null output"); return 2; }
  hipLaunchKernelGGL(ell_tiles_kernel, dim3(b.ntiles, 2), dim3(256), 0, as_stream(stream), a);
  return check_launch("ell_tiles_build");
}

extern "C" int dss2_deg_pows(const int32_t* rowptr, const int32_t* col, const float* w, const float* deg, int64_t n_nodes,
                             float* out, double* work, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_deg_pows");
  if (n_nodes <= 0) return 0;
  hipStream_t s = as_stream(stream);
  double *v0 = work, *v1 = work + n_nodes;
  hipLaunchKernelGGL(deg_pows_init_kernel, dim3(grid_for(n_nodes)), dim3(256), 0, s, deg, v0, out, n_nodes);
  for (int m = 1; m < 4; ++m) {
    hipLaunchKernelGGL(deg_pows_kernel, dim3(grid_for(n_nodes)), dim3(256), 0, s, rowptr, col, w, v0, v1, out, m, n_nodes);
    double* t = v0; v0 = v1; v1 = t;
  }
  return check_launch("deg_pows");
}
