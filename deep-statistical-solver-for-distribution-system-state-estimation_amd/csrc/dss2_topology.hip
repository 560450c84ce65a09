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
      auto cur = [&](int i) { return p.meta[i]; };
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
  int uniform_rows; long long N; int32_t* tile_start_w;      // (uniform_rows > 0: tile t = rows [t * uniform_rows, ..) and tile_start is WRITTEN here)
};
__global__ void __launch_bounds__(256) ell_tiles_kernel(const EllArgs a) {
  const int which = blockIdx.y, tile = blockIdx.x;
  const int D = a.D[which];
  int ts, R;
  if (a.uniform_rows > 0) {      // tiles of equal-size graphs: closed form, and this launch also writes tile_start (dss2_tiles_uniform folded in)
    const long long t0 = (long long)tile * a.uniform_rows, t1 = t0 + a.uniform_rows;
    ts = (int)(t0 < a.N ? t0 : a.N); R = (int)(t1 < a.N ? t1 : a.N) - ts;
    if (which == 0 && threadIdx.x == 0) { a.tile_start_w[tile] = ts; if (tile == a.ntiles - 1) a.tile_start_w[a.ntiles] = (int)a.N; }
  } else { ts = a.tile_start[tile]; R = a.tile_start[tile + 1] - ts; }
  const int32_t* rp = a.rowptr[which];
  if (threadIdx.x == 0 && a.uniform_rows == 0) {      // (closed-form tilings take their entry bound from the hint: no statistic.  A running maximum: read first -- thousands of tiles hammering one word with atomics cost this launch 38 us at C5;
                               //  a stale read only costs a redundant atomic)
    const int nnz = rp[ts + R] - rp[ts];
    if (nnz > a.meta[6 + which]) atomicMax(a.meta + 6 + which, nnz);
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

// ---- the whole CSR part in ONE launch for batches of equal-size graphs whose edge ranges are known (round 6) --------------------------
// What dss2_csr_build does in 14 launches (count, three scans of the whole batch, cut positions, their running maximum, fill, finalize)
// and dss2_deg_pows in 4 more is LOCAL to a graph once the graph's edge range [e0, e1) in the stored list is known: every global offset is
// (1 or 2) * e0 plus a prefix sum over the graph's own <= 192 nodes.  The device loader knows that range (uniform samples: g * e; a mixed
// batch: the prefix sum the host formed to lay the batch out), so a fresh batch -- BASELINE config C5: a new structure every step -- builds
// its structure with ONE wave per graph: counts, prefix sums, slot claims, per-row order, gcn_norm weights, incidence lists, legal cuts,
// statistics and the folded-bias row scales, all in LDS, same definitions and therefore the same bits as the general build.
// (Assembly of a C5 batch: ~26 launches, 0.19 ms of kernels + their gaps -> 5; tests/test_gpu_topology.py compares the two builds.)
struct GraphBuildArgs {
  BuildPtrs p; FinalPtrs f; int32_t* efrom_w; int32_t* eto_w;      // (efrom / eto written here, read through f)
  int32_t *rowptr, *rowptrT, *inc_rowptr, *lastcut;
  const long long* edge_ptr; int n, e_uniform, emax; long long G;
  float* deg_pows;      // [N][4] or NULL
};

// scans over a GROUP of W consecutive lanes (W = 16, 32 or 64: one graph per group)
template <int W>
__device__ __forceinline__ int group_excl_scan_add(int v, int sl) {
  int s = v;
#pragma unroll
  for (int o = 1; o < W; o <<= 1) { const int t = __shfl_up(s, o, W); if (sl >= o) s += t; }
  return s - v;
}
template <int W>
__device__ __forceinline__ int group_incl_scan_max(int v, int sl) {
#pragma unroll
  for (int o = 1; o < W; o <<= 1) { const int t = __shfl_up(v, o, W); if (sl >= o) v = v > t ? v : t; }
  return v;
}

// W lanes per graph: small graphs share a wave (15-bus graphs: four per wave -- with one graph per wave 15 of 64 lanes worked and the launch
// was bound by instruction issue: 80 us for 4 096 graphs)
template <int W>
__global__ void __launch_bounds__(256) topo_build_graphs_kernel(const GraphBuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) int32_t gsm[];
  constexpr int GPW = 64 / W;                       // graphs per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sl = lane & (W - 1), gi = lane / W;     // lane inside its group, group inside the wave
  const long long g = ((long long)blockIdx.x * 4 + wv) * GPW + gi;
  const bool live = g < a.G;                        // (groups beyond the batch idle through the wave's steps: no early return, the syncs are wave-wide)
  const int n = a.n, emax = a.emax;
  const BuildPtrs& p = a.p;
  const long long gg = live ? g : 0;
  const long long e0 = a.edge_ptr ? a.edge_ptr[gg] : gg * a.e_uniform, e1 = a.edge_ptr ? a.edge_ptr[gg + 1] : (gg + 1) * (long long)a.e_uniform;
  int ne = live ? (int)(e1 - e0) : 0;
  const long long nb = gg * n;
  const int dbl = p.doubled ? 2 : 1;
  // per-group LDS: 11 arrays of n + 2 ints, three key arrays, the local endpoints, two double vectors
  const int per_group = 11 * (n + 2) + 2 * (2 * emax) + 2 * emax + 2 * emax + 4 * (n + 2) + 2;
  int32_t* w = gsm + (size_t)(wv * GPW + gi) * per_group;
  int32_t *cnt = w, *cntT = cnt + (n + 2), *cntI = cntT + (n + 2), *cover = cntI + (n + 2), *cur = cover + (n + 2), *curT = cur + (n + 2),
          *curI = curT + (n + 2), *rp = curI + (n + 2), *rpT = rp + (n + 2), *rpI = rpT + (n + 2), *cut = rpI + (n + 2);
  int32_t *keys = cut + (n + 2), *keysT = keys + 2 * emax, *keysI = keysT + 2 * emax;
  int32_t *ela = keysI + 2 * emax, *elb = ela + emax;      // the graph's stored edges, local endpoints: everything after the counting pass reads these
  double* v0 = reinterpret_cast<double*>(elb + emax + ((size_t)(elb + emax) & 4 ? 1 : 0));      // (8-byte aligned)
  double* v1 = v0 + n;
  if (ne < 0 || ne > emax) { if (sl == 0) p.meta[5] = 1; ne = 0; }
  for (int i = sl; i < 7 * (n + 2); i += W) cnt[i] = 0;      // cnt .. curI
  wave_lds_sync();
  // ---- counts (dss2_csr_build: topo_count_kernel)
  for (int j = sl; j < ne; j += W) {
    const long long A = p.ei[e0 + j], B = p.ei[p.E + e0 + j];
    long long la = A - nb, lb = B - nb;
    if (la < 0 || la >= n || lb < 0 || lb >= n) { p.meta[5] = 1; la = lb = 0; }      // an endpoint outside this graph's rows: the error flag (the arrays are then meaningless, not out of bounds)
    a.efrom_w[e0 + j] = (int32_t)A;
    a.eto_w[e0 + j] = (int32_t)B;
    ela[j] = (int)la; elb[j] = (int)lb;
    atomicAdd(cnt + lb, 1); atomicAdd(cntT + la, 1);
    if (p.doubled) { atomicAdd(cnt + la, 1); atomicAdd(cntT + lb, 1); }
    atomicAdd(cntI + la, 1); atomicAdd(cntI + lb, 1);
    const int lo = (int)(la < lb ? la : lb), hi = (int)(la < lb ? lb : la);
    atomicAdd(cover + lo + 1, 1); atomicAdd(cover + hi + 1, -1);
  }
  wave_lds_sync();
  // ---- exclusive sums over the graph's nodes (chunks of W with running carries), global offsets added where they are stored
  {
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int base = 0; base <= n + 1; base += W) {      // (uniform trip count: n is the same for every group)
      const int i = base + sl;
      const int x0 = i <= n ? cnt[i] : 0, x1 = i <= n ? cntT[i] : 0, x2 = i <= n ? cntI[i] : 0, x3 = i <= n + 1 ? cover[i] : 0;
      const int s0 = group_excl_scan_add<W>(x0, sl) + c0, s1 = group_excl_scan_add<W>(x1, sl) + c1, s2 = group_excl_scan_add<W>(x2, sl) + c2,
                s3 = group_excl_scan_add<W>(x3, sl) + c3;
      // sum of cover[0 .. q] = number of edges with lo < q <= hi; a cut before local row q is legal iff that is 0 (or q = 0 / q = n)
      if (i <= n) { rp[i] = s0; rpT[i] = s1; rpI[i] = s2; cut[i] = (i == 0 || i == n || s3 + x3 == 0) ? i : -1; }
      c0 = __shfl(s0 + x0, W - 1, W); c1 = __shfl(s1 + x1, W - 1, W); c2 = __shfl(s2 + x2, W - 1, W); c3 = __shfl(s3 + x3, W - 1, W);
    }
  }
  wave_lds_sync();
  if (live) {
    const int32_t b2 = (int32_t)(dbl * e0), bI = (int32_t)(2 * e0);
    for (int i = sl; i <= n; i += W) { a.rowptr[nb + i] = b2 + rp[i]; a.rowptrT[nb + i] = b2 + rpT[i]; a.inc_rowptr[nb + i] = bI + rpI[i]; }
  }
  // ---- legal cuts: running maximum inside the graph (its first row is always one)
  {
    int carry = 0;
    for (int base = 0; base <= n; base += W) {
      const int q = base + sl;
      int v = q <= n ? cut[q] : -1;
      v = group_incl_scan_max<W>(v, sl);
      v = v > carry ? v : carry;
      if (q <= n) { if (live) a.lastcut[nb + q] = (int32_t)(nb + v); cut[q] = v; }
      carry = __shfl(v, W - 1, W);
    }
  }
  // ---- slot claims (topo_fill_kernel): keys hold GLOBAL directed edge ids
  for (int j = sl; j < ne; j += W) {
    const int la = ela[j], lb = elb[j];
    const int32_t d = (int32_t)(e0 + j);
    keys[rp[lb] + atomicAdd(cur + lb, 1)] = d;
    keysT[rpT[la] + atomicAdd(curT + la, 1)] = d;
    if (p.doubled) {
      keys[rp[la] + atomicAdd(cur + la, 1)] = (int32_t)(p.E + d);
      keysT[rpT[lb] + atomicAdd(curT + lb, 1)] = (int32_t)(p.E + d);
    }
    keysI[rpI[la] + atomicAdd(curI + la, 1)] = d;
    keysI[rpI[lb] + atomicAdd(curI + lb, 1)] = (int32_t)((uint32_t)d | kFlip);
  }
  wave_lds_sync();
  // 1 / sqrt(in-degree) of every node of the graph, once (two correctly rounded operations each; the slot counters are dead: their space)
  float* isd = reinterpret_cast<float*>(cur);
  for (int r = sl; r < n; r += W) isd[r] = inv_sqrt_deg(rp[r + 1] - rp[r]);
  wave_lds_sync();
  // ---- per-row order and the final arrays (topo_finalize_kernel)
  int mdeg = 0, mdegT = 0, smax = 0, smin = 0x7fffffff, scnt = 0;
  const FinalPtrs& f = a.f;
  if (live) {
    for (int r = sl; r < n; r += W) {
      const int32_t b2 = (int32_t)(dbl * e0);
#pragma unroll 1
      for (int which = 0; which < 2; ++which) {
        const int32_t* rpp = which == 0 ? rp : rpT;
        int32_t* kk = which == 0 ? keys : keysT;
        const int l0 = rpp[r], cntr = rpp[r + 1] - l0;
        sort_row(kk + l0, cntr);
        for (int k = 0; k < cntr; ++k) {
          const int d = kk[l0 + k];
          const bool flip = d >= p.E;
          const int e = flip ? d - (int)p.E : d;
          const int A = ela[e - (int)e0], B = elb[e - (int)e0];
          const int ls = flip ? B : A, lt = flip ? A : B;
          const int src = (int)nb + ls, tgt = (int)nb + lt;
          // gcn_norm(add_self_loops=False): in-degree on the (doubled) directed list, both CSRs carry the same weight
          const float wv_ = __fmul_rn(isd[ls], isd[lt]);
          const int en = (int)((uint32_t)e | ((flip && !p.no_flip) ? kFlip : 0u));
          const int o = b2 + l0 + k;
          if (which == 0) { f.col[o] = src; f.ent[o] = en; f.perm[o] = d; f.w[o] = wv_; }
          else { f.colT[o] = tgt; f.entT[o] = en; f.permT[o] = d; f.wT[o] = wv_; }
        }
      }
      const int i0 = rpI[r], ni = rpI[r + 1] - i0;
      sort_row(keysI + i0, ni);
      for (int k = 0; k < ni; ++k) f.inc_ent[2 * e0 + i0 + k] = keysI[i0 + k];
      const int dg = rp[r + 1] - rp[r];
      f.deg[nb + r] = (float)dg;
      mdeg = max(mdeg, dg);
      mdegT = max(mdegT, rpT[r + 1] - rpT[r]);
      const int q = r + 1;                         // a legal cut at q closes the segment [lastcut[q - 1], q)
      if (cut[q] == q) { const int len = q - cut[q - 1]; smax = max(smax, len); smin = min(smin, len); ++scnt; }
    }
  }
  int nlive = live && sl == 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) {
    mdeg = max(mdeg, __shfl_xor(mdeg, o)); mdegT = max(mdegT, __shfl_xor(mdegT, o));
    smax = max(smax, __shfl_xor(smax, o)); smin = min(smin, __shfl_xor(smin, o)); scnt += __shfl_xor(scnt, o); nlive += __shfl_xor(nlive, o);
  }
  if (lane == 0) {
    // (PLAIN loads: an agent-scope atomic load of one word from thousands of waves is served as slowly as an atomic -- 36 us for 4 096
    //  of them in ell_tiles_kernel, round 6; a stale value only costs a redundant atomic)
    auto curm = [&](int i) { return p.meta[i]; };
    if (mdeg > curm(0)) atomicMax(p.meta + 0, mdeg);
    if (mdegT > curm(1)) atomicMax(p.meta + 1, mdegT);
    if (scnt) {
      if (smax > curm(2)) atomicMax(p.meta + 2, smax);
      if (smin < curm(4)) atomicMin(p.meta + 4, smin);
    }
    // the number of segments: every graph has at least one (its rows end at a legal cut), so the wave of graph 0 posts G for all of them and a
    // wave posts only what its graphs have beyond one each -- thousands of same-address atomic adds, one per graph, cost this launch 200 us
    if (blockIdx.x == 0 && wv == 0) atomicAdd(p.meta + 3, (int)a.G);
    if (scnt > nlive) atomicAdd(p.meta + 3, scnt - nlive);
  }
  // ---- [deg, A deg, A^2 deg, A^3 deg] in float64 (dss2_deg_pows), CSR by target, in the order just written
  if (a.deg_pows) {
    wave_lds_sync();      // (the sorted keys of every row are in LDS)
    for (int r = sl; r < n; r += W) { const float dg = (float)(rp[r + 1] - rp[r]); v0[r] = (double)dg; if (live) a.deg_pows[(nb + r) * 4] = dg; }
    wave_lds_sync();
    for (int m = 1; m < 4; ++m) {
      for (int r = sl; r < n; r += W) {
        double sacc = 0.0;
        const int l0 = rp[r], cntr = rp[r + 1] - l0;
        for (int k = 0; k < cntr; ++k) {
          const int d = keys[l0 + k];
          const bool flip = d >= p.E;
          const int e = flip ? d - (int)p.E : d;
          const int A = ela[e - (int)e0], B = elb[e - (int)e0];
          const int ls = flip ? B : A, lt = flip ? A : B;
          const float wv_ = __fmul_rn(isd[ls], isd[lt]);
          sacc = __dadd_rn(sacc, __dmul_rn((double)wv_, v0[ls]));
        }
        v1[r] = sacc;
        if (live) a.deg_pows[(nb + r) * 4 + m] = (float)sacc;
      }
      wave_lds_sync();
      double* t = v0; v0 = v1; v1 = t;
    }
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

static int graphs_group_width(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }
static size_t graphs_lds_bytes(int n, int emax) {
  const int gpw = 64 / graphs_group_width(n);
  return ((size_t)4 * gpw * (11 * (n + 2) + 2 * (2 * emax) + 2 * emax + 2 * emax + 4 * (n + 2) + 2) + 8) * 4;
}

extern "C" int dss2_csr_build_graphs_supported(int32_t nodes_per_graph, int32_t max_edges_per_graph) {
  return nodes_per_graph >= 1 && nodes_per_graph <= 192 && max_edges_per_graph >= 1 && max_edges_per_graph <= 1024 &&
         graphs_lds_bytes(nodes_per_graph, max_edges_per_graph) <= (size_t)kMaxLdsBytes / 2;
}

extern "C" int dss2_csr_build_graphs(const dss2_csr_build_args* ap, int32_t nodes_per_graph, const int64_t* edge_ptr, int32_t edges_per_graph,
                                     int32_t max_edges_per_graph, float* deg_pows, void* stream) {
  DSS2_NOT_IN_PLAN("dss2_csr_build_graphs");
  if (!ap) { set_error("csr_build_graphs: null argument"); return 2; }
  const dss2_csr_build_args& a = *ap;
  if (a.n_edges <= 0 || a.n_nodes <= 0 || nodes_per_graph <= 0 || a.n_nodes % nodes_per_graph) { set_error("csr_build_graphs: a batch of whole graphs of nodes_per_graph rows expected"); return 2; }
  if (!dss2_csr_build_graphs_supported(nodes_per_graph, max_edges_per_graph)) { set_error("csr_build_graphs: graphs of %d nodes / %d edges are beyond this build (use dss2_csr_build)", nodes_per_graph, max_edges_per_graph); return 2; }
  if (!edge_ptr && (edges_per_graph <= 0 || (int64_t)edges_per_graph * (a.n_nodes / nodes_per_graph) != a.n_edges)) { set_error("csr_build_graphs: uniform edges_per_graph does not match n_edges"); return 2; }
  if (!a.edge_index || !a.rowptr || !a.col || !a.ent || !a.perm || !a.w || !a.rowptrT || !a.colT || !a.entT || !a.permT || !a.wT ||
      !a.inc_rowptr || !a.inc_ent || !a.efrom || !a.eto || !a.deg || !a.lastcut || !a.meta) { set_error("csr_build_graphs: null argument"); return 2; }
  hipStream_t s = as_stream(stream);
  GraphBuildArgs g = {};
  g.p.ei = a.edge_index; g.p.E = a.n_edges; g.p.N = a.n_nodes; g.p.E2 = a.doubled ? 2 * a.n_edges : a.n_edges; g.p.doubled = a.doubled ? 1 : 0;
  g.p.no_flip = a.no_flip ? 1 : 0; g.p.meta = a.meta;
  g.f.col = a.col; g.f.ent = a.ent; g.f.perm = a.perm; g.f.w = a.w; g.f.colT = a.colT; g.f.entT = a.entT; g.f.permT = a.permT; g.f.wT = a.wT;
  g.f.inc_ent = a.inc_ent; g.f.deg = a.deg;
  g.efrom_w = a.efrom; g.eto_w = a.eto; g.rowptr = a.rowptr; g.rowptrT = a.rowptrT; g.inc_rowptr = a.inc_rowptr; g.lastcut = a.lastcut;
  g.edge_ptr = reinterpret_cast<const long long*>(edge_ptr); g.n = nodes_per_graph; g.e_uniform = edges_per_graph; g.emax = max_edges_per_graph;
  g.G = a.n_nodes / nodes_per_graph; g.deg_pows = deg_pows;
  hipLaunchKernelGGL(meta_init_kernel, dim3(1), dim3(64), 0, s, a.meta);
  const int W = graphs_group_width(nodes_per_graph), per_wg = 4 * (64 / W);
  const unsigned grid = (unsigned)((g.G + per_wg - 1) / per_wg);
  const size_t lds = graphs_lds_bytes(nodes_per_graph, max_edges_per_graph);
#define DSS2_GRAPHS(WW) { static std::atomic<uint32_t> lds_done{0}; \
    if (ensure_max_lds(reinterpret_cast<const void*>(topo_build_graphs_kernel<WW>), lds_done, "csr_build_graphs")) return 1; \
    hipLaunchKernelGGL(topo_build_graphs_kernel<WW>, dim3(grid), dim3(256), lds, s, g); }
  if (W == 16) DSS2_GRAPHS(16) else if (W == 32) DSS2_GRAPHS(32) else DSS2_GRAPHS(64)
#undef DSS2_GRAPHS
  return check_launch("csr_build_graphs");
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
  a.uniform_rows = b.uniform_rows; a.N = b.n_nodes; a.tile_start_w = const_cast<int32_t*>(b.tile_start);
  if (b.uniform_rows < 0 || (b.uniform_rows > 0 && (b.n_nodes <= 0 || b.uniform_rows > b.tm || (int64_t)b.uniform_rows * b.ntiles < b.n_nodes))) {
    set_error("ell_tiles_build: uniform_rows / n_nodes do not describe ntiles tiles of <= tm rows"); return 2;
  }
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
