// The weight-space work of a training step in TWO launches instead of four (round 5; VERDICT r4 #2b).
//
// A step of a folded MPN block starts with  fold (dss2_small_gemm: Wf_m = W_m W2, bf_m = W_m b2)  ->  pack (dss2_pack_weights: MFMA
// fragment layouts of every weight, the folded ones included)  and ends with  slab reductions (dss2_reduce_slabs_multi)  ->  chain
// rule of the fold (dss2_small_gemm on the reduced gradient of Wf).  Each pair is a dependent chain of two latency-bound launches
// (5 + 6.6 us and 17.6 + 9..13 us at C2), most of whose work does NOT depend on the first half: of ~40 packing descriptors a few read
// folded weights, and the chain rule needs one of the five reductions.  Here each pair is one launch:
//   dss2_prep_weights    workgroups [0, F) fold, the next P_dep pack the folded matrices (the first n_dep descriptors) after all F
//                        have published, the rest pack everything else beside them;
//   dss2_finish_weights  workgroups [0, D) run the reductions the chain rule reads (the first n_dep descriptors), the next Q the
//                        chain rule -- after all D have published, and BESIDE the remaining reductions, which take the ids after.
// Hand-off (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed):
//   producer  results stored write-through (sc1: dss2_weightspace.hpp, COH), every wave drains its stores (s_waitcnt vmcnt(0)), the
//             workgroup's barrier, then ONE lane adds 1 to the counter shard (workgroup id & 7), a relaxed agent-scope atomic: no L2
//             write-back (a release fence per workgroup cost the first form of these kernels +0.4 ms per step: one buffer_wbl2 per
//             producer and an acq_rel atomic in every workgroup);
//   consumer  lanes 0..7 of wave 0 poll the eight shards with RELAXED loads (s_sleep between polls) until each holds its share of the
//             producers, ONE agent-scope acquire (this CU's L1), drain, the workgroup's barrier, then plain loads.
// Eight shards on lines of their own: 775 arrivals on one word are ~9 us of serialized atomics (11-13 ns each).
// Producers have the LOWER workgroup ids: workgroups are dispatched in id order, so every producer holds (or has left) its slot before a
// waiting workgroup takes one -- a waiting workgroup never keeps a producer from being scheduled, whatever the launch's size.
// Each waiting workgroup adds to a ninth word after its wait; the one whose add comes last (all producers have published, nobody
// polls any more) zeroes the nine words: the launch is re-entrant, hipGraph and launch-plan replays included.  Workgroups that neither
// publish nor wait touch no counter.
// Same bodies as the separate launches (dss2_weightspace.hpp): bitwise the same results.
#include <stdlib.h>

#include "dss2_weightspace.hpp"

namespace dss2 {

constexpr int WS_SHARDS = 8, WS_LINE = 32;      // counter words: shard k at word 32 k (a 128-byte line each), the waiters' word at 32 * 8
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void publish(unsigned* counters) {      // after the workgroup's last write-through store
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add((gu32*)counters + (blockIdx.x & (WS_SHARDS - 1)) * WS_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// producers = workgroups [0, n_prod): shard k receives those with id & 7 == k
__device__ __forceinline__ void wait_for(unsigned* counters, int n_prod, int n_wait) {
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    const unsigned want = k < WS_SHARDS ? (unsigned)((n_prod - k + WS_SHARDS - 1) / WS_SHARDS) : 0u;
    const gu32* w = (const gu32*)counters + (k & (WS_SHARDS - 1)) * WS_LINE;
    while (true) {
      const bool ok = k >= WS_SHARDS || __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(4);
    }
    // the waiters' count: the last one re-arms the words (every producer has added, every other waiter has left its poll loop)
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add((gu32*)counters + WS_SHARDS * WS_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)n_wait - 1) {
        for (int q = 0; q <= WS_SHARDS; ++q) __hip_atomic_store((gu32*)counters + q * WS_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE per workgroup: drops this CU's stale L1 lines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256, 4) prep_weights_kernel(const dss2_sgemm_desc* __restrict__ fold, int fold_tiles, int n_fold,
                                                           const dss2_pack_desc* __restrict__ pack, int pack_x, int n_dep,
                                                           unsigned* counters) {
  __shared__ __attribute__((aligned(16))) float As[SG_T][SG_LDA];
  __shared__ float Bs[SG_KC][SG_T + 1];
  const int F = n_fold * fold_tiles, b = blockIdx.x;
  if (b < F) {
    small_gemm_tile<true, 1>(fold + b / fold_tiles, nullptr, As, Bs, b % fold_tiles);
    if (n_dep > 0) publish(counters);
  } else {
    const int q = b - F;
    const int di = q / pack_x;
    if (di < n_dep) wait_for(counters, F, n_dep * pack_x);      // (uniform) the source is a folded weight of this launch
    pack_weights_body(pack[di], q % pack_x);
  }
}

__global__ void __launch_bounds__(256, 4) finish_weights_kernel(const ReduceTable tab, int n_red, int n_dep, int red_x,
                                                             const dss2_sgemm_desc* __restrict__ rule, int rule_tiles, int n_rule,
                                                             float* base_out, unsigned* counters) {
  __shared__ __attribute__((aligned(16))) float As[SG_T][SG_LDA];
  __shared__ float Bs[SG_KC][SG_T + 1];
  // workgroup ids: [0, D) the reductions the chain rule reads, [D, D + Q) the chain rule, the rest the other reductions
  const int D = n_dep * red_x, Q = n_rule * rule_tiles, b = blockIdx.x;
  f32x4 (&part)[16][16] = *reinterpret_cast<f32x4(*)[16][16]>(&As[0][0]);
  if (b < D) {
    const int di = b / red_x;
    reduce_slabs_body<true>(tab.d[di], ((tab.scalar_mask >> di) & 1u) != 0, b % red_x, part);
    publish(counters);
  } else if (b < D + Q) {
    const int q = b - D;
    wait_for(counters, D, Q);
    small_gemm_tile<false, 2>(rule + q / rule_tiles, base_out, As, Bs, q % rule_tiles);
  } else {
    const int r = b - Q;
    const int di = r / red_x;
    reduce_slabs_body(tab.d[di], ((tab.scalar_mask >> di) & 1u) != 0, r % red_x, part);
  }
}

}  // namespace dss2

using namespace dss2;

static int dss2_prep_weights_launch(const dss2_sgemm_desc* fold, int n_fold, int fold_tiles, const dss2_pack_desc* pack, int n_pack, int n_dep,
                                    int max_elems, uint32_t* counters, void* stream);
extern "C" int dss2_prep_weights(const dss2_sgemm_desc* fold, int n_fold, int fold_tiles, const dss2_pack_desc* pack, int n_pack, int n_dep,
                                 int max_elems, uint32_t* counters, void* stream) {
  DSS2_RECORD([fold, n_fold, fold_tiles, pack, n_pack, n_dep, max_elems, counters](void* s_) { return dss2_prep_weights_launch(fold, n_fold, fold_tiles, pack, n_pack, n_dep, max_elems, counters, s_); });
  return dss2_prep_weights_launch(fold, n_fold, fold_tiles, pack, n_pack, n_dep, max_elems, counters, stream);
}
static int dss2_prep_weights_launch(const dss2_sgemm_desc* fold, int n_fold, int fold_tiles, const dss2_pack_desc* pack, int n_pack, int n_dep,
                                    int max_elems, uint32_t* counters, void* stream) {
  if (n_fold < 0 || n_pack < 0 || (n_fold > 0 && (!fold || fold_tiles <= 0)) || (n_pack > 0 && !pack) || n_dep < 0 || n_dep > n_pack ||
      (n_dep > 0 && (n_fold == 0 || !counters))) { set_error("prep_weights: bad arguments"); return 2; }
  const int pack_x = (max_elems + 255) / 256;
  const long long total = (long long)n_fold * fold_tiles + (long long)n_pack * pack_x;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), fold, fold_tiles, n_fold, pack, pack_x > 0 ? pack_x : 1,
                     n_dep, reinterpret_cast<unsigned*>(counters));
  return check_launch("prep_weights");
}

static int dss2_finish_weights_launch(const dss2_reduce_desc* descs_host, int n_red, int n_dep, const dss2_sgemm_desc* rule, int n_rule, int rule_tiles,
                                      float* base_out, uint32_t* counters, void* stream);
extern "C" int dss2_finish_weights(const dss2_reduce_desc* descs_host, int n_red, int n_dep, const dss2_sgemm_desc* rule, int n_rule, int rule_tiles,
                                   float* base_out, uint32_t* counters, void* stream) {
  DSS2_RECORD([d = dss2::plan_keep(descs_host, (size_t)(n_red > 0 ? n_red : 0)), n_red, n_dep, rule, n_rule, rule_tiles, base_out, counters](void* s_) {
    return dss2_finish_weights_launch(dss2::plan_ptr(d), n_red, n_dep, rule, n_rule, rule_tiles, base_out, counters, s_); });
  return dss2_finish_weights_launch(descs_host, n_red, n_dep, rule, n_rule, rule_tiles, base_out, counters, stream);
}
static int dss2_finish_weights_launch(const dss2_reduce_desc* descs_host, int n_red, int n_dep, const dss2_sgemm_desc* rule, int n_rule, int rule_tiles,
                                      float* base_out, uint32_t* counters, void* stream) {
  if (n_red < 1 || n_red > REDUCE_MAX_DESC || n_dep < 1 || n_dep > n_red || !descs_host || n_rule < 1 || !rule || rule_tiles <= 0 || !counters) { set_error("finish_weights: bad arguments"); return 2; }
  ReduceTable tab = {};
  int64_t max_len = 0;
  for (int i = 0; i < n_red; ++i) {
    tab.d[i] = descs_host[i];
    if (!tab.d[i].slab || !tab.d[i].out || tab.d[i].n_slabs <= 0 || tab.d[i].len < 0) { set_error("finish_weights: descriptor %d is incomplete", i); return 2; }
    if (tab.d[i].len > max_len) max_len = tab.d[i].len;
    if ((tab.d[i].stride % 4 != 0) || (reinterpret_cast<uintptr_t>(tab.d[i].slab) & 15) || (reinterpret_cast<uintptr_t>(tab.d[i].out) & 15)) tab.scalar_mask |= 1u << i;
  }
  int red_x = (int)((max_len + 63) / 64);
  for (int i = 0; i < n_red; ++i) { const int64_t b = reduce_blocks(tab.d[i], ((tab.scalar_mask >> i) & 1u) != 0); if (b > red_x) red_x = (int)b; }
  if (red_x <= 0) { set_error("finish_weights: empty reductions"); return 2; }
  const long long total = (long long)n_red * red_x + (long long)n_rule * rule_tiles;
  hipLaunchKernelGGL(finish_weights_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), tab, n_red, n_dep, red_x, rule, rule_tiles, n_rule, base_out,
                     reinterpret_cast<unsigned*>(counters));
  return check_launch("finish_weights");
}
