// Split-plane layer chain for 96-row tiles (C3: 70-bus graphs, one per tile): dss2_gemm_chain_sp.hip with three row blocks per wave.
//
// One wave per 32-column group owns all 96 rows of its columns (144 accumulator registers: one wave per SIMD, 512 registers), so
// the Horner hops stay wave-private -- no barriers besides the two per layer -- and every weight fragment is fetched ONCE per
// column group.  The first form of this file ran three waves per column group (one row block each, 168 registers: source in
// profiles/experiments/r03_gemm_chain_sp3_three_waves_per_group.hip.txt): its GEMM phase took 26 K cycles for 13.8 K of matrix-pipe
// time -- the three waves of a group each fetch the group's weight fragments (864 KB per tile and layer through a 64 B/clk L1) --
// and the slot hand-offs cost five workgroup barriers per layer with 8.5 K cycles of skew at the first one: 222 -> 207 us per chain.
// The fp32-tile form with one wave per SIMD (gemm_chain_kernel<3,3,4,1,true>, round 2) was issue-bound by the in-register operand
// split; with the planes in LDS the GEMM phase has no VALU work and a lone wave keeps the pipe fed (one request per MFMA gap).
// LDS: 4 x 24 KB + the ELL slice, one workgroup of four waves per CU.
// Compiled without packed fp32 ops like the other bf16x6 translation units (build.sh, dss2_gemm_chain16.hip).
#include <stdlib.h>

#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

#define S3STAMP(slot) CSTAMP(slot)
constexpr int S3_TM = 96;
constexpr int S3_RS = 40;                      // bf16 per plane row: 32 k + 8 pad (80-byte rows: conflict-free ds_read_b128)
constexpr int S3_PLANE = S3_TM * S3_RS;        // bf16 per plane
constexpr int S3_SLOT = S3_TM * 32;            // floats per Horner slot
constexpr int S3_REGION = 2 * S3_SLOT;         // floats per column group: two slots = 24 KB >= three planes (22.5 KB)

__device__ __forceinline__ void s3_barrier() {      // LDS-only hand-off: the Y stores of the epilogue stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef uint32_t u32x2_s3 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void s3_store_split(__bf16* dst, const f32x4 v) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(v[0], v[1], h0, m0, l0);
  split3_pair(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<u32x2_s3*>(dst) = u32x2_s3{h0, h1};
  *reinterpret_cast<u32x2_s3*>(dst + S3_PLANE) = u32x2_s3{m0, m1};
  *reinterpret_cast<u32x2_s3*>(dst + 2 * S3_PLANE) = u32x2_s3{l0, l1};
}

template <int NMAT>
__global__ void __launch_bounds__(256) gemm_chain_sp3_kernel(const dss2_gemm_prop_args p, const ChainTable ct) {
  constexpr int TM = S3_TM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int ncg = nthreads >> 6;
  const int cg = wave;
  const int tile = blockIdx.x;
  const uint64_t drop_seed = p.drop_state ? p.drop_state[0] : 0, drop_off = p.drop_state ? p.drop_state[1] : 0;
  __bf16* xpl = reinterpret_cast<__bf16*>(smem);               // stripe s: xpl + s * (2 * S3_REGION)
  int2* ell = reinterpret_cast<int2*>(smem + ncg * S3_REGION);
  const int D = p.ell_width;
  const int ts = p.tile_start[tile];
  const int R = p.tile_start[tile + 1] - ts;
  const int kq = p.kpad >> 2;
  const int c32 = lane & 31, half = lane >> 5;
  const int nks = p.kpad >> 4;
  const __bf16* xa = xpl + c32 * S3_RS + half * 8;
  float* slot0 = smem + cg * S3_REGION;
  __bf16* own_planes = xpl + cg * (2 * S3_REGION);
  const int cq = (lane & 7) * 4, r8 = lane >> 3;
  const int col0 = cg * 32 + cq;
  const bool col_ok = col0 < p.hout;
  constexpr int NRP = 12;              // row pieces per lane: rows r8 + 8 i

  // ---- stage the tile's ELL slice and the first layer's input tile as split planes (zero padded to 96 x kpad)
  {
    const int2* src = reinterpret_cast<const int2*>(p.ell_tiles) + (size_t)tile * D * TM;
    for (int idx = tid; idx < D * TM; idx += nthreads) ell[idx] = src[idx];
  }
  for (int idx = tid; idx < TM * kq; idx += nthreads) {
    const int r = idx / kq, c = (idx - r * kq) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < R && c < p.kreal) v = *reinterpret_cast<const f32x4*>(p.X + (size_t)(ts + r) * p.ldx + c);
    s3_store_split(xpl + (c >> 5) * (2 * S3_REGION) + r * S3_RS + (c & 31), v);
  }
  bf16x8 b0[3][NMAT];
  auto load_b = [&](const bf16x8* __restrict__ bp16, bf16x8 (&bb)[3][NMAT], int ks) {
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bb[pl][m] = bp16[(((size_t)(m * ncg + cg) * nks + ks) * 3 + pl) * 64 + lane];
  };
  load_b(reinterpret_cast<const bf16x8*>(ct.l[0].Bp), b0, 0);
  S3STAMP(0);
  s3_barrier();
  S3STAMP(1);

  for (int li = 0; li < ct.n; ++li) {
    const dss2_chain_layer& L = ct.l[li];    // uniform: scalar loads from the kernel-argument segment
    const bf16x8* __restrict__ bp16 = reinterpret_cast<const bf16x8*>(L.Bp);
    f32x16 acc[3][NMAT];

    // ---- tile GEMM, 16 k per step: B fragments (L2) ping-pong one step ahead, A fragments (LDS planes) one row block ahead;
    // one memory request per MFMA gap (dss2_gemm_chain_sp.hip)
    {
      bf16x8 b1[3][NMAT], a[3][3];
      auto load_a = [&](bf16x8 (&af)[3], int rb, int ks) {
        const __bf16* src = xa + (ks >> 1) * (2 * S3_REGION) + rb * 32 * S3_RS + (ks & 1) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(src + pl * S3_PLANE);
      };
      auto mma = [&](const bf16x8 (&af)[3], const bf16x8 (&b)[3][NMAT], f32x16 (&c)[NMAT], const bool first) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0][m], first ? zero : c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[2][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1][m], c[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < NMAT; ++m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0][m], c[m], 0, 0, 0);
      };
      // a[rb] holds row block rb's fragment of the current step; it is re-requested for the next step right after its MFMAs
      // (branch-free: the last step re-requests its own operands)
      auto step = [&](const bf16x8 (&bc)[3][NMAT], bf16x8 (&bn)[3][NMAT], int ks, const bool first) {
        const int kn = ks + 1 < nks ? ks + 1 : ks;
        load_b(bp16, bn, kn);
        mma(a[0], bc, acc[0], first);
        load_a(a[0], 0, kn);
        mma(a[1], bc, acc[1], first);
        load_a(a[1], 1, kn);
        mma(a[2], bc, acc[2], first);
        load_a(a[2], 2, kn);
        // gaps 1-9: the next step's weight fragments; gaps 19-21, 37-39: the re-requests of row blocks 0 and 1; row block 2's after the last MFMA
#pragma unroll
        for (int i = 0; i < 3 * NMAT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMAT - 3 * NMAT, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMAT - 3, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NMAT - 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      };
      load_a(a[0], 0, 0); load_a(a[1], 1, 0); load_a(a[2], 2, 0);
      step(b0, b1, 0, true);
      int ks = 1;
      for (; ks + 2 <= nks; ks += 2) {
        step(b1, b0, ks, false);
        step(b0, b1, ks + 1, false);
      }
      if (ks < nks) step(b1, b0, ks, false);
    }
    S3STAMP(2 + li * 6 + 0);      // GEMM phase done
    // ---- what the epilogue reads from HBM per row, requested before the hops (rowv opaque: see dss2_gemm_chain_sp.hip)
    int rowv = r8;
    asm volatile("" : "+v"(rowv));
    const bool has_pre = L.prebias != nullptr, has_dm = L.dmask != nullptr, has_rs = L.relu_src != nullptr, has_add = L.add_src != nullptr;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (L.bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(L.bias + col0);
    auto grow_of = [&](int i) { const int row = rowv + 8 * i; return (size_t)(ts + (row < R ? row : 0)); };      // (clamped: loads only)
    f32x4 gate[NRP];
    if (has_rs && col_ok) {
#pragma unroll
      for (int i = 0; i < NRP; ++i) gate[i] = *reinterpret_cast<const f32x4*>(L.relu_src + grow_of(i) * p.ld_relu + col0);
    }
    s3_barrier();      // every wave is done with this layer's planes: the slots below go over the wave's own stripe
    S3STAMP(2 + li * 6 + 1);

    // ---- Horner on row pieces, wave-private: T in one slot, G_m in the other; U = G_m + P T replaces G_m
    f32x4 U[NRP];
    {
      auto put = [&](float* slot, int m) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) slot[(rb * 32 + acc_row(r, half)) * 32 + c32] = acc[rb][m][r];
      };
      put(slot0, NMAT - 1);
#pragma unroll
      for (int m = NMAT - 2; m >= 0; --m) {
        float* cur = slot0 + (((NMAT - 2 - m) & 1) ? S3_SLOT : 0);      // holds T
        float* oth = slot0 + (((NMAT - 2 - m) & 1) ? 0 : S3_SLOT);      // receives G_m, then U
        put(oth, m);
        wave_lds_sync();
        // (two passes of six row pieces: twelve gathers of 16 bytes + their entries in flight would need 100 more registers)
#pragma unroll
        for (int h6 = 0; h6 < 2; ++h6) {
          int2 en[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) { const int row = r8 + 8 * (6 * h6 + i); en[i] = ell[row]; U[6 * h6 + i] = *reinterpret_cast<const f32x4*>(oth + row * 32 + cq); }
          for (int k = 0; k < D; ++k) {
            const int kn = k + 1 < D ? k + 1 : k;
            f32x4 z[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) z[i] = *reinterpret_cast<const f32x4*>(cur + en[i].x * 32 + cq);
            int2 en_next[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) en_next[i] = ell[kn * TM + r8 + 8 * (6 * h6 + i)];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const float w = __int_as_float(en[i].y);
#pragma unroll
              for (int q = 0; q < 4; ++q) U[6 * h6 + i][q] = fmaf(w, z[i][q], U[6 * h6 + i][q]);
              en[i] = en_next[i];
            }
          }
        }
        if (m > 0) {
          wave_lds_sync();      // (every lane's gathers of `oth`'s own rows are done before they are overwritten: they read `cur`, but own rows of `oth` above)
#pragma unroll
          for (int i = 0; i < NRP; ++i) *reinterpret_cast<f32x4*>(oth + (r8 + 8 * i) * 32 + cq) = U[i];
        }
        wave_lds_sync();      // the other lanes' gathers precede the next writes into `cur` (m > 0) / into the planes (m == 0)
      }
    }

    S3STAMP(2 + li * 6 + 2);      // hops done
    S3STAMP(2 + li * 6 + 5);
    // ---- epilogue: bias / folded bias / masks / dropout / ReLU / gate / residual -> HBM and, split, the next layer's planes
    const bool keep = li + 1 < ct.n;
#pragma unroll
    for (int i = 0; i < NRP; ++i) U[i] += bias4;
    if (col_ok) {
      if (has_pre) {
        f32x4 pb4[NMAT];
#pragma unroll
        for (int m = 0; m < NMAT; ++m) pb4[m] = *reinterpret_cast<const f32x4*>(L.prebias + (size_t)m * p.hout + col0);
#pragma unroll
        for (int i = 0; i < NRP; ++i) {
          const f32x4 ps = *reinterpret_cast<const f32x4*>(p.pre_rowscale + grow_of(i) * 4);
#pragma unroll
          for (int m = 0; m < NMAT; ++m) U[i] += pb4[m] * ps[m];
        }
      }
      if (has_dm) {
#pragma unroll
        for (int i = 0; i < NRP; ++i) U[i] *= *reinterpret_cast<const f32x4*>(L.dmask + grow_of(i) * p.ld_dmask + col0);
      }
      if (L.drop_id) {
#pragma unroll      // (fully unrolled: a rolled loop indexes U at run time and sends the whole array to scratch memory)
        for (int i = 0; i < NRP; ++i)
          U[i] *= dropout_mult4(drop_seed, drop_off, (uint32_t)L.drop_id, (uint32_t)grow_of(i), (uint32_t)(col0 >> 2), p.drop_thr, p.drop_scale);
      }
      if (L.relu & 1) {
#pragma unroll
        for (int i = 0; i < NRP; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) U[i][q] = fmaxf(U[i][q], 0.f);
      }
      if (has_rs) {
#pragma unroll
        for (int i = 0; i < NRP; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) U[i][q] = gate[i][q] > 0.f ? U[i][q] : 0.f;
      }
      if (has_add) {
#pragma unroll
        for (int i = 0; i < NRP; ++i) U[i] += *reinterpret_cast<const f32x4*>(L.add_src + grow_of(i) * p.ld_add + col0);
      }
#pragma unroll
      for (int i = 0; i < NRP; ++i)
        if (rowv + 8 * i < R) *reinterpret_cast<f32x4*>(L.Y + (size_t)(ts + rowv + 8 * i) * p.ldy + col0) = U[i];
    }
    if (keep) {
      load_b(reinterpret_cast<const bf16x8*>(ct.l[li + 1].Bp), b0, 0);      // the next layer's first fragments
#pragma unroll
      for (int i = 0; i < NRP; ++i) {
        const int row = rowv + 8 * i;
        s3_store_split(own_planes + row * S3_RS + cq, (row < R && col_ok) ? U[i] : f32x4{0.f, 0.f, 0.f, 0.f});
      }
      S3STAMP(2 + li * 6 + 3);
      s3_barrier();   // the next layer's planes are complete
      S3STAMP(2 + li * 6 + 4);
    } else {
      S3STAMP(2 + li * 6 + 3);
      S3STAMP(2 + li * 6 + 4);
    }
  }
}

static size_t chain_sp3_lds_bytes(int ncg, int ell_width) { return (size_t)ncg * S3_REGION * 4 + (size_t)S3_TM * ell_width * 8; }

bool chain_sp3_supported(const dss2_gemm_prop_args& a) {
  static const int on = [] { const char* e = getenv("DSS2_CHAIN_SP"); return e ? atoi(e) : 1; }();
  return on && a.b_format == 1 && a.nrb == 3 && a.nmat >= 2 && a.nmat <= 3 && (a.kpad & 15) == 0 && a.kpad <= 32 * a.ncg &&
         a.ncg >= 2 && a.ncg <= 4 && chain_sp3_lds_bytes(a.ncg, a.ell_width) <= (size_t)kMaxLdsBytes;
}

template <int NMAT>
static int launch_sp3(const dss2_gemm_prop_args& a, const ChainTable& ct, hipStream_t stream) {
  static std::atomic<uint32_t> lds_done{0};
  auto kern = gemm_chain_sp3_kernel<NMAT>;
  if (ensure_max_lds(reinterpret_cast<const void*>(kern), lds_done, "gemm_prop_chain(split planes, 96 rows)")) return 1;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * a.ncg), chain_sp3_lds_bytes(a.ncg, a.ell_width), stream, a, ct);
  return check_launch("gemm_prop_chain(split planes, 96 rows)");
}

int launch_chain_sp3(const dss2_gemm_prop_args& a, const ChainTable& ct, hipStream_t s) {
  return a.nmat == 2 ? launch_sp3<2>(a, ct, s) : launch_sp3<3>(a, ct, s);
}

}  // namespace dss2

#ifdef DSS2_CHAIN_STAMPS
extern "C" int dss2_debug_read_cstamps_sp3(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_cstamps), sizeof(unsigned long long) * n);
}
#endif
