// Shared helpers for the gfx950 kernels of libdss2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <functional>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dss2_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace dss2 {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- launch plans (dss2_plan_*, dss2_api.hip): while a plan records, every launch entry point of the library appends a closure --
// its own launch function bound to COPIES of its host-side arguments -- and dss2_plan_run re-issues the closures in order from ONE C
// call.  DSS2_RECORD sits in the extern "C" wrapper of an entry point, in front of the call of its *_launch body.
bool plan_recording();
void plan_record(std::function<int(void*)> op);
template <class T> inline std::vector<T> plan_keep(const T* p, size_t n) { return p ? std::vector<T>(p, p + n) : std::vector<T>(); }
template <class T> inline const T* plan_ptr(const std::vector<T>& v) { return v.empty() ? nullptr : v.data(); }
#define DSS2_RECORD(...) do { if (dss2::plan_recording()) dss2::plan_record(__VA_ARGS__); } while (0)
// Entry points that are NOT part of a step (structure build, measurement model, z-score: they size their outputs from host-side
// facts of the batch) refuse to run while a plan records: a plan that silently skipped them would replay on stale structure.
#define DSS2_NOT_IN_PLAN(name) do { if (dss2::plan_recording()) { dss2::set_error(name ": not available while a launch plan records (it is not a launch of a training step; build it before dss2_plan_begin)"); return 3; } } while (0)

// 32x32 MFMA accumulator register r of lane -> row inside the 32-row block
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); CDNA4 C/D layout.
// ---- fp32 on the bf16 matrix pipe ("bf16x6") -----------------------------------------------------------------------------
// v = h + m + l with three bf16 pieces of 8 significant bits each (every residual is exact in fp32), so
//   a * b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-27 |a b|):
// six v_mfma_f32_32x32x16_bf16 (fp32 accumulation inside the MFMA) reproduce the fp32 product below fp32's own rounding,
// at 6 x 2 = 12 cycles per unit of k against 32 for v_mfma_f32_32x32x2_f32 (MI355X: the fp32 MFMA runs at 1/16 of bf16).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// Non-finite inputs: +-inf gives h = +-inf and the residual inf - inf = NaN, so a bf16x6 product with an infinite operand is
// NaN where the fp32 MFMA would return +-inf (NaN stays NaN).  Non-finite in => non-finite out either way (tested:
// tests/test_gpu_bf16x6.py::test_non_finite_inputs_stay_non_finite); a guard would cost two VALU ops per value in the hot loop.
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
// The same split for two values at once, each piece delivered as the packed pair {lo = piece of a, hi = piece of b}: one
// v_cvt_pk_bf16_f32 per piece and pair (the scalar form costs one conversion per piece and VALUE, plus the packing).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(ra, rb);
  l = cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

__device__ __forceinline__ void split3x4(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __bf16 a, b, c;
    split3(v[q], a, b, c);
    h[q] = a; m[q] = b; l[q] = c;
  }
}

// ---- f16x3 (round 5): an fp32 operand as TWO fp16 pieces, hi = fp16(x), lo = fp16(x - hi), after an exact power-of-two scale that
// puts the operand's maximum into [2^14, 2^15); a product is three fp16 MFMAs (lo hi + hi lo + hi hi).  See dss2_wgrad16h.hip.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// the two pieces of (a, b), each packed {a, b}
__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& h, uint32_t& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{a, b}, f16x2);
  const float ra = a - (float)hh[0], rb = b - (float)hh[1];
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra, rb}, f16x2));
}
// 2^e as a float, e in [-126, 127] (the kernels' scale exponents are within +-114: exp_of clamps).  Scaling by it with v_mul_f32 is bit for bit what
// v_ldexp_f32 does -- at HALF the issue cost: tools/micro/valu_rates.hip (end of round 6) measures v_mul / v_add / v_sub / v_fmac / v_and / v_mov at ~1.0
// wave-instructions per SIMD and ns, v_ldexp / v_cvt_* / v_max / v_min / v_cmp / v_cndmask / v_lshl_or at ~0.53, v_fma_mixlo_f16 at 0.27.
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
// v & (bit k of `word` ? ~0 : 0): a gate as v_bfe_i32 (the bit sign-extended: 0 or -1) + v_and_b32 -- six issue cycles per wave against ten for the
// `(word >> k) & 1 ? v : 0` form (v_and + v_cmp + v_cndmask); a closed gate gives +0 whatever v holds, an open one v's bits (a NaN stays a NaN).
__device__ __forceinline__ float gate_bit(float v, uint32_t word, int k) {
  return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)word, k, 1));
}
// four values gated by bits k0 .. k0 + 3 of `word` in one block: the four masks live for four instructions (left to the scheduler, a lane's 32 .. 96
// v_bfe_i32 are hoisted in front of their v_and_b32 -- as many live registers, which the 192-row chain does not have: 29 -> 205 spilled)
template <int K0>
__device__ __forceinline__ void gate_bits4(f32x4& v, uint32_t word) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  uint32_t t0, t1, t2, t3;
  asm("v_bfe_i32 %4, %8, %9, 1\n\tv_bfe_i32 %5, %8, %10, 1\n\tv_bfe_i32 %6, %8, %11, 1\n\tv_bfe_i32 %7, %8, %12, 1\n\t"
      "v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %5\n\tv_and_b32 %2, %2, %6\n\tv_and_b32 %3, %3, %7"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(word), "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
  v = f32x4{a, b, c, d};
}
// ReLU with torch's non-finite semantics (nn.ReLU, /root/reference/networks.py:269: a NaN stays a NaN): `!(v <= 0) ? v : 0` is one
// v_cmp_nle_f32 + one v_cndmask_b32 -- what fmaxf(v, 0.f) costs too (the compiler quiets fmaxf's operand with a second v_max_f32), but
// v_max_f32 returns the OTHER operand for a NaN: max(NaN, 0) = 0 turned a NaN weight into a dead column and a FINITE loss where the
// reference's loss is NaN (round 6, tools/nonfinite_probe.py; hardware NaNs are 0xFFC00000 on gfx950, so an integer max does not do either).
__device__ __forceinline__ float relu_nan(float v) { return !(v <= 0.f) ? v : 0.f; }
// The gate of ReLU's backward with torch's semantics (threshold_backward: `self <= 0 ? 0 : grad`): a NaN activation lets the gradient
// through (it is NaN there anyway: the loss is), where `y > 0` would zero it and leave a FINITE parameter gradient beside a NaN loss.
__device__ __forceinline__ bool relu_open(float y) { return !(y <= 0.f); }
// relu_nan on four values with the compares and the selects interleaved by hand: the lane masks go to four SGPR pairs and every select
// follows its compare by three instructions (the VALU-writes-SGPR -> VALU-reads-it-as-a-mask wait the compiler otherwise pads with s_nop 1).
__device__ __forceinline__ void relu_nan4(f32x4& v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  unsigned long long m0, m1, m2, m3;
  asm("v_cmp_nge_f32_e64 %4, 0, %0\n\tv_cmp_nge_f32_e64 %5, 0, %1\n\tv_cmp_nge_f32_e64 %6, 0, %2\n\tv_cmp_nge_f32_e64 %7, 0, %3\n\t"
      "v_cndmask_b32_e64 %0, 0, %0, %4\n\tv_cndmask_b32_e64 %1, 0, %1, %5\n\tv_cndmask_b32_e64 %2, 0, %2, %6\n\tv_cndmask_b32_e64 %3, 0, %3, %7"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3));
  v = f32x4{a, b, c, d};
}
// (two v_max3_f32 with |.| source modifiers, NaN operands ignored as by v_max_f32.  The fmaxf(fabsf(..)) form compiled to SEVEN four-cycle
//  instructions per vector: the compiler canonicalises every |x| with a v_max_f32 |x|, |x| of its own before the three that do the work --
//  56 instructions per layer and wave in the chains' tile maximum, 42 per tile and lane in the weight gradient's: end of round 6, tools/micro/valu_rates.hip)
// relu_nan4 that also shifts the four compare masks -- "open" = !(x <= 0), which IS the sign-bit word's bit of the value ReLU stores (+0, positive or
// NaN) -- into `word` from below, element 3 first: word = 16 word + (m3 m2 m1 m0).  Called for the lane's pieces 7 .. 0 it leaves bit 4 i + q = element q
// of piece i, the layout of y_bits / gate_bits.  v_addc_co_u32 takes the lane mask as its carry: one instruction per value where v_min_u32 +
// v_lshl_or_b32 on the stored value were two (four issue cycles each: tools/micro/valu_rates.hip).
__device__ __forceinline__ void relu_nan4_bits(f32x4& v, uint32_t& word) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  unsigned long long m0, m1, m2, m3, co;
  asm("v_cmp_nge_f32_e64 %4, 0, %0\n\tv_cmp_nge_f32_e64 %5, 0, %1\n\tv_cmp_nge_f32_e64 %6, 0, %2\n\tv_cmp_nge_f32_e64 %7, 0, %3\n\t"
      "v_cndmask_b32_e64 %0, 0, %0, %4\n\tv_cndmask_b32_e64 %1, 0, %1, %5\n\tv_cndmask_b32_e64 %2, 0, %2, %6\n\tv_cndmask_b32_e64 %3, 0, %3, %7\n\t"
      "v_addc_co_u32_e64 %8, %9, %8, %8, %7\n\tv_addc_co_u32_e64 %8, %9, %8, %8, %6\n\tv_addc_co_u32_e64 %8, %9, %8, %8, %5\n\tv_addc_co_u32_e64 %8, %9, %8, %8, %4"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "+v"(word), "=&s"(co));
  v = f32x4{a, b, c, d};
}
__device__ __forceinline__ float absmax4(float m, const f32x4 v) {
#ifdef DSS2_ABSMAX_C
  return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#endif
  asm("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(m) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  return m;
}
// max over the wave of non-negative values, in every lane (DPP row shifts and broadcasts: six VALU instructions; six ds_bpermute
// round trips cost the weight-gradient kernel ~1000 cycles per tile).  v_max_f32 ignores NaN operands.
// (one v_max_f32 with the DPP operand instead of v_mov_dpp + a canonicalising v_max + v_max; s_nop 1: the two wait states between a VALU write of a
//  register and a DPP read of it, which the compiler cannot insert inside an asm block.  Rows outside ROWS keep their value; a lane without a source reads 0.)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_max(float v) {
#ifdef DSS2_DPP_C
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true)));
#endif
  static_assert((CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118) ? ROWS == 0xf : ((CTRL == 0x142 && ROWS == 0xa) || (CTRL == 0x143 && ROWS == 0xc)), "dpp_max: control / row mask");
  if constexpr (CTRL == 0x111) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
  else if constexpr (CTRL == 0x112) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
  else if constexpr (CTRL == 0x114) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
  else if constexpr (CTRL == 0x118) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
  else if constexpr (CTRL == 0x142) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf bound_ctrl:1" : "+v"(v));
  else asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf bound_ctrl:1" : "+v"(v));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max<0x111, 0xf>(v);      // row_shr:1
  v = dpp_max<0x112, 0xf>(v);      // row_shr:2
  v = dpp_max<0x114, 0xf>(v);      // row_shr:4
  v = dpp_max<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row of 16: the row's maximum
  v = dpp_max<0x142, 0xa>(v);      // row_bcast:15 into rows 1, 3
  v = dpp_max<0x143, 0xc>(v);      // row_bcast:31 into rows 2, 3 -> lane 63: the wave's maximum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// floor(log2 m) for a finite positive fp32, clamped to [-100, 127] (0, subnormals -> -100; Inf / NaN -> 128: the scaled values stay Inf / NaN)
__device__ __forceinline__ int exp_of(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 127;
  return e < -100 ? -100 : e;
}

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Orders LDS traffic between lanes of ONE wave (DS ops of a wave execute in order; this keeps
// the compiler from moving accesses across the hand-off and drains lgkmcnt).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kMaxLdsBytes = 160 * 1024;

// ---- in-kernel dropout (nn.Dropout between the TAGConv layers, /root/reference/networks.py:268) ---------------------------
// Counter-based: the keep/drop decision of element (row, col) of layer `id` is a pure function of (seed, offset, id, row,
// col), so the forward epilogue and the backward epilogue regenerate the same mask instead of storing and re-reading an
// [N, H] fp32 tensor per layer.  Philox4x32-10 (Salmon et al., SC'11), one call per 4 consecutive columns.
struct DropSpec {            // by value inside the kernel argument structs
  const uint64_t* state;     // device {seed, offset} snapshot of this forward call (dss2_rng_next), or NULL = no dropout
  uint32_t thr;              // keep iff random uint32 >= thr  (thr = p * 2^32)
  float scale;               // 1 / (1 - p)  (0 for p >= 1: everything is dropped)
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// multipliers (0 or scale) of columns 4*cgroup .. 4*cgroup+3 of `row` in the mask of layer `id`
__device__ __forceinline__ f32x4 dropout_mult4(uint64_t seed, uint64_t offset, uint32_t id, uint32_t row, uint32_t cgroup,
                                               uint32_t thr, float scale) {
  uint32_t r[4];
  philox4x32_10(row, cgroup, id, (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32), r);
  return f32x4{r[0] >= thr ? scale : 0.f, r[1] >= thr ? scale : 0.f, r[2] >= thr ? scale : 0.f, r[3] >= thr ? scale : 0.f};
}

// Opt a kernel into > 64 KB of dynamic LDS, once per (kernel, device): `done` is the kernel's own bitmask of the devices
// that already have the attribute (the attribute belongs to the device's code object, and the autograd thread may
// launch beside the main thread, hence the atomic).  A process that drives several GPUs sets it on each of them.
inline int ensure_max_lds(const void* kern, std::atomic<uint32_t>& done, const char* what) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
  if (dev >= 0 && ((done.load(std::memory_order_acquire) >> dev) & 1u)) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    return 1;
  }
  if (dev >= 0) done.fetch_or(1u << dev, std::memory_order_release);
  return 0;
}

}  // namespace dss2
