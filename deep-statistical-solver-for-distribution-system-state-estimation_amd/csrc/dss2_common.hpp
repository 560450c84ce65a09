// Shared helpers for the gfx950 kernels of libdss2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// 32x32 MFMA accumulator register r of lane -> row inside the 32-row block
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); CDNA4 C/D layout.
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Orders LDS traffic between lanes of ONE wave (DS ops of a wave execute in order; this keeps
// the compiler from moving accesses across the hand-off and drains lgkmcnt).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kMaxLdsBytes = 160 * 1024;

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
