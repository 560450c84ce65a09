// Shared helpers for the gfx950 kernels of libdss2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
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

}  // namespace dss2
