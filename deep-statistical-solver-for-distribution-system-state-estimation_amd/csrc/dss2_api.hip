// Library-level entry points of libdss2_hip.so: error string, version, topology hash.
#include <stdarg.h>

#include "dss2_common.hpp"

namespace dss2 {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// Position-dependent element hashes, combined with a wrapping integer sum (commutative, so the
// result does not depend on the order in which workgroups arrive).
__global__ void __launch_bounds__(256) topology_hash_kernel(const int64_t* __restrict__ v, int64_t n,
                                                            unsigned long long* __restrict__ out) {
  uint64_t s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += mix64((uint64_t)v[i] * 0x100000001b3ull + mix64((uint64_t)i));
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((unsigned long long)s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)s);
}

}  // namespace dss2

extern "C" const char* dss2_last_error(void) { return dss2::g_err; }

extern "C" int dss2_version(void) { return 1; }

extern "C" int dss2_topology_hash(const int64_t* edge_index, int64_t n_elems, uint64_t* hash_out, void* stream) {
  if (n_elems <= 0) return 0;
  int64_t blocks = (n_elems + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(dss2::topology_hash_kernel, dim3((unsigned)blocks), dim3(256), 0, dss2::as_stream(stream),
                     edge_index, n_elems, reinterpret_cast<unsigned long long*>(hash_out));
  return dss2::check_launch("topology_hash");
}
