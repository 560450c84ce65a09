#pragma once
// Shared by dss2_wgrad.hip (fp32 MFMA) and dss2_wgrad16.hip (bf16x6).
#include "dss2_common.hpp"

namespace dss2 {

// Several layers of identical shape in ONE launch (blockIdx.z = layer): the layers of a block are
// independent once all output gradients exist, and at small H one layer alone cannot fill the chip.
constexpr int WGRAD_MAX_BATCH = 8;
struct WgradBatch {
  const float* G[WGRAD_MAX_BATCH]; const float* X[WGRAD_MAX_BATCH]; float* slab[WGRAD_MAX_BATCH];
  const float* rowscale2[WGRAD_MAX_BATCH];   // per layer (NULL: plain layer)
  int n; long long slab_stride;
};

// dss2_wgrad16.hip: the bf16x6 kernel.  wgrad16_lds_bytes: dynamic LDS of its launch (0: shape not covered)
size_t wgrad16_lds_bytes(int nrb, int nmat, int hout, int hin, int ell_width);
bool wgrad16_covers(const dss2_wgrad_args& a);
int wgrad16_y_slices(int nrb, int hout, int hin);      // workgroups per tile-list slice (grid.y)
int launch_wgrad16(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb);

// dss2_wgrad16h.hip: the f16x3 kernel (32-row tiles; args.mfma_bf16 = 2 | headroom bits << 8)
bool wgrad16h_covers(const dss2_wgrad_args& a);
int launch_wgrad16h(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb);

// dss2_wgrad16th.hip: the f16x3 kernel of 96- / 192-row tiles (same args.mfma_bf16 convention)
bool wgrad16th_covers(const dss2_wgrad_args& a);
int launch_wgrad16th(const dss2_wgrad_args& a, hipStream_t stream, const WgradBatch& wb);

}  // namespace dss2
