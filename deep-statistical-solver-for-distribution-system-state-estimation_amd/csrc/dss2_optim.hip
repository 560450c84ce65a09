// Fused multi-tensor Adamax (torch.optim.Adamax semantics, the optimizer of the reference driver,
// /root/reference/dss2_run.py:91-92,143): one launch updates every parameter tensor of the model.
//   exp_avg = b1*exp_avg + (1-b1)*g ;  exp_inf = max(b2*exp_inf, |g| + eps) ;
//   p -= lr / (1 - b1^t) * exp_avg / exp_inf           (weight_decay: g += wd * p first)
#include "dss2_weightspace.hpp"

namespace dss2 {

__global__ void adamax_tick_kernel(float* __restrict__ step_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_dev[0] += 1.f;
}

// step_dev != NULL: the 1-based step count lives on the device (already advanced by adamax_tick_kernel), so the launch can
// sit inside a hipGraph and every replay uses the next count; NULL: bias_corr1 was computed by the host.
constexpr int ADAMAX_CHUNK = 96;      // descriptors per launch, by value in the kernel arguments (96 x 40 B < 4 KB)
struct AdamaxTable { dss2_adamax_desc d[ADAMAX_CHUNK]; };

__global__ void __launch_bounds__(256) adamax_kernel(const AdamaxTable tab, float lr, float beta1,
                                                     float beta2, float eps, float weight_decay, float bias_corr1,
                                                     const float* __restrict__ step_dev) {
  const dss2_adamax_desc& d = tab.d[blockIdx.y];
  if (step_dev) bias_corr1 = 1.f - powf(beta1, step_dev[0]);
  const float clr = lr / bias_corr1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = d.grad[i];
    const float p = d.param[i];
    if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
    const float m = fmaf(beta1, d.exp_avg[i], (1.f - beta1) * g);        // lerp(exp_avg, g, 1-b1)
    const float u = fmaxf(beta2 * d.exp_inf[i], fabsf(g) + eps);
    d.exp_avg[i] = m;
    d.exp_inf[i] = u;
    d.param[i] = p - clr * (m / u);
  }
}

// One launch for every tensor of a flat gradient bucket (descriptor table in device memory, see dss2_hip.h).
__global__ void __launch_bounds__(256) adamax_flat_kernel(const dss2_adamax_flat_desc* __restrict__ descs,
                                                          const float* __restrict__ grad_base, float lr, float beta1, float beta2,
                                                          float eps, float weight_decay, float bias_corr1, float* step_dev,
                                                          unsigned* counter) {
  const dss2_adamax_flat_desc d = descs[blockIdx.y];
  if (step_dev) bias_corr1 = 1.f - powf(beta1, step_dev[0] + 1.f);      // this step's (1-based) count
  const float clr = lr / bias_corr1;
  const float* __restrict__ grad = grad_base + d.grad_off;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = grad[i];
    const float p = d.param[i];
    if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
    const float m = fmaf(beta1, d.exp_avg[i], (1.f - beta1) * g);
    const float u = fmaxf(beta2 * d.exp_inf[i], fabsf(g) + eps);
    d.exp_avg[i] = m;
    d.exp_inf[i] = u;
    d.param[i] = p - clr * (m / u);
  }
  if (!step_dev) return;
  // The last workgroup to arrive advances the device-side count.  Every workgroup has READ the count before it arrives: the
  // barrier drains its loads (s_waitcnt vmcnt(0)), and only then is the arrival posted.  Nothing is published through memory, so
  // a relaxed device-scope atomic is enough -- a release fence here would be an L2 write-back per workgroup on MI355X.
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1) {
      step_dev[0] += 1.f;
      *counter = 0u;
    }
  }
}

// (small_gemm_body / small_gemm_tile: dss2_weightspace.hpp)
__global__ void __launch_bounds__(256) small_gemm_kernel(const dss2_sgemm_desc* __restrict__ descs, float* base_out) {
  __shared__ __attribute__((aligned(16))) float As[SG_T][SG_LDA];   // As[i][k]
  __shared__ float Bs[SG_KC][SG_T + 1];                             // Bs[k][j]
  small_gemm_tile(descs + blockIdx.y, base_out, As, Bs, (int)blockIdx.x);
}

}  // namespace dss2

static int dss2_small_gemm_launch(const dss2_sgemm_desc* descs, int n_desc, int max_tiles, float* base_out, void* stream);
extern "C" int dss2_small_gemm(const dss2_sgemm_desc* descs, int n_desc, int max_tiles, float* base_out, void* stream) {
  DSS2_RECORD([descs, n_desc, max_tiles, base_out](void* s_) { return dss2_small_gemm_launch(descs, n_desc, max_tiles, base_out, s_); });
  return dss2_small_gemm_launch(descs, n_desc, max_tiles, base_out, stream);
}
static int dss2_small_gemm_launch(const dss2_sgemm_desc* descs, int n_desc, int max_tiles, float* base_out, void* stream) {
  if (n_desc <= 0) return 0;
  if (max_tiles <= 0) { dss2::set_error("small_gemm: max_tiles must be positive"); return 2; }
  hipLaunchKernelGGL(dss2::small_gemm_kernel, dim3(max_tiles, n_desc), dim3(256), 0, dss2::as_stream(stream),
                     descs, base_out);
  return dss2::check_launch("small_gemm");
}

static int adamax_launch(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float bc1, const float* step_dev, hipStream_t s) {
  for (int c0 = 0; c0 < n_desc; c0 += dss2::ADAMAX_CHUNK) {
    const int n = n_desc - c0 < dss2::ADAMAX_CHUNK ? n_desc - c0 : dss2::ADAMAX_CHUNK;
    dss2::AdamaxTable tab = {};
    int64_t max_n = 0;
    for (int i = 0; i < n; ++i) {
      tab.d[i] = descs_host[c0 + i];
      if (!tab.d[i].param || !tab.d[i].grad || !tab.d[i].exp_avg || !tab.d[i].exp_inf) { dss2::set_error("adamax_step: descriptor %d is incomplete", c0 + i); return 2; }
      if (tab.d[i].n > max_n) max_n = tab.d[i].n;
    }
    int64_t bx = (max_n + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(dss2::adamax_kernel, dim3((unsigned)bx, n), dim3(256), 0, s, tab, lr, beta1, beta2, eps, weight_decay, bc1, step_dev);
  }
  return dss2::check_launch("adamax_step");
}

static int dss2_adamax_step_launch(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
extern "C" int dss2_adamax_step(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  DSS2_RECORD([d = dss2::plan_keep(descs_host, (size_t)(n_desc > 0 ? n_desc : 0)), n_desc, lr, beta1, beta2, eps, weight_decay, step](void* s_) { return dss2_adamax_step_launch(dss2::plan_ptr(d), n_desc, lr, beta1, beta2, eps, weight_decay, step, s_); });
  return dss2_adamax_step_launch(descs_host, n_desc, lr, beta1, beta2, eps, weight_decay, step, stream);
}
static int dss2_adamax_step_launch(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  if (n_desc <= 0) return 0;
  if (!descs_host) { dss2::set_error("adamax_step: null descriptor table"); return 2; }
  if (step < 1) { dss2::set_error("adamax_step: step must be >= 1"); return 2; }
  return adamax_launch(descs_host, n_desc, lr, beta1, beta2, eps, weight_decay, 1.f - powf(beta1, (float)step), nullptr, dss2::as_stream(stream));
}

static int dss2_adamax_step_flat_launch(const dss2_adamax_flat_desc* descs_dev, int n_desc, int64_t max_n, const float* grad_base, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* step_dev, uint32_t* counter, void* stream);
extern "C" int dss2_adamax_step_flat(const dss2_adamax_flat_desc* descs_dev, int n_desc, int64_t max_n, const float* grad_base, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* step_dev, uint32_t* counter, void* stream) {
  DSS2_RECORD([descs_dev, n_desc, max_n, grad_base, lr, beta1, beta2, eps, weight_decay, step, step_dev, counter](void* s_) { return dss2_adamax_step_flat_launch(descs_dev, n_desc, max_n, grad_base, lr, beta1, beta2, eps, weight_decay, step, step_dev, counter, s_); });
  return dss2_adamax_step_flat_launch(descs_dev, n_desc, max_n, grad_base, lr, beta1, beta2, eps, weight_decay, step, step_dev, counter, stream);
}
static int dss2_adamax_step_flat_launch(const dss2_adamax_flat_desc* descs_dev, int n_desc, int64_t max_n, const float* grad_base, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* step_dev, uint32_t* counter, void* stream) {
  if (n_desc <= 0) return 0;
  if (!descs_dev || !grad_base || n_desc > 65535) { dss2::set_error("adamax_step_flat: bad arguments"); return 2; }
  if (step < 0 || (step == 0 && (!step_dev || !counter))) { dss2::set_error("adamax_step_flat: step >= 1, or step == 0 with step_dev and counter"); return 2; }
  int64_t bx = (max_n + 4095) / 4096;      // a workgroup walks up to 16 elements per thread: few, fat workgroups (and few
  if (bx > 64) bx = 64;                    // arrivals at the step counter's word)
  if (bx < 1) bx = 1;
  const float bc1 = step > 0 ? 1.f - powf(beta1, (float)step) : 1.f;
  hipLaunchKernelGGL(dss2::adamax_flat_kernel, dim3((unsigned)bx, n_desc), dim3(256), 0, dss2::as_stream(stream), descs_dev, grad_base,
                     lr, beta1, beta2, eps, weight_decay, bc1, step > 0 ? nullptr : step_dev, counter);
  return dss2::check_launch("adamax_step_flat");
}

static int dss2_adamax_step_dev_launch(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, float* step_dev, void* stream);
extern "C" int dss2_adamax_step_dev(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, float* step_dev, void* stream) {
  DSS2_RECORD([d = dss2::plan_keep(descs_host, (size_t)(n_desc > 0 ? n_desc : 0)), n_desc, lr, beta1, beta2, eps, weight_decay, step_dev](void* s_) { return dss2_adamax_step_dev_launch(dss2::plan_ptr(d), n_desc, lr, beta1, beta2, eps, weight_decay, step_dev, s_); });
  return dss2_adamax_step_dev_launch(descs_host, n_desc, lr, beta1, beta2, eps, weight_decay, step_dev, stream);
}
static int dss2_adamax_step_dev_launch(const dss2_adamax_desc* descs_host, int n_desc, float lr, float beta1, float beta2, float eps, float weight_decay, float* step_dev, void* stream) {
  if (n_desc <= 0) return 0;
  if (!descs_host || !step_dev) { dss2::set_error("adamax_step_dev: null argument"); return 2; }
  hipLaunchKernelGGL(dss2::adamax_tick_kernel, dim3(1), dim3(64), 0, dss2::as_stream(stream), step_dev);
  return adamax_launch(descs_host, n_desc, lr, beta1, beta2, eps, weight_decay, 1.f, step_dev, dss2::as_stream(stream));
}


// ---- dropout random state (see dss2_hip.h) -------------------------------------------------------------------------------
namespace dss2 {
__global__ void rng_next_kernel(unsigned long long* __restrict__ state, unsigned long long* __restrict__ snap,
                                unsigned long long host_seed, int use_host_seed) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (use_host_seed) { snap[0] = host_seed; snap[1] = 0; }
  else { snap[0] = state[0]; snap[1] = state[1]; state[1] = state[1] + 1; }
}

__global__ void __launch_bounds__(256) dropout_mask_kernel(const uint64_t* __restrict__ snap, uint32_t id, uint32_t thr, float scale,
                                                           int64_t n_rows, int h, float* __restrict__ out, int64_t ldo) {
  const uint64_t seed = snap[0], off = snap[1];
  const int groups = (h + 3) >> 2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * groups; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / groups;
    const int g = (int)(t - row * groups);
    const f32x4 m = dropout_mult4(seed, off, id, (uint32_t)row, (uint32_t)g, thr, scale);
    for (int q = 0; q < 4; ++q)
      if (4 * g + q < h) out[row * ldo + 4 * g + q] = m[q];
  }
}
}  // namespace dss2

static int dss2_rng_next_launch(uint64_t* state, uint64_t* snapshot, uint64_t host_seed, int use_host_seed, void* stream);
extern "C" int dss2_rng_next(uint64_t* state, uint64_t* snapshot, uint64_t host_seed, int use_host_seed, void* stream) {
  DSS2_RECORD([state, snapshot, host_seed, use_host_seed](void* s_) { return dss2_rng_next_launch(state, snapshot, host_seed, use_host_seed, s_); });
  return dss2_rng_next_launch(state, snapshot, host_seed, use_host_seed, stream);
}
static int dss2_rng_next_launch(uint64_t* state, uint64_t* snapshot, uint64_t host_seed, int use_host_seed, void* stream) {
  if (!snapshot || (!use_host_seed && !state)) { dss2::set_error("rng_next: null argument"); return 2; }
  hipLaunchKernelGGL(dss2::rng_next_kernel, dim3(1), dim3(64), 0, dss2::as_stream(stream),
                     reinterpret_cast<unsigned long long*>(state), reinterpret_cast<unsigned long long*>(snapshot),
                     (unsigned long long)host_seed, use_host_seed);
  return dss2::check_launch("rng_next");
}

// p -> (threshold, scale) exactly as the Python side computes them for the kernels (one definition: this one)
extern "C" void dss2_dropout_params(float p, uint32_t* thr, float* scale) {
  if (p <= 0.f) { *thr = 0u; *scale = 1.f; }
  else if (p >= 1.f) { *thr = 0u; *scale = 0.f; }
  else {
    const double t = (double)p * 4294967296.0;
    *thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    *scale = 1.f / (1.f - p);
  }
}

static int dss2_dropout_mask_launch(const uint64_t* snapshot, int32_t drop_id, float p, int64_t n_rows, int h, float* out, int64_t ldo, void* stream);
extern "C" int dss2_dropout_mask(const uint64_t* snapshot, int32_t drop_id, float p, int64_t n_rows, int h, float* out,
                                 int64_t ldo, void* stream) {
  DSS2_RECORD([=](void* s_) { return dss2_dropout_mask_launch(snapshot, drop_id, p, n_rows, h, out, ldo, s_); });
  return dss2_dropout_mask_launch(snapshot, drop_id, p, n_rows, h, out, ldo, stream);
}
static int dss2_dropout_mask_launch(const uint64_t* snapshot, int32_t drop_id, float p, int64_t n_rows, int h, float* out, int64_t ldo, void* stream) {
  if (!snapshot || !out || drop_id <= 0 || h <= 0) { dss2::set_error("dropout_mask: bad arguments"); return 2; }
  if (n_rows <= 0) return 0;
  uint32_t thr; float scale;
  dss2_dropout_params(p, &thr, &scale);
  int64_t blocks = (n_rows * ((h + 3) / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dss2::dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, dss2::as_stream(stream), snapshot,
                     (uint32_t)drop_id, thr, scale, n_rows, h, out, ldo);
  return dss2::check_launch("dropout_mask");
}


// ---- gradient through "dropout, then ReLU" of a layer output (networks.py:268-269 between the layers of the Multi* variants):
//      gpre = g * mask * (y > 0), y = the layer's post-activation output, mask regenerated from the layer's dropout spec.
namespace dss2 {
__global__ void __launch_bounds__(256) gate_grad_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out,
                                                        int64_t n_rows, int h, const uint64_t* __restrict__ snap, uint32_t id,
                                                        uint32_t thr, float scale, int relu) {
  const uint64_t seed = snap ? snap[0] : 0, off = snap ? snap[1] : 0;
  const int groups = (h + 3) >> 2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * groups; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / groups;
    const int gq = (int)(t - row * groups);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (snap) m = dropout_mult4(seed, off, id, (uint32_t)row, (uint32_t)gq, thr, scale);
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * gq + q;
      if (c < h) {
        const int64_t i = row * h + c;
        out[i] = (!relu || relu_open(y[i])) ? g[i] * m[q] : 0.f;
      }
    }
  }
}
}  // namespace dss2

static int dss2_gate_grad_launch(const float* g, const float* y, float* out, int64_t n_rows, int h, const uint64_t* snapshot, int32_t drop_id, float p, int relu, void* stream);
extern "C" int dss2_gate_grad(const float* g, const float* y, float* out, int64_t n_rows, int h, const uint64_t* snapshot, int32_t drop_id, float p, int relu, void* stream) {
  DSS2_RECORD([g, y, out, n_rows, h, snapshot, drop_id, p, relu](void* s_) { return dss2_gate_grad_launch(g, y, out, n_rows, h, snapshot, drop_id, p, relu, s_); });
  return dss2_gate_grad_launch(g, y, out, n_rows, h, snapshot, drop_id, p, relu, stream);
}
static int dss2_gate_grad_launch(const float* g, const float* y, float* out, int64_t n_rows, int h, const uint64_t* snapshot, int32_t drop_id, float p, int relu, void* stream) {
  if (!g || !out || (relu && !y) || h <= 0) { dss2::set_error("gate_grad: bad arguments"); return 2; }
  if (n_rows <= 0) return 0;
  uint32_t thr = 0; float scale = 1.f;
  if (snapshot) dss2_dropout_params(p, &thr, &scale);
  int64_t blocks = (n_rows * ((h + 3) / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dss2::gate_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, dss2::as_stream(stream), g, y, out, n_rows, h,
                     snapshot, (uint32_t)drop_id, thr, scale, relu);
  return dss2::check_launch("gate_grad");
}
