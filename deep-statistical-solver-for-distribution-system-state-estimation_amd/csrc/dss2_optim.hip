// Fused multi-tensor Adamax (torch.optim.Adamax semantics, the optimizer of the reference driver,
// /root/reference/dss2_run.py:91-92,143): one launch updates every parameter tensor of the model.
//   exp_avg = b1*exp_avg + (1-b1)*g ;  exp_inf = max(b2*exp_inf, |g| + eps) ;
//   p -= lr / (1 - b1^t) * exp_avg / exp_inf           (weight_decay: g += wd * p first)
#include "dss2_common.hpp"

namespace dss2 {

__global__ void __launch_bounds__(256) adamax_kernel(const dss2_adamax_desc* __restrict__ descs, float lr, float beta1,
                                                     float beta2, float eps, float weight_decay, float bias_corr1) {
  const dss2_adamax_desc d = descs[blockIdx.y];
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

}  // namespace dss2

extern "C" int dss2_adamax_step(const dss2_adamax_desc* descs, int n_desc, int64_t max_n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, void* stream) {
  if (n_desc <= 0) return 0;
  if (step < 1) { dss2::set_error("adamax_step: step must be >= 1"); return 2; }
  int64_t bx = (max_n + 255) / 256;
  if (bx > 64) bx = 64;
  const float bc1 = 1.f - powf(beta1, (float)step);
  hipLaunchKernelGGL(dss2::adamax_kernel, dim3((unsigned)bx, n_desc), dim3(256), 0, dss2::as_stream(stream), descs, lr,
                     beta1, beta2, eps, weight_decay, bc1);
  return dss2::check_launch("adamax_step");
}
