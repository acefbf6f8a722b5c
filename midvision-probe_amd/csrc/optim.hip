// Fused AdamW over a flat fp32 parameter buffer (torch.optim.AdamW defaults as used by
// train_depth.py:624-627).  lr and the bias corrections travel BY VALUE with the launch (hyper == NULL), or come
// from device scalars (hyper != NULL) for a caller that replays a captured hipGraph with a changing schedule.
#include "mvp_common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(const mvp_adamw_args p) {
  const float lr = p.hyper ? p.hyper[0] : p.lr, bc1 = p.hyper ? p.hyper[1] : p.bias_c1, bc2 = p.hyper ? p.hyper[2] : p.bias_c2;
  const float step_size = lr / bc1;
  const float rbc2 = 1.0f / sqrtf(bc2);
  const float decay = 1.0f - lr * p.weight_decay;
  const int64_t n4 = p.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 w = ((float4*)p.param)[i];
    const float4 g4 = ((const float4*)p.grad)[i];
    float4 m = ((float4*)p.exp_avg)[i];
    float4 v = ((float4*)p.exp_avg_sq)[i];
    float* wp = (float*)&w; const float* gp = (const float*)&g4; float* mp = (float*)&m; float* vp = (float*)&v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = gp[e] * p.grad_scale;
      wp[e] *= decay;
      mp[e] = p.beta1 * mp[e] + (1.f - p.beta1) * g;
      vp[e] = p.beta2 * vp[e] + (1.f - p.beta2) * g * g;
      wp[e] -= step_size * mp[e] / (sqrtf(vp[e]) * rbc2 + p.eps);
    }
    ((float4*)p.param)[i] = w;
    ((float4*)p.exp_avg)[i] = m;
    ((float4*)p.exp_avg_sq)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (p.n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float g = p.grad[i] * p.grad_scale;
    float w = p.param[i] * decay;
    const float m = p.beta1 * p.exp_avg[i] + (1.f - p.beta1) * g;
    const float v = p.beta2 * p.exp_avg_sq[i] + (1.f - p.beta2) * g * g;
    w -= step_size * m / (sqrtf(v) * rbc2 + p.eps);
    p.param[i] = w; p.exp_avg[i] = m; p.exp_avg_sq[i] = v;
  }
}

}  // namespace

extern "C" int mvp_adamw_step(const mvp_adamw_args* a, void* stream) {
  if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || a->n <= 0) return MVP_EINVAL;
  if (!a->hyper && !(a->bias_c1 > 0.f && a->bias_c2 > 0.f)) return MVP_EINVAL;
  if (((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) & 15) return MVP_EINVAL;
  int64_t g = ((a->n >> 2) + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
