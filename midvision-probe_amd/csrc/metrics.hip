// Validation metrics as fused masked reductions (per-image fp64 partials, fixed-order combine):
//   depth : evaluate_depth global metrics + optional match_scale_and_shift  (evals/utils/metrics.py:106-178,742-780)
//   snorm : evaluate_surface_norm global metrics                             (evals/utils/metrics.py:397-440)
#include "mvp_common.h"

namespace {

constexpr int MT_NCH = 32;  // chunks per image

__device__ __forceinline__ double blk_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// pass 1 (scale-invariant only): normal equations of the per-image affine fit
__global__ __launch_bounds__(256) void dm_fit_partial(const mvp_depth_metrics_args p, double* part) {
  __shared__ double red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int64_t per = (p.HW + MT_NCH - 1) / MT_NCH, i0 = ch * per, i1 = min(p.HW, i0 + per);
  double s[5] = {0, 0, 0, 0, 0};
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float g = p.gt[(int64_t)b * p.HW + i], q = p.pred[(int64_t)b * p.HW + i];
    if (g > 0.f) { s[0] += (double)q * q; s[1] += q; s[2] += 1.0; s[3] += (double)q * g; s[4] += g; }
  }
  for (int k = 0; k < 5; ++k) {
    const double v = blk_sum(s[k], red);
    if (threadIdx.x == 0) part[((int64_t)b * MT_NCH + ch) * 5 + k] = v;
  }
}

__global__ void dm_fit_final(const mvp_depth_metrics_args p, const double* part) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  double s[5] = {0, 0, 0, 0, 0};
  for (int ch = 0; ch < MT_NCH; ++ch)
    for (int k = 0; k < 5; ++k) s[k] += part[((int64_t)b * MT_NCH + ch) * 5 + k];
  // the reference solves in fp32: keep the same conditioning decisions (det != 0) on fp32 values
  const float a00 = (float)s[0], a01 = (float)s[1], a11 = (float)s[2], b0 = (float)s[3], b1 = (float)s[4];
  const float det = a00 * a11 - a01 * a01;
  float scale = 1.f, shift = 0.f;
  if (det != 0.f) { scale = (a11 * b0 - a01 * b1) / det; shift = (-a01 * b0 + a00 * b1) / det; }
  p.scale_shift[2 * b] = scale;
  p.scale_shift[2 * b + 1] = shift;
}

__global__ __launch_bounds__(256) void dm_partial(const mvp_depth_metrics_args p, double* part) {
  __shared__ double red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int64_t per = (p.HW + MT_NCH - 1) / MT_NCH, i0 = ch * per, i1 = min(p.HW, i0 + per);
  const float sc = p.scale_invariant ? p.scale_shift[2 * b] : 1.f, sh = p.scale_invariant ? p.scale_shift[2 * b + 1] : 0.f;
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float t1 = 1.25f, t2 = 1.25f * 1.25f, t3 = 1.25f * 1.25f * 1.25f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float g = p.gt[(int64_t)b * p.HW + i];
    if (!(g > 0.f)) continue;
    float q = p.pred[(int64_t)b * p.HW + i];
    if (p.scale_invariant) q = q * sc + sh;
    const float th = fmaxf(g / fmaxf(q, 1e-9f), q / fmaxf(g, 1e-9f));
    const float d = g - q;
    s[0] += 1.0; s[1] += q; s[2] += (double)q * q; s[3] += g; s[4] += (double)g * g;
    s[5] += th < t1 ? 1.0 : 0.0; s[6] += th < t2 ? 1.0 : 0.0; s[7] += th < t3 ? 1.0 : 0.0; s[8] += (double)d * d;
  }
  for (int k = 0; k < 9; ++k) {
    const double v = blk_sum(s[k], red);
    if (threadIdx.x == 0) part[((int64_t)b * MT_NCH + ch) * 9 + k] = v;
  }
}

__global__ void dm_final(const mvp_depth_metrics_args p, const double* part) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int ch = 0; ch < MT_NCH; ++ch)
    for (int k = 0; k < 9; ++k) s[k] += part[((int64_t)b * MT_NCH + ch) * 9 + k];
  const double n = s[0], ne = (n == 0.0) ? 1e-6 : n;
  const double mp = s[1] / ne, mg = s[3] / ne;
  const double vp = (s[2] - 2.0 * mp * s[1] + mp * mp * n) / ne;
  const double vg = (s[4] - 2.0 * mg * s[3] + mg * mg * n) / ne;
  float* o = p.out + (int64_t)b * 12;
  o[0] = (float)(s[5] / ne); o[1] = (float)(s[6] / ne); o[2] = (float)(s[7] / ne); o[3] = (float)sqrt(s[8] / ne);
  o[4] = (float)mp; o[5] = (float)sqrt(fmax(vp, 0.0)); o[6] = (float)vp;
  o[7] = (float)mg; o[8] = (float)sqrt(fmax(vg, 0.0)); o[9] = (float)vg;
  o[10] = (float)(vp / (vg == 0.0 ? 1e-6 : vg));
  o[11] = (float)n;
}

__global__ __launch_bounds__(256) void sm_partial(const mvp_snorm_metrics_args p, double* part) {
  __shared__ double red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int64_t per = (p.HW + MT_NCH - 1) / MT_NCH, i0 = ch * per, i1 = min(p.HW, i0 + per);
  double s[5] = {0, 0, 0, 0, 0};
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float* pr = p.pred + ((int64_t)b * p.Cp) * p.HW + i;
    const float* gt = p.gt + ((int64_t)b * 3) * p.HW + i;
    const float g0 = gt[0], g1 = gt[p.HW], g2 = gt[2 * p.HW];
    if (!(fabsf(g0) + fabsf(g1) + fabsf(g2) > 0.f)) continue;
    const float p0 = pr[0], p1 = pr[p.HW], p2 = pr[2 * p.HW];
    const float np = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-8f), ng = fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-8f);
    const float c = fminf(fmaxf((p0 / np) * (g0 / ng) + (p1 / np) * (g1 / ng) + (p2 / np) * (g2 / ng), -1.f), 1.f);
    const float e = acosf(c) * 57.29577951308232f;
    s[0] += 1.0; s[1] += (double)e * e; s[2] += e < p.t1 ? 1.0 : 0.0; s[3] += e < p.t2 ? 1.0 : 0.0; s[4] += e < p.t3 ? 1.0 : 0.0;
  }
  for (int k = 0; k < 5; ++k) {
    const double v = blk_sum(s[k], red);
    if (threadIdx.x == 0) part[((int64_t)b * MT_NCH + ch) * 5 + k] = v;
  }
}

__global__ void sm_final(const mvp_snorm_metrics_args p, const double* part) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  double s[5] = {0, 0, 0, 0, 0};
  for (int ch = 0; ch < MT_NCH; ++ch)
    for (int k = 0; k < 5; ++k) s[k] += part[((int64_t)b * MT_NCH + ch) * 5 + k];
  const double n = s[0] < 1.0 ? 1.0 : s[0];
  float* o = p.out + (int64_t)b * 5;
  o[0] = (float)(s[2] / n); o[1] = (float)(s[3] / n); o[2] = (float)(s[4] / n); o[3] = (float)sqrt(s[1] / n); o[4] = (float)s[0];
}

}  // namespace

extern "C" int64_t mvp_metrics_workspace_bytes(int B) { return (int64_t)B * MT_NCH * 9 * 8 + 64; }

extern "C" int mvp_depth_metrics(const mvp_depth_metrics_args* a, void* stream) {
  if (!a || !a->pred || !a->gt || !a->out || !a->workspace || a->B <= 0 || a->HW <= 0) return MVP_EINVAL;
  if (a->workspace_bytes < mvp_metrics_workspace_bytes(a->B) || (a->scale_invariant && !a->scale_shift)) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)a->workspace;
  if (a->scale_invariant) {
    hipLaunchKernelGGL(dm_fit_partial, dim3(MT_NCH, a->B), dim3(256), 0, s, *a, part);
    hipLaunchKernelGGL(dm_fit_final, dim3((a->B + 63) / 64), dim3(64), 0, s, *a, part);
  }
  hipLaunchKernelGGL(dm_partial, dim3(MT_NCH, a->B), dim3(256), 0, s, *a, part);
  hipLaunchKernelGGL(dm_final, dim3((a->B + 63) / 64), dim3(64), 0, s, *a, part);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_snorm_metrics(const mvp_snorm_metrics_args* a, void* stream) {
  if (!a || !a->pred || !a->gt || !a->out || !a->workspace || a->B <= 0 || a->HW <= 0 || a->Cp < 3) return MVP_EINVAL;
  if (a->workspace_bytes < mvp_metrics_workspace_bytes(a->B)) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)a->workspace;
  hipLaunchKernelGGL(sm_partial, dim3(MT_NCH, a->B), dim3(256), 0, s, *a, part);
  hipLaunchKernelGGL(sm_final, dim3((a->B + 63) / 64), dim3(64), 0, s, *a, part);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
