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

// Per-image affine map with optional clamp, forward and adjoint (match_scale_and_shift's last line, metrics.py:775-777, and the
// clamp of the scale-invariant training branch, train_depth.py:116-118).  scale / shift carry no gradient (detached in the
// reference), so the adjoint is grad * scale, gated where the clamp is inactive (torch.clamp passes gradients on lo <= y <= hi).
__global__ __launch_bounds__(256) void ss_kernel(const mvp_scale_shift_args p, const int backward) {
  const int64_t total = (int64_t)p.B * p.HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / p.HW);
    const float sc = p.scale_shift[2 * b], sh = p.scale_shift[2 * b + 1];
    const float y = p.x[i] * sc + sh;
    if (!backward) p.out[i] = p.clamp ? fminf(fmaxf(y, p.lo), p.hi) : y;
    else p.out[i] = (!p.clamp || (y >= p.lo && y <= p.hi)) ? p.grad_out[i] * sc : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Per-level / per-segment breakdown (evaluate_depth metrics.py:179-358, evaluate_surface_norm metrics.py:441-577) as ONE
// segmented masked reduction per batch: every pixel is binned once by its centroid level and once by its segment id.
//   level bins  [B, L, 5]   = {n_valid, d1, d2, d3, sum err^2}
//   segment bins [B, S, 6]  = {n_all, n_valid, d1, d2, d3, sum err^2}      (n_all: torch.unique sees invalid pixels too)
// Workgroup-local bins live in LDS (ds_add_f64), chunk partials are combined in a fixed order.  The reference's
// per-segment Python loop (metrics.py:325-355: one full-image masked pass per id and per metric) becomes one read of
// pred / gt / seg.  stuff / things groups and the 1e-6 / clamp(1) normalisations are finished on the host from the bins.
constexpr int MB_NCH = 32;
constexpr int MB_MAX_LEVELS = 16;

template <bool SNORM>
__global__ __launch_bounds__(256) void mb_partial(const mvp_metrics_breakdown_args p, double* part) {
  extern __shared__ double bins[];  // [L*5 + S*6]
  const int b = blockIdx.y, ch = blockIdx.x;
  const int L = p.num_levels, S = p.seg ? p.num_ids : 0, nb = L * 5 + S * 6;
  for (int i = threadIdx.x; i < nb; i += 256) bins[i] = 0.0;
  __shared__ int offs[MB_MAX_LEVELS];
  if (threadIdx.x < L) offs[threadIdx.x] = (p.H / L) * (L - (threadIdx.x + 1)) / 2;
  __syncthreads();
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t per = (HW + MB_NCH - 1) / MB_NCH, i0 = ch * per, i1 = min(HW, i0 + per);
  const float sc = (!SNORM && p.scale_shift) ? p.scale_shift[2 * b] : 1.f, sh = (!SNORM && p.scale_shift) ? p.scale_shift[2 * b + 1] : 0.f;
  const float t1 = SNORM ? p.t1 : 1.25f, t2 = SNORM ? p.t2 : 1.25f * 1.25f, t3 = SNORM ? p.t3 : 1.25f * 1.25f * 1.25f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    bool valid;
    float x = 0.f, se = 0.f;  // x: the quantity compared with the thresholds; se: squared error
    if (SNORM) {
      const float* gt = p.gt + ((int64_t)b * 3) * HW + i;
      const float g0 = gt[0], g1 = gt[HW], g2 = gt[2 * HW];
      valid = fabsf(g0) + fabsf(g1) + fabsf(g2) > 0.f;
      if (valid) {
        const float* pr = p.pred + ((int64_t)b * p.Cp) * HW + i;
        const float p0 = pr[0], p1 = pr[HW], p2 = pr[2 * HW];
        const float np = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-8f), ng = fmaxf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2), 1e-8f);
        const float c = fminf(fmaxf((p0 / np) * (g0 / ng) + (p1 / np) * (g1 / ng) + (p2 / np) * (g2 / ng), -1.f), 1.f);
        x = acosf(c) * 57.29577951308232f;
        se = x * x;
      }
    } else {
      const float g = p.gt[(int64_t)b * HW + i];
      valid = g > 0.f;
      if (valid) {
        float q = p.pred[(int64_t)b * HW + i];
        if (p.scale_shift) q = q * sc + sh;
        x = fmaxf(g / fmaxf(q, 1e-9f), q / fmaxf(g, 1e-9f));
        se = (g - q) * (g - q);
      }
    }
    const double h1 = x < t1 ? 1.0 : 0.0, h2 = x < t2 ? 1.0 : 0.0, h3 = x < t3 ? 1.0 : 0.0;
    if (valid) {
      const int y = (int)(i / p.W), xx = (int)(i - (int64_t)y * p.W);
      int lev = L - 1;  // the last level has offset 0: it covers the whole image
      for (int l = 0; l < L; ++l) {
        const int o = offs[l];
        if (y >= o && y < p.H - o && xx >= o && xx < p.W - o) { lev = l; break; }
      }
      double* lb = bins + lev * 5;
      atomicAdd(lb, 1.0); atomicAdd(lb + 1, h1); atomicAdd(lb + 2, h2); atomicAdd(lb + 3, h3); atomicAdd(lb + 4, (double)se);
    }
    if (S) {
      const int id = p.seg[(int64_t)b * HW + i];
      if (id >= 0 && id < S) {
        double* sb = bins + L * 5 + id * 6;
        atomicAdd(sb, 1.0);
        if (valid) { atomicAdd(sb + 1, 1.0); atomicAdd(sb + 2, h1); atomicAdd(sb + 3, h2); atomicAdd(sb + 4, h3); atomicAdd(sb + 5, (double)se); }
      }
    }
  }
  __syncthreads();
  double* out = part + ((int64_t)b * MB_NCH + ch) * nb;
  for (int i = threadIdx.x; i < nb; i += 256) out[i] = bins[i];
}

__global__ __launch_bounds__(256) void mb_final(const mvp_metrics_breakdown_args p, const double* part) {
  const int b = blockIdx.y;
  const int L = p.num_levels, S = p.seg ? p.num_ids : 0, nb = L * 5 + S * 6;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nb) return;
  double s = 0.0;
  for (int ch = 0; ch < MB_NCH; ++ch) s += part[((int64_t)b * MB_NCH + ch) * nb + i];
  if (i < L * 5) p.level_sums[(int64_t)b * L * 5 + i] = s;
  else p.seg_sums[(int64_t)b * S * 6 + (i - L * 5)] = s;
}

}  // namespace

extern "C" int mvp_scale_shift(const mvp_scale_shift_args* a, void* stream) {
  if (!a || !a->x || !a->scale_shift || !a->out || a->B <= 0 || a->HW <= 0 || (a->backward && !a->grad_out)) return MVP_EINVAL;
  int64_t g = ((int64_t)a->B * a->HW + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(ss_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *a, a->backward);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int64_t mvp_metrics_breakdown_workspace_bytes(int B, int num_levels, int num_ids) {
  if (B <= 0 || num_levels <= 0 || num_ids < 0) return 0;
  return (int64_t)B * MB_NCH * (num_levels * 5 + num_ids * 6) * 8 + 64;
}

extern "C" int mvp_metrics_breakdown(const mvp_metrics_breakdown_args* a, void* stream) {
  if (!a || !a->pred || !a->gt || !a->level_sums || !a->workspace || a->B <= 0 || a->H <= 0 || a->W <= 0) return MVP_EINVAL;
  if (a->num_levels <= 0 || a->num_levels > MB_MAX_LEVELS || (a->Cp != 0 && a->Cp < 3)) return MVP_EINVAL;
  if (a->seg && (!a->seg_sums || a->num_ids <= 0 || a->num_ids > 2048)) return MVP_EINVAL;
  const int S = a->seg ? a->num_ids : 0, nb = a->num_levels * 5 + S * 6;
  if (a->workspace_bytes < mvp_metrics_breakdown_workspace_bytes(a->B, a->num_levels, S) || ((uintptr_t)a->workspace & 7)) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)a->workspace;
  const size_t lds = (size_t)nb * sizeof(double);
  if (a->Cp) hipLaunchKernelGGL(mb_partial<true>, dim3(MB_NCH, a->B), dim3(256), lds, s, *a, part);
  else hipLaunchKernelGGL(mb_partial<false>, dim3(MB_NCH, a->B), dim3(256), lds, s, *a, part);
  hipLaunchKernelGGL(mb_final, dim3((nb + 255) / 256, a->B), dim3(256), 0, s, *a, part);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

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
