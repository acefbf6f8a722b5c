// SPair-71k correspondence core (evaluate_spair_correspondence.py:59-83 + argmax_2d,
// evals/utils/correspondence.py:179-190): L2-normalise both feature maps over C, bilinearly
// sample the normalised source map at K keypoints (grid_sample, align_corners=True, zero
// padding), heat-map = <descriptor, normalised target pixel>, 2-D argmax (first maximum wins,
// as torch.argmax), returned as (col, row).
#include "mvp_common.h"

namespace {

// inverse L2 norms over C for both maps: inv[0..hw) source, inv[hw..2hw) target
__global__ __launch_bounds__(256) void corr_norm_kernel(const mvp_corr_argmax_args p, float* inv) {
  const int hw = p.h * p.w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * hw) return;
  const float* f = (i < hw) ? p.src_feat + i : p.tgt_feat + (i - hw);
  float s = 0.f;
  for (int c = 0; c < p.C; ++c) { const float v = f[(size_t)c * hw]; s += v * v; }
  inv[i] = 1.0f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize eps
}

// one block per keypoint
__global__ __launch_bounds__(256) void corr_kernel(const mvp_corr_argmax_args p, const float* inv, float* desc_all) {
  __shared__ float best_v[256];
  __shared__ int best_i[256];
  const int k = blockIdx.x, hw = p.h * p.w;
  float* desc = desc_all + (size_t)k * p.C;
  // grid_sample bilinear, align_corners=True, padding zeros
  const float gx = p.kp_xy[k * 2], gy = p.kp_xy[k * 2 + 1];
  const float fx = (gx + 1.f) * 0.5f * (float)(p.w - 1), fy = (gy + 1.f) * 0.5f * (float)(p.h - 1);
  const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  const float tx = fx - (float)x0, ty = fy - (float)y0;
  const int xs[2] = {x0, x0 + 1}, ys[2] = {y0, y0 + 1};
  const float wx[2] = {1.f - tx, tx}, wy[2] = {1.f - ty, ty};
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float acc = 0.f;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        if (ys[a] < 0 || ys[a] >= p.h || xs[b] < 0 || xs[b] >= p.w) continue;
        const int pix = ys[a] * p.w + xs[b];
        acc += wy[a] * wx[b] * p.src_feat[(size_t)c * hw + pix] * inv[pix];
      }
    desc[c] = acc;
  }
  __syncthreads();
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int pix = threadIdx.x; pix < hw; pix += 256) {
    float acc = 0.f;
    for (int c = 0; c < p.C; ++c) acc += desc[c] * p.tgt_feat[(size_t)c * hw + pix];
    acc *= inv[hw + pix];
    if (acc > bv) { bv = acc; bi = pix; }  // strided visit order is increasing: first max kept
  }
  best_v[threadIdx.x] = bv;
  best_i[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float ov = best_v[threadIdx.x + s];
      const int oi = best_i[threadIdx.x + s];
      if (ov > best_v[threadIdx.x] || (ov == best_v[threadIdx.x] && oi < best_i[threadIdx.x])) {
        best_v[threadIdx.x] = ov;
        best_i[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int idx = best_i[0];
    p.out_xy[k * 2] = idx % p.w;      // col
    p.out_xy[k * 2 + 1] = idx / p.w;  // row
    if (p.out_val) p.out_val[k] = best_v[0];
  }
}

}  // namespace

extern "C" int mvp_corr_argmax(const mvp_corr_argmax_args* a, void* stream) {
  if (!a || !a->src_feat || !a->tgt_feat || !a->kp_xy || !a->out_xy || !a->workspace) return MVP_EINVAL;
  if (a->C <= 0 || a->h <= 0 || a->w <= 0 || a->K <= 0) return MVP_EINVAL;
  const int64_t need = ((int64_t)2 * a->h * a->w + (int64_t)a->K * a->C) * 4;
  if (a->workspace_bytes < need) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* inv = a->workspace;
  float* desc = inv + 2 * a->h * a->w;
  hipLaunchKernelGGL(corr_norm_kernel, dim3((2 * a->h * a->w + 255) / 256), dim3(256), 0, s, *a, inv);
  hipLaunchKernelGGL(corr_kernel, dim3(a->K), dim3(256), 0, s, *a, inv, desc);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
