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
  int c = 0;
  for (; c + 8 <= p.C; c += 8) {  // 8 channel loads in flight; summation order unchanged
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = f[(size_t)(c + u) * hw];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u] * v[u];
  }
  for (; c < p.C; ++c) { const float v = f[(size_t)c * hw]; s += v * v; }
  inv[i] = 1.0f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize eps
}

// descriptors: one block per keypoint, desc[k][c] = bilinear sample of the normalised source map; also resets the
// keypoint's running (value, index) key.
__global__ __launch_bounds__(256) void corr_desc_kernel(const mvp_corr_argmax_args p, const float* inv, float* desc_all, unsigned long long* keys) {
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
  if (threadIdx.x == 0) keys[k] = 0ull;
}

// (value, index) packed so that an unsigned 64-bit max = "largest value, smallest index among ties" (torch.argmax keeps
// the first maximum); max is order-independent, so the atomic keeps the result deterministic.
__device__ __forceinline__ unsigned long long corr_key(float v, int idx) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  const unsigned ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx);
}

// heat map + argmax: grid (pixel chunks of 256, K).  Lanes run along the pixels (coalesced reads of the [C, hw] target map),
// the descriptor sits in LDS, 8 channel loads in flight per thread.  (One block per keypoint with a serial channel loop
// took 1.78 ms at 50x50x768; this form runs on K * hw / 256 workgroups.)
__global__ __launch_bounds__(256) void corr_heat_kernel(const mvp_corr_argmax_args p, const float* inv, const float* desc_all, unsigned long long* keys) {
  extern __shared__ float desc[];
  __shared__ unsigned long long best[4];
  const int k = blockIdx.y, hw = p.h * p.w;
  for (int c = threadIdx.x; c < p.C; c += 256) desc[c] = desc_all[(size_t)k * p.C + c];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;
  unsigned long long key = 0ull;
  if (pix < hw) {
    const float* t = p.tgt_feat + pix;
    float acc = 0.f;  // same channel order as a serial loop: bit-identical to the previous kernel
    int c = 0;
    for (; c + 8 <= p.C; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = t[(size_t)(c + u) * hw];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += desc[c + u] * v[u];
    }
    for (; c < p.C; ++c) acc += desc[c] * t[(size_t)c * hw];
    acc *= inv[hw + pix];
    if (p.heat_out) p.heat_out[(size_t)k * hw + pix] = acc;
    key = corr_key(acc, pix);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(key >> 32), o, 64) << 32) | (unsigned long long)__shfl_xor((unsigned)key, o, 64);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long m = best[0];
    for (int i = 1; i < 4; ++i) m = best[i] > m ? best[i] : m;
    atomicMax(keys + k, m);
  }
}

__global__ void corr_final_kernel(const mvp_corr_argmax_args p, const unsigned long long* keys) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.K) return;
  const unsigned long long key = keys[k];
  const int idx = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
  const unsigned ord = (unsigned)(key >> 32);
  const unsigned b = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
  p.out_xy[k * 2] = idx % p.w;      // col
  p.out_xy[k * 2 + 1] = idx / p.w;  // row
  if (p.out_val) p.out_val[k] = __builtin_bit_cast(float, b);
}

// argmax_2d (correspondence.py:179-190) on materialised maps x [K, h, w]: one workgroup per map, (value, index) keys so that
// ties resolve to the lowest flat index exactly as torch.argmax / torch.argmin do; max_value == 0 orders by -value.
__global__ __launch_bounds__(256) void argmax2d_kernel(const mvp_argmax_2d_args p) {
  __shared__ unsigned long long best[4];
  const int k = blockIdx.x, hw = p.h * p.w;
  const float* x = p.x + (size_t)k * hw;
  unsigned long long key = 0ull;
  for (int i = threadIdx.x; i < hw; i += 256) {
    const float v = x[i];
    const unsigned long long c = corr_key(p.max_value ? v : -v, i);
    key = c > key ? c : key;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(key >> 32), o, 64) << 32) | (unsigned long long)__shfl_xor((unsigned)key, o, 64);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long m = best[0];
    for (int i = 1; i < 4; ++i) m = best[i] > m ? best[i] : m;
    const int idx = (int)(0xffffffffu - (unsigned)(m & 0xffffffffull));
    p.out_xy[k * 2] = idx % p.w;
    p.out_xy[k * 2 + 1] = idx / p.w;
  }
}

}  // namespace

extern "C" int mvp_argmax_2d(const mvp_argmax_2d_args* a, void* stream) {
  if (!a || !a->x || !a->out_xy || a->K <= 0 || a->h <= 0 || a->w <= 0 || (int64_t)a->h * a->w > 0x7fffffff) return MVP_EINVAL;
  hipLaunchKernelGGL(argmax2d_kernel, dim3(a->K), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int64_t mvp_corr_workspace_bytes(int C, int h, int w, int K) {
  if (C <= 0 || h <= 0 || w <= 0 || K <= 0) return 0;
  // inverse norms (2hw) + descriptors (K*C), then K 64-bit keys on an 8-byte boundary
  return ((((int64_t)2 * h * w + (int64_t)K * C) * 4 + 7) & ~(int64_t)7) + (int64_t)K * 8;
}

extern "C" int mvp_corr_argmax(const mvp_corr_argmax_args* a, void* stream) {
  if (!a || !a->src_feat || !a->tgt_feat || !a->kp_xy || !a->out_xy || !a->workspace) return MVP_EINVAL;
  if (a->C <= 0 || a->h <= 0 || a->w <= 0 || a->K <= 0 || a->C > 16384) return MVP_EINVAL;
  if (a->workspace_bytes < mvp_corr_workspace_bytes(a->C, a->h, a->w, a->K) || ((uintptr_t)a->workspace & 7)) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int hw = a->h * a->w;
  float* inv = a->workspace;
  float* desc = inv + 2 * hw;
  unsigned long long* keys = (unsigned long long*)((char*)a->workspace + ((((int64_t)2 * hw + (int64_t)a->K * a->C) * 4 + 7) & ~(int64_t)7));
  hipLaunchKernelGGL(corr_norm_kernel, dim3((2 * hw + 255) / 256), dim3(256), 0, s, *a, inv);
  hipLaunchKernelGGL(corr_desc_kernel, dim3(a->K), dim3(256), 0, s, *a, inv, desc, keys);
  hipLaunchKernelGGL(corr_heat_kernel, dim3((hw + 255) / 256, a->K), dim3(256), a->C * sizeof(float), s, *a, inv, desc, keys);
  hipLaunchKernelGGL(corr_final_kernel, dim3((a->K + 63) / 64), dim3(64), 0, s, *a, keys);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
