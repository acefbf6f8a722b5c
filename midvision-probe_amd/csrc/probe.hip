// Probe-side pointwise kernels: depth predictors (bins / sigmoid) fwd+bwd, column sums.
#include "mvp_common.h"

namespace {

__device__ __forceinline__ float bin_value(int k, int K, float lo, float hi) {
  // torch.linspace: start + step*i for the lower half, end - step*(K-1-i) for the upper half
  const float step = (hi - lo) / (float)(K - 1);
  return (k < K / 2) ? lo + step * (float)k : hi - step * (float)(K - 1 - k);
}

// One wave per pixel; K logits contiguous (token-major), float4 per lane per iteration.
__global__ __launch_bounds__(256) void depth_bins_fwd(const mvp_depth_predict_args p) {
  const int lane = threadIdx.x & 63;
  const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (px >= p.P) return;
  const float* l = p.logits + px * p.K;
  float s = 0.f, d = 0.f;
  for (int k = lane * 4; k < p.K; k += 256) {
    const float4 v = *(const float4*)(l + k);
    const float a = fmaxf(v.x, 0.f) + 0.1f, b = fmaxf(v.y, 0.f) + 0.1f, c = fmaxf(v.z, 0.f) + 0.1f, e = fmaxf(v.w, 0.f) + 0.1f;
    s += (a + b) + (c + e);
    d += a * bin_value(k, p.K, p.min_depth, p.max_depth) + b * bin_value(k + 1, p.K, p.min_depth, p.max_depth) +
         c * bin_value(k + 2, p.K, p.min_depth, p.max_depth) + e * bin_value(k + 3, p.K, p.min_depth, p.max_depth);
  }
  s = wave_sum(s);
  d = wave_sum(d);
  if (lane == 0) {
    const float inv = 1.0f / s;
    p.depth[px] = d * inv;
    if (p.inv_sum) p.inv_sum[px] = inv;
  }
}

// d depth / d l_k = [l_k > 0] * (bin_k - depth) / sum
__global__ __launch_bounds__(256) void depth_bins_bwd(const mvp_depth_predict_args p) {
  const int lane = threadIdx.x & 63;
  const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (px >= p.P) return;
  const float* l = p.logits + px * p.K;
  float* g = p.grad_logits + px * p.K;
  const float depth = p.depth[px];
  const float gs = p.grad_depth[px] * p.inv_sum[px];
  for (int k = lane * 4; k < p.K; k += 256) {
    const float4 v = *(const float4*)(l + k);
    float4 o;
    o.x = v.x > 0.f ? gs * (bin_value(k, p.K, p.min_depth, p.max_depth) - depth) : 0.f;
    o.y = v.y > 0.f ? gs * (bin_value(k + 1, p.K, p.min_depth, p.max_depth) - depth) : 0.f;
    o.z = v.z > 0.f ? gs * (bin_value(k + 2, p.K, p.min_depth, p.max_depth) - depth) : 0.f;
    o.w = v.w > 0.f ? gs * (bin_value(k + 3, p.K, p.min_depth, p.max_depth) - depth) : 0.f;
    *(float4*)(g + k) = o;
  }
}

__global__ __launch_bounds__(256) void depth_sigmoid_fwd(const mvp_depth_predict_args p) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.P; i += (int64_t)gridDim.x * 256) {
    const float s = 1.0f / (1.0f + expf(-p.logits[i]));
    p.depth[i] = p.min_depth + s * (p.max_depth - p.min_depth);
  }
}

__global__ __launch_bounds__(256) void depth_sigmoid_bwd(const mvp_depth_predict_args p) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.P; i += (int64_t)gridDim.x * 256) {
    const float s = 1.0f / (1.0f + expf(-p.logits[i]));
    p.grad_logits[i] = p.grad_depth[i] * (p.max_depth - p.min_depth) * s * (1.0f - s);
  }
}

// Column sums, deterministic two-level reduction: level 1 = (64-column strip) x (row chunk)
// workgroups write partial sums into the workspace, level 2 adds the chunks in a fixed order.
constexpr int CS_CHUNK = 64;  // minimum rows per level-1 workgroup

// rows per level-1 chunk: 64 at ViT sizes, growing with M so that level 2 folds at most 256 chunks per column (the DPT
// bias gradients have M = 131k..524k rows: 2048+ chunks made the second level a 30 us serial sum)
__host__ __device__ inline int cs_chunk_rows(int M) {
  const int r = (M + 255) / 256;
  return r < CS_CHUNK ? CS_CHUNK : ((r + 3) & ~3);
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const mvp_colsum_args p, float* part, int chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int r0 = blockIdx.y * chunk, r1 = min(p.M, r0 + chunk);
  float s = 0.f;
  if (c < p.N) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // 4 loads in flight; fixed combination order
    int r = r0 + rg;
    for (; r + 12 < r1; r += 16) {
      s0 += p.x[(size_t)r * p.ld + c];
      s1 += p.x[(size_t)(r + 4) * p.ld + c];
      s2 += p.x[(size_t)(r + 8) * p.ld + c];
      s3 += p.x[(size_t)(r + 12) * p.ld + c];
    }
    for (; r < r1; r += 4) s0 += p.x[(size_t)r * p.ld + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < p.N)
    part[(size_t)blockIdx.y * p.N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const mvp_colsum_args p, const float* part, int nchunk) {
  // 64 columns x 4 chunk groups per workgroup: the chunk loads of a thread are independent (4 accumulators), the
  // combination order is fixed -> deterministic.
  __shared__ float red[4][64];
  const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < p.N) {
    int k = rg;
    for (; k + 12 < nchunk; k += 16) {
      s0 += part[(size_t)k * p.N + c];
      s1 += part[(size_t)(k + 4) * p.N + c];
      s2 += part[(size_t)(k + 8) * p.N + c];
      s3 += part[(size_t)(k + 12) * p.N + c];
    }
    for (; k < nchunk; k += 4) s0 += part[(size_t)k * p.N + c];
  }
  red[rg][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < p.N) {
    const float s = (red[0][lc] + red[1][lc]) + (red[2][lc] + red[3][lc]);
    p.out[c] = p.accumulate ? p.out[c] + s : s;
  }
}

inline int grid_for(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int mvp_depth_predict_fwd(const mvp_depth_predict_args* a, void* stream) {
  if (!a || !a->logits || !a->depth || a->P <= 0 || a->K <= 0) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->kind == 0) {
    if ((a->K & 3) || a->K < 2) return MVP_EINVAL;
    hipLaunchKernelGGL(depth_bins_fwd, dim3((unsigned)((a->P + 3) / 4)), dim3(256), 0, s, *a);
  } else {
    if (a->K != 1) return MVP_EINVAL;
    hipLaunchKernelGGL(depth_sigmoid_fwd, dim3(grid_for(a->P)), dim3(256), 0, s, *a);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_depth_predict_bwd(const mvp_depth_predict_args* a, void* stream) {
  if (!a || !a->logits || !a->grad_depth || !a->grad_logits || a->P <= 0 || a->K <= 0) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->kind == 0) {
    if ((a->K & 3) || !a->depth || !a->inv_sum) return MVP_EINVAL;
    hipLaunchKernelGGL(depth_bins_bwd, dim3((unsigned)((a->P + 3) / 4)), dim3(256), 0, s, *a);
  } else {
    if (a->K != 1) return MVP_EINVAL;
    hipLaunchKernelGGL(depth_sigmoid_bwd, dim3(grid_for(a->P)), dim3(256), 0, s, *a);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int64_t mvp_colsum_workspace_bytes(int M, int N) {
  const int chunk = cs_chunk_rows(M);
  return (int64_t)((M + chunk - 1) / chunk) * N * 4;
}

extern "C" int mvp_colsum(const mvp_colsum_args* a, void* stream) {
  if (!a || !a->x || !a->out || !a->workspace || a->M <= 0 || a->N <= 0 || a->ld < a->N) return MVP_EINVAL;
  if (a->workspace_bytes < mvp_colsum_workspace_bytes(a->M, a->N)) return MVP_EINVAL;
  const int chunk = cs_chunk_rows(a->M);
  const int nchunk = (a->M + chunk - 1) / chunk;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((a->N + 63) / 64, nchunk), dim3(256), 0, s, *a, (float*)a->workspace, chunk);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((a->N + 63) / 64), dim3(256), 0, s, *a, (const float*)a->workspace, nchunk);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

// ----------------------------------------------------------------------------- mask + split
// g[M,N] fp32 (optionally gated by a byte mask) -> fp32 copy (may alias) + bf16 pair with row
// stride ldo (>= N; pad columns are written as zeros so the buffer can feed the TN kernel).
namespace {
__global__ __launch_bounds__(256) void mask_split_kernel(const mvp_mask_split_args p) {
  const int64_t total = (int64_t)p.M * p.ldo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % p.ldo);
    const int64_t r = i / p.ldo;
    float v = 0.f;
    if (c < p.N) {
      v = p.src[r * p.lds + c];
      if (p.mask && !p.mask[r * p.ldm + c]) v = 0.f;
      if (p.relu_mask_out) { const bool on = v > 0.f; v = on ? v : 0.f; p.relu_mask_out[r * p.ldm + c] = on; }
      if (p.dst_f32) p.dst_f32[r * p.lds + c] = v;
    }
    uint16_t h, l;
    split_bf16(v, h, l);
    if (p.dst_hi) p.dst_hi[i] = h;
    if (p.dst_lo) p.dst_lo[i] = l;
  }
}

// 4 columns per thread (N, ldo, lds, ldm all multiples of 4): float4 / 4-byte mask loads, 16-byte and 8-byte stores.
__global__ __launch_bounds__(256) void mask_split_vec4_kernel(const mvp_mask_split_args p) {
  const int q = p.ldo >> 2;
  const int64_t total = (int64_t)p.M * q;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) << 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < p.N) {
      v = *(const float4*)(p.src + r * p.lds + c);
      if (p.mask) {
        const uint32_t m = *(const uint32_t*)(p.mask + r * p.ldm + c);
        if (!(m & 0xffu)) v.x = 0.f;
        if (!(m & 0xff00u)) v.y = 0.f;
        if (!(m & 0xff0000u)) v.z = 0.f;
        if (!(m & 0xff000000u)) v.w = 0.f;
      }
      if (p.relu_mask_out) {
        const uint32_t m = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 0x100u : 0u) | (v.z > 0.f ? 0x10000u : 0u) | (v.w > 0.f ? 0x1000000u : 0u);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *(uint32_t*)(p.relu_mask_out + r * p.ldm + c) = m;
      }
      if (p.dst_f32) *(float4*)(p.dst_f32 + r * p.lds + c) = v;
    }
    uint32_t h01, l01, h23, l23;
    split2_bf16(v.x, v.y, h01, l01);
    split2_bf16(v.z, v.w, h23, l23);
    const int64_t o = r * p.ldo + c;
    if (p.dst_hi) *(u32x2_t*)(p.dst_hi + o) = u32x2_t{h01, h23};
    if (p.dst_lo) *(u32x2_t*)(p.dst_lo + o) = u32x2_t{l01, l23};
  }
}
}  // namespace

extern "C" int mvp_mask_split(const mvp_mask_split_args* a, void* stream) {
  if (!a || !a->src || a->M <= 0 || a->N <= 0 || a->ldo < a->N || a->lds < a->N) return MVP_EINVAL;
  const bool vec = !((a->N | a->ldo | a->lds) & 3) && ((!a->mask && !a->relu_mask_out) || !(a->ldm & 3)) &&
                   !(((uintptr_t)a->src | (uintptr_t)a->dst_f32) & 15) && !(((uintptr_t)a->dst_hi | (uintptr_t)a->dst_lo) & 7) &&
                   !(((uintptr_t)a->mask | (uintptr_t)a->relu_mask_out) & 3);
  int64_t g = ((int64_t)a->M * (vec ? a->ldo >> 2 : a->ldo) + 255) / 256;
  if (g > 16384) g = 16384;
  if (vec)
    hipLaunchKernelGGL(mask_split_vec4_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(mask_split_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
