// Nearest / bilinear / bicubic resampling, forward and adjoint (gather form: deterministic,
// no atomics).  Index arithmetic follows ATen (area_pixel_compute_source_index,
// nearest_neighbor_compute_source_index, cubic A = -0.75 with clamped taps), which is what
// F.interpolate runs in the reference (train_depth.py:114, train_snorm.py:110,
// probes.py:255-258,388,396-398,431).
//
// Layouts: planar [planes, H, W] (lanes along W), or channels_last [B, H, W, C] (lanes along
// C, float4) for the token-major probe logits.
#include "mvp_common.h"

namespace {

struct Taps {
  int idx[4];
  float w[4];
  int n;
};

__device__ __forceinline__ float src_scale(int in, int out, int align, float sf) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return sf > 0.f ? 1.0f / sf : (float)in / (float)out;
}

__device__ __forceinline__ Taps taps_1d(int mode, int align, float scale, int o, int in) {
  Taps t;
  if (mode == MVP_RESIZE_NEAREST) {
    t.n = 1;
    t.idx[0] = min((int)floorf((float)o * scale), in - 1);
    t.w[0] = 1.f;
  } else if (mode == MVP_RESIZE_BILINEAR) {
    float s = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    int i0 = min((int)s, in - 1);
    int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    float l1 = s - (float)i0;
    t.n = 2;
    t.idx[0] = i0; t.idx[1] = i1;
    t.w[0] = 1.f - l1; t.w[1] = l1;
  } else {
    float s = align ? scale * (float)o : scale * ((float)o + 0.5f) - 0.5f;
    float fl = floorf(s);
    float x = s - fl;
    int i = (int)fl;
    const float A = -0.75f;
    float x0 = x + 1.f, x1 = x, x2 = 1.f - x, x3 = 2.f - x;
    t.n = 4;
    t.w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    t.w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    t.w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    t.w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.idx[k] = max(0, min(i - 1 + k, in - 1));
  }
  return t;
}

// Candidate output range [lo, hi] whose taps may touch input index i (a superset; the exact
// membership test is done per candidate with taps_1d).
__device__ __forceinline__ void cand_range(int mode, int align, float scale, int i, int in, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float r = (mode == MVP_RESIZE_BICUBIC) ? 2.f : 1.f;
  const float off = (align || mode == MVP_RESIZE_NEAREST) ? 0.f : 0.5f;
  float flo = ((float)i - r - 1.f + off) / scale - off;
  float fhi = ((float)i + r + 1.f + off) / scale - off;
  lo = max(0, (int)floorf(flo) - 1);
  hi = min(out - 1, (int)ceilf(fhi) + 1);
  if (i == 0) lo = 0;
  if (i == in - 1) hi = out - 1;
}

__global__ __launch_bounds__(256) void resize_fwd_planar(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int64_t total = (int64_t)p.planes * p.Ho * p.Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % p.Wo);
    const int64_t r = i / p.Wo;
    const int oy = (int)(r % p.Ho);
    const int64_t pl = r / p.Ho;
    const Taps ty = taps_1d(p.mode, p.align_corners, sh, oy, p.Hi);
    const Taps tx = taps_1d(p.mode, p.align_corners, sw, ox, p.Wi);
    const float* s = p.src + pl * p.Hi * p.Wi;
    float acc = 0.f;
    for (int a = 0; a < ty.n; ++a) {
      float rowacc = 0.f;
      for (int b = 0; b < tx.n; ++b) rowacc += tx.w[b] * s[(size_t)ty.idx[a] * p.Wi + tx.idx[b]];
      acc += ty.w[a] * rowacc;
    }
    p.dst[i] = acc;
  }
}

// Adjoint weight of output index o on input index i along one axis (0 when o does not touch i).
__device__ __forceinline__ float adj_w(int mode, int align, float scale, int o, int in, int i) {
  const Taps t = taps_1d(mode, align, scale, o, in);
  float w = 0.f;
  for (int a = 0; a < t.n; ++a) w += (t.idx[a] == i) ? t.w[a] : 0.f;
  return w;
}

constexpr int ADJ_MAXW = 24;  // candidate columns whose weights are cached in registers

__global__ __launch_bounds__(256) void resize_bwd_planar(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  // 8 lanes cooperate on one input element (candidate rows strided by 8, xor-shuffle reduce): the
  // planar adjoint runs on few, small planes (B x 56 x 56 depth maps), i.e. it is latency-bound.
  const int64_t total = (int64_t)p.planes * p.Hi * p.Wi;
  const int sub = threadIdx.x & 7;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; i < total; i += ((int64_t)gridDim.x * 256) >> 3) {
    const int ix = (int)(i % p.Wi);
    const int64_t r = i / p.Wi;
    const int iy = (int)(r % p.Hi);
    const int64_t pl = r / p.Hi;
    int ylo, yhi, xlo, xhi;
    cand_range(p.mode, p.align_corners, sh, iy, p.Hi, p.Ho, ylo, yhi);
    cand_range(p.mode, p.align_corners, sw, ix, p.Wi, p.Wo, xlo, xhi);
    const float* g = p.src + pl * p.Ho * p.Wo;
    const int nx = xhi - xlo + 1;
    float wxs[ADJ_MAXW];
    const bool cached = nx <= ADJ_MAXW;
    if (cached) {
#pragma unroll
      for (int k = 0; k < ADJ_MAXW; ++k) wxs[k] = (k < nx) ? adj_w(p.mode, p.align_corners, sw, xlo + k, p.Wi, ix) : 0.f;
    }
    float acc = 0.f;
    for (int oy = ylo + sub; oy <= yhi; oy += 8) {
      const float wy = adj_w(p.mode, p.align_corners, sh, oy, p.Hi, iy);
      if (wy == 0.f) continue;
      const float* grow = g + (size_t)oy * p.Wo + xlo;
      float rowacc = 0.f;
      if (cached) {
#pragma unroll
        for (int k = 0; k < ADJ_MAXW; ++k)
          if (k < nx) rowacc += wxs[k] * grow[k];
      } else {
        for (int k = 0; k < nx; ++k) rowacc += adj_w(p.mode, p.align_corners, sw, xlo + k, p.Wi, ix) * grow[k];
      }
      acc += wy * rowacc;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) p.dst[i] = acc;
  }
}

// channels_last: one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void resize_fwd_cl(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int C4 = p.C >> 2;
  const int64_t total = (int64_t)p.planes * p.Ho * p.Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int ox = (int)(r % p.Wo); r /= p.Wo;
    const int oy = (int)(r % p.Ho);
    const int64_t b = r / p.Ho;
    const Taps ty = taps_1d(p.mode, p.align_corners, sh, oy, p.Hi);
    const Taps tx = taps_1d(p.mode, p.align_corners, sw, ox, p.Wi);
    const float4* s = (const float4*)p.src + b * p.Hi * p.Wi * C4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < ty.n; ++a)
      for (int q = 0; q < tx.n; ++q) {
        const float w = ty.w[a] * tx.w[q];
        const float4 v = s[((size_t)ty.idx[a] * p.Wi + tx.idx[q]) * C4];
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      }
    ((float4*)p.dst)[i] = acc;
  }
}

__global__ __launch_bounds__(256) void resize_bwd_cl(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int C4 = p.C >> 2;
  const int64_t total = (int64_t)p.planes * p.Hi * p.Wi * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int ix = (int)(r % p.Wi); r /= p.Wi;
    const int iy = (int)(r % p.Hi);
    const int64_t b = r / p.Hi;
    int ylo, yhi, xlo, xhi;
    cand_range(p.mode, p.align_corners, sh, iy, p.Hi, p.Ho, ylo, yhi);
    cand_range(p.mode, p.align_corners, sw, ix, p.Wi, p.Wo, xlo, xhi);
    const float4* g = (const float4*)p.src + b * p.Ho * p.Wo * C4 + c;
    const int nx = xhi - xlo + 1;
    float wxs[ADJ_MAXW];
    const bool cached = nx <= ADJ_MAXW;
    if (cached) {
#pragma unroll
      for (int k = 0; k < ADJ_MAXW; ++k) wxs[k] = (k < nx) ? adj_w(p.mode, p.align_corners, sw, xlo + k, p.Wi, ix) : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = adj_w(p.mode, p.align_corners, sh, oy, p.Hi, iy);
      if (wy == 0.f) continue;
      const float4* grow = g + ((size_t)oy * p.Wo + xlo) * C4;
      if (cached) {
#pragma unroll
        for (int k = 0; k < ADJ_MAXW; ++k) {
          if (k < nx && wxs[k] != 0.f) {
            const float w = wy * wxs[k];
            const float4 v = grow[(size_t)k * C4];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
          }
        }
      } else {
        for (int k = 0; k < nx; ++k) {
          const float w = wy * adj_w(p.mode, p.align_corners, sw, xlo + k, p.Wi, ix);
          if (w == 0.f) continue;
          const float4 v = grow[(size_t)k * C4];
          acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
      }
    }
    ((float4*)p.dst)[i] = acc;
  }
}

inline int grid_for(int64_t work) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

int validate(const mvp_resize_args* a) {
  if (!a || !a->src || !a->dst) return MVP_EINVAL;
  if (a->planes <= 0 || a->Hi <= 0 || a->Wi <= 0 || a->Ho <= 0 || a->Wo <= 0) return MVP_EINVAL;
  if (a->mode < 0 || a->mode > 2) return MVP_EINVAL;
  if (a->channels_last && (a->C <= 0 || (a->C & 3))) return MVP_EINVAL;
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_resize_fwd(const mvp_resize_args* a, void* stream) {
  if (int e = validate(a)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (a->channels_last)
    hipLaunchKernelGGL(resize_fwd_cl, dim3(grid_for((int64_t)a->planes * a->Ho * a->Wo * (a->C >> 2))), dim3(256), 0, s, *a);
  else
    hipLaunchKernelGGL(resize_fwd_planar, dim3(grid_for((int64_t)a->planes * a->Ho * a->Wo)), dim3(256), 0, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_resize_bwd(const mvp_resize_args* a, void* stream) {
  if (int e = validate(a)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (a->channels_last)
    hipLaunchKernelGGL(resize_bwd_cl, dim3(grid_for((int64_t)a->planes * a->Hi * a->Wi * (a->C >> 2))), dim3(256), 0, s, *a);
  else
    hipLaunchKernelGGL(resize_bwd_planar, dim3(grid_for((int64_t)a->planes * a->Hi * a->Wi * 8)), dim3(256), 0, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
