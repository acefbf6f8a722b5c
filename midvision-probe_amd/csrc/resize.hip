// Nearest / bilinear / bicubic resampling, forward and adjoint (gather form: deterministic,
// no atomics).  Index arithmetic follows ATen (area_pixel_compute_source_index,
// nearest_neighbor_compute_source_index, cubic A = -0.75 with clamped taps), which is what
// F.interpolate runs in the reference (train_depth.py:114, train_snorm.py:110,
// probes.py:255-258,388,396-398,431).
//
// Layouts: planar [planes, H, W] (lanes along W), or channels_last [B, H, W, C] (lanes along
// C, float4) for the token-major probe logits.
//
// Every kernel is instantiated per MODE so that the tap count is a compile-time constant: the tap loops unroll, the
// tap arrays stay in registers (no scratch) and the gathers of one output are all in flight together.
#include "mvp_common.h"

namespace {

template <int MODE>
struct Taps {
  static constexpr int N = (MODE == MVP_RESIZE_NEAREST) ? 1 : (MODE == MVP_RESIZE_BILINEAR) ? 2 : 4;
  int idx[N];
  float w[N];
};

__device__ __forceinline__ float src_scale(int in, int out, int align, float sf) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return sf > 0.f ? 1.0f / sf : (float)in / (float)out;
}

template <int MODE>
__device__ __forceinline__ Taps<MODE> taps_1d(int align, float scale, int o, int in) {
  Taps<MODE> t;
  if constexpr (MODE == MVP_RESIZE_NEAREST) {
    t.idx[0] = min((int)floorf((float)o * scale), in - 1);
    t.w[0] = 1.f;
  } else if constexpr (MODE == MVP_RESIZE_BILINEAR) {
    float s = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    int i0 = min((int)s, in - 1);
    int i1 = i0 + (i0 < in - 1 ? 1 : 0);
    float l1 = s - (float)i0;
    t.idx[0] = i0; t.idx[1] = i1;
    t.w[0] = 1.f - l1; t.w[1] = l1;
  } else {
    float s = align ? scale * (float)o : scale * ((float)o + 0.5f) - 0.5f;
    float fl = floorf(s);
    float x = s - fl;
    int i = (int)fl;
    const float A = -0.75f;
    float x0 = x + 1.f, x1 = x, x2 = 1.f - x, x3 = 2.f - x;
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
// membership test is done per candidate with adj_w).
template <int MODE>
__device__ __forceinline__ void cand_range(int align, float scale, int i, int in, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float r = (MODE == MVP_RESIZE_BICUBIC) ? 2.f : 1.f;
  const float off = (align || MODE == MVP_RESIZE_NEAREST) ? 0.f : 0.5f;
  float flo = ((float)i - r + off) / scale - off;
  float fhi = ((float)i + r + off) / scale - off;
  lo = max(0, (int)floorf(flo) - 1);
  hi = min(out - 1, (int)ceilf(fhi) + 1);
  if (i == 0) lo = 0;
  if (i == in - 1) hi = out - 1;
}

// Adjoint weight of output index o on input index i along one axis (0 when o does not touch i).
template <int MODE>
__device__ __forceinline__ float adj_w(int align, float scale, int o, int in, int i) {
  const Taps<MODE> t = taps_1d<MODE>(align, scale, o, in);
  float w = 0.f;
#pragma unroll
  for (int a = 0; a < Taps<MODE>::N; ++a) w += (t.idx[a] == i) ? t.w[a] : 0.f;
  return w;
}

// One output row segment per thread group: cw = 2^cw_log2 columns x (256 / cw) rows per block, so the row taps are
// computed without any per-pixel division and stores are row-contiguous.
template <int MODE>
__global__ __launch_bounds__(256) void resize_fwd_planar(const mvp_resize_args p, const int cw_log2) {
  constexpr int NT = Taps<MODE>::N;
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int cw = 1 << cw_log2, rb = 256 >> cw_log2;
  const int lx = threadIdx.x & (cw - 1), ly = threadIdx.x >> cw_log2;
  const int64_t rows = (int64_t)p.planes * p.Ho;
  for (int64_t row = (int64_t)blockIdx.x * rb + ly; row < rows; row += (int64_t)gridDim.x * rb) {
    const int64_t pl = row / p.Ho;
    const int oy = (int)(row - pl * p.Ho);
    const Taps<MODE> ty = taps_1d<MODE>(p.align_corners, sh, oy, p.Hi);
    const float* s = p.src + pl * p.Hi * p.Wi;
    for (int ox = lx; ox < p.Wo; ox += cw) {
      const Taps<MODE> tx = taps_1d<MODE>(p.align_corners, sw, ox, p.Wi);
      float v[NT][NT];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) v[a][b] = s[(size_t)ty.idx[a] * p.Wi + tx.idx[b]];
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int b = 0; b < NT; ++b) rowacc += tx.w[b] * v[a][b];
        acc += ty.w[a] * rowacc;
      }
      p.dst[row * p.Wo + ox] = acc;
    }
  }
}

// Separable adjoint, one workgroup per input row (plane, iy): first the y-adjoint of the candidate output rows into an
// LDS row buffer (coalesced row reads, the row weights computed once per workgroup), then the x-adjoint out of LDS.
constexpr int ADJ_ROWS = 64, ADJ_WO = 8192;

template <int MODE>
__global__ __launch_bounds__(256) void resize_bwd_planar_rows(const mvp_resize_args p) {
  __shared__ float rowbuf[ADJ_WO];
  __shared__ float wys[ADJ_ROWS];
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int t = threadIdx.x;
  const int64_t rows = (int64_t)p.planes * p.Hi;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int64_t pl = row / p.Hi;
    const int iy = (int)(row - pl * p.Hi);
    int ylo, yhi;
    cand_range<MODE>(p.align_corners, sh, iy, p.Hi, p.Ho, ylo, yhi);
    const int ny = yhi - ylo + 1;
    const bool cached = ny <= ADJ_ROWS;  // uniform
    if (cached && t < ny) wys[t] = adj_w<MODE>(p.align_corners, sh, ylo + t, p.Hi, iy);
    __syncthreads();
    const float* g = p.src + pl * p.Ho * p.Wo;
    for (int ox = t; ox < p.Wo; ox += 256) {
      const float* gc = g + (size_t)ylo * p.Wo + ox;
      float acc = 0.f;
      int k = 0;
      if (cached) {
        for (; k + 4 <= ny; k += 4) {  // 4 row loads in flight
          const float v0 = gc[(size_t)k * p.Wo], v1 = gc[(size_t)(k + 1) * p.Wo], v2 = gc[(size_t)(k + 2) * p.Wo], v3 = gc[(size_t)(k + 3) * p.Wo];
          acc += wys[k] * v0;
          acc += wys[k + 1] * v1;
          acc += wys[k + 2] * v2;
          acc += wys[k + 3] * v3;
        }
      }
      for (; k < ny; ++k) {
        const float w = cached ? wys[k] : adj_w<MODE>(p.align_corners, sh, ylo + k, p.Hi, iy);
        acc += w * gc[(size_t)k * p.Wo];
      }
      rowbuf[ox] = acc;
    }
    __syncthreads();
    for (int ix = t; ix < p.Wi; ix += 256) {
      int xlo, xhi;
      cand_range<MODE>(p.align_corners, sw, ix, p.Wi, p.Wo, xlo, xhi);
      float acc = 0.f;
      for (int ox = xlo; ox <= xhi; ++ox) acc += adj_w<MODE>(p.align_corners, sw, ox, p.Wi, ix) * rowbuf[ox];
      p.dst[row * p.Wi + ix] = acc;
    }
    __syncthreads();
  }
}

constexpr int ADJ_MAXW = 24;  // candidate columns whose weights are cached in registers

// Fallback for rows wider than the LDS row buffer: 8 lanes per input element, candidate rows strided by 8.
template <int MODE>
__global__ __launch_bounds__(256) void resize_bwd_planar(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const int64_t total = (int64_t)p.planes * p.Hi * p.Wi;
  const int sub = threadIdx.x & 7;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; i < total; i += ((int64_t)gridDim.x * 256) >> 3) {
    const int ix = (int)(i % p.Wi);
    const int64_t r = i / p.Wi;
    const int iy = (int)(r % p.Hi);
    const int64_t pl = r / p.Hi;
    int ylo, yhi, xlo, xhi;
    cand_range<MODE>(p.align_corners, sh, iy, p.Hi, p.Ho, ylo, yhi);
    cand_range<MODE>(p.align_corners, sw, ix, p.Wi, p.Wo, xlo, xhi);
    const float* g = p.src + pl * p.Ho * p.Wo;
    float acc = 0.f;
    for (int oy = ylo + sub; oy <= yhi; oy += 8) {
      const float wy = adj_w<MODE>(p.align_corners, sh, oy, p.Hi, iy);
      if (wy == 0.f) continue;
      const float* grow = g + (size_t)oy * p.Wo;
      float rowacc = 0.f;
      for (int ox = xlo; ox <= xhi; ++ox) rowacc += adj_w<MODE>(p.align_corners, sw, ox, p.Wi, ix) * grow[ox];
      acc += wy * rowacc;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (sub == 0) p.dst[i] = acc;
  }
}

// channels_last: one thread per (pixel, 4 channels); IDX = unsigned when the element count fits 32 bits (the
// decomposition costs three runtime divisions per element, 64-bit ones are several times dearer).
template <int MODE, typename IDX>
__global__ __launch_bounds__(256) void resize_fwd_cl(const mvp_resize_args p) {
  constexpr int NT = Taps<MODE>::N;
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const IDX C4 = (IDX)(p.C >> 2);
  const IDX total = (IDX)p.planes * (IDX)p.Ho * (IDX)p.Wo * C4;
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const IDX pix = i / C4;
    const int c = (int)(i - pix * C4);
    const IDX r = pix / (IDX)p.Wo;
    const int ox = (int)(pix - r * (IDX)p.Wo);
    const IDX b = r / (IDX)p.Ho;
    const int oy = (int)(r - b * (IDX)p.Ho);
    const Taps<MODE> ty = taps_1d<MODE>(p.align_corners, sh, oy, p.Hi);
    const Taps<MODE> tx = taps_1d<MODE>(p.align_corners, sw, ox, p.Wi);
    const float4* s = (const float4*)p.src + (size_t)b * p.Hi * p.Wi * C4 + c;
    float4 v[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < NT; ++q) v[a][q] = s[((size_t)ty.idx[a] * p.Wi + tx.idx[q]) * C4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const float w = ty.w[a] * tx.w[q];
        acc.x += w * v[a][q].x; acc.y += w * v[a][q].y; acc.z += w * v[a][q].z; acc.w += w * v[a][q].w;
      }
    ((float4*)p.dst)[i] = acc;
  }
}

template <int MODE, typename IDX>
__global__ __launch_bounds__(256) void resize_bwd_cl(const mvp_resize_args p) {
  const float sh = src_scale(p.Hi, p.Ho, p.align_corners, p.scale_h);
  const float sw = src_scale(p.Wi, p.Wo, p.align_corners, p.scale_w);
  const IDX C4 = (IDX)(p.C >> 2);
  const IDX total = (IDX)p.planes * (IDX)p.Hi * (IDX)p.Wi * C4;
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const IDX pix = i / C4;
    const int c = (int)(i - pix * C4);
    const IDX r = pix / (IDX)p.Wi;
    const int ix = (int)(pix - r * (IDX)p.Wi);
    const IDX b = r / (IDX)p.Hi;
    const int iy = (int)(r - b * (IDX)p.Hi);
    int ylo, yhi, xlo, xhi;
    cand_range<MODE>(p.align_corners, sh, iy, p.Hi, p.Ho, ylo, yhi);
    cand_range<MODE>(p.align_corners, sw, ix, p.Wi, p.Wo, xlo, xhi);
    const float4* g = (const float4*)p.src + (size_t)b * p.Ho * p.Wo * C4 + c;
    const int nx = xhi - xlo + 1;
    float wxs[ADJ_MAXW];
    const bool cached = nx <= ADJ_MAXW;
    if (cached) {
#pragma unroll
      for (int k = 0; k < ADJ_MAXW; ++k) wxs[k] = (k < nx) ? adj_w<MODE>(p.align_corners, sw, xlo + k, p.Wi, ix) : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = adj_w<MODE>(p.align_corners, sh, oy, p.Hi, iy);
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
          const float w = wy * adj_w<MODE>(p.align_corners, sw, xlo + k, p.Wi, ix);
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

template <int MODE>
void launch_fwd(const mvp_resize_args* a, hipStream_t s) {
  if (a->channels_last) {
    const int64_t total = (int64_t)a->planes * a->Ho * a->Wo * (a->C >> 2);
    if (total < (int64_t)1 << 31)
      hipLaunchKernelGGL((resize_fwd_cl<MODE, unsigned>), dim3(grid_for(total)), dim3(256), 0, s, *a);
    else
      hipLaunchKernelGGL((resize_fwd_cl<MODE, int64_t>), dim3(grid_for(total)), dim3(256), 0, s, *a);
  } else {
    int cw_log2 = 0;
    while (cw_log2 < 8 && (1 << cw_log2) < a->Wo) ++cw_log2;
    const int64_t rows = (int64_t)a->planes * a->Ho, rb = 256 >> cw_log2;
    const int64_t g = (rows + rb - 1) / rb;
    hipLaunchKernelGGL((resize_fwd_planar<MODE>), dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, s, *a, cw_log2);
  }
}

template <int MODE>
void launch_bwd(const mvp_resize_args* a, hipStream_t s) {
  if (a->channels_last) {
    const int64_t total = (int64_t)a->planes * a->Hi * a->Wi * (a->C >> 2);
    if (total < (int64_t)1 << 31)
      hipLaunchKernelGGL((resize_bwd_cl<MODE, unsigned>), dim3(grid_for(total)), dim3(256), 0, s, *a);
    else
      hipLaunchKernelGGL((resize_bwd_cl<MODE, int64_t>), dim3(grid_for(total)), dim3(256), 0, s, *a);
  } else if (a->Wo <= ADJ_WO) {
    const int64_t rows = (int64_t)a->planes * a->Hi;
    hipLaunchKernelGGL((resize_bwd_planar_rows<MODE>), dim3((unsigned)(rows > 65536 ? 65536 : rows)), dim3(256), 0, s, *a);
  } else {
    hipLaunchKernelGGL((resize_bwd_planar<MODE>), dim3(grid_for((int64_t)a->planes * a->Hi * a->Wi * 8)), dim3(256), 0, s, *a);
  }
}

// Antialiased bilinear resampling, forward only (torchvision transforms.Resize on a tensor = F.interpolate(bilinear,
// align_corners=False, antialias=True): dino_res50.py:80,85 when an input axis is LARGER than fixed_size).  ATen's
// _upsample_bilinear2d_aa: per axis, scale = in/out, support = max(scale, 1), the triangle filter is stretched by the
// scale and its taps renormalised; an axis that up-samples or keeps its size reduces to plain bilinear / identity.
struct AaAxis { int lo, n; float center, inv, total; };
__device__ __forceinline__ float aa_w(const AaAxis& a, int j) {
  const float x = fabsf(((float)(j + a.lo) - a.center + 0.5f) * a.inv);
  return x < 1.f ? 1.f - x : 0.f;
}
__device__ __forceinline__ AaAxis aa_axis(int o, int in, float scale) {
  AaAxis a;
  const float support = scale >= 1.f ? scale : 1.f;
  a.center = scale * ((float)o + 0.5f);
  a.inv = scale >= 1.f ? 1.f / scale : 1.f;
  a.lo = max((int)(a.center - support + 0.5f), 0);
  a.n = min((int)(a.center + support + 0.5f), in) - a.lo;
  a.total = 0.f;
  for (int j = 0; j < a.n; ++j) a.total += aa_w(a, j);
  return a;
}
__global__ __launch_bounds__(256) void resize_aa_fwd_planar(const mvp_resize_args p) {
  const int64_t total = (int64_t)p.planes * p.Ho * p.Wo;
  const float sh = (float)p.Hi / (float)p.Ho, sw = (float)p.Wi / (float)p.Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % p.Wo), yo = (int)((i / p.Wo) % p.Ho);
    const int64_t pl = i / ((int64_t)p.Wo * p.Ho);
    const AaAxis ay = aa_axis(yo, p.Hi, sh), ax = aa_axis(xo, p.Wi, sw);
    const float* src = p.src + pl * p.Hi * p.Wi;
    float acc = 0.f;
    for (int jy = 0; jy < ay.n; ++jy) {
      const float* row = src + (int64_t)(ay.lo + jy) * p.Wi + ax.lo;
      float r = 0.f;  // horizontal pass first, as ATen's separable kernel
      for (int jx = 0; jx < ax.n; ++jx) r += (aa_w(ax, jx) / ax.total) * row[jx];
      acc += (aa_w(ay, jy) / ay.total) * r;
    }
    p.dst[i] = acc;
  }
}

}  // namespace

extern "C" int mvp_resize_aa_fwd(const mvp_resize_args* a, void* stream) {
  if (int e = validate(a)) return e;
  if (a->mode != MVP_RESIZE_BILINEAR || a->channels_last || a->align_corners || a->scale_h != 0.f || a->scale_w != 0.f) return MVP_EINVAL;
  hipLaunchKernelGGL(resize_aa_fwd_planar, dim3(grid_for((int64_t)a->planes * a->Ho * a->Wo)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_resize_fwd(const mvp_resize_args* a, void* stream) {
  if (int e = validate(a)) return e;
  hipStream_t s = (hipStream_t)stream;
  switch (a->mode) {
    case MVP_RESIZE_NEAREST: launch_fwd<MVP_RESIZE_NEAREST>(a, s); break;
    case MVP_RESIZE_BILINEAR: launch_fwd<MVP_RESIZE_BILINEAR>(a, s); break;
    default: launch_fwd<MVP_RESIZE_BICUBIC>(a, s); break;
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_resize_bwd(const mvp_resize_args* a, void* stream) {
  if (int e = validate(a)) return e;
  hipStream_t s = (hipStream_t)stream;
  switch (a->mode) {
    case MVP_RESIZE_NEAREST: launch_bwd<MVP_RESIZE_NEAREST>(a, s); break;
    case MVP_RESIZE_BILINEAR: launch_bwd<MVP_RESIZE_BILINEAR>(a, s); break;
    default: launch_bwd<MVP_RESIZE_BICUBIC>(a, s); break;
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
