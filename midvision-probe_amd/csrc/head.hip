// Fused tail of the linear depth-bin probe (probes.py:427-432 with k = 1 + probes.py:176-200):
//
//   forward : depth[b,y,x] = E_bins( normalise( relu( bilinear_x_f( L0 )[b,y,x,:] ) + 0.1 ) )
//             L0 = token-resolution logits [B,h,w,K] (the 1x1 conv commutes with the resample).
//             The f-times upsampled logits (51 MB at B=16, 224^2) are never written: each wave
//             rebuilds its pixel's K logits from the 4 neighbouring token rows (L2 resident) and
//             keeps only depth, 1/sum and ONE GATE BIT per logit (relu'(l) = [l > 0]).
//   backward: dL0[b,ty,tx,k] = sum_px w(px -> token) * gs[px] * gate[px,k] * (bin_k - depth[px])
//                            = bin_k * S1[k] - S2[k],  S1 = sum w*gs*gate,  S2 = sum w*gs*depth*gate
//             one wave per token gathering its <= (2f+3)^2 candidate pixels: 3 scalars + K/8 gate
//             bytes per pixel instead of a K-float row (25x less traffic than the unfused chain).
#include "mvp_common.h"

namespace {

__device__ __forceinline__ float bin_value(int k, int K, float lo, float hi) {
  const float step = (hi - lo) / (float)(K - 1);
  return (k < K / 2) ? lo + step * (float)k : hi - step * (float)(K - 1 - k);
}

// ATen bilinear (align_corners = False) source taps of output index o
__device__ __forceinline__ void bil_taps(float scale, int o, int in, int& i0, int& i1, float& w0, float& w1) {
  const float s = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = min((int)s, in - 1);
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}

__global__ __launch_bounds__(256) void linear_bins_fwd_kernel(const mvp_linear_bins_args p) {
  const int lane = threadIdx.x & 63;
  const int Ho = p.h * p.f, Wo = p.w * p.f;
  const int64_t px = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (px >= (int64_t)p.B * Ho * Wo) return;
  const int x = (int)(px % Wo);
  const int64_t r = px / Wo;
  const int y = (int)(r % Ho);
  const int64_t b = r / Ho;
  const float sc = 1.0f / (float)p.f;
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bil_taps(sc, y, p.h, y0, y1, wy0, wy1);
  bil_taps(sc, x, p.w, x0, x1, wx0, wx1);
  const float* base = p.l0 + b * p.h * p.w * p.K;
  const float* r00 = base + ((size_t)y0 * p.w + x0) * p.K;
  const float* r01 = base + ((size_t)y0 * p.w + x1) * p.K;
  const float* r10 = base + ((size_t)y1 * p.w + x0) * p.K;
  const float* r11 = base + ((size_t)y1 * p.w + x1) * p.K;
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
  float s = 0.f, d = 0.f;
  uint8_t* gate = p.gate + px * (p.K >> 3);
  for (int k = lane * 4; k < p.K; k += 256) {
    const float4 a = *(const float4*)(r00 + k), bq = *(const float4*)(r01 + k), c = *(const float4*)(r10 + k), e = *(const float4*)(r11 + k);
    // same association as the planar/channels-last resize kernels: rows first, then columns
    float l[4] = {wy0 * (wx0 * a.x + wx1 * bq.x) + wy1 * (wx0 * c.x + wx1 * e.x), wy0 * (wx0 * a.y + wx1 * bq.y) + wy1 * (wx0 * c.y + wx1 * e.y),
                  wy0 * (wx0 * a.z + wx1 * bq.z) + wy1 * (wx0 * c.z + wx1 * e.z), wy0 * (wx0 * a.w + wx1 * bq.w) + wy1 * (wx0 * c.w + wx1 * e.w)};
    (void)w00; (void)w01; (void)w10; (void)w11;
    unsigned nib = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      nib |= (l[q] > 0.f ? 1u : 0u) << q;
      const float pq = fmaxf(l[q], 0.f) + 0.1f;
      s += pq;
      d += pq * bin_value(k + q, p.K, p.min_depth, p.max_depth);
    }
    const unsigned other = __shfl_xor(nib, 1, 64);
    if ((lane & 1) == 0) gate[k >> 3] = (uint8_t)(nib | (other << 4));
  }
  s = wave_sum(s);
  d = wave_sum(d);
  if (lane == 0) {
    const float inv = 1.0f / s;
    p.depth[px] = d * inv;
    p.inv_sum[px] = inv;
  }
}

__global__ __launch_bounds__(256) void linear_bins_bwd_kernel(const mvp_linear_bins_args p) {
  const int lane = threadIdx.x & 63;
  const int Ho = p.h * p.f, Wo = p.w * p.f;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (int64_t)p.B * p.h * p.w) return;
  const int tx = (int)(tok % p.w);
  const int64_t r = tok / p.w;
  const int ty = (int)(r % p.h);
  const int64_t b = r / p.h;
  const float sc = 1.0f / (float)p.f;
  // candidate output range touching this token (superset; exact weights computed per candidate)
  // output o reads tokens floor(s), floor(s)+1 with s = (o + 0.5)/f - 0.5  =>  token t is touched by
  // o in (f*(t - 0.5) - 0.5, f*(t + 1.5) - 0.5); one extra candidate on each side for safety.
  const float ff = (float)p.f;
  int ylo = max(0, (int)floorf(ff * ((float)ty - 0.5f) - 0.5f) - 1), yhi = min(Ho - 1, (int)ceilf(ff * ((float)ty + 1.5f) - 0.5f) + 1);
  int xlo = max(0, (int)floorf(ff * ((float)tx - 0.5f) - 0.5f) - 1), xhi = min(Wo - 1, (int)ceilf(ff * ((float)tx + 1.5f) - 0.5f) + 1);
  if (ty == 0) ylo = 0;
  if (ty == p.h - 1) yhi = Ho - 1;
  if (tx == 0) xlo = 0;
  if (tx == p.w - 1) xhi = Wo - 1;
  const int KB = p.K >> 3;
  for (int k = lane * 4; k < p.K; k += 256) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oy = ylo; oy <= yhi; ++oy) {
      int i0, i1; float w0, w1;
      bil_taps(sc, oy, p.h, i0, i1, w0, w1);
      const float wy = (i0 == ty ? w0 : 0.f) + (i1 == ty ? w1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        bil_taps(sc, ox, p.w, i0, i1, w0, w1);
        const float wx = (i0 == tx ? w0 : 0.f) + (i1 == tx ? w1 : 0.f);
        if (wx == 0.f) continue;
        const int64_t px = (b * Ho + oy) * Wo + ox;
        const float a = wy * wx * p.grad_depth[px] * p.inv_sum[px];
        const float bb = a * p.depth[px];
        const unsigned byte = p.gate[px * KB + (k >> 3)];
        const unsigned nib = (lane & 1) ? (byte >> 4) : (byte & 15u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float m = (float)((nib >> q) & 1u);
          s1[q] += a * m;
          s2[q] += bb * m;
        }
      }
    }
    float4 o;
    o.x = bin_value(k, p.K, p.min_depth, p.max_depth) * s1[0] - s2[0];
    o.y = bin_value(k + 1, p.K, p.min_depth, p.max_depth) * s1[1] - s2[1];
    o.z = bin_value(k + 2, p.K, p.min_depth, p.max_depth) * s1[2] - s2[2];
    o.w = bin_value(k + 3, p.K, p.min_depth, p.max_depth) * s1[3] - s2[3];
    *(float4*)(p.grad_l0 + tok * p.K + k) = o;
  }
}

// ------------------------------------------------------------------ fast paths (K == 256, f <= 4)
// Both generic kernels above are LATENCY-bound, not instruction-bound (PMC: waves issue-stalled / parked ~75 % of their
// life): the forward issued 3 tiny stores per pixel behind two wave reductions, the backward ran <= 144 dependent
// load -> use iterations per token.  The fast paths keep the arithmetic (same association, bit-identical results) and
// restructure the memory traffic.

__device__ __forceinline__ int bil_cell(float scale, int o, int in) {
  const float s = scale * ((float)o + 0.5f) - 0.5f;
  return (s < 0.f) ? -1 : min((int)s, in - 1);
}

// Forward: one wave per upsampling CELL = the output pixels whose source corner (y0, x0) is the same token pair
// ((h + 1) x (w + 1) cells per image, <= (f + 1)^2 <= 16 pixels).  The four corner token rows are loaded once; the
// cell's results are collected in LDS and leave with 3 store instructions (depth, 1/sum, 16 x 32-byte gate rows).
__global__ __launch_bounds__(256) void linear_bins_fwd_cells(const mvp_linear_bins_args p) {
  __shared__ __attribute__((aligned(16))) uint8_t gate_s[4][16][32];
  __shared__ float dep_s[4][16], inv_s[4][16];
  __shared__ int px_s[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Ho = p.h * p.f, Wo = p.w * p.f;
  const int cells_x = p.w + 1, cells_y = p.h + 1;
  const int64_t cell = (int64_t)blockIdx.x * 4 + wave;
  if (cell >= (int64_t)p.B * cells_y * cells_x) return;  // wave-uniform; no block barrier below
  const int cx = (int)(cell % cells_x) - 1;
  const int64_t r = cell / cells_x;
  const int cy = (int)(r % cells_y) - 1;
  const int64_t b = r / cells_y;
  const float sc = 1.0f / (float)p.f, ff = (float)p.f;
  const int r0 = max(cy, 0), r1 = min(r0 + 1, p.h - 1), c0 = max(cx, 0), c1 = min(c0 + 1, p.w - 1);
  const float* base = p.l0 + b * p.h * p.w * 256;
  const int k = lane * 4;
  const float4 a = *(const float4*)(base + ((size_t)r0 * p.w + c0) * 256 + k);
  const float4 bq = *(const float4*)(base + ((size_t)r0 * p.w + c1) * 256 + k);
  const float4 c = *(const float4*)(base + ((size_t)r1 * p.w + c0) * 256 + k);
  const float4 e = *(const float4*)(base + ((size_t)r1 * p.w + c1) * 256 + k);
  float bins[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bins[q] = bin_value(k + q, 256, p.min_depth, p.max_depth);
  // candidate outputs of this cell (superset; exact membership via bil_cell)
  const int ylo = max(0, (int)floorf(ff * ((float)cy + 0.5f) - 0.5f) - 1), yhi = min(Ho - 1, (int)ceilf(ff * ((float)cy + 1.5f) - 0.5f) + 1);
  const int xlo = max(0, (int)floorf(ff * ((float)cx + 0.5f) - 0.5f) - 1), xhi = min(Wo - 1, (int)ceilf(ff * ((float)cx + 1.5f) - 0.5f) + 1);
  int npix = 0;  // wave-uniform
  for (int y = ylo; y <= yhi; ++y) {
    if (bil_cell(sc, y, p.h) != cy) continue;
    int y0, y1; float wy0, wy1;
    bil_taps(sc, y, p.h, y0, y1, wy0, wy1);
    for (int x = xlo; x <= xhi; ++x) {
      if (bil_cell(sc, x, p.w) != cx) continue;
      int x0, x1; float wx0, wx1;
      bil_taps(sc, x, p.w, x0, x1, wx0, wx1);
      // same association as the planar/channels-last resize kernels: rows first, then columns
      const float l[4] = {wy0 * (wx0 * a.x + wx1 * bq.x) + wy1 * (wx0 * c.x + wx1 * e.x), wy0 * (wx0 * a.y + wx1 * bq.y) + wy1 * (wx0 * c.y + wx1 * e.y),
                          wy0 * (wx0 * a.z + wx1 * bq.z) + wy1 * (wx0 * c.z + wx1 * e.z), wy0 * (wx0 * a.w + wx1 * bq.w) + wy1 * (wx0 * c.w + wx1 * e.w)};
      float s = 0.f, d = 0.f;
      unsigned nib = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        nib |= (l[q] > 0.f ? 1u : 0u) << q;
        const float pq = fmaxf(l[q], 0.f) + 0.1f;
        s += pq;
        d += pq * bins[q];
      }
      const unsigned other = __shfl_xor(nib, 1, 64);
      if ((lane & 1) == 0) gate_s[wave][npix][lane >> 1] = (uint8_t)(nib | (other << 4));
      s = wave_sum(s);
      d = wave_sum(d);
      if (lane == 0) {
        const float inv = 1.0f / s;
        dep_s[wave][npix] = d * inv;
        inv_s[wave][npix] = inv;
        px_s[wave][npix] = (int)(((int)b * Ho + y) * Wo + x);
      }
      ++npix;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < npix) {
    const int px = px_s[wave][lane];
    p.depth[px] = dep_s[wave][lane];
    p.inv_sum[px] = inv_s[wave][lane];
  }
  if (lane < 2 * npix) {  // 16 pixels x two 16-byte halves of the 32-byte gate row
    const int pi = lane >> 1, half = lane & 1;
    *(u32x4_t*)(p.gate + (size_t)px_s[wave][pi] * 32 + half * 16) = *(const u32x4_t*)&gate_s[wave][pi][half * 16];
  }
}

// Backward: one wave per token.  Phase 1: the <= 144 candidate pixels are spread over the lanes, each lane loads its
// pixel's three scalars ONCE (all loads in flight together) and keeps a = w * gs / sum, bb = a * depth.  Phase 2:
// the pixels with a != 0 are walked 8 at a time, their 32-byte gate rows loaded as a batch before use.
__global__ __launch_bounds__(256) void linear_bins_bwd_fast(const mvp_linear_bins_args p) {
  const int lane = threadIdx.x & 63;
  const int Ho = p.h * p.f, Wo = p.w * p.f;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= (int64_t)p.B * p.h * p.w) return;
  const int tx = (int)(tok % p.w);
  const int64_t r = tok / p.w;
  const int ty = (int)(r % p.h);
  const int64_t b = r / p.h;
  const float sc = 1.0f / (float)p.f, ff = (float)p.f;
  int ylo = max(0, (int)floorf(ff * ((float)ty - 0.5f) - 0.5f) - 1), yhi = min(Ho - 1, (int)ceilf(ff * ((float)ty + 1.5f) - 0.5f) + 1);
  int xlo = max(0, (int)floorf(ff * ((float)tx - 0.5f) - 0.5f) - 1), xhi = min(Wo - 1, (int)ceilf(ff * ((float)tx + 1.5f) - 0.5f) + 1);
  if (ty == 0) ylo = 0;
  if (ty == p.h - 1) yhi = Ho - 1;
  if (tx == 0) xlo = 0;
  if (tx == p.w - 1) xhi = Wo - 1;
  const int ny = yhi - ylo + 1, nx = xhi - xlo + 1, ncand = ny * nx;  // <= 12 x 12 at f = 4 (host checks f <= 4)
  const int k = lane * 4;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < ncand; c0 += 64) {
    // ---- phase 1: lane <-> candidate
    const int ci = c0 + lane;
    float a_l = 0.f, bb_l = 0.f;
    int px_l = 0;
    if (ci < ncand) {
      const int oy = ylo + ci / nx, ox = xlo + ci % nx;
      int i0, i1; float w0, w1;
      bil_taps(sc, oy, p.h, i0, i1, w0, w1);
      const float wy = (i0 == ty ? w0 : 0.f) + (i1 == ty ? w1 : 0.f);
      bil_taps(sc, ox, p.w, i0, i1, w0, w1);
      const float wx = (i0 == tx ? w0 : 0.f) + (i1 == tx ? w1 : 0.f);
      if (wy != 0.f && wx != 0.f) {
        px_l = (int)(((int)b * Ho + oy) * Wo + ox);
        a_l = wy * wx * p.grad_depth[px_l] * p.inv_sum[px_l];
        bb_l = a_l * p.depth[px_l];
      }
    }
    // ---- phase 2: touching pixels (ballot order = candidate order -> same summation order as the generic kernel)
    unsigned long long live = __ballot(a_l != 0.f || bb_l != 0.f);
    while (live) {
      int idx[8];
      unsigned byte[8];
      int n = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        idx[u] = -1;
        if (live) {
          idx[u] = __builtin_ctzll(live);
          live &= live - 1;
          n = u + 1;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < n) byte[u] = p.gate[(size_t)__shfl(px_l, idx[u], 64) * 32 + (lane >> 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < n) {
          const float av = __shfl(a_l, idx[u], 64), bv = __shfl(bb_l, idx[u], 64);
          const unsigned nib = (lane & 1) ? (byte[u] >> 4) : (byte[u] & 15u);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float m = (float)((nib >> q) & 1u);
            s1[q] += av * m;
            s2[q] += bv * m;
          }
        }
      }
    }
  }
  float4 o;
  o.x = bin_value(k, 256, p.min_depth, p.max_depth) * s1[0] - s2[0];
  o.y = bin_value(k + 1, 256, p.min_depth, p.max_depth) * s1[1] - s2[1];
  o.z = bin_value(k + 2, 256, p.min_depth, p.max_depth) * s1[2] - s2[2];
  o.w = bin_value(k + 3, 256, p.min_depth, p.max_depth) * s1[3] - s2[3];
  *(float4*)(p.grad_l0 + tok * 256 + k) = o;
}

}  // namespace

extern "C" int mvp_linear_bins_fwd(const mvp_linear_bins_args* a, void* stream) {
  if (!a || !a->l0 || !a->depth || !a->inv_sum || !a->gate) return MVP_EINVAL;
  if (a->B <= 0 || a->h <= 0 || a->w <= 0 || a->f < 1 || a->K < 8 || (a->K & 7)) return MVP_EINVAL;
  const int64_t P = (int64_t)a->B * a->h * a->f * a->w * a->f;
  static const bool generic = getenv("MVP_BINS_GENERIC") != nullptr;  // diagnostic: the one-wave-per-pixel kernel (used to bisect the
                                                                      // packed-fp32 problem described in the Makefile)
  if (!generic && a->K == 256 && a->f <= 4 && P < ((int64_t)1 << 31)) {
    const int64_t cells = (int64_t)a->B * (a->h + 1) * (a->w + 1);
    hipLaunchKernelGGL(linear_bins_fwd_cells, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  } else {
    hipLaunchKernelGGL(linear_bins_fwd_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_linear_bins_bwd(const mvp_linear_bins_args* a, void* stream) {
  if (!a || !a->grad_depth || !a->grad_l0 || !a->depth || !a->inv_sum || !a->gate) return MVP_EINVAL;
  if (a->B <= 0 || a->h <= 0 || a->w <= 0 || a->f < 1 || a->K < 8 || (a->K & 7)) return MVP_EINVAL;
  const int64_t T = (int64_t)a->B * a->h * a->w;
  if (a->K == 256 && a->f <= 4 && T * a->f * a->f < ((int64_t)1 << 31))
    hipLaunchKernelGGL(linear_bins_bwd_fast, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(linear_bins_bwd_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
