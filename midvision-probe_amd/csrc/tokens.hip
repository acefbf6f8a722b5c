// Token-stream plumbing kernels (all HBM-bound, coalesced, no atomics):
//   split_bf16        fp32 -> bf16 pair
//   patch_gather      NCHW image -> [patches, C*P*P] bf16 pair (im2col of the 16x16/16 conv,
//                     zero "center padding" applied on the fly)
//   cls_rows          x[b,0,:] = cls + pos[0]
//   bn_tokens         train-mode BatchNorm over tokens fused with token->NCHW transpose and
//                     the token-major bf16 packing the probe-head GEMMs consume
//   pack_nchw_tokens  NCHW fp32 -> token-major bf16 pair (+ transposed copy)
#include "mvp_common.h"

namespace {

// ----------------------------------------------------------------------------- split
__global__ __launch_bounds__(256) void split_kernel(const mvp_split_bf16_args p) {
  const int64_t n4 = p.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = ((const float4*)p.src)[i];
    uint16_t h[4], l[4];
    split_bf16(v.x, h[0], l[0]); split_bf16(v.y, h[1], l[1]);
    split_bf16(v.z, h[2], l[2]); split_bf16(v.w, h[3], l[3]);
    ((u32x2_t*)p.hi)[i] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
    if (p.lo) ((u32x2_t*)p.lo)[i] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
  }
  if (blockIdx.x == 0 && threadIdx.x < (p.n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    uint16_t h, l;
    split_bf16(p.src[i], h, l);
    p.hi[i] = h;
    if (p.lo) p.lo[i] = l;
  }
}

// ----------------------------------------------------------------------------- patch gather
__global__ __launch_bounds__(256) void patch_gather_kernel(const mvp_patch_gather_args p) {
  const int P = p.P, PP = P * P, Kc = p.C * PP;
  const int64_t rows = (int64_t)p.B * p.gh * p.gw;
  const int64_t total4 = rows * (Kc >> 2);
  const bool vec = ((p.W & 3) == 0) && ((p.pad_left & 3) == 0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / (Kc >> 2);
    const int col = (int)(i - row * (Kc >> 2)) << 2;
    const int c = col / PP, rem = col - c * PP, ky = rem / P, kx = rem - ky * P;
    const int b = (int)(row / (p.gh * p.gw));
    const int pr = (int)(row - (int64_t)b * p.gh * p.gw);
    const int py = pr / p.gw, px = pr - py * p.gw;
    const int y = py * P + ky - p.pad_top;
    const int x = px * P + kx - p.pad_left;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < p.H) {
      const float* src = p.images + (((size_t)b * p.C + c) * p.H + y) * p.W;
      if (vec && x >= 0 && x + 3 < p.W) {
        const float4 t = *(const float4*)(src + x);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (x + e >= 0 && x + e < p.W) v[e] = src[x + e];
      }
    }
    uint16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16(v[e], h[e], l[e]);
    const size_t o = (size_t)row * Kc + col;
    *(u32x2_t*)(p.out_hi + o) = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
    if (p.out_lo) *(u32x2_t*)(p.out_lo + o) = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
  }
}

// ----------------------------------------------------------------------------- cls rows
__global__ __launch_bounds__(256) void cls_rows_kernel(const mvp_cls_rows_args p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.B * p.C) return;
  const int b = i / p.C, c = i - b * p.C;
  p.x[(size_t)b * p.N * p.C + c] = p.cls[c] + p.pos0[c];
}

// ----------------------------------------------------------------------------- BN over tokens
constexpr int BN_RB = 8;   // rows per load batch of the partial-statistics pass

// rows per slab: 8 for the ViT taps (M ~ 3k -> ~400 workgroups), growing with M so that the finalize pass folds at
// most ~2k slabs per channel (ResNet taps have M = 230k rows: 8-row slabs meant 28.8k slabs and a 57 us finalize)
__host__ __device__ inline int bn_slab_rows(int M) {
  int rb = BN_RB;
  while ((M + rb - 1) / rb > 2048) rb <<= 1;
  return rb;
}

// Pass 1: per slab and channel, shifted sums (shift = first row of the slab) so that the later variance is free of
// catastrophic cancellation even when |mean| >> std.
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int M, int C, int rb) {
  x += (size_t)blockIdx.z * M * C;                      // batch blockIdx.z of a grouped call: its own rows,
  part += (size_t)blockIdx.z * gridDim.x * C * 3;       // its own partials
  const int r0 = blockIdx.x * rb;
  const int nr = min(rb, M - r0);
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const float* col = x + (size_t)r0 * C + c;
  const float shift = col[0];
  float s1 = 0.f, s2 = 0.f;
  for (int rr = 0; rr < nr; rr += BN_RB) {
    float v[BN_RB];
#pragma unroll
    for (int r = 0; r < BN_RB; ++r) v[r] = col[(size_t)min(rr + r, nr - 1) * C];  // all loads in flight together
#pragma unroll
    for (int r = 0; r < BN_RB; ++r) {
      const float d = (rr + r < nr) ? v[r] - shift : 0.f;
      s1 += d;
      s2 += d * d;
    }
  }
  float* o = part + ((size_t)blockIdx.x * C + c) * 3;
  o[0] = shift; o[1] = s1; o[2] = s2;
}

// Pass 2: a workgroup = 16 channels x 64 slab groups (48 workgroups at C = 768, so the 3.6 MB of partials are pulled
// by 48 CUs instead of 12).  Each thread folds its slabs (stride 64) into fp64 moments about the channel's
// first-slab shift (no per-slab division; fp64 keeps the centred second moment exact to ~1e-13 relative), the 64
// groups are combined through LDS in a fixed order (deterministic) -> mean, biased var; running stats (momentum,
// unbiased var) and the affine scale/shift used by the apply pass.
constexpr int BN_FC = 16, BN_FG = 64;

// The momentum update of a running statistic, written with explicit roundings (no FMA contraction) so that the in-kernel form and the
// deferred form (bn_running_update_kernel) give the same bits: (1 - m) * old + m * val, every operation rounded as torch's CPU kernel does.
__device__ __forceinline__ float bn_running(float old, float val, float momentum) {
#pragma clang fp contract(off)  // (the _rn intrinsics alone were still fused differently in the two kernels by one build)
  const float a = (1.f - momentum) * old;
  const float b = momentum * val;
  return a + b;
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(mvp_bn_tokens_args p, const float* __restrict__ part,
                                                           float* __restrict__ ss, int M, int nslab, int rb) {
  __shared__ double red[BN_FG][BN_FC][2];
  part += (size_t)blockIdx.y * nslab * p.C * 3;  // batch blockIdx.y of a grouped call
  ss += (size_t)blockIdx.y * 2 * p.C;
  p.stats += (size_t)blockIdx.y * p.stats_gstride;
  const int lc = threadIdx.x & (BN_FC - 1), grp = threadIdx.x / BN_FC;
  const int c = blockIdx.x * BN_FC + lc;
  const bool ok = c < p.C;
  const double ref = ok ? (double)part[(size_t)c * 3] : 0.0;  // slab 0's shift: common origin of the moments
  double s1 = 0.0, s2 = 0.0;
  if (ok) {
    for (int s0 = grp; s0 < nslab; s0 += BN_FG * 4) {
      float o0[4], o1[4], o2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // 12 independent loads in flight
        const int s = min(s0 + u * BN_FG, nslab - 1);
        const float* o = part + ((size_t)s * p.C + c) * 3;
        o0[u] = o[0]; o1[u] = o[1]; o2[u] = o[2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * BN_FG;
        if (s < nslab) {
          const double nb = (double)min(rb, M - s * rb);
          const double d = (double)o0[u] - ref, a1 = (double)o1[u], a2 = (double)o2[u];
          // sum (x - ref) and sum (x - ref)^2 of this slab from its own shifted sums
          s1 += a1 + nb * d;
          s2 += a2 + 2.0 * d * a1 + nb * d * d;
        }
      }
    }
  }
  red[grp][lc][0] = s1;
  red[grp][lc][1] = s2;
  __syncthreads();
  if (grp != 0 || !ok) return;
  s1 = 0.0; s2 = 0.0;
#pragma unroll 8
  for (int g = 0; g < BN_FG; ++g) { s1 += red[g][lc][0]; s2 += red[g][lc][1]; }
  const double n = (double)M;
  const double dm = s1 / n;
  const double mean = ref + dm;
  const double m2 = fmax(s2 - s1 * dm, 0.0);
  const double var = m2 / n;
  p.stats[c] = (float)mean;
  p.stats[p.C + c] = (float)var;
  const double unb = (n > 1.0) ? m2 / (n - 1.0) : var;
  if (p.defer_running) {
    p.stats[2 * p.C + c] = (float)unb;  // mvp_bn_running_update applies it later, in batch order
  } else if (p.running_mean) {
    p.running_mean[c] = bn_running(p.running_mean[c], (float)mean, p.momentum);
    p.running_var[c] = bn_running(p.running_var[c], (float)unb, p.momentum);
  }
  const float rstd = rsqrtf((float)var + p.eps);
  const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
  ss[c] = rstd * g;
  ss[p.C + c] = b - (float)mean * rstd * g;
  if (c == 0 && p.num_batches_tracked && !p.defer_running) *p.num_batches_tracked += 1;  // nn.BatchNorm train-mode bookkeeping, no extra launch
}

__global__ __launch_bounds__(256) void bn_running_update_kernel(const mvp_bn_running_update_args p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
  if (c >= p.C) return;
  p.running_mean[c] = bn_running(p.running_mean[c], p.stats[c], p.momentum);
  p.running_var[c] = bn_running(p.running_var[c], p.stats[2 * p.C + c], p.momentum);
}

struct bn_running_update_set { mvp_bn_running_update_args item[MVP_BN_RUNNING_MAX]; };
__global__ __launch_bounds__(256) void bn_running_update_n_kernel(const bn_running_update_set s) {
  const mvp_bn_running_update_args& p = s.item[blockIdx.y];  // wave-uniform: scalar loads from the kernel-argument segment
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && p.num_batches_tracked) *p.num_batches_tracked += 1;
  if (c >= p.C) return;
  p.running_mean[c] = bn_running(p.running_mean[c], p.stats[c], p.momentum);
  p.running_var[c] = bn_running(p.running_var[c], p.stats[2 * p.C + c], p.momentum);
}

// eval (running stats) / identity modes: scale & shift without a statistics pass.
__global__ __launch_bounds__(256) void bn_prep_kernel(const mvp_bn_tokens_args p, float* __restrict__ ss) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.C) return;
  ss += (size_t)blockIdx.y * 2 * p.C;  // (every batch of a grouped call gets the same eval-mode / identity scale and shift)
  if (p.mode == 2) {
    ss[c] = 1.f; ss[p.C + c] = 0.f;
    return;
  }
  const float rstd = rsqrtf(p.running_var[c] + p.eps);
  const float g = p.gamma ? p.gamma[c] : 1.f, b = p.beta ? p.beta[c] : 0.f;
  ss[c] = rstd * g;
  ss[p.C + c] = b - p.running_mean[c] * rstd * g;
}

// Pass 3: y = x*scale + shift on the spatial tokens; 32-token x 64-channel tiles go through
// LDS so that both the token-major reads and the NCHW / transposed writes are coalesced.
__global__ __launch_bounds__(256) void bn_apply_kernel(mvp_bn_tokens_args p, const float* __restrict__ ss) {
  __shared__ float tile[32][65];
  if (blockIdx.z) {  // batch blockIdx.z of a grouped call: its rows, its scale / shift, its outputs
    const size_t g = blockIdx.z;
    p.x += g * p.B * p.N * p.C;
    ss += g * 2 * p.C;
    if (p.nchw) p.nchw += g * p.nchw_gstride;
    if (p.tok_hi) p.tok_hi += g * p.tok_gstride;
    if (p.tok_lo) p.tok_lo += g * p.tok_gstride;
    if (p.cls_out) p.cls_out += g * p.cls_gstride;
  }
  const int t = threadIdx.x;
  const int ptiles = (p.hw + 31) / 32;
  const int b = blockIdx.x / ptiles, pt = blockIdx.x - b * ptiles;
  const int p0 = pt * 32, c0 = blockIdx.y * 64;
  const int tok0 = p.N - p.hw;  // leading (CLS / register) tokens are dropped
  {
    const int c = c0 + (t & 63);
    const float sc = (c < p.C) ? ss[c] : 0.f, sh = (c < p.C) ? ss[p.C + c] : 0.f;
    if (p.cls_out && pt == 0 && t < 64 && c < p.C) p.cls_out[(size_t)b * p.C + c] = p.x[(size_t)b * p.N * p.C + c] * sc + sh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tp = i * 4 + (t >> 6), pp = p0 + tp;
      float y = 0.f;
      if (pp < p.hw && c < p.C) {
        y = p.x[((size_t)b * p.N + tok0 + pp) * p.C + c] * sc + sh;
        if (p.tok_hi) {
          uint16_t h, l;
          split_bf16(y, h, l);
          const size_t o = ((size_t)b * p.hw + pp) * p.ld_tok + p.col_off + c;
          p.tok_hi[o] = h;
          if (p.tok_lo) p.tok_lo[o] = l;
        }
      }
      tile[tp][t & 63] = y;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cc = i * 8 + (t >> 5), tp = t & 31;
    const int c = c0 + cc, pp = p0 + tp;
    if (c < p.C && pp < p.hw) {
      const float y = tile[tp][cc];
      if (p.nchw) p.nchw[((size_t)b * p.C + c) * p.hw + pp] = y;
      if (p.tokT_hi) {
        uint16_t h, l;
        split_bf16(y, h, l);
        const size_t o = (size_t)(p.col_off + c) * p.ldT + (size_t)b * p.hw + pp;
        p.tokT_hi[o] = h;
        if (p.tokT_lo) p.tokT_lo[o] = l;
      }
    }
  }
}

// ----------------------------------------------------------------------------- pack NCHW
__global__ __launch_bounds__(256) void pack_nchw_kernel(const mvp_pack_nchw_args p) {
  __shared__ float tile[64][33];
  const int t = threadIdx.x;
  const int ptiles = (p.hw + 31) / 32;
  const int b = blockIdx.x / ptiles, pt = blockIdx.x - b * ptiles;
  const int p0 = pt * 32, c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cc = i * 8 + (t >> 5), tp = t & 31;
    const int c = c0 + cc, pp = p0 + tp;
    float y = 0.f;
    if (c < p.C && pp < p.hw) {
      y = p.nchw[((size_t)b * p.C + c) * p.hw + pp];
      if (p.tokT_hi) {
        uint16_t h, l;
        split_bf16(y, h, l);
        const size_t o = (size_t)(p.col_off + c) * p.ldT + (size_t)b * p.hw + pp;
        p.tokT_hi[o] = h;
        if (p.tokT_lo) p.tokT_lo[o] = l;
      }
    }
    tile[cc][tp] = y;
  }
  __syncthreads();
  if (!p.tok_hi) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int tp = i * 4 + (t >> 6), cc = t & 63;
    const int c = c0 + cc, pp = p0 + tp;
    if (c < p.C && pp < p.hw) {
      uint16_t h, l;
      split_bf16(tile[cc][tp], h, l);
      const size_t o = ((size_t)b * p.hw + pp) * p.ld_tok + p.col_off + c;
      p.tok_hi[o] = h;
      if (p.tok_lo) p.tok_lo[o] = l;
    }
  }
}

inline int grid_for(int64_t work, int cap = 2048) {
  int64_t g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

extern "C" int mvp_split_bf16(const mvp_split_bf16_args* a, void* stream) {
  if (!a || !a->src || !a->hi || a->n <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(split_kernel, dim3(grid_for(a->n >> 2)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_patch_gather(const mvp_patch_gather_args* a, void* stream) {
  if (!a || !a->images || !a->out_hi) return MVP_EINVAL;
  if (a->P <= 0 || (a->P & 3) || a->B <= 0 || a->C <= 0) return MVP_EINVAL;
  if (a->gh * a->P < a->H + a->pad_top || a->gw * a->P < a->W + a->pad_left) return MVP_EINVAL;
  const int64_t total4 = (int64_t)a->B * a->gh * a->gw * ((a->C * a->P * a->P) >> 2);
  hipLaunchKernelGGL(patch_gather_kernel, dim3(grid_for(total4, 4096)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_cls_rows(const mvp_cls_rows_args* a, void* stream) {
  if (!a || !a->cls || !a->pos0 || !a->x || a->B <= 0 || a->C <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(cls_rows_kernel, dim3((a->B * a->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int64_t mvp_bn_tokens_workspace_bytes(int M, int C) {
  const int rb = bn_slab_rows(M);
  const int64_t nslab = (M + rb - 1) / rb;
  return (nslab * C * 3 + 2 * (int64_t)C) * 4;
}

extern "C" int mvp_bn_tokens_to_nchw_fwd(const mvp_bn_tokens_args* a, void* stream) {
  if (!a || !a->x || !a->workspace) return MVP_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->C <= 0 || a->hw <= 0 || a->hw > a->N) return MVP_EINVAL;
  const int M = a->B * a->N;
  const int G = a->groups > 1 ? a->groups : 1;
  if (a->workspace_bytes < G * mvp_bn_tokens_workspace_bytes(M, a->C)) return MVP_EINVAL;
  if (a->mode == 0 && !a->stats) return MVP_EINVAL;
  if (a->defer_running && a->mode != 0) return MVP_EINVAL;
  if (a->mode == 1 && (!a->running_mean || !a->running_var)) return MVP_EINVAL;
  if (G > 1) {  // per-batch statistics of a grouped call: running statistics only through the deferred update, no transposed packing
    if ((a->mode == 0 && !a->defer_running) || a->tokT_hi || G > 65535) return MVP_EINVAL;
    if (a->stats_gstride < 0 || a->nchw_gstride < 0 || a->tok_gstride < 0 || a->cls_gstride < 0) return MVP_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  const int rb = bn_slab_rows(M);
  const int nslab = (M + rb - 1) / rb;
  float* part = (float*)a->workspace;
  float* ss = part + (size_t)G * nslab * a->C * 3;  // [G][2 * C] scale / shift behind the partials of all batches
  if (a->mode == 0) {
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nslab, (a->C + 255) / 256, G), dim3(256), 0, s, a->x, part, M, a->C, rb);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((a->C + BN_FC - 1) / BN_FC, G), dim3(1024), 0, s, *a, part, ss, M, nslab, rb);
  } else {
    hipLaunchKernelGGL(bn_prep_kernel, dim3((a->C + 255) / 256, G), dim3(256), 0, s, *a, ss);
  }
  if (a->cls_out && a->N <= a->hw) return MVP_EINVAL;
  if (a->nchw || a->tok_hi || a->tokT_hi || a->cls_out) {
    const int ptiles = (a->hw + 31) / 32;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(a->B * ptiles, (a->C + 63) / 64, G), dim3(256), 0, s, *a, ss);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_bn_running_update(const mvp_bn_running_update_args* a, void* stream) {
  if (!a || !a->stats || !a->running_mean || !a->running_var || a->C <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((a->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_bn_running_update_n(const mvp_bn_running_update_args* items, int n, void* stream) {
  if (!items || n <= 0 || n > MVP_BN_RUNNING_MAX) return MVP_EINVAL;
  bn_running_update_set s;
  int cmax = 0;
  for (int i = 0; i < n; ++i) {
    const mvp_bn_running_update_args& a = items[i];
    if (!a.stats || !a.running_mean || !a.running_var || a.C <= 0) return MVP_EINVAL;
    for (int j = 0; j < i; ++j)  // two items on one module would race (and the order of their updates matters)
      if (items[j].running_mean == a.running_mean || items[j].running_var == a.running_var) return MVP_EINVAL;
    s.item[i] = a;
    cmax = a.C > cmax ? a.C : cmax;
  }
  for (int i = n; i < MVP_BN_RUNNING_MAX; ++i) s.item[i] = items[0];  // never indexed (grid.y = n)
  hipLaunchKernelGGL(bn_running_update_n_kernel, dim3((cmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, s);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_pack_nchw_tokens(const mvp_pack_nchw_args* a, void* stream) {
  if (!a || !a->nchw || (!a->tok_hi && !a->tokT_hi)) return MVP_EINVAL;
  if (a->B <= 0 || a->C <= 0 || a->hw <= 0) return MVP_EINVAL;
  const int ptiles = (a->hw + 31) / 32;
  hipLaunchKernelGGL(pack_nchw_kernel, dim3(a->B * ptiles, (a->C + 63) / 64), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
