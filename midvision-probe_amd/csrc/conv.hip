// Convolution support kernels for the DPT / ResNet path (channels-last bf16 pairs):
//   conv_weight_pack   torch [Cout, Cin, kh, kw] fp32 -> GEMM operand [rows, K] bf16 pair
//                      mode 0 (forward):        rows = Cout, k = (t)*Cin + c
//                      mode 1 (data gradient):  rows = Cin,  k = (t')*Cout + n with t' the flipped tap
//   upsample_nearest_cl fwd / bwd (integer factor) on [B, H, W, C] fp32 (+ bf16 pair copy)
//   gemm_tn_conv       weight gradient  dW[i, (t, c)] = sum_m G[m, i] * X[pix(m, t), c]
//                      both operands are pixel-major (K = pixels), staged by LDS-DMA and consumed
//                      through ds_read_b64_tr_b16; split-K over pixel ranges into fp32 partial slabs
//   splitk_reduce      sum of the slabs (fixed order) -> torch-layout gradient (optionally +=)
#include "mvp_common.h"

namespace {

// ----------------------------------------------------------------------------- weight pack
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const mvp_conv_weight_pack_args p) {
  const int T = p.kh * p.kw;
  const int64_t total = (int64_t)p.Cout * p.Cin * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // destination-major indexing so that writes are coalesced
    float v;
    if (p.mode == 0) {
      const int c = (int)(i % p.Cin);
      const int64_t r = i / p.Cin;
      const int t = (int)(r % T), n = (int)(r / T);
      v = p.w[((int64_t)n * p.Cin + c) * T + t];
    } else {
      const int n = (int)(i % p.Cout);
      const int64_t r = i / p.Cout;
      const int t = (int)(r % T), c = (int)(r / T);
      v = p.w[((int64_t)n * p.Cin + c) * T + (T - 1 - t)];
    }
    uint16_t h, l;
    split_bf16(v, h, l);
    p.out_hi[i] = h;
    if (p.out_lo) p.out_lo[i] = l;
  }
}

// ----------------------------------------------------------------------------- nearest upsample (channels-last)
__global__ __launch_bounds__(256) void upsample_nearest_cl_fwd(const mvp_upsample_cl_args p) {
  const int C4 = p.C >> 2, Ho = p.H * p.f, Wo = p.W * p.f;
  const int64_t total = (int64_t)p.B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int64_t b = r / Ho;
    const float4 v = ((const float4*)p.src)[((b * p.H + y / p.f) * p.W + x / p.f) * C4 + c];
    if (p.dst_f32) ((float4*)p.dst_f32)[i] = v;
    if (p.dst_hi) {
      uint16_t h[4], l[4];
      split_bf16(v.x, h[0], l[0]); split_bf16(v.y, h[1], l[1]); split_bf16(v.z, h[2], l[2]); split_bf16(v.w, h[3], l[3]);
      ((u32x2_t*)p.dst_hi)[i] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (p.dst_lo) ((u32x2_t*)p.dst_lo)[i] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
  }
}

// adjoint: dst[b, y, x, c] = sum over the f x f block of src (src is the fine grid)
__global__ __launch_bounds__(256) void upsample_nearest_cl_bwd(const mvp_upsample_cl_args p) {
  const int C4 = p.C >> 2, Wo = p.W * p.f, Ho = p.H * p.f;
  const int64_t total = (int64_t)p.B * p.H * p.W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int x = (int)(r % p.W); r /= p.W;
    const int y = (int)(r % p.H);
    const int64_t b = r / p.H;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < p.f; ++dy)
      for (int dx = 0; dx < p.f; ++dx) {
        const float4 v = ((const float4*)p.src)[((b * Ho + y * p.f + dy) * Wo + x * p.f + dx) * C4 + c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    if (p.dst_f32) ((float4*)p.dst_f32)[i] = a;
    if (p.dst_hi) {
      uint16_t h[4], l[4];
      split_bf16(a.x, h[0], l[0]); split_bf16(a.y, h[1], l[1]); split_bf16(a.z, h[2], l[2]); split_bf16(a.w, h[3], l[3]);
      ((u32x2_t*)p.dst_hi)[i] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (p.dst_lo) ((u32x2_t*)p.dst_lo)[i] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
  }
}

// ----------------------------------------------------------------------------- backward of "nearest x f, then 3x3 conv" at the COARSE grid
// y = conv3x3(up_f(x), W) (pad 1): the upsampled input is constant over f x f blocks, so both gradients fold onto the coarse grid.
// With g = dL/dy on the fine grid and, per coarse pixel (i, j) and tap (ky, kx), the box sum
//     G[(i,j), tap, co] = sum of g[p, co] over the fine pixels p with p + (ky-1, kx-1) inside block (i, j)   (an f x f box, shifted),
// dW[co, ci, tap] = sum_(i,j) G[(i,j), tap, co] * x[(i,j), ci]  and  dx[(i,j), ci] = sum_(tap, co) W[co, ci, tap] * G[(i,j), tap, co]:
// one TN GEMM and one NT GEMM over B*H*W coarse pixels with K = 9*C instead of two convolutions over f^2 as many fine pixels (16x fewer
// MFMA flops at f = 4: the DPT probe's out_conv, probes.py:384-398).  This kernel makes G (bf16 pair, column tap*C + c): one thread
// = one coarse pixel x 4 channels, the (f+2)^2 window of g read once per thread (rows left to right, then top to bottom: fixed order).
template <int F>
__global__ __launch_bounds__(256) void upconv3_boxsum_kernel(const mvp_upconv_boxsum_args p) {
  const int C4 = p.C >> 2, Hf = p.H * F, Wf = p.W * F;
  const int64_t total = (int64_t)p.B * p.H * p.W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int x = (int)(r % p.W); r /= p.W;
    const int y = (int)(r % p.H);
    const int64_t b = r / p.H;
    float4 G[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int d = 0; d < 3; ++d) G[a][d] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ry = 0; ry < F + 2; ++ry) {
      const int Y = y * F - 1 + ry;
      if (Y < 0 || Y >= Hf) continue;
      float4 v[F + 2];
#pragma unroll
      for (int rx = 0; rx < F + 2; ++rx) {
        const int X = x * F - 1 + rx;
        v[rx] = (X >= 0 && X < Wf) ? ((const float4*)p.g)[((b * Hf + Y) * Wf + X) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float4 R[3];  // row sums for kx = 0, 1, 2 (dx = -1, 0, +1): window columns [1 - dx, F - dx]
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rx = 2 - kx; rx < 2 - kx + F; ++rx) { a.x += v[rx].x; a.y += v[rx].y; a.z += v[rx].z; a.w += v[rx].w; }
        R[kx] = a;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
        if (ry >= 2 - ky && ry < 2 - ky + F) {  // window rows [1 - dy, F - dy]
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { G[ky][kx].x += R[kx].x; G[ky][kx].y += R[kx].y; G[ky][kx].z += R[kx].z; G[ky][kx].w += R[kx].w; }
        }
    }
    const int64_t row = (b * p.H + y) * p.W + x;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 a = G[ky][kx];
        uint16_t h[4], l[4];
        split_bf16(a.x, h[0], l[0]); split_bf16(a.y, h[1], l[1]); split_bf16(a.z, h[2], l[2]); split_bf16(a.w, h[3], l[3]);
        const int64_t o = (row * 9 + ky * 3 + kx) * C4 + c;
        ((u32x2_t*)p.out_hi)[o] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
        if (p.out_lo) ((u32x2_t*)p.out_lo)[o] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
      }
  }
}

// ----------------------------------------------------------------------------- forward of "nearest x f, then 3x3 conv" from COARSE tap products
// y[p] = b + sum_tap W_tap · x[cell(p + d_tap)]: the per-tap products T[(i,j), tap, :] = W_tap · x[(i,j)] are ONE GEMM over the coarse
// pixels (N = 9*C); this kernel sums, per fine pixel, the (at most 9) tap products of the coarse cells its taps fall into (taps outside
// the image are the zero padding), adds the bias, applies ReLU and writes what the convolution's epilogue would: bf16 pair, gate
// mask, fp32.  One thread = one fine pixel x 4 channels; taps in (ky, kx) order (fixed).  9x less MFMA work than the convolution over
// the fine pixels at f = 4 (1/16 of the rows, 9x the columns), paid with 9 L2-resident reads per output.
template <int F>
__global__ __launch_bounds__(256) void upconv3_gather_kernel(const mvp_upconv_gather_args p) {
  const int C4 = p.C >> 2, Hf = p.H * F, Wf = p.W * F;
  const int64_t total = (int64_t)p.B * Hf * Wf * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int X = (int)(r % Wf); r /= Wf;
    const int Y = (int)(r % Hf);
    const int64_t b = r / Hf;
    float4 a = p.bias ? ((const float4*)p.bias)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = Y + ky - 1;
      if (yy < 0 || yy >= Hf) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = X + kx - 1;
        if (xx < 0 || xx >= Wf) continue;
        const float4 v = ((const float4*)p.t)[(((b * p.H + yy / F) * p.W + xx / F) * 9 + ky * 3 + kx) * C4 + c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    if (p.act == MVP_ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    if (p.out_mask)
      ((uint32_t*)p.out_mask)[i] = (a.x > 0.f ? 1u : 0u) | (a.y > 0.f ? 0x100u : 0u) | (a.z > 0.f ? 0x10000u : 0u) | (a.w > 0.f ? 0x1000000u : 0u);
    if (p.out_f32) ((float4*)p.out_f32)[i] = a;
    if (p.out_hi) {
      uint16_t h[4], l[4];
      split_bf16(a.x, h[0], l[0]); split_bf16(a.y, h[1], l[1]); split_bf16(a.z, h[2], l[2]); split_bf16(a.w, h[3], l[3]);
      ((u32x2_t*)p.out_hi)[i] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (p.out_lo) ((u32x2_t*)p.out_lo)[i] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
  }
}

// ----------------------------------------------------------------------------- im2col (NCHW fp32 image -> GEMM rows)
// For convs whose Cin is not a multiple of 32 (the 7x7/2 RGB stem): row m = (b, yo, xo),
// col k = (ky*kw + kx)*C + c, zero padded up to ldk columns.
// One thread = 8 consecutive k of one output row (one 16-byte store per output array; the per-row index arithmetic is
// shared by the 8 elements).
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const mvp_im2col_args p) {
  const int Kc = p.kh * p.kw * p.C;
  const int k8 = p.ldk >> 3;
  const int64_t total = (int64_t)p.B * p.Ho * p.Wo * k8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / k8;
    const int k0 = (int)(i - m * k8) << 3;
    const int xo = (int)(m % p.Wo);
    const int64_t r = m / p.Wo;
    const int yo = (int)(r % p.Ho);
    const int64_t b = r / p.Ho;
    const int ybase = yo * p.stride - p.pad, xbase = xo * p.stride - p.pad;
    const float* img = p.src + b * p.C * p.H * p.W;
    uint16_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      float v = 0.f;
      if (k < Kc) {
        const int t = k / p.C, c = k - t * p.C, ky = t / p.kw, kx = t - ky * p.kw;
        const int y = ybase + ky, x = xbase + kx;
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) v = img[((size_t)c * p.H + y) * p.W + x];
      }
      split_bf16(v, h[e], l[e]);
    }
    const size_t o = (size_t)m * p.ldk + k0;
    *(u32x4_t*)(p.out_hi + o) = u32x4_t{pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])};
    if (p.out_lo) *(u32x4_t*)(p.out_lo + o) = u32x4_t{pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])};
  }
}

// ----------------------------------------------------------------------------- max-pool (channels-last)
__global__ __launch_bounds__(256) void maxpool_cl_kernel(const mvp_maxpool_cl_args p) {
  const int C4 = p.C >> 2;
  const int64_t total = (int64_t)p.B * p.Ho * p.Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int xo = (int)(r % p.Wo); r /= p.Wo;
    const int yo = (int)(r % p.Ho);
    const int64_t b = r / p.Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < p.k; ++ky)
      for (int kx = 0; kx < p.k; ++kx) {
        const int y = yo * p.stride + ky - p.pad, x = xo * p.stride + kx - p.pad;
        if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) continue;
        const float4 v = ((const float4*)p.src)[((b * p.H + y) * p.W + x) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    if (p.dst_f32) ((float4*)p.dst_f32)[i] = m;
    if (p.dst_hi) {
      uint16_t h[4], l[4];
      split_bf16(m.x, h[0], l[0]); split_bf16(m.y, h[1], l[1]); split_bf16(m.z, h[2], l[2]); split_bf16(m.w, h[3], l[3]);
      ((u32x2_t*)p.dst_hi)[i] = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (p.dst_lo) ((u32x2_t*)p.dst_lo)[i] = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
    }
  }
}

// ----------------------------------------------------------------------------- TN weight-gradient GEMM
// Tile: 128 output channels (i) x 128 input channels (j) of ONE tap; K-tile = 32 pixels.
// LDS images [32 pixel rows][128 channels] (256-B rows), 16-B chunk c of row r stored at
// chunk c ^ (f(r) << 1), f(r) = (r & 3) | ((r >> 1) & 4): the 8 rows a half-wave touches in one
// ds_read_b64_tr_b16 (rows 8g..8g+3 of two lane groups) land on 8 distinct 32-B bank slots.
// MFMA roles: A operand = X fragment (rows = input channel j), B operand = G fragment
// (cols = output channel i): each lane then owns 4 consecutive j of one i -> 16-byte stores
// into dW's [i][tap][j] rows.
// NW waves per workgroup: 4 (2x2 wave grid, 64x64 per wave) or 8 (4x2, 32(j) x 64(i) per wave: same tile and LDS, more
// waves per SIMD to hide the staging latency).
// KT = pixels per K-tile (32 or 64): 64 halves the barriers and LDS-DMA waits per MFMA (one stage = 64 KB in bf16x3).
template <int SPLIT, int NST, int NW = 4, int KT = 32>
__global__ __launch_bounds__(NW * 64) void gemm_tn_conv_kernel(const mvp_gemm_tn_args p) {
  constexpr int RPW = KT / NW;   // pixel rows of a K-tile staged by one wave
  constexpr int WJ = 256 / NW;   // j (input-channel) extent of a wave tile: 64 or 32
  constexpr int JT = WJ / 16;    // 16-wide j sub-tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int TILE = KT * 256;             // one [KT][128] bf16 image
  constexpr int STAGE = 2 * NARR * TILE;     // G (hi[,lo]) then X (hi[,lo])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, c16 = lane & 15;

  const int T = p.kh * p.kw;
  const int jt_per_tap = p.Cin / 128;
  const int tiles_j = T * jt_per_tap;
  const int tiles_i = (p.Cout + 127) / 128;
  int bid = blockIdx.x;
  const int split = bid / (tiles_i * tiles_j);
  bid -= split * tiles_i * tiles_j;
  const int ti = bid / tiles_j, tj = bid - ti * tiles_j;
  const int tap = tj / jt_per_tap, c0 = (tj - tap * jt_per_tap) * 128, i0 = ti * 128;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  // pixel range of this split (multiple of KT)
  const int64_t per = ((p.M + p.splits - 1) / p.splits + KT - 1) / KT * KT;
  const int64_t mbeg = (int64_t)split * per, mend = min<int64_t>(p.M, mbeg + per);
  const int nk = (mend > mbeg) ? (int)((mend - mbeg + KT - 1) / KT) : 0;
  const int Hs = p.H >> p.up, Ws = p.W >> p.up;

  // staging: a piece = 4 rows x 256 B; wave w stages rows w*8 .. w*8+7 (two pieces)
  const int rsub = lane >> 4;                      // row inside a piece (0..3)
  auto fsw = [](int r) { return (r & 3) | ((r >> 1) & 4); };
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int ps = 0; ps < RPW / 4; ++ps) {
      const int r = wave * RPW + ps * 4 + rsub;    // tile row (pixel)
      const int64_t m = mbeg + (int64_t)kt * KT + r;
      const int cs = (c16 ^ (fsw(r) << 1)) << 3;   // swizzled source chunk (elements)
      const bool inr = m < mend;
      // G row (buffer-form LDS-DMA: rows past the split's range and padding taps use an out-of-range offset -> zeros)
      constexpr int OOB = 0x7fffff80;
      constexpr unsigned NREC = 0x7fffff00u;
      const int goff = inr ? (int)((m * p.ldg + i0 + cs) * 2) : OOB;
      lds_dma16(p.g_hi, NREC, base + (wave * RPW + ps * 4) * 256, goff, 0);
      if (SPLIT == 3) lds_dma16(p.g_lo, NREC, base + TILE + (wave * RPW + ps * 4) * 256, goff, 0);
      // X row of the tap's source pixel
      int xoff = OOB;
      if (inr) {
        const int x = (int)(m % p.Wo);
        const int64_t t2 = m / p.Wo;
        const int y = (int)(t2 % p.Ho);
        const int64_t b = t2 / p.Ho;
        const int yy = y * p.stride + ky - p.pad, xx = x * p.stride + kx - p.pad;
        if (((unsigned)yy < (unsigned)p.H) && ((unsigned)xx < (unsigned)p.W))
          xoff = (int)((((b * Hs + (yy >> p.up)) * Ws + (xx >> p.up)) * p.ldx + c0 + cs) * 2);
      }
      lds_dma16(p.x_hi, NREC, base + NARR * TILE + (wave * RPW + ps * 4) * 256, xoff, 0);
      if (SPLIT == 3) lds_dma16(p.x_lo, NREC, base + 3 * TILE + (wave * RPW + ps * 4) * 256, xoff, 0);
    }
  };

  // wave tile 64 (j) x 64 (i): wave>>1 picks the j half, wave&1 the i half
  const int wj0 = (wave >> 1) * WJ, wi0 = (wave & 1) * 64;
  f32x4_t acc[JT][4];  // [jt][it]
#pragma unroll
  for (int a = 0; a < JT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: group g4 supplies rows 8*g4 + q (+4 for the second read), 8-B slot pp
  const int q = c16 >> 2, pp = lane & 3;
  auto tr_addr = [&](int row, int col0) {  // col0 multiple of 16 (elements): 32-B chunk index = col0/16
    const int c32 = col0 >> 4;
    return row * 256 + ((c32 ^ fsw(row)) << 5) + ((pp >> 1) << 4) + ((pp & 1) << 3);
  };

  auto mma_tile = [&](const char* gb) {
    const char* xb = gb + NARR * TILE;
#pragma unroll
    for (int ks = 0; ks < KT / 32; ++ks) {
    const int r0 = ks * 32 + g4 * 8 + q;
    bf16x8_t xf_hi[JT], xf_lo[JT], gf_hi[4], gf_lo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < JT) {
        const int a0 = tr_addr(r0, wj0 + t * 16), a1 = tr_addr(r0 + 4, wj0 + t * 16);
        const bf16x4_t u0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(xb + a0));
        const bf16x4_t u1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(xb + a1));
        xf_hi[t] = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (SPLIT == 3) {
          const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(xb + TILE + a0));
          const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(xb + TILE + a1));
          xf_lo[t] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      {
        const int a0 = tr_addr(r0, wi0 + t * 16), a1 = tr_addr(r0 + 4, wi0 + t * 16);
        const bf16x4_t u0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(gb + a0));
        const bf16x4_t u1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(gb + a1));
        gf_hi[t] = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
        if (SPLIT == 3) {
          const bf16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(gb + TILE + a0));
          const bf16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(gb + TILE + a1));
          gf_lo[t] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < JT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (SPLIT == 3) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf_lo[a], gf_hi[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf_hi[a], gf_lo[b], acc[a][b], 0, 0, 0);
        }
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf_hi[a], gf_hi[b], acc[a][b], 0, 0, 0);
      }
    }
  };
  if (NST == 1) {
    // one LDS stage (32 KB): up to 4-5 workgroups resident per CU overlap each other's load and MFMA phases
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      stage(0, kt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mma_tile(smem);
    }
  } else {
    if (nk > 0) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      mma_tile(smem + (kt & 1) * STAGE);
    }
  }

  // partial slab [split][Cout][T*Cin]: lane owns i = col (lane & 15), 4 consecutive j (rows 4*g4 ..)
  float* slab = p.partial + (size_t)split * p.Cout * T * p.Cin;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int i = i0 + wi0 + b * 16 + c16;
    if (i >= p.Cout) continue;
#pragma unroll
    for (int a = 0; a < JT; ++a) {
      const int j = c0 + wj0 + a * 16 + g4 * 4;
      *(float4*)(slab + ((size_t)i * T + tap) * p.Cin + j) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    }
  }
}

// slabs [splits][Cout][T][Cin] -> torch layout [Cout][Cin][T] (+= when accumulate)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const mvp_gemm_tn_args p) {
  const int T = p.kh * p.kw;
  const int64_t total = (int64_t)p.Cout * p.Cin * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // source-major (coalesced reads): i = (n*T + t)*Cin + c
    const int c = (int)(i % p.Cin);
    const int64_t r = i / p.Cin;
    const int t = (int)(r % T), n = (int)(r / T);
    float s = 0.f;
    for (int k = 0; k < p.splits; ++k) s += p.partial[(size_t)k * total + i];
    float* d = p.dw + ((int64_t)n * p.Cin + c) * T + t;
    *d = p.accumulate ? *d + s : s;
  }
}

inline int grid_for(int64_t work, int cap = 4096) {
  int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <int SPLIT, int NST, int NW = 4, int KT = 32>
int launch_tn(const mvp_gemm_tn_args* a, hipStream_t s) {
  constexpr int SMEM = NST * 2 * ((SPLIT == 3) ? 2 : 1) * KT * 256;
  static int configured = [] {
    return (int)hipFuncSetAttribute((const void*)gemm_tn_conv_kernel<SPLIT, NST, NW, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }();
  if (configured != 0) return MVP_ELAUNCH;
  const int T = a->kh * a->kw;
  const int blocks = a->splits * ((a->Cout + 127) / 128) * T * (a->Cin / 128);
  hipLaunchKernelGGL((gemm_tn_conv_kernel<SPLIT, NST, NW, KT>), dim3(blocks), dim3(NW * 64), SMEM, s, *a);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for((int64_t)a->Cout * a->Cin * T)), dim3(256), 0, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_conv_weight_pack(const mvp_conv_weight_pack_args* a, void* stream) {
  if (!a || !a->w || !a->out_hi || a->Cout <= 0 || a->Cin <= 0 || a->kh <= 0 || a->kw <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid_for((int64_t)a->Cout * a->Cin * a->kh * a->kw)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_upsample_nearest_cl(const mvp_upsample_cl_args* a, void* stream) {
  if (!a || !a->src || (!a->dst_f32 && !a->dst_hi) || a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 3) || a->f < 1) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (a->backward)
    hipLaunchKernelGGL(upsample_nearest_cl_bwd, dim3(grid_for((int64_t)a->B * a->H * a->W * (a->C >> 2), 8192)), dim3(256), 0, s, *a);
  else
    hipLaunchKernelGGL(upsample_nearest_cl_fwd, dim3(grid_for((int64_t)a->B * a->H * a->W * a->f * a->f * (a->C >> 2), 8192)), dim3(256), 0, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_upconv3_fwd_gather(const mvp_upconv_gather_args* a, void* stream) {
  if (!a || !a->t || (!a->out_hi && !a->out_f32)) return MVP_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 3) || (a->f != 2 && a->f != 4)) return MVP_EINVAL;
  if (a->act != MVP_ACT_NONE && a->act != MVP_ACT_RELU) return MVP_EINVAL;
  const int64_t total = (int64_t)a->B * a->H * a->W * a->f * a->f * (a->C >> 2);
  if (a->f == 4)
    hipLaunchKernelGGL(upconv3_gather_kernel<4>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(upconv3_gather_kernel<2>, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_upconv3_grad_boxsum(const mvp_upconv_boxsum_args* a, void* stream) {
  if (!a || !a->g || !a->out_hi) return MVP_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->C <= 0 || (a->C & 3) || (a->f != 2 && a->f != 4)) return MVP_EINVAL;
  const int64_t total = (int64_t)a->B * a->H * a->W * (a->C >> 2);
  if (a->f == 4)
    hipLaunchKernelGGL(upconv3_boxsum_kernel<4>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(upconv3_boxsum_kernel<2>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_im2col_nchw(const mvp_im2col_args* a, void* stream) {
  if (!a || !a->src || !a->out_hi || a->B <= 0 || a->C <= 0 || a->kh <= 0 || a->kw <= 0 || a->stride <= 0) return MVP_EINVAL;
  if (a->ldk < a->kh * a->kw * a->C || (a->ldk & 7) || a->Ho <= 0 || a->Wo <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(grid_for((int64_t)a->B * a->Ho * a->Wo * (a->ldk >> 3), 16384)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int mvp_maxpool_cl(const mvp_maxpool_cl_args* a, void* stream) {
  if (!a || !a->src || (!a->dst_f32 && !a->dst_hi) || a->B <= 0 || a->C <= 0 || (a->C & 3) || a->k <= 0 || a->stride <= 0) return MVP_EINVAL;
  hipLaunchKernelGGL(maxpool_cl_kernel, dim3(grid_for((int64_t)a->B * a->Ho * a->Wo * (a->C >> 2), 8192)), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

extern "C" int64_t mvp_gemm_tn_workspace_bytes(int Cout, int Cin, int kh, int kw, int splits) {
  return (int64_t)splits * Cout * Cin * kh * kw * 4;
}

extern "C" int mvp_gemm_tn_conv(const mvp_gemm_tn_args* a, void* stream) {
  if (!a || !a->g_hi || !a->x_hi || !a->partial || !a->dw || !a->zero_page) return MVP_EINVAL;
  if (a->Cout <= 0 || a->Cin <= 0 || (a->Cin & 127) || a->M <= 0 || a->splits < 1) return MVP_EINVAL;
  if (a->kh <= 0 || a->kw <= 0 || a->stride <= 0 || a->Ho <= 0 || a->Wo <= 0 || (a->M % ((int64_t)a->Ho * a->Wo))) return MVP_EINVAL;
  if ((a->ldg & 7) || (a->ldx & 7) || a->ldx < a->Cin || a->ldg < ((a->Cout + 127) / 128) * 128) return MVP_EINVAL;
  // 32-bit byte offsets in the buffer-form staging: both operand arrays must stay below 2 GiB
  if (a->M * a->ldg * 2 >= 0x7fffff00ll || (a->M / ((int64_t)a->Ho * a->Wo)) * (a->H >> a->up) * (a->W >> a->up) * a->ldx * 2 >= 0x7fffff00ll) return MVP_EINVAL;
  if (a->precision == MVP_PREC_BF16X3) {
    if (!a->g_lo || !a->x_lo) return MVP_EINVAL;
    // one LDS stage (+1.3 % over two on the DPT step) and 8 waves on the 128x128 tile (+5 %: 618 -> 649 img/s);
    // 64-pixel K-tiles: half the barriers / LDS-DMA waits per MFMA of the 32-pixel form (tools/tn_bench.py, DPT conv shapes: +2...13 %)
#ifndef MVP_TN_KT
#define MVP_TN_KT 64
#endif
#ifndef MVP_TN_NST
#define MVP_TN_NST 1  // LDS stages (diagnostic builds: tools/tn_bench.py --variants)
#endif
    return launch_tn<3, MVP_TN_NST, 8, MVP_TN_KT>(a, (hipStream_t)stream);
  }
  if (a->precision != MVP_PREC_BF16) return MVP_EINVAL;
  return launch_tn<1, 2>(a, (hipStream_t)stream);
}
