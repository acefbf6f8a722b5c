// ResNet-50 stem in one kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels, BatchNorm folded) + ReLU + max-pool 3x3 / stride 2 /
// pad 1, straight from the NCHW fp32 image to the channels-last map the bottlenecks consume
// (reference: torchvision resnet50 conv1 / bn1 / relu / maxpool as driven by evals/models/dino_res50.py:38-44, 83-90).
//
// Why: the im2col form wrote a [B*240*240, 160] bf16-pair matrix (590 MB at B=16, 480^2) for a 17 GFLOP convolution, then a
// [B*240*240, 64] fp32 map (236 MB) only for the pool to read it back: 257 + 215 + 71 us of a 4.75 ms forward.  Here a
// workgroup owns a 3 x 8 tile of POOLED pixels: it stages the 19 x 39 x 3 input patch in LDS (fp32), builds the im2col
// fragments of the 7 x 17 conv pixels it needs on the fly (8 scattered ds_read_b32 per lane and k-step, split into a bf16
// pair in registers), runs them against the 64 x 160 weight image kept in LDS for the workgroup's whole life (persistent
// grid), writes bias + ReLU results to an LDS tile and pools from there.  HBM traffic = the image once + the pooled map once.
//
// MFMA 16x16x32 bf16, "swapped" operands as in gemm.hip (A = weight fragment, B = pixel fragment): a lane's 4 accumulators
// are 4 consecutive channels of one conv pixel.  8 waves = 8 m-tiles of 16 conv pixels (119 used of 128).
#include "mvp_common.h"

namespace {

constexpr int ST_C0 = 64;             // output channels
constexpr int ST_K = 160;             // 7*7*3 = 147 padded to 5 k-steps of 32
constexpr int ST_PTH = 3, ST_PTW = 8; // pooled tile
constexpr int ST_CR = 2 * ST_PTH + 1, ST_CC = 2 * ST_PTW + 1;  // conv pixels per tile: 7 x 17 = 119
constexpr int ST_PH = 2 * ST_CR + 5, ST_PW = 2 * ST_CC + 5;    // input patch: 19 x 39
constexpr int ST_PWS = 40;            // patch row stride (floats)
constexpr int ST_WROW = 168;          // weight row stride in LDS (bf16): 336 B -> the 16 rows of a fragment read hit distinct banks
constexpr int ST_OROW = 68;           // conv-tile row stride (floats)
constexpr int ST_NW = 8;

template <int SPLIT>
__global__ __launch_bounds__(ST_NW * 64) void stem_kernel(const mvp_stem_args p, const int tiles_y, const int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W_BYTES = ST_C0 * ST_WROW * 2;
  uint16_t* w_hi = (uint16_t*)smem;
  uint16_t* w_lo = (uint16_t*)(smem + W_BYTES);
  float* patch = (float*)(smem + (SPLIT == 3 ? 2 : 1) * W_BYTES);
  float* ctile = patch + 3 * ST_PH * ST_PWS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int Ho = (p.H - 1) / 2 + 1, Wo = (p.W - 1) / 2 + 1;
  const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;

  // weights -> LDS once per workgroup (global layout [64][160] bf16, K contiguous)
  for (int i = tid; i < ST_C0 * (ST_K / 8); i += ST_NW * 64) {
    const int n = i / (ST_K / 8), c8 = (i - n * (ST_K / 8)) * 8;
    *(u32x4_t*)(w_hi + n * ST_WROW + c8) = *(const u32x4_t*)(p.w_hi + n * ST_K + c8);
    if (SPLIT == 3) *(u32x4_t*)(w_lo + n * ST_WROW + c8) = *(const u32x4_t*)(p.w_lo + n * ST_K + c8);
  }
  // this lane's im2col offsets: fragment element e of k-step ks is k = ks*32 + fq*8 + e -> (c, ky, kx) -> patch offset
  int koff[5][8];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 32 + fq * 8 + e;
      const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
      koff[ks][e] = (k < 147) ? (c * ST_PH + ky) * ST_PWS + kx : -1;
    }
  // this lane's conv pixel inside the tile (m-tile = wave)
  const int pix = wave * 16 + frow;
  const bool pix_ok = pix < ST_CR * ST_CC;
  const int pr = pix_ok ? pix / ST_CC : 0, pc = pix_ok ? pix - pr * ST_CC : 0;
  const int pbase = (2 * pr) * ST_PWS + 2 * pc;
  float bias4[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias4[nt][e] = p.bias ? p.bias[nt * 16 + fq * 4 + e] : 0.f;

  const int tiles_per_img = tiles_y * tiles_x;
  const int total = p.B * tiles_per_img;
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int b = t / tiles_per_img, tt = t - b * tiles_per_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int py0 = ty * ST_PTH, px0 = tx * ST_PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // conv origin of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;  // input origin of the patch
    __syncthreads();  // previous tile's pooling has finished reading ctile / nobody still reads the patch
    // ---- stage the input patch (zero outside the image)
    const float* img = p.images + (size_t)b * 3 * p.H * p.W;
    for (int i = tid; i < 3 * ST_PH * ST_PWS; i += ST_NW * 64) {
      const int c = i / (ST_PH * ST_PWS), r = (i - c * ST_PH * ST_PWS) / ST_PWS, x = i - (c * ST_PH + r) * ST_PWS;
      const int yy = iy0 + r, xx = ix0 + x;
      float v = 0.f;
      if (x < ST_PW && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) v = img[((size_t)c * p.H + yy) * p.W + xx];
      patch[i] = v;
    }
    __syncthreads();
    // ---- 16 conv pixels x 64 channels per wave: 5 k-steps
    f32x4_t acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (koff[ks][e] >= 0) ? patch[pbase + koff[ks][e]] : 0.f;
      uint32_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2_bf16(v[2 * e], v[2 * e + 1], h[e], l[e]);
      const bf16x8_t a_hi = __builtin_bit_cast(bf16x8_t, u32x4_t{h[0], h[1], h[2], h[3]});
      const bf16x8_t a_lo = __builtin_bit_cast(bf16x8_t, u32x4_t{l[0], l[1], l[2], l[3]});
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int wo = (nt * 16 + frow) * ST_WROW + ks * 32 + fq * 8;
        const bf16x8_t wf_hi = *(const bf16x8_t*)(w_hi + wo);
        if (SPLIT == 3) {
          const bf16x8_t wf_lo = *(const bf16x8_t*)(w_lo + wo);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_lo, a_hi, acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_hi, a_lo, acc[nt], 0, 0, 0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf_hi, a_hi, acc[nt], 0, 0, 0);
      }
    }
    // ---- bias + ReLU -> LDS conv tile; conv pixels outside the image count as 0 (every pool window holds a real pixel and
    // ReLU outputs are >= 0, so 0 is as good as the max-pool's -inf padding)
    {
      const int cy = cy0 + pr, cx = cx0 + pc;
      const bool in_img = pix_ok && (unsigned)cy < (unsigned)Ho && (unsigned)cx < (unsigned)Wo;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = in_img ? fmaxf(acc[nt][e] + bias4[nt][e], 0.f) : 0.f;
        *(f32x4_t*)(ctile + pix * ST_OROW + nt * 16 + fq * 4) = o;
      }
    }
    __syncthreads();
    // ---- 3x3 / 2 max-pool of the tile: 24 pooled pixels x 16 channel quads
    for (int i = tid; i < ST_PTH * ST_PTW * 16; i += ST_NW * 64) {
      const int c4 = (i & 15) * 4, q = i >> 4;
      const int qy = q / ST_PTW, qx = q - qy * ST_PTW;
      const int py = py0 + qy, px = px0 + qx;
      if (py >= Hp || px >= Wp) continue;
      f32x4_t m = *(const f32x4_t*)(ctile + ((2 * qy) * ST_CC + 2 * qx) * ST_OROW + c4);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x4_t v = *(const f32x4_t*)(ctile + ((2 * qy + dy) * ST_CC + 2 * qx + dx) * ST_OROW + c4);
          m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
        }
      const size_t o = (((size_t)b * Hp + py) * Wp + px) * ST_C0 + c4;
      if (p.out_f32) *(float4*)(p.out_f32 + o) = make_float4(m[0], m[1], m[2], m[3]);
      if (p.out_hi) {
        uint32_t h01, l01, h23, l23;
        split2_bf16(m[0], m[1], h01, l01);
        split2_bf16(m[2], m[3], h23, l23);
        *(u32x2_t*)(p.out_hi + o) = u32x2_t{h01, h23};
        if (p.out_lo) *(u32x2_t*)(p.out_lo + o) = u32x2_t{l01, l23};
      }
    }
  }
}

template <int SPLIT>
int launch_stem(const mvp_stem_args* a, hipStream_t s) {
  constexpr int SMEM = (SPLIT == 3 ? 2 : 1) * ST_C0 * ST_WROW * 2 + 3 * ST_PH * ST_PWS * 4 + ST_NW * 16 * ST_OROW * 4;
  static int configured = (int)hipFuncSetAttribute((const void*)stem_kernel<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  if (configured != 0) return MVP_ELAUNCH;
  const int Ho = (a->H - 1) / 2 + 1, Wo = (a->W - 1) / 2 + 1;
  const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;
  const int ty = (Hp + ST_PTH - 1) / ST_PTH, tx = (Wp + ST_PTW - 1) / ST_PTW;
  const long total = (long)a->B * ty * tx;
  const int grid = (int)(total < 256 ? total : 256);  // persistent, one workgroup per CU (87 KB of LDS): the weight image is loaded once per workgroup
  hipLaunchKernelGGL((stem_kernel<SPLIT>), dim3(grid), dim3(ST_NW * 64), SMEM, s, *a, ty, tx);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_stem7x7_pool(const mvp_stem_args* a, void* stream) {
  if (!a || !a->images || !a->w_hi || (!a->out_f32 && !a->out_hi) || a->B <= 0 || a->H < 7 || a->W < 7) return MVP_EINVAL;
  if (a->precision == MVP_PREC_BF16X3) {
    if (!a->w_lo) return MVP_EINVAL;
    return launch_stem<3>(a, (hipStream_t)stream);
  }
  if (a->precision != MVP_PREC_BF16) return MVP_EINVAL;
  return launch_stem<1>(a, (hipStream_t)stream);
}
