// LayerNorm forward, one wave64 per row, row held in registers (two-pass mean / variance,
// the same algorithm torch's CPU kernel uses up to summation order), output written as the
// bf16 pair that feeds the next MFMA GEMM.  HBM-bound: reads 4 B/elem, writes 2-4 B/elem.
#include "mvp_common.h"

#ifndef MVP_LN_NT
#define MVP_LN_NT 0
#endif
#ifndef MVP_LN_LD_NT  // the x loads non-temporal (A/B builds)
#define MVP_LN_LD_NT 0
#endif
namespace {

constexpr int LN_MAXV = 4;  // 8-element chunks per lane: C <= 64 * 8 * 4 = 2048

// Each lane owns chunks of 8 consecutive elements (two float4 loads, ONE 16-byte store per output array: 8-byte
// stores run at 0.5-0.7x the 16-byte rate on gfx950).
__global__ __launch_bounds__(256) void layernorm_kernel(const mvp_layernorm_args p) {
  f16_saturate_mode();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nv = p.C >> 3;
  const float4* xr = (const float4*)(p.x + (size_t)row * p.C);
  float4 v[LN_MAXV][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
#if MVP_LN_LD_NT
      typedef __attribute__((ext_vector_type(4))) float f4v;
      const f4v t0 = __builtin_nontemporal_load((const f4v*)&xr[c * 2]), t1 = __builtin_nontemporal_load((const f4v*)&xr[c * 2 + 1]);
      v[i][0] = make_float4(t0[0], t0[1], t0[2], t0[3]);
      v[i][1] = make_float4(t1[0], t1[1], t1[2], t1[3]);
#else
      v[i][0] = xr[c * 2];
      v[i][1] = xr[c * 2 + 1];
#endif
      s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
  }
  const float mean = wave_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float a = v[i][h].x - mean, b = v[i][h].y - mean, cc = v[i][h].z - mean, d = v[i][h].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
  const float4* g4 = (const float4*)p.gamma;
  const float4* b4 = (const float4*)p.beta;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float y[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 g = g4[c * 2 + h], b = b4[c * 2 + h];
        y[h * 4 + 0] = (v[i][h].x - mean) * rstd * g.x + b.x;
        y[h * 4 + 1] = (v[i][h].y - mean) * rstd * g.y + b.y;
        y[h * 4 + 2] = (v[i][h].z - mean) * rstd * g.z + b.z;
        y[h * 4 + 3] = (v[i][h].w - mean) * rstd * g.w + b.w;
      }
      const size_t o = (size_t)row * p.C + c * 8;
      uint32_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (p.out_f16) split2_f16_comp(y[2 * e], y[2 * e + 1], h[e], l[e]);
        else split2_bf16(y[2 * e], y[2 * e + 1], h[e], l[e]);
      }
      if (p.out_layout == MVP_PAIR_A_ILV32) {  // one array, hi | lo interleaved per 32 columns (an 8-element chunk never straddles a block)
        const size_t oi = (size_t)row * 2 * p.C + ilv32_col(c * 8);
        store_out((u32x4_t*)(p.out_hi + oi), u32x4_t{h[0], h[1], h[2], h[3]}, MVP_LN_NT);
        store_out((u32x4_t*)(p.out_hi + oi + 32), u32x4_t{l[0], l[1], l[2], l[3]}, MVP_LN_NT);
      } else {
        store_out((u32x4_t*)(p.out_hi + o), u32x4_t{h[0], h[1], h[2], h[3]}, MVP_LN_NT);
        if (p.out_lo) store_out((u32x4_t*)(p.out_lo + o), u32x4_t{l[0], l[1], l[2], l[3]}, MVP_LN_NT);
      }
      if (p.out_f32) {
        *(float4*)(p.out_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
        *(float4*)(p.out_f32 + o + 4) = make_float4(y[4], y[5], y[6], y[7]);
      }
    }
  }
}

}  // namespace

extern "C" int mvp_layernorm_fwd(const mvp_layernorm_args* a, void* stream) {
  if (!a || !a->x || !a->gamma || !a->beta || !a->out_hi) return MVP_EINVAL;
  if (a->M <= 0 || a->C <= 0 || (a->C & 7) || a->C > 64 * 8 * LN_MAXV) return MVP_EINVAL;
  if (a->out_layout != MVP_PAIR_SEPARATE && (a->out_layout != MVP_PAIR_A_ILV32 || (a->C & 31))) return MVP_EINVAL;
  hipLaunchKernelGGL(layernorm_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
