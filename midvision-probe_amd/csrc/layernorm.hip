// LayerNorm forward, one wave64 per row, row held in registers (two-pass mean / variance,
// the same algorithm torch's CPU kernel uses up to summation order), output written as the
// bf16 pair that feeds the next MFMA GEMM.  HBM-bound: reads 4 B/elem, writes 2-4 B/elem.
#include "mvp_common.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane: C <= 64 * 4 * 8 = 2048

__global__ __launch_bounds__(256) void layernorm_kernel(const mvp_layernorm_args p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int nv = p.C >> 2;
  const float4* xr = (const float4*)(p.x + (size_t)row * p.C);
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)p.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
  const float4* g4 = (const float4*)p.gamma;
  const float4* b4 = (const float4*)p.beta;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = g4[c], b = b4[c];
      float y[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                    (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
      const size_t o = (size_t)row * p.C + c * 4;
      uint16_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_bf16(y[e], h[e], l[e]);
      *(u32x2_t*)(p.out_hi + o) = u32x2_t{pack2(h[0], h[1]), pack2(h[2], h[3])};
      if (p.out_lo) *(u32x2_t*)(p.out_lo + o) = u32x2_t{pack2(l[0], l[1]), pack2(l[2], l[3])};
      if (p.out_f32) *(float4*)(p.out_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
}

}  // namespace

extern "C" int mvp_layernorm_fwd(const mvp_layernorm_args* a, void* stream) {
  if (!a || !a->x || !a->gamma || !a->beta || !a->out_hi) return MVP_EINVAL;
  if (a->M <= 0 || a->C <= 0 || (a->C & 3) || a->C > 64 * 4 * LN_MAXV) return MVP_EINVAL;
  hipLaunchKernelGGL(layernorm_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
