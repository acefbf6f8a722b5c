// Flash-style multi-head self-attention forward for gfx950, head_dim = 64.
//
// Work decomposition: grid = (ceil(N/128) query blocks, B*H); a workgroup = 4 waves, each
// wave owns 32 query rows (two 16-column MFMA tiles) and sweeps the keys in 64-key tiles
// that the 4 waves stage cooperatively into LDS by LDS-DMA (global_load_lds, 16 B/lane),
// double-buffered.
//
// MFMA orientation ("swapped", so nothing crosses lanes between the two products):
//   Sᵀ[key, q]  = K · Qᵀ      A operand = K fragment (ds_read_b128 from the swizzled K tile),
//                             B operand = Q fragment (registers, loaded once per wave)
//     -> the accumulator has q on the lane (col = lane & 15) and keys on the registers, so
//        the online-softmax row statistics are lane-local plus two shuffles (xor 16, 32).
//   Oᵀ[d, q]    = Vᵀ · Pᵀ     B operand = P, taken from the Sᵀ accumulator registers as they
//                             stand (k-slot j<4 -> key 4g+j of the even 16-key sub-tile,
//                             j>=4 -> the odd sub-tile); A operand = Vᵀ fragment in the SAME
//                             permuted k order, produced by two ds_read_b64_tr_b16
//                             (hardware-transposed LDS reads) from the row-major V tile.
//     -> the output accumulator again has q on the lane: the 1/l normalisation and the
//        running-max rescale are per-lane scalars, and each lane stores 4 consecutive d.
//
// LDS images (128-B rows = one key's 64 d):  K: 16-B chunk c of row r at chunk c ^ (r & 7)
// (conflict-free ds_read_b128);  V: chunk c at c ^ (((r >> 1) & 3) << 1) (conflict-free
// transposed reads: 8 consecutive rows x 32 B cover all 64 banks).  Both swizzles are applied
// to the per-lane global SOURCE address of the LDS-DMA (its LDS destination is lane-linear).
//
// SPLIT == 3: Q/K/V/P are bf16 pairs, each product runs hi*hi + hi*lo + lo*hi.
#include "mvp_common.h"

namespace {

template <int SPLIT>
__global__ __launch_bounds__(256) void attention_kernel(const mvp_attention_args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int TILE = 64 * 128;             // one 64-key x 64-d bf16 tile
  constexpr int STAGE = 2 * NARR * TILE;     // K (hi[,lo]) then V (hi[,lo])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int N = p.N;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int nkt = (N + 63) >> 6;
  const int HD = p.H * 64;
  const size_t rowbase = (size_t)b * N;
  const int g = lane >> 4, c16 = lane & 15;

  // ---- Q fragments (B operand): lane holds Q[q = tile*16 + (lane&15)][d = ks*32 + 8g .. +7]
  bf16x8_t q_hi[2][2], q_lo[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = min(q0 + qt * 16 + c16, N - 1);
    const size_t off = (rowbase + qrow) * p.ld_qkv + h * 64 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      q_hi[qt][ks] = *(const bf16x8_t*)(p.qkv_hi + off + ks * 32);
      if (SPLIT == 3) q_lo[qt][ks] = *(const bf16x8_t*)(p.qkv_lo + off + ks * 32);
    }
  }

  // ---- staging: each wave moves 2 pieces (8 rows x 128 B) of every array per tile
  const int rsub = lane >> 3;
  const int kc = ((lane & 7) ^ rsub) << 3;                      // K source chunk (elements)
  const int vc = ((lane & 7) ^ (((lane >> 4) & 3) << 1)) << 3;  // V source chunk
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int r = ps * 32 + wave * 8;
      const int krow = min(kt * 64 + r + rsub, N - 1);
      const size_t ro = (rowbase + krow) * p.ld_qkv + h * 64;
      __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_hi + ro + HD + kc), LDS_PTR(base + r * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_hi + ro + 2 * HD + vc), LDS_PTR(base + NARR * TILE + r * 128), 16, 0, 0);
      if (SPLIT == 3) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_lo + ro + HD + kc), LDS_PTR(base + TILE + r * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_lo + ro + 2 * HD + vc), LDS_PTR(base + 3 * TILE + r * 128), 16, 0, 0);
      }
    }
  };

  f32x4_t o_acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) o_acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-1e30f, -1e30f};
  float l_run[2] = {0.f, 0.f};
  const float cs = p.scale * 1.44269504088896340736f;  // softmax(x*scale) via exp2
  const bool active = q0 < N;                           // wave-uniform

  // transposed-read lane constants: group g supplies rows 4g + ((lane&15)>>2), 8-B slot lane&3
  const int tr_q = c16 >> 2, tr_p = lane & 3;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    if (active) {
      const char* kb = smem + (kt & 1) * STAGE;
      const char* vb = kb + NARR * TILE;
      // ---------------- Sᵀ = K · Qᵀ over the 64-key tile
      f32x4_t s[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) s[t][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = (((ks << 2) + g) ^ (lane & 7)) << 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const char* ka = kb + (t * 16 + c16) * 128 + coff;
          const bf16x8_t k_hi = *(const bf16x8_t*)ka;
          if (SPLIT == 3) {
            const bf16x8_t k_lo = *(const bf16x8_t*)(ka + TILE);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_lo, q_hi[qt][ks], s[t][qt], 0, 0, 0);
              s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_lo[qt][ks], s[t][qt], 0, 0, 0);
            }
          }
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_hi[qt][ks], s[t][qt], 0, 0, 0);
        }
      }
      // ---------------- online softmax (q on the lane; keys on registers + lane groups)
      const int kbase = kt * 64 + g * 4;
      bf16x8_t p_hi[2][2], p_lo[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float tmax = -1e30f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool valid = (kbase + t * 16 + j) < N;
            s[t][qt][j] = valid ? s[t][qt][j] : -1e30f;
            tmax = fmaxf(tmax, s[t][qt][j]);
          }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run[qt], tmax);
        const float alpha = exp2f((m_run[qt] - m_new) * cs);
        const float mc = m_new * cs;
        m_run[qt] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float pv = exp2f(s[t][qt][j] * cs - mc);
            psum += pv;
            const __bf16 ph = (__bf16)pv;
            p_hi[qt][t >> 1][(t & 1) * 4 + j] = ph;
            if (SPLIT == 3) p_lo[qt][t >> 1][(t & 1) * 4 + j] = (__bf16)(pv - (float)ph);
          }
        l_run[qt] = l_run[qt] * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o_acc[dt][qt] *= alpha;
      }
      // ---------------- Oᵀ += Vᵀ · Pᵀ
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row = ks * 32 + g * 4 + tr_q;  // +16 for the second read
        const int fsw = ((row >> 1) & 3) << 1;   // identical for row and row + 16
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int off = row * 128 + ((((dt << 1) + (tr_p >> 1)) ^ fsw) << 4) + ((tr_p & 1) << 3);
          const bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + off));
          const bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + off + 16 * 128));
          const bf16x8_t v_hi = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          if (SPLIT == 3) {
            const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off));
            const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off + 16 * 128));
            const bf16x8_t v_lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo, p_hi[qt][ks], o_acc[dt][qt], 0, 0, 0);
              o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_lo[qt][ks], o_acc[dt][qt], 0, 0, 0);
            }
          }
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_hi[qt][ks], o_acc[dt][qt], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = q0 + qt * 16 + c16;
    if (qrow >= N) continue;
    const size_t ob = (rowbase + qrow) * p.ld_out + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint16_t hh[4], ll[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_bf16(o_acc[dt][qt][j] * inv, hh[j], ll[j]);
      *(u32x2_t*)(p.out_hi + ob + dt * 16) = u32x2_t{pack2(hh[0], hh[1]), pack2(hh[2], hh[3])};
      if (p.out_lo) *(u32x2_t*)(p.out_lo + ob + dt * 16) = u32x2_t{pack2(ll[0], ll[1]), pack2(ll[2], ll[3])};
    }
  }
}

template <int SPLIT>
int launch_attention(const mvp_attention_args* a, hipStream_t s) {
  constexpr int SMEM = 2 * 2 * ((SPLIT == 3) ? 2 : 1) * 64 * 128;
  static int configured = [] {
    return (int)hipFuncSetAttribute((const void*)attention_kernel<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }();
  if (configured != 0) return MVP_ELAUNCH;
  dim3 grid((a->N + 127) / 128, a->B * a->H);
  hipLaunchKernelGGL((attention_kernel<SPLIT>), grid, dim3(256), SMEM, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_attention_fwd(const mvp_attention_args* a, void* stream) {
  if (!a || !a->qkv_hi || !a->out_hi) return MVP_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->H <= 0) return MVP_EINVAL;
  if ((a->ld_qkv & 7) || (a->ld_out & 3) || a->ld_qkv < 3 * a->H * 64 || a->ld_out < a->H * 64) return MVP_EINVAL;
  if (a->precision == MVP_PREC_BF16X3) {
    if (!a->qkv_lo || !a->out_lo) return MVP_EINVAL;
    return launch_attention<3>(a, (hipStream_t)stream);
  }
  if (a->precision != MVP_PREC_BF16) return MVP_EINVAL;
  return launch_attention<1>(a, (hipStream_t)stream);
}
