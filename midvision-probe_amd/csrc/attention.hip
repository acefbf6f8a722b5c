// Flash-style multi-head self-attention forward for gfx950, head_dim = 64.
//
// Two kernels share one per-tile routine (attn_tile):
//   attention_stream_kernel  any N: grid = (ceil(N/128) query blocks, B*H); a workgroup = 4 waves, each wave owns
//                            32 query rows and sweeps the keys in 64-key tiles staged cooperatively into LDS by
//                            LDS-DMA (global_load_lds, 16 B/lane), double-buffered.
//   attention_resident_kernel N <= 256 (the 224^2 / 197-token case): grid = B*H; a workgroup = 8 waves = all
//                            query rows of one (batch, head); ALL keys and values are staged once (<= 128 KB):
//                            tile 0 + Q, barrier, the remaining tiles land under tile 0's compute, one more
//                            barrier, then every wave runs its tiles back to back.  K/V are read from L2 once
//                            per head instead of once per query block.
//
// MFMA orientation ("swapped", so nothing crosses lanes between the two products):
//   S^T[key, q] = K . Q^T     A operand = K fragment (ds_read_b128 from the swizzled K tile),
//                             B operand = Q fragment (registers, loaded once per wave)
//     -> the accumulator has q on the lane (col = lane & 15) and keys on the registers, so the online-softmax
//        row statistics are lane-local plus two shuffles (xor 16, 32).
//   O^T[d, q]   = V^T . P^T   B operand = P, taken from the S^T accumulator registers as they stand (k-slot j<4 ->
//                             key 4g+j of the even 16-key sub-tile, j>=4 -> the odd sub-tile); A operand = V^T
//                             fragment in the SAME permuted k order, produced by two ds_read_b64_tr_b16
//                             (hardware-transposed LDS reads) from the row-major V tile.
//     -> the output accumulator again has q on the lane: the 1/l normalisation and the running-max rescale are
//        per-lane scalars, and each lane stores 4 consecutive d.
//
// LDS images (128-B rows = one key's 64 d):  K: 16-B chunk c of row r at chunk c ^ (r & 7) (conflict-free
// ds_read_b128);  V: chunk c at c ^ (((r >> 1) & 3) << 1) (conflict-free transposed reads: 8 consecutive rows x
// 32 B cover all 64 banks).  Both swizzles are applied to the per-lane global SOURCE address of the LDS-DMA (its
// LDS destination is lane-linear).
//
// SPLIT == 3: Q/K/V/P are bf16 pairs, each product runs hi*hi + hi*lo + lo*hi.
// VF16 (SPLIT == 3 only; mvp_attention_args.v_format == MVP_ATT_V_F16, round 4): the kernels are VALU-bound — per 64-key tile and wave
// ~340 vector instructions against 96 MFMAs, 3 of the ~7 per score spent splitting the probability into a bf16 pair.  A probability
// lies in [0, 2^6] and needs no exponent range: ONE fp16 value carries it to 2^-12 relative.  So the V third of qkv arrives as
// hi = fp16(v), lo = bf16(v - hi) (written that way by the qkv GEMM's epilogue, mvp_gemm_args.out_f16_col0) and
//   O^T += v_hi . p16 (v_mfma_f32_16x16x32_f16)  +  v_lo . bf16(p) (the bf16 MFMA; p's rounding to 8 bits meets a factor already 2^-12 small)
// — two products instead of three, one conversion per product instead of the split.  (lo stays bf16: an fp16 lo of a small v would
// be a denormal.)  The running maximum is then only an exponent offset, so it is moved — and O, l rescaled — only when a row's
// maximum rises by more than 2^6 above it (always at a pair's first tile): probabilities stay below 2^6 = 64, far inside fp16.
#include "mvp_common.h"

#ifndef MVP_ATT_NT
#define MVP_ATT_NT 0
#endif
#ifndef MVP_ATT_LD_AUX  // cache policy of the K / V LDS-DMA loads (2 = nt): A/B builds, profiles/r04_store_policy.txt
#define MVP_ATT_LD_AUX 0
#endif
namespace {

// Diagnostic knobs (tools/attn_bench.py builds variants): MVP_ATT_PRIO 1 = s_setprio 1 around the two MFMA clusters of a tile (a wave in
// its matrix phase is then not starved of issue slots by its SIMD partner's softmax VALU stream); MVP_ATT_STAGGER n = waves 4-7 of the
// resident kernel sleep n x 64 cycles after every barrier (SIMD partners out of phase: one's MFMAs beside the other's VALU).
#ifndef MVP_ATT_PRIO
#define MVP_ATT_PRIO 0
#endif
#ifndef MVP_ATT_STAGGER
#define MVP_ATT_STAGGER 0
#endif
#if MVP_ATT_PRIO
#define ATT_PRIO(v) __builtin_amdgcn_s_setprio(v)
#else
#define ATT_PRIO(v)
#endif
#ifndef MVP_ATT_ABLATE
#define MVP_ATT_ABLATE 0
#endif
constexpr int TILE = 64 * 128;  // one 64-key x 64-d bf16 tile
constexpr float ATT_DEFER = 6.0f;  // VF16: the running maximum moves only when a row's maximum rises by more than this (exp2 units)

template <int SPLIT>
struct AttnState {
  bf16x8_t q_hi[2][2], q_lo[2][2];
  f32x4_t o_acc[4][2];
  float m_run[2], l_run[2];
};

// One 64-key tile for one wave (32 query rows).  kb / vb: LDS bases of the K and V images of this tile
// (lo images follow at +TILE when SPLIT == 3).  last: whether keys beyond N must be masked.
// The kernels are VALU-bound (per 64-key tile and wave: ~1100 vector instructions against 96 MFMAs), so the softmax
// body is kept lean: raw v_exp_f32 (__builtin_amdgcn_exp2f: no denormal range fix-up, 5 instructions fewer per
// element; probabilities below 2^-126 are zero either way) and the key-validity mask compiled only into the LAST
// tile's instantiation.
template <int SPLIT, bool LAST, int VF16 = 0>
__device__ __forceinline__ void attn_tile(AttnState<SPLIT>& st, const char* kb, const char* vb, int key0, int N, float cs, int lane) {
  static_assert(!VF16 || SPLIT == 3, "the fp16-probability form belongs to the bf16x3 mode");
  const int g = lane >> 4, c16 = lane & 15;
  // LAST tile: only the first nsub 16-key sub-tiles hold real keys (N = 197: 5 keys of the 4th tile -> nsub = 1); the
  // products of the all-padding sub-tiles are skipped (wave-uniform branches) and their probabilities set to zero.
  const int nsub = LAST ? min(4, (N - key0 + 15) >> 4) : 4;
  f32x4_t s[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) s[t][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  ATT_PRIO(1);
#pragma unroll
  for (int ks = 0; ks < (MVP_ATT_ABLATE == 3 ? 0 : 2); ++ks) {
    const int coff = (((ks << 2) + g) ^ (lane & 7)) << 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (LAST && t >= nsub) continue;
      const char* ka = kb + (t * 16 + c16) * 128 + coff;
      const bf16x8_t k_hi = *(const bf16x8_t*)ka;
      if (VF16 == 2) {
        // MVP_ATT_V_F16_QK_F16: Q is the compensated activation pair, K the compensated weight-side pair of MVP_PREC_F16X2 (both written
        // by the qkv GEMM's epilogue): S^T = k_lo . q_lo + k_hi . q_hi on the f16 MFMA — two products instead of three, in the order
        // of the GEMM kernels; the sum is q . k + (q - hi_q) . 64 d_k (~2^-18 relative per term: mvp_common.h, split2_f16_comp).
        const f16x8_t k_lo = *(const f16x8_t*)(ka + TILE);
        const f16x8_t k_h = __builtin_bit_cast(f16x8_t, k_hi);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_lo, __builtin_bit_cast(f16x8_t, st.q_lo[qt][ks]), s[t][qt], 0, 0, 0);
          s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_h, __builtin_bit_cast(f16x8_t, st.q_hi[qt][ks]), s[t][qt], 0, 0, 0);
        }
        continue;
      }
      if (SPLIT == 3) {
        const bf16x8_t k_lo = *(const bf16x8_t*)(ka + TILE);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_lo, st.q_hi[qt][ks], s[t][qt], 0, 0, 0);
          s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, st.q_lo[qt][ks], s[t][qt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        s[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, st.q_hi[qt][ks], s[t][qt], 0, 0, 0);
    }
  }
  ATT_PRIO(0);
  // ---------------- online softmax (q on the lane; keys on registers + lane groups)
  const int kbase = key0 + g * 4;
  bf16x8_t p_hi[2][2], p_lo[2][2];  // VF16: p_lo = bf16(p) (the partner of v_lo), p_h = fp16(p)
  f16x8_t p_h[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (VF16 && MVP_ATT_ABLATE != 1) {
      float tmax = -1e30f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (LAST) s[t][qt][j] = ((kbase + t * 16 + j) < N) ? s[t][qt][j] : -1e30f;
          tmax = fmaxf(tmax, s[t][qt][j]);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      // move the running maximum only if some row of the wave needs it (wave-uniform branch; every lane of a row sees the same tmax)
      if (__builtin_amdgcn_ballot_w64((tmax - st.m_run[qt]) * cs > ATT_DEFER) != 0) {
        const float m_new = fmaxf(st.m_run[qt], tmax);
        const float alpha = __builtin_amdgcn_exp2f((st.m_run[qt] - m_new) * cs);
        st.m_run[qt] = m_new;
        st.l_run[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st.o_acc[dt][qt] *= alpha;
      }
      const float mc = st.m_run[qt] * cs;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pv = (LAST && t >= nsub) ? 0.f : __builtin_amdgcn_exp2f(s[t][qt][j] * cs - mc);
          psum += pv;
          p_h[qt][t >> 1][(t & 1) * 4 + j] = (_Float16)pv;
          p_lo[qt][t >> 1][(t & 1) * 4 + j] = (__bf16)pv;
        }
      st.l_run[qt] += psum;
      continue;
    }
    if (MVP_ATT_ABLATE == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          p_hi[qt][t >> 1][(t & 1) * 4 + j] = (__bf16)s[t][qt][j];
          if (SPLIT == 3) p_lo[qt][t >> 1][(t & 1) * 4 + j] = (__bf16)(s[t][qt][j] * cs);
        }
      st.l_run[qt] = 1.f;
      continue;
    }
    float tmax = -1e30f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (LAST) s[t][qt][j] = ((kbase + t * 16 + j) < N) ? s[t][qt][j] : -1e30f;
        tmax = fmaxf(tmax, s[t][qt][j]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(st.m_run[qt], tmax);
    const float alpha = __builtin_amdgcn_exp2f((st.m_run[qt] - m_new) * cs);
    const float mc = m_new * cs;
    st.m_run[qt] = m_new;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = (LAST && t >= nsub) ? 0.f : __builtin_amdgcn_exp2f(s[t][qt][j] * cs - mc);
        psum += pv;
        const __bf16 ph = (__bf16)pv;
        p_hi[qt][t >> 1][(t & 1) * 4 + j] = ph;
        if (SPLIT == 3) p_lo[qt][t >> 1][(t & 1) * 4 + j] = (__bf16)(pv - (float)ph);
      }
    st.l_run[qt] = st.l_run[qt] * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.o_acc[dt][qt] *= alpha;
  }
  // ---------------- O^T += V^T . P^T
  const int tr_q = c16 >> 2, tr_p = lane & 3;
  ATT_PRIO(1);
#pragma unroll
  for (int ks = 0; ks < (MVP_ATT_ABLATE == 2 ? 0 : 2); ++ks) {
    if (LAST && 2 * ks >= nsub) continue;    // both 16-key sub-tiles of this half are padding (P = 0)
    const int row = ks * 32 + g * 4 + tr_q;  // +16 for the second read
    const int fsw = ((row >> 1) & 3) << 1;   // identical for row and row + 16
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int off = row * 128 + ((((dt << 1) + (tr_p >> 1)) ^ fsw) << 4) + ((tr_p & 1) << 3);
      const bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + off));
      const bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + off + 16 * 128));
      const bf16x8_t v_hi = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      if (VF16) {  // v_hi holds fp16 bits, v_lo bf16: O^T += v_hi . p16 + v_lo . bf16(p)
        const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off));
        const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off + 16 * 128));
        const bf16x8_t v_lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        const f16x8_t v_h = __builtin_bit_cast(f16x8_t, v_hi);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          st.o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo, p_lo[qt][ks], st.o_acc[dt][qt], 0, 0, 0);
          st.o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_h, p_h[qt][ks], st.o_acc[dt][qt], 0, 0, 0);
        }
        continue;
      }
      if (SPLIT == 3) {
        const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off));
        const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(vb + TILE + off + 16 * 128));
        const bf16x8_t v_lo = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          st.o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo, p_hi[qt][ks], st.o_acc[dt][qt], 0, 0, 0);
          st.o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_lo[qt][ks], st.o_acc[dt][qt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
        st.o_acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_hi[qt][ks], st.o_acc[dt][qt], 0, 0, 0);
    }
  }
  ATT_PRIO(0);
}

template <int SPLIT>
__device__ __forceinline__ void attn_load_q(AttnState<SPLIT>& st, const mvp_attention_args& p, size_t rowbase, int q0, int h, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = min(q0 + qt * 16 + c16, p.N - 1);
    const size_t off = (rowbase + qrow) * p.ld_qkv + h * 64 + g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      st.q_hi[qt][ks] = *(const bf16x8_t*)(p.qkv_hi + off + ks * 32);
      if (SPLIT == 3) st.q_lo[qt][ks] = *(const bf16x8_t*)(p.qkv_lo + off + ks * 32);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) st.o_acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  st.m_run[0] = st.m_run[1] = -1e30f;
  st.l_run[0] = st.l_run[1] = 0.f;
}

template <int SPLIT>
__device__ __forceinline__ void attn_store(AttnState<SPLIT>& st, const mvp_attention_args& p, size_t rowbase, int q0, int h, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = st.l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = q0 + qt * 16 + c16;
    if (qrow >= p.N) continue;
    const bool ilv = p.out_layout == MVP_PAIR_A_ILV32;  // one array, hi | lo interleaved per 32 columns (column h*64 + dt*16 + g*4)
    const size_t ob = (rowbase + qrow) * p.ld_out + (ilv ? h * 128 : h * 64) + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint32_t h01, l01, h23, l23;
      if (p.out_f16) {  // the compensated fp16 pair: the activation operand of a two-product (MVP_PREC_F16X2) proj GEMM
        split2_f16_comp(st.o_acc[dt][qt][0] * inv, st.o_acc[dt][qt][1] * inv, h01, l01);
        split2_f16_comp(st.o_acc[dt][qt][2] * inv, st.o_acc[dt][qt][3] * inv, h23, l23);
      } else {
        split2_bf16(st.o_acc[dt][qt][0] * inv, st.o_acc[dt][qt][1] * inv, h01, l01);
        split2_bf16(st.o_acc[dt][qt][2] * inv, st.o_acc[dt][qt][3] * inv, h23, l23);
      }
      if (ilv) {
        const size_t o = ob + (dt >> 1) * 64 + (dt & 1) * 16;
        store_out((u32x2_t*)(p.out_hi + o), u32x2_t{h01, h23}, MVP_ATT_NT);
        store_out((u32x2_t*)(p.out_hi + o + 32), u32x2_t{l01, l23}, MVP_ATT_NT);
      } else {
        store_out((u32x2_t*)(p.out_hi + ob + dt * 16), u32x2_t{h01, h23}, MVP_ATT_NT);
        if (p.out_lo) store_out((u32x2_t*)(p.out_lo + ob + dt * 16), u32x2_t{l01, l23}, MVP_ATT_NT);
      }
    }
  }
}

// Stage the 8-row piece (rows r .. r+7 of the 64-key tile starting at key0) of K and V for this wave.
template <int SPLIT>
__device__ __forceinline__ void attn_stage_piece(const mvp_attention_args& p, size_t rowbase, int h, int key0, int r, char* kdst, char* vdst, int lane) {
  const int rsub = lane >> 3;
  const int kc = ((lane & 7) ^ rsub) << 3;                      // K source chunk (elements)
  const int vc = ((lane & 7) ^ (((lane >> 4) & 3) << 1)) << 3;  // V source chunk
  const int HD = p.H * 64;
  if (MVP_ATT_ABLATE == 4) return;
  const int krow = min(key0 + r + rsub, p.N - 1);
  const size_t ro = (rowbase + krow) * p.ld_qkv + h * 64;
  __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_hi + ro + HD + kc), LDS_PTR(kdst + r * 128), 16, 0, MVP_ATT_LD_AUX);
  __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_hi + ro + 2 * HD + vc), LDS_PTR(vdst + r * 128), 16, 0, MVP_ATT_LD_AUX);
  if (SPLIT == 3) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_lo + ro + HD + kc), LDS_PTR(kdst + TILE + r * 128), 16, 0, MVP_ATT_LD_AUX);
    __builtin_amdgcn_global_load_lds(GLB_PTR(p.qkv_lo + ro + 2 * HD + vc), LDS_PTR(vdst + TILE + r * 128), 16, 0, MVP_ATT_LD_AUX);
  }
}

template <int SPLIT, int VF16 = 0>
__global__ __launch_bounds__(256) void attention_stream_kernel(const mvp_attention_args p) {
  f16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int STAGE = 2 * NARR * TILE;  // K (hi[,lo]) then V (hi[,lo])
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int nkt = (p.N + 63) >> 6;
  const size_t rowbase = (size_t)b * p.N;
  AttnState<SPLIT> st;
  attn_load_q<SPLIT>(st, p, rowbase, q0, h, lane);
  const float cs = p.scale * 1.44269504088896340736f;  // softmax(x*scale) via exp2
  const bool active = q0 < p.N;                         // wave-uniform

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) attn_stage_piece<SPLIT>(p, rowbase, h, kt * 64, ps * 32 + wave * 8, base, base + NARR * TILE, lane);
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    if (active) {
      const char* kb = smem + (kt & 1) * STAGE;
      if (kt == nkt - 1) attn_tile<SPLIT, true, VF16>(st, kb, kb + NARR * TILE, kt * 64, p.N, cs, lane);
      else attn_tile<SPLIT, false, VF16>(st, kb, kb + NARR * TILE, kt * 64, p.N, cs, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (active) attn_store<SPLIT>(st, p, rowbase, q0, h, lane);
}

// N <= 256: all keys / values of one (batch, head) resident in LDS, 8 waves = 256 query rows.
//
// Persistent over (batch, head) pairs (round 3): a workgroup takes pairs blockIdx.x, blockIdx.x + gridDim.x, ... and keeps a ring of
// nkt + 1 tile slots in LDS, so that the NEXT pair's first K/V tile and its Q fragments are fetched while the current pair's tiles
// 1 .. nkt-1 are computed, and the current pair's output stores drain under the next pair's first tile.  Why: with one workgroup of
// 100+ KB per CU nothing overlapped a pair's memory phase — at B = 96 (1152 pairs, 4.5 per CU) the launch took 81 us of which 39 us
// were staging + Q loads + stores with no tile computed at all (tools/attn_bench.py, ablation 5): 232 MB at 6 TB/s, then the MFMAs.
//   slot(i, t) = (nkt * i + t) % (nkt + 1) for tile t of the workgroup's i-th pair: the slot pair i leaves unused is exactly pair
//   i+1's slot 0, and pair i+1's tiles t >= 1 fall into slots pair i has finished with when the pair-boundary barrier is passed.
// Per pair: [tile 0 resident, Q in registers] -> issue tiles 1.. (land under tile 0's compute) -> tile 0 -> vmcnt(0) + barrier ->
// issue next pair's tile 0 + Q -> tiles 1.. -> vmcnt(0) (the prefetch, issued long before) -> output stores -> barrier.
template <int SPLIT, int VF16 = 0>
__global__ __launch_bounds__(512) void attention_resident_kernel(const mvp_attention_args p) {
  f16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int STAGE = 2 * NARR * TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = wave * 32;
  const int nkt = (p.N + 63) >> 6, nslot = nkt + 1;
  const int npair = p.B * p.H;
  const float cs = p.scale * 1.44269504088896340736f;
  const bool active = q0 < p.N;  // wave-uniform
  const int g = lane >> 4, c16 = lane & 15;

  auto stage_tile = [&](int bh, int kt, int slot) {  // 8 waves x 8 rows = one 64-key tile
    const int b = bh / p.H, h = bh - b * p.H;
    char* base = smem + slot * STAGE;
    attn_stage_piece<SPLIT>(p, (size_t)b * p.N, h, kt * 64, wave * 8, base, base + NARR * TILE, lane);
  };
  bf16x8_t qn_hi[2][2], qn_lo[2][2];
  auto fetch_q = [&](int bh) {  // Q fragments of this wave's 32 query rows (global -> registers)
    const int b = bh / p.H, h = bh - b * p.H;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qrow = min(q0 + qt * 16 + c16, p.N - 1);
      const size_t off = ((size_t)b * p.N + qrow) * p.ld_qkv + h * 64 + g * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qn_hi[qt][ks] = *(const bf16x8_t*)(p.qkv_hi + off + ks * 32);
        if (SPLIT == 3) qn_lo[qt][ks] = *(const bf16x8_t*)(p.qkv_lo + off + ks * 32);
      }
    }
  };

  int bh = blockIdx.x;
  if (bh >= npair) return;
  stage_tile(bh, 0, 0);
  fetch_q(bh);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  AttnState<SPLIT> st;
  for (int i = 0; bh < npair; ++i, bh += gridDim.x) {
    const int b = bh / p.H, h = bh - b * p.H;
    const size_t rowbase = (size_t)b * p.N;
    const int s0 = (nkt * i) % nslot;  // slot of this pair's tile 0
    // this pair's remaining tiles: their slots were released at the barrier that ended the previous pair
    for (int kt = 1; kt < nkt; ++kt) stage_tile(bh, kt, (s0 + kt) % nslot);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        st.q_hi[qt][ks] = qn_hi[qt][ks];
        if (SPLIT == 3) st.q_lo[qt][ks] = qn_lo[qt][ks];
      }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) st.o_acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    st.m_run[0] = st.m_run[1] = -1e30f;
    st.l_run[0] = st.l_run[1] = 0.f;
    if (active && MVP_ATT_ABLATE != 5) {
      const char* kb = smem + s0 * STAGE;
      if (nkt == 1) attn_tile<SPLIT, true, VF16>(st, kb, kb + NARR * TILE, 0, p.N, cs, lane);
      else attn_tile<SPLIT, false, VF16>(st, kb, kb + NARR * TILE, 0, p.N, cs, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tiles 1.. landed (and the previous pair's stores have left)
    __syncthreads();
    if (MVP_ATT_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(MVP_ATT_STAGGER);
    const int nbh = bh + gridDim.x;
    if (nbh < npair) {  // the next pair's first tile into the one free slot, its Q into registers: under this pair's remaining tiles
      stage_tile(nbh, 0, (s0 + nkt) % nslot);
      fetch_q(nbh);
    }
    if (active) {
      for (int kt = 1; kt < (MVP_ATT_ABLATE == 5 ? 0 : nkt); ++kt) {
        const char* kb = smem + ((s0 + kt) % nslot) * STAGE;
        if (kt == nkt - 1) attn_tile<SPLIT, true, VF16>(st, kb, kb + NARR * TILE, kt * 64, p.N, cs, lane);
        else attn_tile<SPLIT, false, VF16>(st, kb, kb + NARR * TILE, kt * 64, p.N, cs, lane);
      }
    }
    // The prefetch was issued a whole pair's worth of tiles ago: this wait is free.  It comes BEFORE the output stores on purpose —
    // vmcnt retires in issue order, so a wait placed behind the stores would also wait for their round trip; issued after it, the
    // stores drain under the next pair's first tile (whose vmcnt(0) covers them) and the barrier below does not hold them back.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (active) attn_store<SPLIT>(st, p, rowbase, q0, h, lane);
    if (nbh < npair) __builtin_amdgcn_s_barrier();  // next pair: its tile 0 is resident for everybody; everybody is done with this pair's tiles
    if (MVP_ATT_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(MVP_ATT_STAGGER);
  }
}

template <int SPLIT, int VF16 = 0>
int launch_attention(const mvp_attention_args* a, hipStream_t s) {
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int STAGE = 2 * NARR * TILE;
  constexpr int SMEM_STREAM = 2 * STAGE, SMEM_RES = 5 * STAGE;  // resident kernel: ring of nkt + 1 <= 5 tile slots (160 KiB at bf16x3)
  static int configured = [] {
    int e = (int)hipFuncSetAttribute((const void*)attention_stream_kernel<SPLIT, VF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_STREAM);
    if (e == 0) e = (int)hipFuncSetAttribute((const void*)attention_resident_kernel<SPLIT, VF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_RES);
    return e;
  }();
  if (configured != 0) return MVP_ELAUNCH;
  if (a->N <= 256) {
    const int nkt = (a->N + 63) >> 6;
    // persistent: one workgroup per CU-sized share of LDS (ring of nkt + 1 slots), pairs dealt round-robin
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
      return n > 0 ? n : 256;
    }();
    const int per_cu = (160 * 1024) / ((nkt + 1) * STAGE) > 0 ? (160 * 1024) / ((nkt + 1) * STAGE) : 1;
    const int grid = a->B * a->H < cus * per_cu ? a->B * a->H : cus * per_cu;
    hipLaunchKernelGGL((attention_resident_kernel<SPLIT, VF16>), dim3(grid), dim3(512), (nkt + 1) * STAGE, s, *a);
  } else {
    dim3 grid((a->N + 127) / 128, a->B * a->H);
    hipLaunchKernelGGL((attention_stream_kernel<SPLIT, VF16>), grid, dim3(256), SMEM_STREAM, s, *a);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_attention_fwd(const mvp_attention_args* a, void* stream) {
  if (!a || !a->qkv_hi || !a->out_hi) return MVP_EINVAL;
  if (a->B <= 0 || a->N <= 0 || a->H <= 0) return MVP_EINVAL;
  if ((a->ld_qkv & 7) || (a->ld_out & 3) || a->ld_qkv < 3 * a->H * 64 || a->ld_out < a->H * 64) return MVP_EINVAL;
  if (a->out_layout != MVP_PAIR_SEPARATE && (a->out_layout != MVP_PAIR_A_ILV32 || a->precision != MVP_PREC_BF16X3 || a->ld_out < 2 * a->H * 64)) return MVP_EINVAL;
  if (a->precision == MVP_PREC_BF16X3) {
    if (!a->qkv_lo || (!a->out_lo && a->out_layout == MVP_PAIR_SEPARATE)) return MVP_EINVAL;
    if (a->v_format == MVP_ATT_V_F16) return launch_attention<3, 1>(a, (hipStream_t)stream);
    if (a->v_format == MVP_ATT_V_F16_QK_F16) return launch_attention<3, 2>(a, (hipStream_t)stream);
    if (a->v_format != MVP_ATT_V_BF16_PAIR) return MVP_EINVAL;
    return launch_attention<3>(a, (hipStream_t)stream);
  }
  if (a->precision != MVP_PREC_BF16 || a->v_format != MVP_ATT_V_BF16_PAIR || a->out_f16) return MVP_EINVAL;
  return launch_attention<1>(a, (hipStream_t)stream);
}
