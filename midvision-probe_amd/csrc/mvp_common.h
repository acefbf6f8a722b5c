// Shared device helpers for the gfx950 kernels (wave64, bf16 pair arithmetic, MFMA types).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mvp_hip.h"

#define MVP_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// fp32 -> bf16 round-to-nearest-even (v_cvt_pk_bf16_f32: NaN stays NaN).
__device__ __forceinline__ uint16_t f2bf(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(uint16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
// Split v into a bf16 pair hi + lo (lo = rne(v - hi)); |v - hi - lo| <= 2^-17 |v|.
__device__ __forceinline__ void split_bf16(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}
// Two values at once, already packed (low half = a): one v_cvt_pk_bf16_f32 per array and no re-packing — 14 vector
// instructions per 4 elements instead of 21 through split_bf16 + pack2 (same roundings, bit-identical results).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void split2_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
  const bf16x2_t h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(uint32_t, h);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  const bf16x2_t l = {(__bf16)(a - ha), (__bf16)(b - hb)};
  lo = __builtin_bit_cast(uint32_t, l);
}
// Saturation of the fp16 halves.  MVP_F16_OVFL 1 (shipped): the kernels that write fp16 halves set MODE.FP16_OVFL once (f16_saturate_mode():
// an overflowing fp16 conversion then returns +-65504 instead of inf; inf only from inf) and the conversions need no clamp — 4 v_med3 fewer per
// two values in the fc1 epilogue, which is bound by its vector instructions.  0: explicit v_med3 clamps (the first form; same results).
#ifndef MVP_F16_OVFL
#define MVP_F16_OVFL 1
#endif
__device__ __forceinline__ void f16_saturate_mode() {
#if MVP_F16_OVFL
  __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);  // hwreg(HW_REG_MODE, offset 23, 1 bit) = FP16_OVFL
#endif
}
__device__ __forceinline__ float f16_clamp(float v) {
#if MVP_F16_OVFL
  return v;
#else
  return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
#endif
}
// The mixed form of mvp_gemm_args.out_f16_col0: hi = fp16(v) (rne), lo = bf16(v - hi), two values packed like split2_bf16.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2v_t;
// hi saturates at the largest finite fp16 (one v_med3 per value): an activation outlier beyond 65504 then keeps hi finite and leaves its excess
// to lo, which has fp32's exponent range (hi + lo still carries the value to bf16's 8 bits) instead of turning the whole GEMM row into NaN.
__device__ __forceinline__ void split2_f16_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f16x2_t h = __builtin_convertvector(f32x2v_t{f16_clamp(a), f16_clamp(b)}, f16x2_t);  // one v_cvt_pk_f16_f32; the halves are read back by v_fma_mix
  hi = __builtin_bit_cast(uint32_t, h);
  const bf16x2_t l = {(__bf16)(a - (float)h[0]), (__bf16)(b - (float)h[1])};
  lo = __builtin_bit_cast(uint32_t, l);
}
// The ACTIVATION operand of MVP_PREC_F16X2 (the "compensated" pair; include/mvp_hip.h has the algebra): hi = fp16(v) (rne, saturating like
// split2_f16_bf16), lo = fp16(8 * (v - hi) + hi / 8) — v - hi, hi / 8 and their sum are exact in fp32 (17 significant bits), so lo has ONE
// rounding.  Against the weight pair (fp16((1 - 2^-6) w), fp16((w + 64 d) / 8)), d = (1 - 2^-6) w - hi_w, the two fp16 products sum to
// v * w + (v - hi) * 64 d: relative error ~2^-18 per term instead of the 2^-12 of a single fp16 rounding of w.
__device__ __forceinline__ void split2_f16_comp(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f16x2_t h = __builtin_convertvector(f32x2v_t{f16_clamp(a), f16_clamp(b)}, f16x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  // 8 (v - hi) + hi / 8 = 8 v - 7.875 hi: one multiply (exact) and one fma whose exact result fits fp32 — same bits as the long form
  const float la = __builtin_fmaf((float)h[0], -7.875f, a * 8.f), lb = __builtin_fmaf((float)h[1], -7.875f, b * 8.f);
  const f16x2_t l = __builtin_convertvector(f32x2v_t{f16_clamp(la), f16_clamp(lb)}, f16x2_t);
  lo = __builtin_bit_cast(uint32_t, l);
}
// The WEIGHT-side operand of MVP_PREC_F16X2 computed on the device in fp32 (the K third of qkv for MVP_ATT_QK_F16): hi = fp16((1 - 2^-6) v),
// lo = fp16((v + 64 d) / 8), d = (1 - 2^-6) v - hi.  hi rounds the fp32 product (2^-24: far below the pair's 2^-18) to 11 bits; d is one fma on the
// exact product; v / 8 is exact and the second fma rounds once.  hi + lo / 8 reproduces v to ~2^-17.
__device__ __forceinline__ void split2_f16_wcomp(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float ta = a * 0.984375f, tb = b * 0.984375f;
  const f16x2_t h = __builtin_convertvector(f32x2v_t{f16_clamp(ta), f16_clamp(tb)}, f16x2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  // d from the EXACT product (one fma, one rounding) — written out so that no build's contraction choice decides the bits
  const float da = __builtin_fmaf(a, 0.984375f, -(float)h[0]), db = __builtin_fmaf(b, 0.984375f, -(float)h[1]);
  const float la = __builtin_fmaf(da, 8.f, a * 0.125f), lb = __builtin_fmaf(db, 8.f, b * 0.125f);
  const f16x2_t l = __builtin_convertvector(f32x2v_t{f16_clamp(la), f16_clamp(lb)}, f16x2_t);
  lo = __builtin_bit_cast(uint32_t, l);
}
// Which 16-bit pair form column `col` of a GEMM's pair output takes (mvp_gemm_args.out_f16_col0; wave-uniform wherever col is a wave tile's
// first column: every boundary is a multiple of 64): 0 = bf16 pair, 1 = fp16 hi + bf16 lo (V), 2 = compensated activation pair, 3 = compensated
// weight pair.  out_f16_col0 = -2C' (< -1, C' = 2 * heads * 64 = the V third's first column): Q | K | V = forms 2 | 3 | 1.
__device__ __forceinline__ int out_pair_form(int f16_col0, int col) {
  if (f16_col0 == 0) return 0;
  if (f16_col0 == -1) return 2;
  if (f16_col0 > 0) return col >= f16_col0 ? 1 : 0;
  const int v0 = -f16_col0;
  return col >= v0 ? 1 : (col >= (v0 >> 1) ? 3 : 2);
}
__device__ __forceinline__ void split2_form(int form, float a, float b, uint32_t& hi, uint32_t& lo) {
  if (form == 2) split2_f16_comp(a, b, hi, lo);
  else if (form == 1) split2_f16_bf16(a, b, hi, lo);
  else if (form == 3) split2_f16_wcomp(a, b, hi, lo);
  else split2_bf16(a, b, hi, lo);
}
// Output stores of data that this kernel never reads back and the NEXT kernel reads once (operand pairs): MVP_OUT_NT 1 marks them non-temporal
// (global_store ... nt), so they do not push the operands of the running kernel out of L2.  Per kernel: MVP_LN_NT, MVP_ATT_NT (A/B builds).
template <class T>
__device__ __forceinline__ void store_out(T* ptr, const T& v, const bool nt) {
  if (nt) __builtin_nontemporal_store(v, ptr);
  else *ptr = v;
}
// Bit pattern <-> float through a SCALAR.  (ROCm 7.2's clang miscompiles __builtin_bit_cast(float, vec[e]) written directly on an
// element of an ext_vector: every e yields element 0 — the optimiser then narrows a 16-byte load to its first dword.  Passing the
// element by value makes it a scalar first.)
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ uint32_t pack2(uint16_t a, uint16_t b) {
  return (uint32_t)a | ((uint32_t)b << 16);
}

// Column c of a hi|lo-interleaved pair array (MVP_PAIR_*_ILV32): element offset of the hi half inside the row; the lo half sits 32 further.
__device__ __forceinline__ int ilv32_col(int c) { return ((c >> 5) << 6) | (c & 31); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a linear block id: blocks that the dispatcher deals to the
// same XCD (id % 8 equal) get a contiguous range of logical tile ids, so neighbouring
// tiles (shared operand panels) hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// One LDS-DMA piece in the buffer form (buffer_load_dwordx4 ... lds): SGPR resource {base, num_records = bytes}, 32-bit per-lane
// byte offset, wave-uniform byte offset in an SGPR.  Offsets at or beyond `bytes` read zeros.  (A __device__-only helper: with
// the builtin written directly inside the kernel template's staging lambda the HOST pass of ROCm 7.2's clang silently drops
// every instantiation of the kernel — no diagnostic, undefined __device_stub__ symbols at link time.)
__device__ __forceinline__ void lds_dma16(const void* base, unsigned bytes, char* lds_dst, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
}

#define MVP_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return MVP_ELAUNCH;               \
  } while (0)
