// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_pp.hip): the erf-GELU, the XCD-region tile order and the fused
// LDS-staged epilogue.  Header-only, force-inlined: each kernel keeps its own main loop and calls gemm_epilogue() once its
// accumulators are final and every wave is done with the staging buffers (the caller's barrier).
#pragma once
#include "mvp_common.h"

namespace {

// Branch-free erf GELU: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7), one v_exp + one
// v_rcp.  (ocml erff measured ~17 us of VALU on the fc1 epilogue.)
// Every multiply-add is spelled out and contraction is off: the function is inlined into several epilogues (gemm_epilogue,
// gemm_epilogue_wide) whose surrounding code would otherwise let the compiler fuse differently per call site — and a batch must get
// the same bits whichever GEMM kernel its forward happens to use (grouped forwards use the large-M kernel, single batches the tiles).
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  poly = __builtin_fmaf(t, poly, 1.421413741f);
  poly = __builtin_fmaf(t, poly, -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.254829592f);
  poly = t * poly;
  const float e = __expf(-(ax * ax));
  const float erf_abs = __builtin_fmaf(-poly, e, 1.0f);
  const float erfv = copysignf(erf_abs, x);
  const float hx = 0.5f * x;
  return __builtin_fmaf(hx, erfv, hx);
}

// Logical tile index -> (tm, tn), XCD-region-major.  xcd_remap() hands each XCD one contiguous slice of the logical
// index space; here that space is enumerated region by region, the output being cut into XR x XC = 8 rectangles
// chosen to minimise what one XCD must pull through its L2: rows(A)/XR + rows(W)/XC; when a slice needs more than
// one round of resident workgroups, near-ties go to more column cuts (the smaller W share then stays L2-resident
// across the rounds).  Inside a region tiles run row-major.
// With plain row-major enumeration every XCD streamed ALL of W: fc1 fetched 164 MB from the memory side for 19 MB
// of operands at 26 % L2 misses (profiles/r01_pmc_*).  Uneven divisions only shift a few tiles across slice
// borders; the map stays a bijection.
__device__ __forceinline__ void region_tile(int L, int TM, int TN, int M, int N, bool multi_round, int& tm, int& tn) {
  int XR = 1;
  long best = (long)M * 8 + N;  // 8 * (M / XR + N * XR / 8)
#pragma unroll
  for (int c = 2; c <= 8; c <<= 1) {
    const long cost = (long)M * 8 / c + (long)N * c;
    if (cost + (multi_round ? cost / 16 : 0) < best) { best = cost; XR = c; }
  }
  const int XC = 8 / XR;
  for (int r = 0; r < 8; ++r) {
    const int xr = r / XC, xc = r - xr * XC;
    const int r0 = xr * TM / XR, r1 = (xr + 1) * TM / XR, c0 = xc * TN / XC, c1 = (xc + 1) * TN / XC;
    const int w = c1 - c0, sz = (r1 - r0) * w;
    if (L < sz) {
      const int q = L / w;
      tm = r0 + q;
      tn = c0 + (L - q * w);
      return;
    }
    L -= sz;
  }
  tm = TM - 1; tn = TN - 1;  // unreachable: the regions partition TM x TN
}

// Fused epilogue.  acc[i][j] = 16x16 accumulator of n-fragment i, m-fragment j of this wave's WM x WN output tile (MFMA issued
// "swapped": a lane's 4 registers are 4 consecutive output columns of row lane & 15).  smem = the kernel's dynamic LDS (>= NW * 32 *
// (WN + 4) * 4 bytes), free for reuse.  (m0, n0) = the workgroup's tile origin, (wm0, wn0) = this wave's offset inside it.
template <int NT, int MT, int WN, bool EXT>
__device__ __forceinline__ void gemm_epilogue(const mvp_gemm_args& p, f32x4_t (&acc)[NT][MT], char* smem, const int wave, const int lane,
                                              const int m0, const int n0, const int wm0, const int wn0) {
  const int frow = lane & 15;
  const int fq = lane >> 4;
  // EXT = false compiles the ReLU-gate / second-residual / post-residual-ReLU features out of the
  // hot backbone instantiations (they cost ~2-3 % there, measured A/B in one process).
  const uint8_t* x_relu_mask = EXT ? p.relu_mask : nullptr;
  uint8_t* x_out_mask = EXT ? p.out_mask : nullptr;
  const float* x_residual2 = EXT ? p.residual2 : nullptr;
  const mvp_bf16* x_res_hi = EXT ? p.residual_hi : nullptr;
  const int x_act_after = EXT ? p.act_after_res : 0;
  const int x_mask_mode = EXT ? p.mask_mode : 0;
  // Per-wave private scratch [32 rows][WN + 4] fp32; the trailing barrier of the main loop has
  // retired every read of the staging buffers, so they can be reused.
  constexpr int EPW = WN + 4;                 // padded row, floats (conflict-free b128 write/read)
  constexpr int EP_BYTES = 32 * EPW * 4;      // per wave
  constexpr int LPR = WN / 4;                 // lanes per output row (16 or 8)
  constexpr int RPI = 64 / LPR;               // rows per wave-instruction (4 or 8)
  float* ep = (float*)(smem + wave * EP_BYTES);
  const int er = lane / LPR, ec = (lane % LPR) * 4;
  const int ncol = n0 + wn0 + ec;
  const bool vec_ok = ((p.N & 3) == 0) && (ncol + 3 < p.N);
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ncol + e < p.N) bias4[e] = p.bias[ncol + e];
  }

#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
    // phase 1: accumulators -> LDS (lane: row frow of m-tile, cols i*16 + 4*fq ..)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < NT; ++i)
        *(f32x4_t*)(ep + (jj * 16 + frow) * EPW + i * 16 + fq * 4) = acc[i][h * 2 + jj];
    // phase 2: row-contiguous read-back, fused bias / activation / residual, full-line stores
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int lr = it * RPI + er;
      const int m = m0 + wm0 + h * 32 + lr;
      const f32x4_t a4 = *(const f32x4_t*)(ep + lr * EPW + ec);
      if (m >= p.M || ncol >= p.N) continue;
      int orow = m;
      if (p.row_group > 0) {
        const int gidx = m / p.row_group;
        orow = gidx * p.row_group_stride + p.row_group_off + (m - gidx * p.row_group);
      }
      const int rrow = (p.res_row_mod > 0) ? (m % p.res_row_mod) : orow;
      float v[4] = {a4[0] + bias4[0], a4[1] + bias4[1], a4[2] + bias4[2], a4[3] + bias4[3]};
      if (p.act == MVP_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      } else if (p.act == MVP_ACT_RELU && !x_act_after) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      auto write_mask = [&]() {  // forward: remember which outputs the ReLU kept (its backward gate)
        uint8_t* mo = x_out_mask + (size_t)orow * p.ldm + ncol;
        if (vec_ok && ((p.ldm & 3) == 0)) {
          *(uint32_t*)mo = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 0x100u : 0u) | (v[2] > 0.f ? 0x10000u : 0u) | (v[3] > 0.f ? 0x1000000u : 0u);
        } else {
          for (int e = 0; e < 4; ++e) if (ncol + e < p.N) mo[e] = v[e] > 0.f ? 1 : 0;
        }
      };
      if (x_out_mask && !x_act_after) write_mask();
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (x_relu_mask) {  // backward of ReLU: gate by the saved byte mask
        const uint8_t* mp = x_relu_mask + (size_t)orow * p.ldm + ncol;
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[e] = (ncol + e < p.N && mp[e]) ? 1.f : 0.f;
        if (x_mask_mode == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= keep[e];
        }
      }
      if (p.residual) {
        const float* rp = p.residual + (size_t)rrow * p.ldr + ncol;
        if (vec_ok && ((p.ldr & 3) == 0)) {
          const float4 r = *(const float4*)rp;
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        } else {
          for (int e = 0; e < 4; ++e) if (ncol + e < p.N) v[e] += rp[e];
        }
      }
      if (x_res_hi) {  // residual kept only as a bf16 pair (ResNet identities: no fp32 copy of every block output)
        const size_t ro = (size_t)rrow * p.ldr + ncol;
        if (vec_ok && ((p.ldr & 3) == 0)) {
          const u32x2_t h2 = *(const u32x2_t*)(x_res_hi + ro);
          const u32x2_t l2 = p.residual_lo ? *(const u32x2_t*)(p.residual_lo + ro) : u32x2_t{0u, 0u};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t hw = h2[e >> 1], lw = l2[e >> 1];
            v[e] += __builtin_bit_cast(float, (e & 1) ? (hw & 0xffff0000u) : (hw << 16)) +
                    __builtin_bit_cast(float, (e & 1) ? (lw & 0xffff0000u) : (lw << 16));
          }
        } else {
          for (int e = 0; e < 4; ++e)
            if (ncol + e < p.N) v[e] += bf2f(x_res_hi[ro + e]) + (p.residual_lo ? bf2f(p.residual_lo[ro + e]) : 0.f);
        }
      }
      if (x_residual2) {
        const float* rp = x_residual2 + (size_t)orow * p.ldr + ncol;
        for (int e = 0; e < 4; ++e) if (ncol + e < p.N) v[e] += rp[e];
      }
      if (p.act == MVP_ACT_RELU && x_act_after) {  // ResNet bottleneck: relu(conv3(x) + identity)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (x_out_mask) write_mask();  // gate of the post-residual ReLU (pre-activation fusion blocks)
      }
      if (p.out_f32) {
        float* op = p.out_f32 + (size_t)orow * p.ldo + ncol;
        if (vec_ok && ((p.ldo & 3) == 0)) {
          *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int e = 0; e < 4; ++e) if (ncol + e < p.N) op[e] = v[e];
        }
      }
      if (p.out_hi) {
        if (x_mask_mode == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= keep[e];  // (out_f32 above stayed un-gated)
        }
        uint32_t h01, l01, h23, l23;
        split2_bf16(v[0], v[1], h01, l01);
        split2_bf16(v[2], v[3], h23, l23);
        const bool oilv = p.out_pair_layout == MVP_PAIR_A_ILV32;  // one array, hi | lo interleaved per 32 columns (N % 32 == 0: host check)
        const size_t o = (size_t)orow * p.ldob + (oilv ? ilv32_col(ncol) : ncol);
        if (oilv) {
          if (ncol < p.N) {
            *(u32x2_t*)(p.out_hi + o) = u32x2_t{h01, h23};
            *(u32x2_t*)(p.out_hi + o + 32) = u32x2_t{l01, l23};
          }
        } else if (vec_ok && ((p.ldob & 3) == 0)) {
          *(u32x2_t*)(p.out_hi + o) = u32x2_t{h01, h23};
          if (p.out_lo) *(u32x2_t*)(p.out_lo + o) = u32x2_t{l01, l23};
        } else {
          const uint16_t hh[4] = {(uint16_t)h01, (uint16_t)(h01 >> 16), (uint16_t)h23, (uint16_t)(h23 >> 16)};
          const uint16_t ll[4] = {(uint16_t)l01, (uint16_t)(l01 >> 16), (uint16_t)l23, (uint16_t)(l23 >> 16)};
          for (int e = 0; e < 4; ++e)
            if (ncol + e < p.N) {
              p.out_hi[o + e] = hh[e];
              if (p.out_lo) p.out_lo[o + e] = ll[e];
            }
        }
      }
    }
  }
}

// Wide epilogue of the large-M kernel (gemm_pp.hip) for the combinations the ViT blocks use — compiled per combination, no runtime
// feature branches.  Same arithmetic as gemm_epilogue() (bias, erf-GELU, fp32 residual, fp32 and / or bf16-pair output, row remap),
// hence the same bits, but shaped for the memory system, which is what bounds it: when every CU of a round reaches its epilogue at
// once, 58-232 MB leave the chip in one burst, and that burst ran at 2.8-3.5 TB/s against the 6.9 TB/s a plain fill reaches
// (in-kernel stamps: 27 % of the qkv GEMM, 20 % of fc2).  What it changes:
//   * a lane owns 8 consecutive columns (two b128 reads of the per-wave LDS scratch): every store is 16 bytes per lane — the pair
//     halves left as 8-byte stores before, which run at 0.5-0.7x the 16-byte rate on gfx950;
//   * the residual rows of the NEXT 32-row chunk are loaded before the stores of the current one are issued: vmcnt retires in issue
//     order, so a residual load issued behind a store waits for that store's whole round trip (the generic loop did that once per
//     row group: a dependent load -> store -> load chain).
// N % 8 == 0 and 16-byte-aligned rows are required (the caller checks and falls back to gemm_epilogue()).
// GATE: the backward of a ReLU fused into an input-gradient GEMM (mask_mode 2: v *= the stored byte gate; the DPT probe's
// out_conv[2] input gradient, 200 704 x 512 outputs) — the 8 gate bytes of a lane are one 8-byte load, prefetched like the residual.
template <int NT, int MT, int WN, int ACT, bool RES, bool F32OUT, bool PAIR, bool GATE = false>
__device__ __forceinline__ void gemm_epilogue_wide(const mvp_gemm_args& p, f32x4_t (&acc)[NT][MT], char* smem, const int wave, const int lane,
                                                   const int m0, const int n0, const int wm0, const int wn0) {
  static_assert(WN == 64 && NT == 4 && (MT % 2) == 0, "wave tile 32k x 64");
  constexpr int EPW = WN + 4;                 // padded scratch row, floats
  constexpr int EP_BYTES = 32 * EPW * 4;      // per wave
  const int frow = lane & 15, fq = lane >> 4;
  float* ep = (float*)(smem + wave * EP_BYTES);
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int ncol = n0 + wn0 + ec;
  const bool col_ok = ncol < p.N;  // (N % 8 == 0: the 8 columns are all in or all out)
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_ok) {
    const float4 b0 = *(const float4*)(p.bias + ncol), b1 = *(const float4*)(p.bias + ncol + 4);
    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
  }
  const bool oilv = p.out_pair_layout == MVP_PAIR_A_ILV32;
  auto out_row = [&](int m) {
    if (p.row_group <= 0) return m;
    const int gidx = m / p.row_group;
    return gidx * p.row_group_stride + p.row_group_off + (m - gidx * p.row_group);
  };
  // Work unit = 16 rows (2 wave-instructions of 8 rows): small enough that the kernel's register allocation stays where the main loop
  // put it (the whole 32-row chunk in flight cost 40 more VGPRs: 255, i.e. no other wave can share the SIMDs with this kernel's two —
  // the probe step's small kernels run beside the frozen forward and need that room).
  constexpr int HIT = 2;
  float4 rpre[HIT][2];
  uint64_t gpre[HIT];
  auto load_res = [&](int u) {  // residual rows (and gate bytes) of 16-row unit u (u = 0 .. MT - 1)
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int m = m0 + wm0 + u * 16 + it * 8 + er;
      if (GATE) gpre[it] = (m < p.M && col_ok) ? *(const uint64_t*)(p.relu_mask + (size_t)out_row(m) * p.ldm + ncol) : 0ull;
      if (!RES) continue;
      rpre[it][0] = rpre[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && col_ok) {
        const int rrow = (p.res_row_mod > 0) ? (m % p.res_row_mod) : out_row(m);
        const float* rp = p.residual + (size_t)rrow * p.ldr + ncol;
        rpre[it][0] = *(const float4*)rp;
        rpre[it][1] = *(const float4*)(rp + 4);
      }
    }
  };
  if (RES || GATE) load_res(0);
#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
    // accumulators -> LDS (lane: row frow of the m-fragment, columns i*16 + 4*fq ..)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < NT; ++i) *(f32x4_t*)(ep + (jj * 16 + frow) * EPW + i * 16 + fq * 4) = acc[i][h * 2 + jj];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int u = h * 2 + hh;
      float v[HIT][8];
#pragma unroll
      for (int it = 0; it < HIT; ++it) {
        const int lr = hh * 16 + it * 8 + er;
        const f32x4_t a = *(const f32x4_t*)(ep + lr * EPW + ec), b = *(const f32x4_t*)(ep + lr * EPW + ec + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[it][e] = a[e] + bias8[e]; v[it][4 + e] = b[e] + bias8[4 + e]; }
        if (ACT == MVP_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] = gelu_erf(v[it][e]);
        }
        if (GATE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] *= ((gpre[it] >> (8 * e)) & 0xffull) ? 1.f : 0.f;
        }
        if (RES) {
          v[it][0] += rpre[it][0].x; v[it][1] += rpre[it][0].y; v[it][2] += rpre[it][0].z; v[it][3] += rpre[it][0].w;
          v[it][4] += rpre[it][1].x; v[it][5] += rpre[it][1].y; v[it][6] += rpre[it][1].z; v[it][7] += rpre[it][1].w;
        }
      }
      if ((RES || GATE) && u + 1 < MT) load_res(u + 1);  // issued BEFORE this unit's stores: its data never waits behind them
#pragma unroll
      for (int it = 0; it < HIT; ++it) {
        const int m = m0 + wm0 + u * 16 + it * 8 + er;
        if (m >= p.M || !col_ok) continue;
        const int orow = out_row(m);
        if (F32OUT) {
          float* op = p.out_f32 + (size_t)orow * p.ldo + ncol;
          *(float4*)op = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
          *(float4*)(op + 4) = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
        }
        if (PAIR) {
          uint32_t hw[4], lw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_bf16(v[it][2 * e], v[it][2 * e + 1], hw[e], lw[e]);
          if (oilv) {
            mvp_bf16* o = p.out_hi + (size_t)orow * p.ldob + ilv32_col(ncol);
            *(u32x4_t*)o = u32x4_t{hw[0], hw[1], hw[2], hw[3]};
            *(u32x4_t*)(o + 32) = u32x4_t{lw[0], lw[1], lw[2], lw[3]};
          } else {
            const size_t o = (size_t)orow * p.ldob + ncol;
            *(u32x4_t*)(p.out_hi + o) = u32x4_t{hw[0], hw[1], hw[2], hw[3]};
            if (p.out_lo) *(u32x4_t*)(p.out_lo + o) = u32x4_t{lw[0], lw[1], lw[2], lw[3]};
          }
        }
      }
    }
  }
}

// Which wide-epilogue instantiation serves these arguments (0 = none: the generic epilogue).
__host__ __device__ __forceinline__ int gemm_epilogue_wide_variant(const mvp_gemm_args& p) {
  if (p.relu_mask && p.mask_mode == 2 && !p.out_mask && !p.residual2 && !p.act_after_res && !p.residual_hi && !p.residual && p.act == MVP_ACT_NONE &&
      p.out_f32 && !p.out_hi && !(p.N & 7) && !(p.ldm & 7) && !((size_t)p.relu_mask & 7) && !(p.ldo & 3) && !((size_t)p.out_f32 & 15) &&
      !(p.bias && ((size_t)p.bias & 15)))
    return 6;  // gated input gradient, fp32 out
  if (p.relu_mask || p.out_mask || p.residual2 || p.act_after_res || p.residual_hi) return 0;
  if ((p.N & 7) || (p.act != MVP_ACT_NONE && p.act != MVP_ACT_GELU)) return 0;
  if (p.residual && ((p.ldr & 3) || ((size_t)p.residual & 15))) return 0;
  if (p.out_f32 && ((p.ldo & 3) || ((size_t)p.out_f32 & 15))) return 0;
  if (p.out_hi && ((p.ldob & 7) || ((size_t)p.out_hi & 15) || (p.out_lo && ((size_t)p.out_lo & 15)))) return 0;
  if (p.bias && ((size_t)p.bias & 15)) return 0;
  const bool pair = p.out_hi != nullptr, f32 = p.out_f32 != nullptr, res = p.residual != nullptr, gelu = p.act == MVP_ACT_GELU;
  if (pair && !f32 && !res) return gelu ? 2 : 1;   // qkv / fc1
  if (f32 && !pair && res && !gelu) return 3;      // proj / fc2 / patch embedding
  if (f32 && !pair && !res) return gelu ? 5 : 4;
  return 0;
}

}  // namespace
