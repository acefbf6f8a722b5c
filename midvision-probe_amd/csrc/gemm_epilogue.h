// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_pp.hip): the erf-GELU, the XCD-region tile order and the fused
// LDS-staged epilogue.  Header-only, force-inlined: each kernel keeps its own main loop and calls gemm_epilogue() once its
// accumulators are final and every wave is done with the staging buffers (the caller's barrier).
#pragma once
#include "mvp_common.h"

// Cache policy of the wide epilogue's output stores (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1; profiles/r04_store_policy.txt).
//   MVP_EPI_AUX_PAIR  the pair-only forms (qkv, fc1: written once, read by the NEXT kernel, nothing read back here).  Shipped: 18 = nt sc1 —
//                     the stores no longer push the running launch's weight panels and A rows out of L2: +1.8 % / +3.1 % on the headline
//                     (whole-library A/B builds alternating on one box, two boxes).
//   MVP_EPI_AUX       every other output store (the residual forms: x is read again by LayerNorm and by the next residual add).  Shipped: 0 —
//                     nt costs proj / fc2 3-6 %.  Defining it on the command line (tools/pp_bench.py --store-policy) sets the pair forms too.
//   MVP_EPI_LD_AUX    the fp32 residual loads.  Shipped: 0 (nt: -1.6 % on the headline).
#ifdef MVP_EPI_AUX
#ifndef MVP_EPI_AUX_PAIR
#define MVP_EPI_AUX_PAIR MVP_EPI_AUX
#endif
#else
#define MVP_EPI_AUX 0
#ifndef MVP_EPI_AUX_PAIR
#define MVP_EPI_AUX_PAIR 18
#endif
#endif
#ifndef MVP_EPI_LD_AUX
#define MVP_EPI_LD_AUX 0
#endif
#ifndef MVP_EPI_UNI_NT  // 1: gemm_epilogue_uni's pair-only outputs (conv + ReLU -> pair: ResNet / DPT layers, the tile kernels' qkv / fc1) nt sc1 too (A/B builds)
#define MVP_EPI_UNI_NT 0
#endif

namespace {

// Branch-free erf GELU: Abramowitz-Stegun 7.1.26 for 1 - erf (|erf error| <= 1.5e-7), one v_exp + one v_rcp.  (ocml erff measured
// ~17 us of VALU on the fc1 epilogue.)  Round 4 (the fc1 epilogue is bound by its vector instructions — 26 per value with the operand
// pair — and the two-product kernel is cycle-bound): 13 instructions per value instead of 15, from
//     GELU(x) = x/2 (1 + erf(|x|/sqrt2) sign x) = max(x, 0) - |x| (1 - erf(a)) / 2,   a = |x| / sqrt2,   1 - erf(a) = poly(t) exp(-a^2),  t = 1 / (1 + p a)
// with b = a sqrt(log2 e): exp(-a^2) = exp2(-b b) (raw v_exp_f32: results below 2^-126 are zero either way), p a = (p / sqrt(log2 e)) b, the 1/2
// folded into the polynomial's coefficients; no copysign, no x/2.  Max |error| against fp64 over [-12, 12]: 3.3e-7 (the previous form: 4.7e-7).
// Every multiply-add is spelled out and contraction is off: the function is inlined into several epilogues (gemm_epilogue,
// gemm_epilogue_wide, gemm_epilogue_uni) whose surrounding code would otherwise let the compiler fuse differently per call site — and a batch must
// get the same bits whichever GEMM kernel its forward happens to use (grouped forwards use the large-M kernel, single batches the tiles).
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  const float b = fabsf(x) * 0.8493218002880191f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.2727374808792225f, b, 1.0f));
  float poly = __builtin_fmaf(t, 0.5307027145f, -0.7265760135f);
  poly = __builtin_fmaf(t, poly, 0.7107068705f);
  poly = __builtin_fmaf(t, poly, -0.142248368f);
  poly = __builtin_fmaf(t, poly, 0.127414796f);
  poly = t * poly;
  const float pe = poly * __builtin_amdgcn_exp2f(-(b * b));
  return __builtin_fmaf(-fabsf(x), pe, fmaxf(x, 0.f));
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
// ARGS: mvp_gemm_args, or the same struct read in place from the kernel-argument segment (address space 4; gemm_pp.hip's tile loop).
template <int NT, int MT, int WN, bool EXT, class ARGS>
__device__ __forceinline__ void gemm_epilogue(const ARGS& p, f32x4_t (&acc)[NT][MT], char* smem, const int wave, const int lane,
                                              const int m0, const int n0, const int wm0, const int wn0) {
  f16_saturate_mode();  // (the pair outputs' fp16 forms saturate instead of overflowing: mvp_common.h)
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
  // An unconditional use right here: the compiler waits for the bias loads ONCE, before any store is issued.  Their only other
  // uses sit inside the row loop's `m < M` regions, and its wait-count pass must assume that a skipped row leaves them pending —
  // it then re-waits with s_waitcnt vmcnt(0) at every row group, i.e. for all the stores issued so far (seen in the ISA of every
  // tile kernel in round 4: one store round trip per row group).
#pragma unroll
  for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bias4[e]));

#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
    // phase 1: accumulators -> LDS (lane: row frow of m-tile, cols i*16 + 4*fq ..)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < NT; ++i)
        *(f32x4_t*)(ep + (jj * 16 + frow) * EPW + i * 16 + fq * 4) = acc[i][h * 2 + jj];
    // phase 2: row-contiguous read-back, fused bias / activation / residual, full-line stores
    auto rows = [&](const int it) {
      const int lr = it * RPI + er;
      const int m = m0 + wm0 + h * 32 + lr;
      const f32x4_t a4 = *(const f32x4_t*)(ep + lr * EPW + ec);
      if (m >= p.M || ncol >= p.N) return;
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
        const int form = out_pair_form(p.out_f16_col0, ncol);  // (4 columns: all in one form — every boundary is a multiple of 64)
        if (form == 2) {  // the activation operand of a MVP_PREC_F16X2 GEMM (the compensated fp16 pair): -1 = every column; the Q third of qkv
          split2_f16_comp(v[0], v[1], h01, l01);
          split2_f16_comp(v[2], v[3], h23, l23);
        } else if (form == 1) {  // hi = fp16(v), lo = bf16(v - hi): the V third of qkv
          split2_f16_bf16(v[0], v[1], h01, l01);
          split2_f16_bf16(v[2], v[3], h23, l23);
        } else if (form == 3) {  // the compensated weight-side pair: the K third of qkv for MVP_ATT_QK_F16
          split2_f16_wcomp(v[0], v[1], h01, l01);
          split2_f16_wcomp(v[2], v[3], h23, l23);
        } else {
          split2_bf16(v[0], v[1], h01, l01);
          split2_bf16(v[2], v[3], h23, l23);
        }
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
    };
    // The `h` loop MUST be unrolled (acc[][h * 2 + jj] is a register array: a dynamic index sends all 128 accumulators through
    // scratch — 528 bytes per lane in the EXT instantiations of round 3, where the fully unrolled body exceeded the pragma-unroll
    // size limit and the compiler silently kept `h` as a loop).  The row loop only moves LDS / global addresses: with the large EXT body
    // it stays a loop, which keeps the unrolled size under the limit.
    if constexpr (EXT) {
#pragma nounroll
      for (int it = 0; it < 32 / RPI; ++it) rows(it);
    } else {
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) rows(it);
    }
  }
}

// Wide epilogue of the large-M kernel (gemm_pp.hip) for the combinations the ViT blocks use — compiled per combination, no runtime
// feature branches.  Same arithmetic as gemm_epilogue() (bias, erf-GELU, fp32 residual, fp32 and / or bf16-pair output, row remap),
// hence the same bits, but shaped for the memory system, which is what bounds it: when every CU of a round reaches its epilogue at
// once, 58-232 MB leave the chip in one burst.  What it does about it:
//   * a lane owns 8 consecutive columns (two b128 reads of the per-wave LDS scratch): every store is 16 bytes per lane — the pair
//     halves left as 8-byte stores before, which run at 0.5-0.7x the 16-byte rate on gfx950;
//   * the residual rows of the NEXT 16-row unit are loaded before the stores of the current one are issued: vmcnt retires in issue
//     order, so a residual load issued behind a store waits for that store's whole round trip;
//   * (round 4) NO BRANCHES: every global access is a buffer instruction whose per-lane offset is the out-of-range sentinel for rows
//     past M / columns past N (loads return zeros, stores are dropped), absent operands (no bias, no lo array) are zero-sized
//     resources.  With `if (row < M) { load / add / store }` the compiler sinks the uses of the loaded bias / residual registers into the
//     conditional region, and its wait-count pass — which must assume the skipped path leaves those loads pending — then emits a full
//     s_waitcnt vmcnt(0) in EVERY 16-row unit: each unit waited for the previous unit's stores to be acknowledged (8 store round trips
//     per tile = the 16 k cycles the round-3 stamps show for the qkv epilogue), and, once the tile loop put the main loop behind the
//     epilogue, the same "maybe pending" registers cost a vmcnt(0) per k-step in the main loop.  Straight-line code gets counted waits.
//     A fixed instruction count per tile (2 bias loads, 4 stores per unit, 4 residual / 2 gate loads per unit) also lets the caller
//     count the epilogue's operations in its own vmcnt arithmetic (gemm_pp.hip, tile switch).
// N % 8 == 0, 16-byte-aligned rows and arrays below 2 GiB are required (gemm_epilogue_wide_variant() checks; else gemm_epilogue()).
// GATE: the backward of a ReLU fused into an input-gradient GEMM (mask_mode 2: v *= the stored byte gate; the DPT probe's
// out_conv[2] input gradient, 200 704 x 512 outputs) — the 8 gate bytes of a lane are one 8-byte load, prefetched like the residual.
// LDS scratch: 16 rows x (WN + 4) floats per wave (4352 B; 34 KB for the 8 waves) at `smem` — the persistent large-M kernel points it at
// its second k-step buffer while the first already receives the next tile's operands.
// `after_first_loads()` is called once the bias (and first residual / gate) loads are issued and before anything waits for them: the
// caller's hook for memory operations that must be OLDER than the epilogue's stores but need not delay its first loads (the next
// tile's LDS-DMA prefetch).
// Memory operations issued per wave, for the caller's vmcnt arithmetic: 2 bias loads, then per 16-row unit 4 stores (+ 4 residual
// loads, or 2 gate loads, for the units after the first, issued one unit ahead).
template <bool RES, bool GATE>
struct gemm_epilogue_wide_ops {
  static constexpr int before_hook = 2 + (RES ? 4 : 0) + (GATE ? 2 : 0);              // bias, unit 0's residual / gate rows
  static constexpr int after_hook = 8 * 4 + 7 * ((RES ? 4 : 0) + (GATE ? 2 : 0));     // 32 stores, the other units' residual / gate rows
};
template <int NT, int MT, int WN, int ACT, bool RES, bool F32OUT, bool PAIR, bool GATE = false, class ARGS = mvp_gemm_args, class HOOK>
__device__ __forceinline__ void gemm_epilogue_wide(const ARGS& p, f32x4_t (&acc)[NT][MT], char* smem, const int wave, const int lane,
                                                   const int m0, const int n0, const int wm0, const int wn0, HOOK&& after_first_loads) {
  f16_saturate_mode();  // (the pair outputs' fp16 forms saturate instead of overflowing: mvp_common.h)
  static_assert(WN == 64 && NT == 4 && MT == 8, "wave tile 128 x 64");
  static_assert(F32OUT != PAIR, "one output form per instantiation (4 stores per unit)");
  constexpr int EPW = WN + 4;                 // padded scratch row, floats
  constexpr int EP_BYTES = 16 * EPW * 4;      // per wave
  constexpr int SENT = 0x7fffff00;            // out-of-range byte offset (every resource below has at most that many bytes)
  const int frow = lane & 15, fq = lane >> 4;
  float* ep = (float*)(smem + wave * EP_BYTES);
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int ncol = n0 + wn0 + ec;
  const bool col_ok = ncol < p.N;  // (N % 8 == 0: the 8 columns are all in or all out)
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, ptr ? SENT : 0, 0x00020000); };  // null: every access out of range
  const __amdgpu_buffer_rsrc_t r_bias = rsrc(p.bias), r_res = rsrc(RES ? (const void*)p.residual : nullptr), r_gate = rsrc(GATE ? (const void*)p.relu_mask : nullptr);
  const __amdgpu_buffer_rsrc_t r_o32 = rsrc(F32OUT ? (const void*)p.out_f32 : nullptr), r_ohi = rsrc(PAIR ? (const void*)p.out_hi : nullptr);
  const bool oilv = p.out_pair_layout == MVP_PAIR_A_ILV32;
  const int pform = PAIR ? out_pair_form(p.out_f16_col0, n0 + wn0) : 0;  // the 16-bit form of this wave's 64 columns (wave-uniform)
  const __amdgpu_buffer_rsrc_t r_olo = rsrc(PAIR ? (oilv ? (const void*)p.out_hi : (const void*)p.out_lo) : nullptr);  // interleaved: the lo half sits 64 bytes behind the hi half
  const int lo_soff = oilv ? 64 : 0;
  const int ob = col_ok ? ncol * 4 : SENT;
  const u32x4_t bias_a = __builtin_amdgcn_raw_buffer_load_b128(r_bias, ob, 0, 0), bias_b = __builtin_amdgcn_raw_buffer_load_b128(r_bias, ob, 16, 0);
  // row m of the tile -> byte offsets into the output / residual / gate arrays (the sentinel for rows past M and columns past N)
  const int pcol = PAIR ? (oilv ? ilv32_col(ncol) : ncol) : 0;
  struct row_off { int out, res, gate; };
  auto offsets = [&](int m) {
    row_off o;
    const bool ok = (m < p.M) && col_ok;
    int orow = m;
    if (p.row_group > 0) {  // (wave-uniform: the patch embedding's output-row remap)
      const int gidx = m / p.row_group;
      orow = gidx * p.row_group_stride + p.row_group_off + (m - gidx * p.row_group);
    }
    o.out = ok ? (F32OUT ? (orow * p.ldo + ncol) * 4 : (orow * p.ldob + pcol) * 2) : SENT;
    o.res = SENT; o.gate = SENT;
    if (RES) {
      const int rrow = (p.res_row_mod > 0) ? (m % p.res_row_mod) : orow;
      o.res = ok ? (rrow * p.ldr + ncol) * 4 : SENT;
    }
    if (GATE) o.gate = ok ? (orow * p.ldm + ncol) : SENT;
    return o;
  };
  // Work unit = 16 rows (2 wave-instructions of 8 rows): small enough that the kernel's register allocation stays where the main loop
  // put it (a whole 32-row chunk in flight cost 40 more VGPRs: no other wave could then share the SIMDs with this kernel's two —
  // the probe step's small kernels run beside the frozen forward and need that room).
  constexpr int HIT = 2;
  u32x4_t rpre[HIT][2];
  u32x2_t gpre[HIT];
  row_off ro[HIT], ro_next[HIT];
  auto load_res = [&](int u) {  // offsets of 16-row unit u, and its residual rows / gate bytes
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      ro_next[it] = offsets(m0 + wm0 + u * 16 + it * 8 + er);
      if (GATE) gpre[it] = __builtin_amdgcn_raw_buffer_load_b64(r_gate, ro_next[it].gate, 0, 0);
      if (RES) {
        rpre[it][0] = __builtin_amdgcn_raw_buffer_load_b128(r_res, ro_next[it].res, 0, MVP_EPI_LD_AUX);
        rpre[it][1] = __builtin_amdgcn_raw_buffer_load_b128(r_res, ro_next[it].res, 16, MVP_EPI_LD_AUX);
      }
    }
  };
  load_res(0);
  after_first_loads();
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bias8[e] = u2f(bias_a[e]); bias8[4 + e] = u2f(bias_b[e]); }
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    // (straight-line code: without the fence the scheduler pulls the LDS round trips and residual loads of later units forward until
    // the register file is full — 256 VGPRs and spills)
    __builtin_amdgcn_sched_barrier(0);
    // accumulators of m-fragment u -> LDS (lane: row frow of the m-fragment, columns i*16 + 4*fq ..)
#pragma unroll
    for (int i = 0; i < NT; ++i) *(f32x4_t*)(ep + frow * EPW + i * 16 + fq * 4) = acc[i][u];
    float v[HIT][8];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      ro[it] = ro_next[it];
      const int lr = it * 8 + er;
      const f32x4_t a = *(const f32x4_t*)(ep + lr * EPW + ec), b = *(const f32x4_t*)(ep + lr * EPW + ec + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[it][e] = a[e] + bias8[e]; v[it][4 + e] = b[e] + bias8[4 + e]; }
      if (ACT == MVP_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = gelu_erf(v[it][e]);
      }
      if (GATE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] *= ((gpre[it][e >> 2] >> (8 * (e & 3))) & 0xffu) ? 1.f : 0.f;
      }
      if (RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[it][e] += u2f(rpre[it][0][e]);
          v[it][4 + e] += u2f(rpre[it][1][e]);
        }
      }
    }
    if (u + 1 < MT) load_res(u + 1);  // issued BEFORE this unit's stores: its data never waits behind them
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      if (F32OUT) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{f2u(v[it][0]), f2u(v[it][1]), f2u(v[it][2]), f2u(v[it][3])}, r_o32, ro[it].out, 0, MVP_EPI_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{f2u(v[it][4]), f2u(v[it][5]), f2u(v[it][6]), f2u(v[it][7])}, r_o32, ro[it].out, 16, MVP_EPI_AUX);
      }
      if (PAIR) {
        uint32_t hw[4], lw[4];
        if (pform == 2) {  // (the compensated fp16 pair: -1 = every column; the Q third of qkv)
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_comp(v[it][2 * e], v[it][2 * e + 1], hw[e], lw[e]);
        } else if (pform == 1) {  // (the V third)
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_bf16(v[it][2 * e], v[it][2 * e + 1], hw[e], lw[e]);
        } else if (pform == 3) {  // (the K third as the weight-side pair)
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_wcomp(v[it][2 * e], v[it][2 * e + 1], hw[e], lw[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_bf16(v[it][2 * e], v[it][2 * e + 1], hw[e], lw[e]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hw[0], hw[1], hw[2], hw[3]}, r_ohi, ro[it].out, 0, (!RES && !F32OUT) ? MVP_EPI_AUX_PAIR : MVP_EPI_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{lw[0], lw[1], lw[2], lw[3]}, r_olo, ro[it].out, lo_soff, (!RES && !F32OUT) ? MVP_EPI_AUX_PAIR : MVP_EPI_AUX);
      }
    }
  }
}

// The universal branch-free epilogue (round 4): gemm_epilogue()'s arithmetic — bias, GELU / ReLU before or after the residuals, the
// ReLU gate (relu_mask, both mask modes), an fp32 residual and / or a residual given as a bf16 pair, the byte mask of a ReLU's output,
// fp32 and / or pair output (bf16 or fp16-hi form), row remaps — in gemm_epilogue_wide()'s shape: a lane owns 8 consecutive columns of
// one row, every global access is a buffer instruction (out-of-range sentinel for rows >= M / columns >= N, zero-sized resources for
// absent operands: a load of an absent residual returns zeros and adding them changes nothing), the next 16-row unit's loads are issued
// ahead of the current unit's stores, the only branches are wave-uniform ones around vector arithmetic.  For wave tiles of 64 or 32
// columns: every tile kernel of gemm.hip (the ResNet-50 trunk's 1x1 / 3x3 convolutions, whose residual-adding 1x1 layers at
// K = 64 ... 512 are all epilogue — tools/resnet_bench.py: 35 % of the forward at 0.4 of the HBM rate with the row-guarded epilogue and its
// vmcnt(0) per row group —, the DPT probe's convolutions, the serial loop's qkv / fc1) and the EXT instantiations of the large-M kernel.
// Not covered (gemm_epilogue() stays): residual2, N % 8 != 0, unaligned or >= 2 GiB arrays (gemm_epilogue_uni_ok()).
// The results are gemm_epilogue()'s bit for bit (same operations in the same order per element).
// EXT = false compiles the gate / mask / pair-residual / post-residual-ReLU parts out, as in gemm_epilogue().
__host__ __device__ __forceinline__ bool gemm_epilogue_uni_ok(const mvp_gemm_args& p) {
  if ((p.N & 7) || p.residual2) return false;
  if (p.act != MVP_ACT_NONE && p.act != MVP_ACT_GELU && p.act != MVP_ACT_RELU) return false;
  const int64_t lim = 0x7fffff00ll;
  const int64_t rows = p.row_group > 0 ? ((int64_t)(p.M - 1) / p.row_group) * p.row_group_stride + p.row_group_off + p.row_group : p.M;
  const int64_t rrows = p.res_row_mod > 0 ? (int64_t)p.res_row_mod : rows;
  if (p.bias && (((size_t)p.bias & 15) || (int64_t)p.N * 4 > lim)) return false;
  if (p.out_f32 && ((p.ldo & 3) || ((size_t)p.out_f32 & 15) || rows * p.ldo * 4 > lim)) return false;
  if (p.out_hi && ((p.ldob & 7) || ((size_t)p.out_hi & 15) || (p.out_lo && ((size_t)p.out_lo & 15)) || rows * p.ldob * 2 + 64 > lim)) return false;
  if (p.residual && ((p.ldr & 3) || ((size_t)p.residual & 15) || rrows * p.ldr * 4 > lim)) return false;
  if (p.residual_hi && ((p.ldr & 7) || ((size_t)p.residual_hi & 15) || (p.residual_lo && ((size_t)p.residual_lo & 15)) || rrows * p.ldr * 2 > lim)) return false;
  if ((p.relu_mask || p.out_mask) && ((p.ldm & 7) || rows * p.ldm > lim)) return false;
  if (p.relu_mask && (((size_t)p.relu_mask & 7) || (p.mask_mode != 1 && p.mask_mode != 2))) return false;
  if (p.out_mask && ((size_t)p.out_mask & 7)) return false;
  return true;
}

template <int NT, int MT, int WN, bool EXT, class ARGS = mvp_gemm_args, class HOOK>
__device__ __forceinline__ void gemm_epilogue_uni(const ARGS& p, f32x4_t (&acc)[NT][MT], char* smem, const int wave, const int lane,
                                                  const int m0, const int n0, const int wm0, const int wn0, HOOK&& after_first_loads) {
  f16_saturate_mode();  // (the pair outputs' fp16 forms saturate instead of overflowing: mvp_common.h)
  static_assert((WN == 64 || WN == 32) && NT == WN / 16, "wave tile 16k x 64 or 16k x 32");
  constexpr int EPW = WN + 4;                 // padded scratch row, floats
  constexpr int EP_BYTES = 16 * EPW * 4;      // per wave
  constexpr int SENT = 0x7fffff00;            // out-of-range byte offset (every resource below has at most that many bytes)
  constexpr int LPR = WN / 8;                 // lanes per output row (8 columns each): 8 or 4
  constexpr int RPI = 64 / LPR;               // rows per wave-instruction: 8 or 16
  constexpr int HIT = 16 / RPI;               // wave-instructions per 16-row unit: 2 or 1
  const int frow = lane & 15, fq = lane >> 4;
  float* ep = (float*)(smem + wave * EP_BYTES);
  const int er = lane / LPR, ec = (lane % LPR) * 8;
  const int ncol = n0 + wn0 + ec;
  const bool col_ok = ncol < p.N;  // (N % 8 == 0: the 8 columns are all in or all out)
  auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, ptr ? SENT : 0, 0x00020000); };  // null: every access out of range
  const uint8_t* x_gate = EXT ? p.relu_mask : nullptr;
  uint8_t* x_omask = EXT ? p.out_mask : nullptr;
  const mvp_bf16* x_rhi = EXT ? p.residual_hi : nullptr;
  const mvp_bf16* x_rlo = EXT ? p.residual_lo : nullptr;
  const bool act_after = EXT && p.act_after_res != 0;
  const int mask_mode = EXT ? p.mask_mode : 0;
  const bool oilv = p.out_pair_layout == MVP_PAIR_A_ILV32;
  const __amdgpu_buffer_rsrc_t r_bias = rsrc(p.bias), r_res = rsrc(p.residual), r_rhi = rsrc(x_rhi), r_rlo = rsrc(x_rlo), r_gate = rsrc(x_gate);
  const __amdgpu_buffer_rsrc_t r_o32 = rsrc(p.out_f32), r_ohi = rsrc(p.out_hi), r_omask = rsrc(x_omask);
  const __amdgpu_buffer_rsrc_t r_olo = rsrc(oilv ? (const void*)p.out_hi : (const void*)p.out_lo);  // interleaved: the lo half sits 64 bytes behind the hi half
  const int lo_soff = oilv ? 64 : 0;
  const bool has_pair = p.out_hi != nullptr;
  const bool nt_pair = has_pair && p.out_f32 == nullptr && p.residual == nullptr && x_rhi == nullptr;  // pair-only form (MVP_EPI_UNI_NT builds)
  const int pform = has_pair ? out_pair_form(p.out_f16_col0, n0 + wn0) : 0;  // (wave-uniform: every boundary is a multiple of 64)
  const int ob = col_ok ? ncol * 4 : SENT;
  const u32x4_t bias_a = __builtin_amdgcn_raw_buffer_load_b128(r_bias, ob, 0, 0), bias_b = __builtin_amdgcn_raw_buffer_load_b128(r_bias, ob, 16, 0);
  const int pcol = oilv ? ilv32_col(ncol) : ncol;
  struct row_off { int o32, opair, res, rpair, mask; };
  auto offsets = [&](int m) {
    row_off o;
    const bool ok = (m < p.M) && col_ok;
    int orow = m;
    if (p.row_group > 0) {  // (wave-uniform: the patch embedding's output-row remap)
      const int gidx = m / p.row_group;
      orow = gidx * p.row_group_stride + p.row_group_off + (m - gidx * p.row_group);
    }
    const int rrow = (p.res_row_mod > 0) ? (m % p.res_row_mod) : orow;
    o.o32 = ok ? (orow * p.ldo + ncol) * 4 : SENT;
    o.opair = ok ? (orow * p.ldob + pcol) * 2 : SENT;
    o.res = ok ? (rrow * p.ldr + ncol) * 4 : SENT;
    o.rpair = ok ? (rrow * p.ldr + ncol) * 2 : SENT;
    o.mask = ok ? (orow * p.ldm + ncol) : SENT;
    return o;
  };
  u32x4_t rpre[HIT][2], ppre[HIT][2];  // next unit's fp32 residual (8 floats) and pair residual (8 hi, 8 lo bf16)
  u32x2_t gpre[HIT];                   // next unit's gate bytes
  row_off ro[HIT], ro_next[HIT];
  auto load_next = [&](int u) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      ro_next[it] = offsets(m0 + wm0 + u * 16 + it * RPI + er);
      rpre[it][0] = __builtin_amdgcn_raw_buffer_load_b128(r_res, ro_next[it].res, 0, 0);
      rpre[it][1] = __builtin_amdgcn_raw_buffer_load_b128(r_res, ro_next[it].res, 16, 0);
      if (EXT) {
        ppre[it][0] = __builtin_amdgcn_raw_buffer_load_b128(r_rhi, ro_next[it].rpair, 0, 0);
        ppre[it][1] = __builtin_amdgcn_raw_buffer_load_b128(r_rlo, ro_next[it].rpair, 0, 0);
        gpre[it] = __builtin_amdgcn_raw_buffer_load_b64(r_gate, ro_next[it].mask, 0, 0);
      }
    }
  };
  load_next(0);
  after_first_loads();
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { bias8[e] = u2f(bias_a[e]); bias8[4 + e] = u2f(bias_b[e]); }
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    __builtin_amdgcn_sched_barrier(0);  // (straight-line code: keep the scheduler from pulling later units forward until the registers run out)
#pragma unroll
    for (int i = 0; i < NT; ++i) *(f32x4_t*)(ep + frow * EPW + i * 16 + fq * 4) = acc[i][u];
    float v[HIT][8], vp[HIT][8];  // value for the fp32 output / for the pair output (they differ under mask_mode 1)
    uint32_t mbits[HIT][2];       // the 8 mask bytes of a lane's columns (out_mask)
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      ro[it] = ro_next[it];
      const int lr = it * RPI + er;
      const f32x4_t a = *(const f32x4_t*)(ep + lr * EPW + ec), b = *(const f32x4_t*)(ep + lr * EPW + ec + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[it][e] = a[e] + bias8[e]; v[it][4 + e] = b[e] + bias8[4 + e]; }
      if (p.act == MVP_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = gelu_erf(v[it][e]);
      } else if (p.act == MVP_ACT_RELU && !act_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
      }
      mbits[it][0] = mbits[it][1] = 0u;
      if (EXT && x_omask && !act_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) mbits[it][e >> 2] |= (v[it][e] > 0.f ? 1u : 0u) << (8 * (e & 3));
      }
      float keep[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) keep[e] = 1.f;
      if (EXT && x_gate) {
#pragma unroll
        for (int e = 0; e < 8; ++e) keep[e] = ((gpre[it][e >> 2] >> (8 * (e & 3))) & 0xffu) ? 1.f : 0.f;
        if (mask_mode == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[it][e] *= keep[e];
        }
      }
      if (p.residual) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[it][e] += u2f(rpre[it][0][e]); v[it][4 + e] += u2f(rpre[it][1][e]); }
      }
      if (EXT && x_rhi) {  // residual kept only as a bf16 pair (ResNet identities): v += hi + lo, the sum formed first as in gemm_epilogue()
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t hw = ppre[it][0][e >> 1], lw = ppre[it][1][e >> 1];  // (an absent lo array reads as zeros)
          v[it][e] += u2f((e & 1) ? (hw & 0xffff0000u) : (hw << 16)) + u2f((e & 1) ? (lw & 0xffff0000u) : (lw << 16));
        }
      }
      if (p.act == MVP_ACT_RELU && act_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
        if (x_omask) {
#pragma unroll
          for (int e = 0; e < 8; ++e) mbits[it][e >> 2] |= (v[it][e] > 0.f ? 1u : 0u) << (8 * (e & 3));
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) vp[it][e] = (EXT && mask_mode == 1) ? v[it][e] * keep[e] : v[it][e];
    }
    if (u + 1 < MT) load_next(u + 1);  // issued BEFORE this unit's stores: its data never waits behind them
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{f2u(v[it][0]), f2u(v[it][1]), f2u(v[it][2]), f2u(v[it][3])}, r_o32, ro[it].o32, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{f2u(v[it][4]), f2u(v[it][5]), f2u(v[it][6]), f2u(v[it][7])}, r_o32, ro[it].o32, 16, 0);
      uint32_t hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
      if (has_pair) {
        if (pform == 2) {  // (the compensated fp16 pair: -1 = every column; the Q third of qkv)
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_comp(vp[it][2 * e], vp[it][2 * e + 1], hw[e], lw[e]);
        } else if (pform == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_bf16(vp[it][2 * e], vp[it][2 * e + 1], hw[e], lw[e]);
        } else if (pform == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16_wcomp(vp[it][2 * e], vp[it][2 * e + 1], hw[e], lw[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_bf16(vp[it][2 * e], vp[it][2 * e + 1], hw[e], lw[e]);
        }
      }
      if (MVP_EPI_UNI_NT && nt_pair) {  // (wave-uniform; stores only: both paths leave the same counts behind)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hw[0], hw[1], hw[2], hw[3]}, r_ohi, ro[it].opair, 0, 18);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{lw[0], lw[1], lw[2], lw[3]}, r_olo, ro[it].opair, lo_soff, 18);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{hw[0], hw[1], hw[2], hw[3]}, r_ohi, ro[it].opair, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{lw[0], lw[1], lw[2], lw[3]}, r_olo, ro[it].opair, lo_soff, 0);
      }
      if (EXT) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{mbits[it][0], mbits[it][1]}, r_omask, ro[it].mask, 0, 0);
    }
  }
}

// Which wide-epilogue instantiation serves these arguments (0 = none: the generic epilogue).
__host__ __device__ __forceinline__ int gemm_epilogue_wide_variant(const mvp_gemm_args& p) {
  {  // the wide epilogues address every array through a buffer resource with 32-bit byte offsets: all of them below 2 GiB
    const int64_t lim = 0x7fffff00ll;
    const int64_t rows = p.row_group > 0 ? ((int64_t)(p.M - 1) / p.row_group) * p.row_group_stride + p.row_group_off + p.row_group : p.M;
    if (p.out_f32 && rows * p.ldo * 4 > lim) return 0;
    if (p.out_hi && rows * p.ldob * 2 + 64 > lim) return 0;
    if (p.residual && (p.res_row_mod > 0 ? (int64_t)p.res_row_mod : rows) * p.ldr * 4 > lim) return 0;
    if (p.relu_mask && rows * p.ldm > lim) return 0;
    if ((int64_t)p.N * 4 > lim) return 0;
  }
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
