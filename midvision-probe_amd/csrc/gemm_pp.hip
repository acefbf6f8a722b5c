// bf16x3 MFMA GEMM for large M on gfx950: 256x256 output tile, 8 waves, one workgroup per CU, "ping-pong" main loop.
//
//   Y[M,N] = act(A[M,K] · W[N,K]ᵀ + bias) + residual          (same contract and epilogue as gemm.hip)
//
// Why a second kernel.  gemm.hip's loop (one LDS stage, vmcnt(0) + barrier per k-tile, 2-5 co-resident workgroups) tops out at
// 24-34 % MFMA-pipe utilisation: every LDS-DMA piece costs its SIMD ~5 MFMAs of issue time and a 128x128 bf16x3 tile issues only 6
// MFMAs per piece.  This kernel follows cdna_hip_programming.md §5 ("what does break it"): ~1 workgroup per CU whose LDS-DMA
// prefetch stays in flight ACROSS barriers — all LDS in one array, counted s_waitcnt vmcnt(N) (never 0 in the loop), raw
// s_barrier — on a tile with 12 MFMAs per piece:
//   * tile 256x256, k-step 32; per k-step a CU stages 64 KB (A 256 rows + W 256 rows, hi and lo) and issues 768 MFMAs.
//   * 8 waves = 2 (M) x 4 (N), wave tile 128x64 (acc[4][8] 16x16 accumulators = 128 registers).
//   * the two wave groups (wr = 0 / 1, one wave of each per SIMD) run the SAME stream staggered by one barrier interval:
//     while one wave of a SIMD issues its 48-MFMA cluster (s_setprio 1) its partner reads fragments from LDS and issues
//     LDS-DMA; every barrier swaps the roles.  MFMA issue, LDS reads and DMA issue therefore overlap inside each SIMD.
//   * each k-step has two phases: P1 = rows [0,64) of the wave tile x all 64 columns (reads 8 A + 8 W fragments), P2 = rows
//     [64,128) (reads 8 A fragments, W fragments stay in registers).  So the A half "HA0" and the whole W tile "HB" of a
//     k-step are dead after P1 and HA1 after P2, and with two LDS buffers (by k-step parity) the staging runs 3 phases ahead:
//       P2(t)   issues HA0(t+2) x2, HB(t+2) x4   (per wave; both slots were last read in P1(t))
//       P1(t+1) issues HA1(t+2) x2               (slot last read in P2(t))
//     so the read interval with 16 fragment reads issues 2 pieces and the one with 8 reads issues 6, and every piece has ~1.5
//     k-steps (2.5-3 us) to land — the A operand streams from HBM (~2 us under load; the first version staged HA only one k-step
//     ahead and measured latency-bound: 2.09 us per k-step against 1.64 with the DMA ablated).  The counted wait is vmcnt(8) at the
//     end of every read interval: all but the wave's 8 youngest pieces have landed, which covers exactly what the NEXT phase reads
//     (end of P2(t): HA0(t+1), HB(t+1); end of P1(t): HA1(t)).  The wait precedes a barrier and the reads of that data come after
//     it (RAW); every ds_read is retired by lgkmcnt(0) BEFORE the barrier that ends its interval, so a DMA issued after that
//     barrier cannot overtake it (WAR).
//
// Operand layouts (templates ILVA / ILVW, chosen per operand by mvp_gemm_args.pair_layout):
//   false: separate hi / lo arrays, K contiguous (what the rest of the library produces).  A 32-deep k-step of one array is
//     a 64-byte row segment (half a cache line per LDS-DMA row); LDS image per operand = [hi: 256 rows x 64 B][lo: same],
//     16-row pieces, swizzle chunk ^= {0,2,3,1}[(row >> 2) & 3] (gemm.hip's BK = 32 image).
//   true: ONE array for the operand, hi | lo interleaved per 32-deep k block: row = [k/32][hi 32 | lo 32] bf16, so the k-step
//     of a row is one whole 128-byte line; LDS image [256 rows x 128 B], 8-row pieces, swizzle chunk ^= row & 7 (gemm.hip's
//     BK = 64 image with "k half 0 / 1" = hi / lo).
//
// The accumulation order per output element is gemm.hip's: per 32-deep k-step lo·hi, hi·lo, hi·hi, k ascending — the results are
// bit-identical to the tile kernels'.
#include "gemm_epilogue.h"

#ifndef MVP_PP_ABLATE
#define MVP_PP_ABLATE 0  // diagnostic builds (tools/pp_bench.py): 1 = no MFMA, 2 = no LDS-DMA in the loop, 3 = no fragment reads, 4 = two MFMAs per fragment pair
#endif
// MVP_PP_STAMP 1 (diagnostic build only): s_memtime stamps around the five segments of every phase (issue of reads + LDS-DMA, the
// counted waits, barrier 1, the MFMA cluster, barrier 2), summed per phase type in SGPRs by waves 0 and 4 of every workgroup and
// written to splitk_ws as uint64 [workgroup][group][10] at the end (cdna_hip_programming.md §7, in-kernel stamps).
#ifndef MVP_PP_STAMP
#define MVP_PP_STAMP 0
#endif
// MVP_PP_PRIO (diagnostic A/B, tools/pp_bench.py --prio): 0 = s_setprio 1 / 0 around every MFMA cluster (the shipped form);
// 1 = no per-cluster flips, ONE s_setprio 1 for the later-dispatched wave group (waves 4-7) before the main loop
// (MI355X_MICROARCH.md "Two waves per SIMD", item 4); 2 = no priority instructions at all.
#ifndef MVP_PP_PRIO
#define MVP_PP_PRIO 0
#endif
// MVP_PP_PREFETCH 0 (diagnostic build): persistent over tiles, but every tile runs the cold prologue after the previous epilogue.
#ifndef MVP_PP_PREFETCH
#define MVP_PP_PREFETCH 1
#endif
#ifndef MVP_PP_RELAXED
#define MVP_PP_RELAXED 1  // 0 (diagnostic build): the first phase of a prefetched tile waits with the steady-state count (all stores acknowledged)
#endif
#ifndef MVP_PP_NOLOOP
#define MVP_PP_NOLOOP 0  // 1 (diagnostic build): one tile per workgroup, no tile loop in the code at all
#endif
#if MVP_PP_STAMP
#define PP_STAMP(v) const uint64_t v = __builtin_amdgcn_s_memtime()
#define PP_ACC(ph) \
  do { tacc[ph * 5 + 0] += s_x - s_e; tacc[ph * 5 + 1] += s_b - s_x; tacc[ph * 5 + 2] += s_c - s_b; tacc[ph * 5 + 3] += s_d - s_c; tacc[ph * 5 + 4] += s_f - s_d; s_e = s_f; } while (0)
#else
#define PP_STAMP(v)
#define PP_ACC(ph)
#endif

namespace {

template <int V>
struct ic { static constexpr int value = V; };

constexpr int PP_TILE_B = 256 * 128;     // bytes per operand tile per k-step (hi + lo)
constexpr int PP_BUF_B = 2 * PP_TILE_B;  // A tile + W tile of one k-step
constexpr int PP_SMEM = 2 * PP_BUF_B;    // two k-step buffers = 128 KiB

// CONV (separate hi / lo arrays only): the A operand is the implicit im2col of a channels-last activation, exactly gemm.hip's conv mode
// (K index = tap * C + c, a 32-deep k-step never straddles a tap: C % 32 == 0; padding taps are read as buffer-out-of-range zeros).
//
// Persistent over tiles (round 4).  The grid is min(tiles, CUs) workgroups; workgroup b runs tiles b, b + grid, b + 2 grid, ... — the
// tile a freshly dispatched workgroup of the one-tile-per-workgroup launch would have got in that round, so the XCD-region order (and
// with it what each L2 holds) is unchanged.  What the loop buys at a tile boundary:
//   * the NEXT tile's first k-step (HA0(0), HB(0), HA1(0): 8 pieces per wave) is issued into k-step buffer 0 BEFORE the current tile's
//     epilogue stores: vmcnt retires in issue order, so loads issued ahead of the stores land first and the stores drain behind them;
//     the epilogue's LDS scratch (16 rows x 68 floats per wave, 34 KB) lives in buffer 1 meanwhile.  After the epilogue one barrier
//     (every wave is done with the scratch), then HA0(1), HB(1) go to buffer 1 and the main loop starts on operands that landed long
//     ago — the 7.7-11 k cycles a cold prologue waits for its first k-step (in-kernel stamps, profiles/r03_pp_stamps.txt) are gone for
//     every tile but a workgroup's first;
//   * no workgroup exit / dispatch / kernel-argument load between the tiles of a CU.
// Only the wide epilogues (no other loads in flight but the residual's) prefetch across the epilogue; the generic epilogue (masks, second
// residuals: the DPT convolutions, K >= 1152 there) keeps its 70 KB scratch at the start of LDS and re-runs the cold prologue per tile.
// F16X2 (MVP_PREC_F16X2): two products per fragment pair instead of three — both operands are the compensated fp16 pairs of
// include/mvp_hip.h (activation: hi = fp16(a), lo = fp16(8 (a - hi) + hi / 8); weight: hi = fp16((1 - 2^-6) w), lo = fp16((w + 64 d) / 8)):
// acc += a_lo . w_lo, then acc += a_hi . w_hi, both f16 MFMAs.  Same arrays, layouts, staging and fragment reads (the halves are
// 16-bit either way); per 32-deep k-step lo product first, then hi, k ascending — the order the tile kernels use, so the two families
// stay bit-identical in this mode too.
template <bool ILVA, bool ILVW, bool EXT, bool CONV = false, bool F16X2 = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const mvp_gemm_args p) {
  static_assert(!(CONV && ILVA), "the convolution reads separate hi / lo activation arrays");
  static_assert(!(F16X2 && (CONV || EXT)), "the two-product mode serves the plain linear GEMMs of the ViT blocks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, WM = 128, WN = 64, MT = 8, NT = 4;
#if MVP_PP_STAMP
  const uint64_t t_k0 = __builtin_amdgcn_s_memtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform (SGPR): LDS-DMA destinations, resource choice
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int tiles = tiles_m * tiles_n;
  const int wm0 = wr * WM, wn0 = wc * WN;
  const int nk = p.K >> 5;

  // ---------------------------------------------------------------- staging geometry (per operand: A and W may differ in layout)
  // piece = one wave-instruction of LDS-DMA = 1 KiB of the LDS image: 8 rows x 128 B (interleaved) or 16 rows x 64 B (separate arrays).
  // (lane_s: an opaque copy of the lane id, refreshed per tile — see the note at the epilogue call; the staging geometry is recomputed
  // per tile from it instead of being held in registers across the main loop)
  int lane_s = lane;
  auto prow_of = [&](bool ilv) { return ilv ? (lane_s >> 3) : (lane_s >> 2); };  // row inside the piece
  auto csrc_of = [&](bool ilv) {                                                 // swizzled source chunk, bytes
    const int pr = prow_of(ilv);
    const int csw = ilv ? (pr & 7) : ((0x1320 >> (((pr >> 2) & 3) * 4)) & 3);
    return (((ilv ? (lane_s & 7) : (lane_s & 3)) ^ csw)) << 4;
  };
  constexpr int KSTEP_A = ILVA ? 128 : 64, KSTEP_W = ILVW ? 128 : 64;  // bytes of one k-step inside a source row
  // A pieces of this wave: list index i = 2 * wave + e (e = 0, 1) into the 16 pieces of a half (HA0; HA1 = the next 64 rows)
  int qa[2];  // piece index inside the operand tile image (HA0 piece; HA1 = + 64 rows)
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = 2 * wave + e;
    qa[e] = ILVA ? ((i < 8) ? i : i + 8) : ((i & 3) + ((i >> 2) & 1) * 8 + (i >> 3) * 16);
  }
  constexpr int HALF_STEP = ILVA ? 8 : 4;  // pieces per 64 rows of A
  // W pieces of this wave: 4 consecutive pieces
  auto piece_row0 = [&](bool ilv, int q) { return ilv ? q * 8 : (q & 15) * 16; };  // first tile row of piece q
  auto piece_lo = [&](bool ilv, int q) { return ilv ? 0 : (q >> 4); };             // 1: the piece belongs to the lo array (separate arrays)

  const int cHs = CONV ? (p.cH >> p.cup) : 0, cWs = CONV ? (p.cW >> p.cup) : 0;
  // (conv) the resource covers exactly the activation: an offset past it — what a padding tap gets — reads as zeros
  const unsigned a_bytes = CONV ? (unsigned)min((size_t)0x7fffff00u, ((size_t)(p.M / (p.cHo * p.cWo)) * cHs * cWs * p.lda) * 2) : 0x7fffff00u;

  // ---- state of the tile being STAGED (during an epilogue: already the next tile's)
  int m0 = 0, n0 = 0;
  const mvp_bf16 *pa_hi = p.a_hi, *pa_lo = p.a_hi, *pw_hi = p.w_hi, *pw_lo = p.w_hi;
  int a_voff[4], w_voff[4];  // per-lane byte offsets of this wave's pieces: [HA0 e0, HA0 e1, HA1 e0, HA1 e1], [HB 0..3]
  int cv_img[4], cv_yx[4];   // (conv) per piece row: byte offset of its image, top-left input coordinates (y << 16 | x, biased by 0x4000)
  int ct_ky[2], ct_kx[2], ct_c0[2];  // (conv) tap and channel of the NEXT k-step each half stages (k ascending per half)
  typedef const __attribute__((address_space(4))) mvp_gemm_args kargs_t;  // the arguments, read in place from the kernel-argument segment
  auto setup_tile = [&](int bid) {
    lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    // (the arguments through an opaque pointer, so that the scalars of this block are re-read per tile instead of living — spilled to
    // VGPR lanes — across the main loop)
    kargs_t* kp = (kargs_t*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    kargs_t& p = *kp;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int tiles = tiles_m * tiles_n;
    int tm, tn;
    region_tile(xcd_remap(bid, tiles), tiles_m, tiles_n, p.M, p.N, tiles > 256, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
    const size_t a_base = CONV ? 0 : (size_t)m0 * p.lda, w_base = (size_t)n0 * p.ldw;
    pa_hi = p.a_hi + a_base;
    pa_lo = (ILVA ? p.a_hi : p.a_lo) + a_base;
    pw_hi = p.w_hi + w_base;
    pw_lo = (ILVW ? p.w_hi : p.w_lo) + w_base;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = qa[e & 1] + (e >> 1) * HALF_STEP;
      const int row = min(piece_row0(ILVA, q) + prow_of(ILVA), p.M - 1 - m0);  // rows past M re-read the last row (never stored)
      a_voff[e] = row * p.lda * 2 + csrc_of(ILVA);
      if (CONV) {
        const int m = m0 + row;
        const int x = m % p.cWo, t = m / p.cWo;
        cv_img[e] = (t / p.cHo) * cHs * cWs * p.lda * 2 + csrc_of(false);
        cv_yx[e] = (((t % p.cHo) * p.cstride - p.cpad + 0x4000) << 16) | (x * p.cstride - p.cpad + 0x4000);
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) ct_ky[e] = ct_kx[e] = ct_c0[e] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = 4 * wave + e;
      const int row = min(piece_row0(ILVW, q) + prow_of(ILVW), p.N - 1 - n0);
      w_voff[e] = row * p.ldw * 2 + csrc_of(ILVW);
    }
  };

  // ``live`` = false: the same instructions on a zero-sized resource (every lane out of range: zeros land in the LDS destination, no
  // memory is read) and no state advanced — what the prefetch hook issues when there is no next tile, so that the epilogue's
  // instruction stream holds the same number of memory operations either way (the compiler's counted waits stay exact, see the hook).
  auto stage_a = [&](int par, int kt, int half, bool live = true) {  // this wave's two pieces of HA<half> of k-step kt
#if MVP_PP_ABLATE != 2
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = qa[e] + half * HALF_STEP;
      if (CONV) {
        const int yy = (cv_yx[half * 2 + e] >> 16) - 0x4000 + ct_ky[half], xx = (cv_yx[half * 2 + e] & 0xffff) - 0x4000 + ct_kx[half];
        const bool ok = ((unsigned)yy < (unsigned)p.cH) && ((unsigned)xx < (unsigned)p.cW);
        const int off = cv_img[half * 2 + e] + (((yy >> p.cup) * cWs + (xx >> p.cup)) * p.lda + ct_c0[half]) * 2;
        lds_dma16(piece_lo(ILVA, q) ? pa_lo : pa_hi, live ? a_bytes : 0u, smem + par * PP_BUF_B + q * 1024, ok ? off : 0x7fffff80, 0);
      } else {
        lds_dma16(piece_lo(ILVA, q) ? pa_lo : pa_hi, live ? 0x7fffff00u : 0u, smem + par * PP_BUF_B + q * 1024, a_voff[half * 2 + e], kt * KSTEP_A);
      }
    }
#endif
    if (CONV && live) {  // every half is staged in increasing k order: advance its running tap position
      ct_c0[half] += 32;
      if (ct_c0[half] >= p.cC) {
        ct_c0[half] = 0;
        if (++ct_kx[half] == p.ckw) { ct_kx[half] = 0; ++ct_ky[half]; }
      }
    }
  };
  auto stage_w = [&](int par, int kt, bool live = true) {  // this wave's four pieces of HB of k-step kt
#if MVP_PP_ABLATE != 2
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = 4 * wave + e;
      lds_dma16(piece_lo(ILVW, q) ? pw_lo : pw_hi, live ? 0x7fffff00u : 0u, smem + par * PP_BUF_B + PP_TILE_B + q * 1024, w_voff[e], kt * KSTEP_W);
    }
#endif
  };

  // ---------------------------------------------------------------- fragment reads
  const int frow = lane & 15, fq = lane >> 4;
  // interleaved: byte = row * 128 + (((h * 4 + fq) ^ (row & 7)) << 4);   separate: h * 16 KiB + row * 64 + ((fq ^ sw(row)) << 4)
  auto foff = [&](bool ilv, int h) {
    const int fsw = ilv ? (frow & 7) : ((0x1320 >> (((frow >> 2) & 3) * 4)) & 3);
    return ilv ? (frow * 128 + (((h * 4 + fq) ^ fsw) << 4)) : (h * 16384 + frow * 64 + ((fq ^ fsw) << 4));
  };
  const int fa_hi = foff(ILVA, 0), fa_lo = foff(ILVA, 1), fw_hi = foff(ILVW, 0), fw_lo = foff(ILVW, 1);
  constexpr int AROWB = ILVA ? 128 : 64, WROWB = ILVW ? 128 : 64;  // bytes per LDS row
  const char* const a_rd = smem + wm0 * AROWB;
  const char* const w_rd = smem + PP_TILE_B + wn0 * WROWB;

  f32x4_t acc[NT][MT];
  bf16x8_t a_hi[4], a_lo[4], w_hi[4], w_lo[4];

  auto read_a = [&](int par, int half) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* r = a_rd + par * PP_BUF_B + (half * 4 + j) * 16 * AROWB;
      a_hi[j] = *(const bf16x8_t*)(r + fa_hi);
      a_lo[j] = *(const bf16x8_t*)(r + fa_lo);
    }
  };
  auto read_w = [&](int par) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* r = w_rd + par * PP_BUF_B + i * 16 * WROWB;
      w_hi[i] = *(const bf16x8_t*)(r + fw_hi);
      w_lo[i] = *(const bf16x8_t*)(r + fw_lo);
    }
  };
  auto mma = [&](const int half) {  // 48 MFMAs: 4 n-fragments x 4 m-fragments x (lo·hi, hi·lo, hi·hi); `half` is a literal at every call
#if MVP_PP_ABLATE == 1
#pragma unroll
    for (int j = 0; j < 4; ++j) { asm volatile("" ::"v"(a_hi[j])); asm volatile("" ::"v"(a_lo[j])); asm volatile("" ::"v"(w_hi[j])); asm volatile("" ::"v"(w_lo[j])); }
#else
#if MVP_PP_PRIO == 0
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t c = acc[i][half * 4 + j];
        if constexpr (F16X2) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w_lo[i]), __builtin_bit_cast(f16x8_t, a_lo[j]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w_hi[i]), __builtin_bit_cast(f16x8_t, a_hi[j]), c, 0, 0, 0);
          acc[i][half * 4 + j] = c;
          continue;
        }
#if MVP_PP_ABLATE != 4  // (4: two products per fragment pair instead of three — what a two-product precision mode would issue; wrong results, timing only)
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[i], a_hi[j], c, 0, 0, 0);
#endif
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[i], a_lo[j], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[i], a_hi[j], c, 0, 0, 0);
        acc[i][half * 4 + j] = c;
      }
#if MVP_PP_PRIO == 0
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
  };

#if MVP_PP_STAMP
  uint64_t tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t s_e = 0, s_x = 0, t_pro = 0, t_loop = 0, t_epi = 0, t_mark = t_k0;
#endif
  bool relaxed = false;  // wave-uniform: the tile being computed had its first k-step prefetched across the previous tile's epilogue
  auto kstep = [&](int t, auto PAR, auto FIRST) {
    constexpr int par = decltype(PAR)::value;
    constexpr bool first = decltype(FIRST)::value != 0;  // the peeled k-step 0 (the relaxed wait below must not sit in the loop's code)
    // ---- P1, read interval: fragments of rows [0,64) + all W fragments; stage HA1(t+1)
#if MVP_PP_ABLATE != 3
    read_w(par);
    read_a(par, 0);
#endif
    if (t + 1 < nk) {
      stage_a(par ^ 1, t + 1, 1);
      {
        PP_STAMP(s_x0);
#if MVP_PP_STAMP
        s_x = s_x0;
#endif
      }
      if (MVP_PP_RELAXED && first && relaxed) {
        // first phase of a tile whose first k-step was prefetched across the previous epilogue: HA1(0) is older than that epilogue's
        // >= 32 stores, HA0(1) x2, HB(1) x4 and HA1(1) x2 — the stores get this phase and the next to drain (P2's vmcnt(8) needs them gone)
        asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // HA1(t) landed; HA0(t+1) x2, HB(t+1) x4, HA1(t+1) x2 stay in flight
      }
    } else {
      {
        PP_STAMP(s_x0);
#if MVP_PP_STAMP
        s_x = s_x0;
#endif
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): every fragment read is retired before the barrier (the builtin, so that hipcc counts it)
    __builtin_amdgcn_sched_barrier(0);
    {
      PP_STAMP(s_b);
      __builtin_amdgcn_s_barrier();
      PP_STAMP(s_c);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(s_d);
      __builtin_amdgcn_s_barrier();
      PP_STAMP(s_f);
      PP_ACC(0);
    }
    // ---- P2, read interval: fragments of rows [64,128); stage HA0(t+2), HB(t+2)
#if MVP_PP_ABLATE != 3
    read_a(par, 1);
#endif
    if (t + 2 < nk) {
      stage_a(par, t + 2, 0);
      stage_w(par, t + 2);
    }
    {
      PP_STAMP(s_x0);
#if MVP_PP_STAMP
      s_x = s_x0;
#endif
    }
    if (t + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // HA0(t+1), HB(t+1) landed; HA1(t+1) x2 and this phase's 6 stay in flight
    } else if (t + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // HA1(t+1) x2 may stay in flight
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): every fragment read is retired before the barrier (the builtin, so that hipcc counts it)
    __builtin_amdgcn_sched_barrier(0);
    {
      PP_STAMP(s_b);
      __builtin_amdgcn_s_barrier();
      PP_STAMP(s_c);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(s_d);
      __builtin_amdgcn_s_barrier();
      PP_STAMP(s_f);
      PP_ACC(1);
    }
  };

#ifndef MVP_PP_WIDE_EPILOGUE
#define MVP_PP_WIDE_EPILOGUE 1  // 0 (diagnostic builds): always the generic epilogue
#endif
  const int wide = (!EXT && MVP_PP_WIDE_EPILOGUE) ? gemm_epilogue_wide_variant(p) : 0;  // wave-uniform: kernel arguments only
  // EXT instantiations (masks, pair residuals, post-residual ReLU): the universal branch-free epilogue where it serves the form
  const bool uni = EXT && MVP_PP_WIDE_EPILOGUE && gemm_epilogue_uni_ok(p) && !(p.tile_policy & MVP_TILES_NO_UNI);

  // ---------------------------------------------------------------- first tile: the cold prologue — HA0(0), HB(0), HA1(0), then what "P2(-1)" would issue
  int bid = blockIdx.x;
  setup_tile(bid);
  stage_a(0, 0, 0);
  stage_w(0, 0);
  stage_a(0, 0, 1);
  stage_a(1, 1, 0);  // (nk >= 2: K >= 64, host check)
  stage_w(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // HA0(0), HB(0) landed (this wave's pieces); HA1(0) x2, HA0(1) x2, HB(1) x4 in flight
  __builtin_amdgcn_s_barrier();
#if MVP_PP_PRIO == 1
  if (wr == 1) __builtin_amdgcn_s_setprio(1);
#endif

  for (;;) {
    const int m0c = m0, n0c = n0;  // the tile computed now (the staging state moves on to the next tile before the epilogue)
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier interval behind group 0
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#if MVP_PP_STAMP
    s_e = __builtin_amdgcn_s_memtime();
    t_pro += s_e - t_mark;
    t_mark = s_e;
#endif
    kstep(0, ic<0>{}, ic<1>{});
    int t = 1;
    for (; t + 1 < nk; t += 2) {
      kstep(t, ic<1>{}, ic<0>{});
      kstep(t + 1, ic<0>{}, ic<0>{});
    }
    if (t < nk) kstep(t, ic<1>{}, ic<0>{});
    if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 waits out group 1's last interval
#if MVP_PP_STAMP
    {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      t_loop += now - t_mark;
      t_mark = now;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with the staging buffers -> the epilogue and the next tile's first k-step reuse them

    const int nbid = bid + (int)gridDim.x;
    const bool has_next = nbid < tiles;
    const bool pre = has_next && (wide != 0 || uni) && MVP_PP_PREFETCH;
    if (has_next) setup_tile(nbid);
    // Ahead of the epilogue's stores (vmcnt retires in order), behind its first loads: the next tile's first k-step into buffer 0.
    // Issued UNCONDITIONALLY — without a next tile as 8 dead pieces (zero-sized resource) into buffer 1, which nothing uses then: were
    // the 8 LDS-DMA instructions under a branch, the compiler's wait for the epilogue's first loads (bias, first residual rows; issued
    // BEFORE this hook) would have to assume the path without them, i.e. leave fewer operations in flight than this path has issued
    // behind those loads — and so wait for the prefetch itself, exposing exactly the latency it is there to hide.
    auto prefetch = [&]() {
      const int par = pre ? 0 : 1;
      stage_a(par, 0, 0, pre);
      stage_w(par, 0, pre);
      stage_a(par, 0, 1, pre);
    };
    char* const scratch = smem + (pre ? PP_BUF_B : 0);
    // (an opaque per-tile copy of the lane id: the epilogue's per-lane constants would otherwise be hoisted out of the tile loop and
    // held in registers across the main loop — 250+ VGPRs instead of ~220, and then no other wave fits on the SIMDs beside this kernel's two)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    // (and the kernel arguments read afresh from the kernel-argument segment through an opaque pointer, for the same reason: hoisted
    // out of the tile loop the epilogue's ~60 scalars do not fit beside the main loop's and get spilled to VGPR lanes)
#if MVP_PP_NOLOOP
    const mvp_gemm_args& pe = p;
#else
    kargs_t* kp = (kargs_t*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    kargs_t& pe = *kp;
#endif
    switch (wide) {
      case 1: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_NONE, false, false, true>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      case 2: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_GELU, false, false, true>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      case 3: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_NONE, true, true, false>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      case 4: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_NONE, false, true, false>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      case 5: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_GELU, false, true, false>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      case 6: gemm_epilogue_wide<NT, MT, WN, MVP_ACT_NONE, false, true, false, true>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch); break;
      default:
        if constexpr (EXT) {
          if (uni) {
            gemm_epilogue_uni<NT, MT, WN, true>(pe, acc, scratch, wave, lane_e, m0c, n0c, wm0, wn0, prefetch);
            break;
          }
        }
        gemm_epilogue<NT, MT, WN, EXT>(pe, acc, smem, wave, lane_e, m0c, n0c, wm0, wn0);
        // The generic epilogue guards its rows with branches, and hipcc's wait-count pass must assume that a skipped row leaves that
        // row's bias / residual / mask loads pending: without a wait it can SEE here it would protect their destination registers with
        // an s_waitcnt vmcnt(0) inside the main loop — one drain of the LDS-DMA pipeline per k-step, for every epilogue variant.
        if (has_next) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    }
#if MVP_PP_STAMP
    {
      if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last epilogue's stores have left
      const uint64_t now = __builtin_amdgcn_s_memtime();
      t_epi += now - t_mark;
      t_mark = now;
    }
#endif
    if (!has_next || MVP_PP_NOLOOP) break;
    bid = nbid;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's scratch reads are retired ...
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();        // ... and so are every other wave's: the LDS-DMA below may overwrite the scratch
    if (!pre) {
      stage_a(0, 0, 0);
      stage_w(0, 0);
      stage_a(0, 0, 1);
    }
    stage_a(1, 1, 0);
    stage_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (pre) {
      // HA0(0), HB(0) of this wave must have landed.  They are OLDER than everything the epilogue issued behind the prefetch hook —
      // HA1(0) x2, at least 32 stores (gemm_epilogue_wide_ops::after_hook; more with residual / gate rows) — and than HA0(1) x2,
      // HB(1) x4 just issued: 40 younger operations may stay in flight, the epilogue's stores among them, which drain under the
      // first phases of the next tile instead of being waited for here.
      static_assert(gemm_epilogue_wide_ops<false, false>::after_hook + 2 + 6 == 40 && gemm_epilogue_wide_ops<true, false>::after_hook >= 32 &&
                    gemm_epilogue_wide_ops<false, true>::after_hook >= 32, "the relaxed waits count 32 stores per tile");
      asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    } else {
      // cold path: HA0(0), HB(0) landed; HA1(0) x2, HA0(1) x2, HB(1) x4 stay in flight
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    relaxed = pre;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
#if MVP_PP_STAMP
  const uint64_t t_e1 = __builtin_amdgcn_s_memtime();
  if (p.splitk_ws && wc == 0 && lane == 0) {
    uint64_t* dbg = (uint64_t*)p.splitk_ws + ((size_t)blockIdx.x * 2 + wr) * 16;
    for (int c = 0; c < 10; ++c) dbg[c] = tacc[c];
    dbg[10] = t_pro; dbg[11] = t_loop; dbg[12] = t_epi; dbg[13] = t_k0; dbg[14] = t_e1;
    dbg[15] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// CUs of the device (the persistent grid): read once.
inline int pp_cu_count() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    return cu;
  }();
  return n;
}

inline int pp_grid(const mvp_gemm_args* a) {
  const int tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256);
  // persistent grid: one workgroup per CU (128 KB of LDS each), tiles dealt round by round; MVP_PP_PERSIST=0 (diagnostic): one tile
  // per workgroup, the pre-round-4 launch.  (A multiple of 8 keeps a workgroup's tiles on one XCD slice of the region order.)
  static const int persist = [] { const char* e = getenv("MVP_PP_PERSIST"); return (e ? atoi(e) : 1) && !MVP_PP_NOLOOP; }();
  // MVP_PP_GRID=n (diagnostic): at most n workgroups per launch — a share of the chip, for forward chains running side by side
  static const int cap = [] { const char* e = getenv("MVP_PP_GRID"); return e ? atoi(e) & ~7 : 0; }();
  int cus = pp_cu_count() & ~7;
  if (cap >= 8 && cap < cus) cus = cap;
  return (persist && cus >= 8 && tiles > cus) ? cus : tiles;
}

template <bool ILVA, bool ILVW>
int launch_pp_f16x2(const mvp_gemm_args* a, hipStream_t s) {  // MVP_PREC_F16X2: plain epilogues only (the caller checked)
  static int configured = (int)hipFuncSetAttribute((const void*)gemm_pp_kernel<ILVA, ILVW, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_SMEM);
  if (configured != 0) return MVP_ELAUNCH;
  hipLaunchKernelGGL((gemm_pp_kernel<ILVA, ILVW, false, false, true>), dim3(pp_grid(a)), dim3(512), PP_SMEM, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

template <bool ILVA, bool ILVW, bool CONV = false>
int launch_pp(const mvp_gemm_args* a, hipStream_t s) {
  static int configured = [] {
    int e = (int)hipFuncSetAttribute((const void*)gemm_pp_kernel<ILVA, ILVW, false, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_SMEM);
    if (e == 0) e = (int)hipFuncSetAttribute((const void*)gemm_pp_kernel<ILVA, ILVW, true, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_SMEM);
    return e;
  }();
  if (configured != 0) return MVP_ELAUNCH;
  const int grid = pp_grid(a);
  // The plain instantiation carries the specialised wide epilogues of the ViT blocks (+ the gated input gradient); every other form goes
  // to the EXT instantiation, whose universal branch-free epilogue serves masks, pair residuals, a ReLU before or after the residual ...
  const bool ext = gemm_epilogue_wide_variant(*a) == 0;
  if (ext)
    hipLaunchKernelGGL((gemm_pp_kernel<ILVA, ILVW, true, CONV>), dim3(grid), dim3(512), PP_SMEM, s, *a);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<ILVA, ILVW, false, CONV>), dim3(grid), dim3(512), PP_SMEM, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

}  // namespace

extern "C" int mvp_gemm_pp(const mvp_gemm_args* a, void* stream) {
  if (!a || !a->a_hi || !a->w_hi) return MVP_EINVAL;
  if (a->pair_layout < 0 || a->pair_layout > (MVP_PAIR_A_ILV32 | MVP_PAIR_W_ILV32)) return MVP_EINVAL;
  const bool ilva = a->pair_layout & MVP_PAIR_A_ILV32, ilvw = a->pair_layout & MVP_PAIR_W_ILV32;
  if ((!ilva && !a->a_lo) || (!ilvw && !a->w_lo)) return MVP_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K < 64 || (a->K & 31) || a->splitk > 1) return MVP_EINVAL;
  if (a->conv) {  // gemm.hip's conv contract (mvp_gemm_bias_act_res validates the same)
    if (ilva || a->cC <= 0 || (a->cC & 31) || a->ckh <= 0 || a->ckw <= 0 || a->cstride <= 0 || a->cpad < 0 || a->cup < 0) return MVP_EINVAL;
    if (a->K != a->ckh * a->ckw * a->cC || a->cHo <= 0 || a->cWo <= 0 || (a->M % (a->cHo * a->cWo))) return MVP_EINVAL;
    if ((a->cH & ((1 << a->cup) - 1)) || (a->cW & ((1 << a->cup) - 1)) || a->cH >= 0x4000 || a->cW >= 0x4000 || a->cpad >= 0x2000) return MVP_EINVAL;
    if ((int64_t)(a->M / (a->cHo * a->cWo)) * (a->cH >> a->cup) * (a->cW >> a->cup) * a->lda * 2 >= 0x7fffff00ll) return MVP_EINVAL;
  }
  if ((a->lda & 7) || (a->ldw & 7) || (a->precision != MVP_PREC_BF16X3 && a->precision != MVP_PREC_F16X2)) return MVP_EINVAL;
  const bool f16x2 = a->precision == MVP_PREC_F16X2;
  if (f16x2 && (a->conv || a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi)) return MVP_EINVAL;
  if (!a->out_f32 && !a->out_hi) return MVP_EINVAL;
  if (a->out_pair_layout != MVP_PAIR_SEPARATE && (a->out_pair_layout != MVP_PAIR_A_ILV32 || !a->out_hi || (a->N & 31))) return MVP_EINVAL;
  if (a->out_f16_col0 != 0 && ((a->out_f16_col0 != -1 && ((a->out_f16_col0 < 0 ? -a->out_f16_col0 : a->out_f16_col0) & (a->out_f16_col0 < 0 ? 127 : 63))) || !a->out_hi || (!a->out_lo && a->out_pair_layout == MVP_PAIR_SEPARATE))) return MVP_EINVAL;
  // 32-bit per-lane byte offsets: 256 tile rows of the widest supported row must stay below 2 GiB
  if ((int64_t)256 * a->lda * 2 >= 0x7fffff00ll || (int64_t)256 * a->ldw * 2 >= 0x7fffff00ll) return MVP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#ifdef MVP_PP_ONE  // diagnostic builds: only the interleaved-operand instantiation (fast compiles of experiments)
  return (ilva && ilvw && !a->conv && !f16x2) ? launch_pp<true, true>(a, st) : MVP_EINVAL;
#else
  if (f16x2) {
    if (ilva) return ilvw ? launch_pp_f16x2<true, true>(a, st) : launch_pp_f16x2<true, false>(a, st);
    return ilvw ? launch_pp_f16x2<false, true>(a, st) : launch_pp_f16x2<false, false>(a, st);
  }
  if (a->conv) return ilvw ? launch_pp<false, true, true>(a, st) : launch_pp<false, false, true>(a, st);
  if (ilva) return ilvw ? launch_pp<true, true>(a, st) : launch_pp<true, false>(a, st);
  return ilvw ? launch_pp<false, true>(a, st) : launch_pp<false, false>(a, st);
#endif
}
