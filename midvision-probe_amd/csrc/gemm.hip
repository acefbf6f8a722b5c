// bf16 / split-bf16 MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
//   Y[M,N] = act(A[M,K] · W[N,K]ᵀ + bias) + residual
//
// Both operands are K-contiguous (torch Linear layout), so A and W tiles share one LDS
// image format: [rows][BK bf16], filled with 16-byte global_load_lds (LDS-DMA, no VGPR
// round trip), XOR-swizzled on the SOURCE address so that the ds_read_b128 fragment reads are
// bank-conflict-free (the LDS destination of an LDS-DMA is lane-linear, hence the swizzle
// lives on the global side + the read side):
//   BK = 64 (128-B rows, 8 chunks):  chunk ^= row & 7
//   BK = 32 ( 64-B rows, 4 chunks):  chunk ^= {0,2,3,1}[(row >> 2) & 3]
// BK = 32 halves the LDS footprint, so 2-3 workgroups stay resident per CU (the hot-path
// GEMMs have only 150-900 tiles: residency, not the MFMA pipe, is the first limiter) and one
// workgroup's epilogue overlaps another's main loop.
//
// The MFMA is issued "swapped": the W fragment is the A operand and the activation fragment
// the B operand, so each lane's 4 accumulator registers are 4 CONSECUTIVE output columns of
// one output row.  The epilogue goes through LDS (per-wave private region, reusing the
// staging buffers): accumulators are written as float4, read back row-contiguous, and stored
// as whole 128/256-byte row segments (8-byte-per-lane scattered stores measured 1.2 TB/s and
// cost 36 % of the fc1 GEMM; full-line stores remove that).
//
// SPLIT == 3 runs hi*hi + hi*lo + lo*hi (three MFMAs per tile) on bf16 pairs: operand error
// ~2^-17 instead of 2^-9, which is what the reference's 1e-3 feature parity needs.
#include "mvp_common.h"

// Diagnostic builds only (tools/gemm_bench.py compiles separate .so files with these set):
//   MVP_ABLATE 1: no MFMA (fragments read and kept live), 2: no LDS-DMA after the first tile,
//   3: LDS-DMA + barriers only, 4: epilogue only.  The shipped library is built with 0.
#ifndef MVP_ABLATE
#define MVP_ABLATE 0
#endif
// MVP_S1_REGBUF 1: single-stage loop double-buffered through registers (fragments of tile kt read out, then the DMA of
// tile kt+1 flies under the MFMAs).  Measured +-3 % against the plain loop (the extra VGPRs cost a resident
// workgroup on the 64x128 / 64x64 tiles), so the shipped default is the plain loop.
#ifndef MVP_S1_REGBUF
#define MVP_S1_REGBUF 0
#endif

namespace {

// split-K workspace = [tile counters: fixed 64 KiB region][fp32 partial tiles]; the fixed counter region keeps the
// counters of GEMMs of different shapes that share one workspace apart from each other's partials.
constexpr int SPLITK_CTR_BYTES = 65536;

// Branch-free erf GELU: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7), one v_exp + one
// v_rcp.  (ocml erff measured ~17 us of VALU on the fc1 epilogue.)
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-ax * ax);
  const float erf_abs = 1.0f - poly * e;
  const float erfv = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erfv);
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

// CONV: the A operand is an implicit im2col of a channels-last activation [B, H, W, C]
// (optionally a virtual nearest-upsample by 2^cup of the stored tensor): K index =
// (ky*kw + kx)*C + c, output row m = (b, y, x) on the Ho x Wo grid.  A K-tile never straddles
// a tap (C % BK == 0), so per tile every staged row is ONE 16-byte-aligned run of channels of
// one source pixel — or of the caller's zero page when the tap falls into the padding.
//
// KSPLIT: split-K for GEMMs whose tile count cannot fill 256 CUs (N = 768 projections at M ~ 3k rows: 300 tiles,
// so 44 CUs carry two tiles and set the time).  The grid is tiles x S units; a unit accumulates K-range `split` of
// its tile, stores the raw accumulators to the workspace and bumps the tile's counter; the unit that arrives LAST
// (no unit ever waits) sums all S partials in the fixed order 0..S-1 (deterministic whichever unit does it) and
// runs the normal fused epilogue.  The counter resets itself for the next launch.
// NW = waves per workgroup (4: 2x2 wave grid; 8: 4x2, i.e. the same tile cut into more, smaller wave tiles).
// WNW = waves along N (the wave grid is (NW / WNW) x WNW).
template <int BM, int BN, int BK, int SPLIT, int NSTAGE, bool CONV, bool EXT, bool KSPLIT = false, int NW = 4, int WNW = 2>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(const mvp_gemm_args p) {
  static_assert(!(KSPLIT && (CONV || EXT)), "split-K is compiled for the plain linear GEMMs only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int ROWB = BK * 2;            // bytes per LDS row
  constexpr int CH = ROWB / 16;           // 16-byte chunks per row (8 or 4)
  constexpr int RPP = 1024 / ROWB;        // rows per 1-KiB LDS-DMA piece (8 or 16)
  constexpr int A_BYTES = BM * ROWB;
  constexpr int W_BYTES = BN * ROWB;
  constexpr int STAGE = (A_BYTES + W_BYTES) * NARR;
  constexpr int WM = BM / (NW / WNW), WN = BN / WNW;  // per-wave output tile
  constexpr int NT_THREADS = NW * 64;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int KS = BK / 32;              // MFMA k-steps per tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int unit = xcd_remap(blockIdx.x, gridDim.x);  // units of one tile are neighbours -> same XCD / L2
  const int S = KSPLIT ? p.splitk : 1;
  const int bid = KSPLIT ? unit / S : unit;
  const int split = KSPLIT ? unit - bid * S : 0;
  int tm, tn;
  constexpr int SLOTS = 160 * 1024 / (NSTAGE * STAGE) > 8 ? 8 : 160 * 1024 / (NSTAGE * STAGE);  // resident workgroups per CU
  const int tiles_m = (p.M + BM - 1) / BM;
  region_tile(bid, tiles_m, tiles_n, p.M, p.N, tiles_m * tiles_n > 256 * SLOTS, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm0 = (wave / WNW) * WM, wn0 = (wave % WNW) * WN;
  const int nk_all = p.K / BK;
  const int kt0 = KSPLIT ? (int)(((long)split * nk_all) / S) : 0;
#if MVP_ABLATE == 4
  const int nk = 0;
#else
  const int nk = KSPLIT ? (int)(((long)(split + 1) * nk_all) / S) - kt0 : nk_all;
#endif

  const int rsub = lane / CH;  // row inside an LDS-DMA piece
  const int csw = (BK == 64) ? (rsub & 7) : ((0x1320 >> (((rsub >> 2) & 3) * 4)) & 3);
  const int csrc = (((lane % CH) ^ csw)) << 3;  // swizzled source chunk, in elements

  constexpr int APASS = BM / (NW * RPP);
  static_assert(BM % (NW * RPP) == 0 && BN % (NW * RPP) == 0 && (WM / 16) % 2 == 0, "tile / wave-count combination");
  int cvb[APASS], cvy[APASS], cvx[APASS];  // conv: image index and top-left input coords of this lane's rows
  int ct_ky = 0, ct_kx = 0, ct_c0 = 0;     // conv: (tap, channel) position of the next tile to stage
  if (CONV) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int m = min(m0 + ps * NW * RPP + wave * RPP + rsub, p.M - 1);
      const int x = m % p.cWo, t = m / p.cWo;
      cvb[ps] = t / p.cHo;
      cvy[ps] = (t % p.cHo) * p.cstride - p.cpad;
      cvx[ps] = x * p.cstride - p.cpad;
    }
  }
  const int cHs = CONV ? (p.cH >> p.cup) : 0, cWs = CONV ? (p.cW >> p.cup) : 0;

  // Operand staging = LDS-DMA in the BUFFER form (buffer_load_dwordx4 ... lds): an SGPR resource per operand array, a 32-bit
  // per-lane byte offset that is computed ONCE (row * ld + swizzled chunk) and the k offset of the tile in an SGPR.  Against
  // global_load_lds with 64-bit per-lane addresses this removes the per-tile address arithmetic and costs the issuing SIMD less
  // (tools/micro/fill_bench.hip: +5...9 % fill rate with MFMAs issued beside it); out-of-range offsets return zeros, which is how
  // the convolution's padding taps are read (no zero page, no pointer select).  Resources are rebased to the tile's first row
  // (linear) so that offsets stay far below 2^31 for any M.
  const size_t a_base = CONV ? 0 : (size_t)m0 * p.lda;
  const unsigned a_bytes = CONV ? (unsigned)min((size_t)0x7fffff00u, ((size_t)(p.M / (p.cHo * p.cWo)) * cHs * cWs * p.lda) * 2) : 0x7fffff00u;
  // (the resource descriptors are rebuilt from these base pointers inside the lambda: a lambda capturing a variable of the opaque
  // __amdgpu_buffer_rsrc_t type makes the HOST pass drop the whole kernel template without a diagnostic — ROCm 7.2 clang)
  mvp_bf16* const pa_hi = (mvp_bf16*)p.a_hi + a_base;
  mvp_bf16* const pa_lo = (mvp_bf16*)(SPLIT == 3 ? p.a_lo : p.a_hi) + a_base;
  const size_t w_base = (size_t)n0 * p.ldw;
  mvp_bf16* const pw_hi = (mvp_bf16*)p.w_hi + w_base;
  mvp_bf16* const pw_lo = (mvp_bf16*)(SPLIT == 3 ? p.w_lo : p.w_hi) + w_base;
  int a_voff[APASS], w_voff[BN / (NW * RPP)];
#pragma unroll
  for (int ps = 0; ps < APASS; ++ps) {
    const int grow = min(ps * NW * RPP + wave * RPP + rsub, p.M - 1 - m0);  // rows past M re-read the last row (never stored)
    a_voff[ps] = CONV ? 0 : (grow * p.lda + csrc) * 2;
  }
#pragma unroll
  for (int ps = 0; ps < BN / (NW * RPP); ++ps) {
    const int grow = min(ps * NW * RPP + wave * RPP + rsub, p.N - 1 - n0);
    w_voff[ps] = (grow * p.ldw + csrc) * 2;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
    const int k0b = (kt0 + kt) * BK * 2;  // byte offset of the k-tile inside a row: wave-uniform (SGPR)
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int r = ps * NW * RPP + wave * RPP;
      if (CONV) {
        const int yy = cvy[ps] + ct_ky, xx = cvx[ps] + ct_kx;
        const bool ok = ((unsigned)yy < (unsigned)p.cH) && ((unsigned)xx < (unsigned)p.cW);
        const int off = ((((cvb[ps] * cHs + (yy >> p.cup)) * cWs + (xx >> p.cup)) * p.lda) + ct_c0 + csrc) * 2;
        const int voff = ok ? off : 0x7fffff80;  // padding tap: beyond num_records -> the load returns zeros
        lds_dma16(pa_hi, a_bytes, base + r * ROWB, voff, 0);
        if (SPLIT == 3) lds_dma16(pa_lo, a_bytes, base + A_BYTES + r * ROWB, voff, 0);
      } else {
        lds_dma16(pa_hi, a_bytes, base + r * ROWB, a_voff[ps], k0b);
        if (SPLIT == 3) lds_dma16(pa_lo, a_bytes, base + A_BYTES + r * ROWB, a_voff[ps], k0b);
      }
    }
    if (CONV) {  // tiles are staged in increasing kt order: advance the running tap position
      ct_c0 += BK;
      if (ct_c0 >= p.cC) {
        ct_c0 = 0;
        if (++ct_kx == p.ckw) { ct_kx = 0; ++ct_ky; }
      }
    }
    char* wb = base + A_BYTES * NARR;
#pragma unroll
    for (int ps = 0; ps < BN / (NW * RPP); ++ps) {
      const int r = ps * NW * RPP + wave * RPP;
      lds_dma16(pw_hi, 0x7fffff00u, wb + r * ROWB, w_voff[ps], k0b);
      if (SPLIT == 3) lds_dma16(pw_lo, 0x7fffff00u, wb + W_BYTES + r * ROWB, w_voff[ps], k0b);
    }
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment reads: row = tile row + (lane & 15), chunk = ks*4 + (lane >> 4), XOR swizzle(row)
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int fsw = (BK == 64) ? (lane & 7) : ((0x1320 >> ((((lane & 15) >> 2) & 3) * 4)) & 3);

  // Software pipeline: NSTAGE LDS buffers, NSTAGE-1 tiles of LDS-DMA in flight.  Per iteration:
  // counted vmcnt (this wave's pieces of tile kt have landed; younger tiles stay in flight) ->
  // raw s_barrier (everybody's pieces landed AND everybody finished reading tile kt-1, whose
  // buffer is the one restaged next) -> issue tile kt+NSTAGE-1 -> MFMAs on tile kt.
  // __syncthreads() is NOT used in the loop: its fence would drain the in-flight LDS-DMA.
  constexpr int LOADS = (BM / (NW * RPP) + BN / (NW * RPP)) * NARR;  // LDS-DMA instructions per wave per tile
#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nk) stage(s0, s0);

  auto mma_tile = [&](const char* base) {
    const char* ab = base + (wm0 + frow) * ROWB;
    const char* wb = base + A_BYTES * NARR + (wn0 + frow) * ROWB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int coff = (((ks << 2) + fq) ^ fsw) << 4;
      bf16x8_t a_hi[MT], w_hi[NT];
      bf16x8_t a_lo[MT], w_lo[NT];
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        a_hi[j] = *(const bf16x8_t*)(ab + j * 16 * ROWB + coff);
        if (SPLIT == 3) a_lo[j] = *(const bf16x8_t*)(ab + A_BYTES + j * 16 * ROWB + coff);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        w_hi[i] = *(const bf16x8_t*)(wb + i * 16 * ROWB + coff);
        if (SPLIT == 3) w_lo[i] = *(const bf16x8_t*)(wb + W_BYTES + i * 16 * ROWB + coff);
      }
#if MVP_ABLATE == 1
#pragma unroll
      for (int j = 0; j < MT; ++j) { asm volatile("" ::"v"(a_hi[j])); if (SPLIT == 3) asm volatile("" ::"v"(a_lo[j])); }
#pragma unroll
      for (int i = 0; i < NT; ++i) { asm volatile("" ::"v"(w_hi[i])); if (SPLIT == 3) asm volatile("" ::"v"(w_lo[i])); }
      continue;
#endif
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if (SPLIT == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo[i], a_hi[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[i], a_lo[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi[i], a_hi[j], acc[i][j], 0, 0, 0);
        }
    }
    };

  if constexpr (NSTAGE == 1) {
    // Single LDS buffer (BK = 64: whole 128-byte lines per LDS-DMA row -> half the L2 requests of BK = 32); the 2-5
    // co-resident workgroups of a CU overlap each other's load and MFMA phases.
#if MVP_S1_REGBUF
    bf16x8_t fa_hi[KS][MT], fa_lo[KS][MT], fw_hi[KS][NT], fw_lo[KS][NT];
    const char* ab = smem + (wm0 + frow) * ROWB;
    const char* wb = smem + A_BYTES * NARR + (wn0 + frow) * ROWB;
    if (nk > 0) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile kt has landed (everybody's pieces)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int coff = (((ks << 2) + fq) ^ fsw) << 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          fa_hi[ks][j] = *(const bf16x8_t*)(ab + j * 16 * ROWB + coff);
          if (SPLIT == 3) fa_lo[ks][j] = *(const bf16x8_t*)(ab + A_BYTES + j * 16 * ROWB + coff);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          fw_hi[ks][i] = *(const bf16x8_t*)(wb + i * 16 * ROWB + coff);
          if (SPLIT == 3) fw_lo[ks][i] = *(const bf16x8_t*)(wb + W_BYTES + i * 16 * ROWB + coff);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everybody holds its fragments: the buffer is free
      if (kt + 1 < nk) stage(0, kt + 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            if (SPLIT == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw_lo[ks][i], fa_hi[ks][j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw_hi[ks][i], fa_lo[ks][j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw_hi[ks][i], fa_hi[ks][j], acc[i][j], 0, 0, 0);
          }
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody finished reading the buffer
      }
      stage(0, kt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mma_tile(smem);
    }
#endif
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NSTAGE - 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if MVP_ABLATE != 2
    if (kt + NSTAGE - 1 < nk) stage((kt + NSTAGE - 1) % NSTAGE, kt + NSTAGE - 1);
#endif
#if MVP_ABLATE == 3
    continue;
#endif
    mma_tile(smem + (kt % NSTAGE) * STAGE);
  }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();  // every wave is done with the staging buffers -> reuse them for the epilogue

  if (KSPLIT && S > 1) {
    // Cross-workgroup hand-over: partials leave with 16-byte `sc1` (write-through) stores that the storing wave waits for (vmcnt(0));
    // after the workgroup barrier ONE lane runs an agent-scope release and the workgroup's ONE agent-scope atomic add; only the
    // workgroup whose add returned S-1 (it came last) pays one agent-scope acquire, then reads all partials with `sc1` loads
    // (MI355X_MICROARCH.md, "Valid forms").  A release per workgroup is affordable here: split-K serves few-tile, long-K shapes only.
    int* ctr = (int*)p.splitk_ws;
    constexpr int NACC = NT * MT;
    constexpr int UNIT_BYTES = NACC * NT_THREADS * 16;  // [acc register][thread] float4: 1-KiB wave accesses
    char* part = (char*)p.splitk_ws + SPLITK_CTR_BYTES;
    constexpr int AUX_SC1 = 16;  // gfx940+ cache policy bit 4
    {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part + (size_t)unit * UNIT_BYTES, 0, UNIT_BYTES, 0x00020000);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, ((i * MT + j) * NT_THREADS + tid) * 16, 0, AUX_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
      // Agent-scope RELEASE before the counter add: the architecturally specified hand-over.  (Round 1 relied on the sc1 stores'
      // vmcnt(0) alone, measured valid on an idle chip; the release was added while pipelined runs were being bisected — the
      // culprit turned out to be packed fp32 elsewhere, csrc/Makefile — and stays: 1500 trajectories beside other chains are exact.)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait: MI355X_MICROARCH.md, compiler hazard)
      s_last = (__hip_atomic_fetch_add(ctr + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // Consumer side, MI355X_MICROARCH.md "Valid forms": ONE relaxed RMW told this workgroup it came last -> ONE agent-scope
    // acquire (invalidates this CU's vector L1) -> vmcnt(0) -> workgroup barrier -> loads.  The measured sc1-only hand-off
    // (no acquire) covers one workgroup per CU; several of these are resident per CU, so the acquire stays.  Producer side
    // needs no agent release because every handed-off byte left with an sc1 (write-through) store that its wave waited for
    // before the barrier that precedes the counter add (conditions (2) and (3) there).
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int sp = 0; sp < S; ++sp) {  // fixed order 0..S-1 whichever unit reduces -> bit-reproducible
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part + ((size_t)bid * S + sp) * UNIT_BYTES, 0, UNIT_BYTES, 0x00020000);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const f32x4_t t = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * MT + j) * NT_THREADS + tid) * 16, 0, AUX_SC1));
          acc[i][j] = (sp == 0) ? t : acc[i][j] + t;
        }
    }
    if (tid == 0) __hip_atomic_store(ctr + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }

  // ---------------------------------------------------------------- epilogue through LDS
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
        const size_t o = (size_t)orow * p.ldob + ncol;
        if (vec_ok && ((p.ldob & 3) == 0)) {
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

template <int BM, int BN, int BK, int SPLIT, int NSTAGE, int NW = 4, int WNW = 2>
constexpr int gemm_smem() {
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int stages = NSTAGE * (BM + BN) * BK * 2 * NARR;
  constexpr int epi = NW * 32 * (BN / WNW + 4) * 4;
  return stages > epi ? stages : epi;
}

template <int BM, int BN, int BK, int SPLIT, int NSTAGE, bool CONV = false, int NW = 4, int WNW = 2>
int launch_gemm(const mvp_gemm_args* a, hipStream_t s) {
  constexpr int SMEM = gemm_smem<BM, BN, BK, SPLIT, NSTAGE, NW, WNW>();
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static int configured = [] {
    int e = (int)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, CONV, false, false, NW, WNW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == 0) e = (int)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, CONV, true, false, NW, WNW>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    return e;
  }();
  if (configured != 0) return MVP_ELAUNCH;
  const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  const bool ext = a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi;
  if (ext)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, CONV, true, false, NW, WNW>), dim3(tiles), dim3(NW * 64), SMEM, s, *a);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, CONV, false, false, NW, WNW>), dim3(tiles), dim3(NW * 64), SMEM, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

template <int BM, int BN>
int64_t splitk_ws_bytes(int M, int N, int S) {
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return SPLITK_CTR_BYTES + tiles * S * (int64_t)(BM * BN * 4);
}

template <int BM, int BN, int BK, int SPLIT, int NSTAGE>
int launch_gemm_splitk(const mvp_gemm_args* a, hipStream_t s) {
  constexpr int SMEM = gemm_smem<BM, BN, BK, SPLIT, NSTAGE>();
  static int configured = [] {
    return (int)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, false, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }();
  if (configured != 0) return MVP_ELAUNCH;
  if (!a->splitk_ws || a->splitk_ws_bytes < splitk_ws_bytes<BM, BN>(a->M, a->N, a->splitk)) return MVP_EINVAL;
  const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
  if (tiles > SPLITK_CTR_BYTES / 4) return MVP_EINVAL;
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, SPLIT, NSTAGE, false, false, true>), dim3(tiles * a->splitk), dim3(256), SMEM, s, *a);
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}

// One tile rule for the split-K path (the workspace query must agree with the launch).
inline bool splitk_wide(int N) { return N >= 1024; }

}  // namespace

extern "C" int64_t mvp_gemm_splitk_workspace_bytes(int M, int N, int splits) {
  if (M <= 0 || N <= 0 || splits < 1) return 0;
  return splitk_wide(N) ? splitk_ws_bytes<128, 128>(M, N, splits) : splitk_ws_bytes<128, 64>(M, N, splits);
}

// Diagnostic override (tools/gemm_bench.py): -DMVP_F_BM=.. -DMVP_F_BN=.. -DMVP_F_BK=.. -DMVP_F_ST=..
extern "C" int mvp_gemm_streamk(const mvp_gemm_args* a, void* stream);  // gemm_sk.hip

extern "C" int mvp_gemm_bias_act_res(const mvp_gemm_args* a, void* stream) {
  if (!a || !a->a_hi || !a->w_hi) return MVP_EINVAL;
  if (a->splitk == MVP_GEMM_STREAMK) return mvp_gemm_streamk(a, stream);
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K & (a->conv ? 31 : 63))) {
    // the one non-conv exception: K % 32 == 0 through the BK = 32 two-stage tile (ResNet stem: K = 147 padded to 160)
    if (!(a && !a->conv && a->M > 0 && a->N > 0 && a->K > 0 && (a->K & 31) == 0 && a->precision == MVP_PREC_BF16X3 && a->splitk <= 1)) return MVP_EINVAL;
  }
  if ((a->lda & 7) || (a->ldw & 7)) return MVP_EINVAL;  // 16-byte aligned rows for LDS-DMA
  if (a->precision == MVP_PREC_BF16X3 && (!a->a_lo || !a->w_lo)) return MVP_EINVAL;
  if (a->precision != MVP_PREC_BF16 && a->precision != MVP_PREC_BF16X3) return MVP_EINVAL;
  if (!a->out_f32 && !a->out_hi) return MVP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool x3 = a->precision == MVP_PREC_BF16X3;
  if (a->conv) {
    if (!a->zero_page || a->cC <= 0 || (a->cC & 31) || a->ckh <= 0 || a->ckw <= 0 || a->cstride <= 0) return MVP_EINVAL;
    if (a->K != a->ckh * a->ckw * a->cC || a->cHo <= 0 || a->cWo <= 0 || (a->M % (a->cHo * a->cWo))) return MVP_EINVAL;
    if ((a->cH & ((1 << a->cup) - 1)) || (a->cW & ((1 << a->cup) - 1))) return MVP_EINVAL;
    // the buffer-form staging addresses the activation with 32-bit byte offsets (out-of-range = padding): it must stay below 2 GiB
    if ((int64_t)(a->M / (a->cHo * a->cWo)) * (a->cH >> a->cup) * (a->cW >> a->cup) * a->lda * 2 >= 0x7fffff00ll) return MVP_EINVAL;
    // A K-tile must stay inside one tap: BK = 64 (whole-line rows, single stage: see the tile notes below) when
    // C % 64 == 0, else BK = 32 (any C % 32 == 0), two stages.
    if (x3 && (a->cC & 63) == 0) {
      if (a->N <= 64) return launch_gemm<128, 64, 64, 3, 1, true>(a, s);
      // few-tile, long-K convolutions (ResNet layer3 / layer4 3x3 at 30^2 / 15^2: 226 / 116 tiles of 128x128):
      // smaller tiles so that every CU holds several workgroups of the single-stage loop
      const long c128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128);
      // (tile_policy is not consulted here: 128x128 for every convolution with >= 50 such tiles while ResNet forwards share the chip
      // measured 4638 img/s against 4565 with this rule — not worth a second rule)
      if (c128 >= 300) return launch_gemm<128, 128, 64, 3, 1, true, 8>(a, s);
      if (c128 >= 160) return launch_gemm<64, 128, 64, 3, 1, true>(a, s);
      return launch_gemm<64, 64, 64, 3, 1, true>(a, s);
    }
    if (a->N > 64) return x3 ? launch_gemm<128, 128, 32, 3, 2, true>(a, s) : launch_gemm<128, 128, 32, 1, 2, true>(a, s);
    return x3 ? launch_gemm<128, 64, 32, 3, 2, true>(a, s) : launch_gemm<128, 64, 32, 1, 2, true>(a, s);
  }
  if (a->K & 63) return launch_gemm<128, 64, 32, 3, 2>(a, s);
  const bool ext = a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi;
  if (a->splitk > 1 && !ext) {  // (with the ReLU-gate / second-residual epilogues the request is ignored)
    if (a->splitk > 64 || a->K / 64 < a->splitk) return MVP_EINVAL;
    if (splitk_wide(a->N)) return x3 ? launch_gemm_splitk<128, 128, 64, 3, 1>(a, s) : launch_gemm_splitk<128, 128, 64, 1, 2>(a, s);
    return x3 ? launch_gemm_splitk<128, 64, 64, 3, 1>(a, s) : launch_gemm_splitk<128, 64, 64, 1, 2>(a, s);
  }
#ifdef MVP_F_BM
#ifndef MVP_F_NW
#define MVP_F_NW 4
#endif
#ifndef MVP_F_WNW
#define MVP_F_WNW 2
#endif
  return x3 ? launch_gemm<MVP_F_BM, MVP_F_BN, MVP_F_BK, 3, MVP_F_ST, false, MVP_F_NW, MVP_F_WNW>(a, s)
            : launch_gemm<MVP_F_BM, MVP_F_BN, MVP_F_BK, 1, MVP_F_ST, false, MVP_F_NW, MVP_F_WNW>(a, s);
#else
  // Tile choice, split mode (measured on MI355X, tools/gemm_bench.py --tiles at M = 3152 and M = 12608; us, old -> new):
  //   qkv 41.3 -> 37.2 / 150.8 -> 138.6, proj 21.1 -> 17.2 / 63.7 -> 49.5, fc1 62.7 -> 51.3 / 193.8 -> 180.6,
  //   fc2 67.3 -> 59.1 / 208.6 -> 173.4.
  // What the counters said (profiles/r01_pmc_*): the bf16-pair GEMMs sit at the L2 REQUEST rate (TCC_REQ x 64 B =
  // 8.4 TB/s = half the 128-B-line ceiling) with BK = 32, whose LDS-DMA rows are half lines.  BK = 64 fetches whole
  // lines; keeping ONE LDS stage (no intra-workgroup prefetch) keeps 2-5 workgroups resident per CU, and those
  // overlap each other's load and MFMA phases better than a second stage would (two stages at BK = 64 cost the
  // residency and measured slower everywhere except the few-tile probe head).
  const long t128 = (long)((a->M + 127) / 128) * ((a->N + 127) / 128);
  // Other kernel chains share the chip (mvp_hip.h, MVP_TILES_SHARED): the big tile's lower SIMD time per output wins once the CUs its
  // coarse grid leaves idle are filled by someone else.  Measured (bench.py, B = 16, 224^2, img/s, 64x64-family rule -> 128x128 everywhere):
  // one chain 5517 -> 4382, two chains in flight 6384 -> 6160, three 6609 -> 7228 (128x64 for N < 1024 instead: 7102; 256x128 with
  // 8 waves for N >= 1024: 6504, for every GEMM: 6210 — one 96 KB workgroup per CU leaves no room for a second chain's workgroup).
  // MVP_GEMM_BIG (diagnostic override): 1 / 2 force 128x128 / 128x64-below-1024 whatever the policy, 0 forces the ALONE rule.
  static const int big_env = [] { const char* e = getenv("MVP_GEMM_BIG"); return e ? atoi(e) : -1; }();
  const int big = big_env >= 0 ? big_env : (a->tile_policy == MVP_TILES_SHARED ? 1 : 0);
  if (x3 && big && !(a->N <= 256 && a->K >= 2048)) {
    if (a->N >= 1024 || big == 1) return launch_gemm<128, 128, 64, 3, 1, false, 8>(a, s);
    return launch_gemm<128, 64, 64, 3, 1>(a, s);
  }
  if (x3) {
    if (a->N <= 256 && a->K >= 2048) {  // probe head: few tiles, long K
      if (a->M >= 8192) return launch_gemm<128, 64, 64, 3, 1>(a, s);
      return launch_gemm<64, 64, 64, 3, 2>(a, s);
    }
    if (a->N >= 1024) {
      // one round of 128x128 tiles (2 resident per CU = 512 slots) or many rounds: big tiles; in between the
      // second, mostly empty round costs more than the smaller tile's extra operand traffic
      // 8 waves (4x2) on the 128x128 tile: same LDS / residency, 4 waves per SIMD hide the load phase better
      // (M = 12608: qkv 136.0 -> 128.4 us, fc1 168.2 -> 164.5; M = 3152: 39.0 -> 38.4)
      if (t128 <= 512 || t128 >= 1536) return launch_gemm<128, 128, 64, 3, 1, false, 8>(a, s);
      return launch_gemm<64, 128, 64, 3, 1>(a, s);
    }
    const long t64 = (long)((a->M + 63) / 64) * ((a->N + 63) / 64);
    if (t64 <= 1280) return launch_gemm<64, 64, 64, 3, 1>(a, s);  // 5 resident per CU
    return launch_gemm<128, 64, 64, 3, 1>(a, s);
  }
  if (t128 >= 400) return launch_gemm<128, 128, 64, 1, 2>(a, s);
  return launch_gemm<64, 64, 64, 1, 2>(a, s);
#endif
}
