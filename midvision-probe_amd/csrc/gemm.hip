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
#include "gemm_epilogue.h"

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
  constexpr int NARR = (SPLIT >= 2) ? 2 : 1;  // SPLIT 3: bf16 pairs, three products; 2 (MVP_PREC_F16X2): fp16 hi + bf16 lo activations, two products
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
  mvp_bf16* const pa_lo = (mvp_bf16*)(SPLIT >= 2 ? p.a_lo : p.a_hi) + a_base;
  const size_t w_base = (size_t)n0 * p.ldw;
  mvp_bf16* const pw_hi = (mvp_bf16*)p.w_hi + w_base;
  mvp_bf16* const pw_lo = (mvp_bf16*)(SPLIT >= 2 ? p.w_lo : p.w_hi) + w_base;
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
        if (SPLIT >= 2) lds_dma16(pa_lo, a_bytes, base + A_BYTES + r * ROWB, voff, 0);
      } else {
        lds_dma16(pa_hi, a_bytes, base + r * ROWB, a_voff[ps], k0b);
        if (SPLIT >= 2) lds_dma16(pa_lo, a_bytes, base + A_BYTES + r * ROWB, a_voff[ps], k0b);
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
      if (SPLIT >= 2) lds_dma16(pw_lo, 0x7fffff00u, wb + W_BYTES + r * ROWB, w_voff[ps], k0b);
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
        if (SPLIT >= 2) a_lo[j] = *(const bf16x8_t*)(ab + A_BYTES + j * 16 * ROWB + coff);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        w_hi[i] = *(const bf16x8_t*)(wb + i * 16 * ROWB + coff);
        if (SPLIT >= 2) w_lo[i] = *(const bf16x8_t*)(wb + W_BYTES + i * 16 * ROWB + coff);
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
          if (SPLIT == 2) {  // MVP_PREC_F16X2: two fp16 products, lo . lo then hi . hi (the order gemm_pp.hip uses)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w_lo[i]), __builtin_bit_cast(f16x8_t, a_lo[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w_hi[i]), __builtin_bit_cast(f16x8_t, a_hi[j]), acc[i][j], 0, 0, 0);
            continue;
          }
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

  if constexpr ((WN == 64 || WN == 32) && NT == WN / 16) {
    // the universal branch-free epilogue wherever its alignment / size conditions hold (same bits as the row-guarded one)
    if (gemm_epilogue_uni_ok(p) && !(p.tile_policy & MVP_TILES_NO_UNI)) {
      gemm_epilogue_uni<NT, MT, WN, EXT>(p, acc, smem, wave, lane, m0, n0, wm0, wn0, [] {});
      return;
    }
  }
  gemm_epilogue<NT, MT, WN, EXT>(p, acc, smem, wave, lane, m0, n0, wm0, wn0);
}

template <int BM, int BN, int BK, int SPLIT, int NSTAGE, int NW = 4, int WNW = 2>
constexpr int gemm_smem() {
  constexpr int NARR = (SPLIT >= 2) ? 2 : 1;
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

// When the large-M kernel (gemm_pp.hip: 256x256 tiles, one workgroup per CU) takes a plain bf16x3 GEMM.  Measured on MI355X
// (tools/pp_bench.py, us, tile kernel -> pp; M = 197 * B): B = 16 (39-117 tiles of 256^2): 38.7 -> 53.7 (qkv), 63.4 -> 154.9 (fc2): never;
// B = 64: qkv (450 tiles) 134 -> 118, fc2 (150 tiles, K = 3072) 170 -> 158, proj (150 tiles, K = 768) 57 -> 62, fc1 (600 tiles = 2.3
// rounds of 256 CUs) 175 -> 186; B = 96: qkv 194 -> 171, proj 78 -> 69, fc1 281 -> 252, fc2 298 -> 214.  Its main loop runs at 92 % of
// the MFMA issue rate in cycles (in-kernel stamps); what it loses it loses to the last, partly filled round of tiles and to ~9 us of
// prologue + epilogue per round, so: long K from 128 tiles on, short K from 200 tiles on when the rounds are >= 80 % full.
// MVP_GEMM_PP = 0 / 1 (diagnostic override): never / whenever the kernel supports the arguments.
static bool pp_takes(const mvp_gemm_args* a) {
  if ((a->precision != MVP_PREC_BF16X3 && a->precision != MVP_PREC_F16X2) || a->splitk > 1 || (a->K & 31) || a->K < 64) return false;
  static const int env = [] { const char* e = getenv("MVP_GEMM_PP"); return e ? atoi(e) : -1; }();
  if (a->conv) {
    // convolutions (the DPT probe's 3x3 layers at 8x the token grid: M = 200 704 pixels, K = 4608 / 2304): the large-M kernel with
    // the im2col staging, its fused masks / residuals through the generic epilogue; from two full rounds of 256x256 tiles on
    if (a->pair_layout != MVP_PAIR_SEPARATE || (a->tile_policy & MVP_TILES_NO_PP) || (a->cC & 31) || a->K < 1024 || a->residual_hi) return false;
    if (env >= 0) return env != 0;
    if ((long)((a->N + 255) / 256) * 256 * 7 > (long)a->N * 8) return false;  // (narrow outputs: see below)
    return (long)((a->M + 255) / 256) * ((a->N + 255) / 256) >= 512;
  }
  // masked / pair-residual epilogues of plain GEMMs (the ResNet-50 trunk's residual-adding 1x1 layers: K = 64 ... 512, all epilogue) stay
  // on the tile kernels: with the universal branch-free epilogue on both, 3-5 resident workgroups per CU keep more bytes in flight than
  // one persistent 8-wave workgroup (tools/resnet_bench.py, M = 230 400, K = 64, N = 256 + identity: 135 us against 157)
  if (a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi) return false;
  if (a->pair_layout != MVP_PAIR_SEPARATE) return true;  // only that kernel reads the interleaved layout
  if (a->tile_policy & MVP_TILES_NO_PP) return false;
  if (env >= 0) return env != 0;
  if (a->N <= 256 && a->K >= 2048) return false;  // the probe head: few tiles, split-K
  // a 256-column tile on a narrow output issues MFMAs for columns that do not exist (N = 64: 4x; the ResNet-50 trunk's layer1 / layer2
  // bottlenecks ran 125 us on it for 60 us of operand streaming): at most 1/8 of the columns of the tile grid may be padding
  if ((long)((a->N + 255) / 256) * 256 * 7 > (long)a->N * 8) return false;
  const long t = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
  if (a->tile_policy & MVP_TILES_SHARED) return t >= 96;  // other chains fill what a partly filled round leaves idle
  if (a->K >= 2048) return t >= 128;
  const long rounds = (t + 255) / 256;
  return t >= 200 && (t * 5 >= rounds * 256 * 4 || t >= 1024);
}

extern "C" int mvp_gemm_bias_act_res(const mvp_gemm_args* a, void* stream) {
  if (!a || !a->a_hi || !a->w_hi) return MVP_EINVAL;
  if (a->splitk == MVP_GEMM_STREAMK) return a->out_f16_col0 ? MVP_EINVAL : mvp_gemm_streamk(a, stream);
  if (a->pair_layout < 0 || a->pair_layout > MVP_PAIR_ILV32) return MVP_EINVAL;
  if (a->out_pair_layout != MVP_PAIR_SEPARATE &&
      (a->out_pair_layout != MVP_PAIR_A_ILV32 || !a->out_hi || (a->N & 31) || (a->precision != MVP_PREC_BF16X3 && a->precision != MVP_PREC_F16X2))) return MVP_EINVAL;
  if (a->out_f16_col0 != 0 && ((a->out_f16_col0 != -1 && ((a->out_f16_col0 < 0 ? -a->out_f16_col0 : a->out_f16_col0) & (a->out_f16_col0 < 0 ? 127 : 63))) || (a->precision != MVP_PREC_BF16X3 && a->precision != MVP_PREC_F16X2) || !a->out_hi ||
                               (!a->out_lo && a->out_pair_layout == MVP_PAIR_SEPARATE) || a->splitk > 1))
    return MVP_EINVAL;
  if (a->M > 0 && a->N > 0 && (a->out_f32 || a->out_hi) && pp_takes(a)) return mvp_gemm_pp(a, stream);
  if (a->pair_layout != MVP_PAIR_SEPARATE) return MVP_EINVAL;  // the tile kernels read separate hi / lo arrays
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K & (a->conv ? 31 : 63))) {
    // the one non-conv exception: K % 32 == 0 through the BK = 32 two-stage tile (ResNet stem: K = 147 padded to 160)
    if (!(a && !a->conv && a->M > 0 && a->N > 0 && a->K > 0 && (a->K & 31) == 0 && a->precision == MVP_PREC_BF16X3 && a->splitk <= 1)) return MVP_EINVAL;
  }
  if ((a->lda & 7) || (a->ldw & 7)) return MVP_EINVAL;  // 16-byte aligned rows for LDS-DMA
  if ((a->precision == MVP_PREC_BF16X3 || a->precision == MVP_PREC_F16X2) && (!a->a_lo || !a->w_lo)) return MVP_EINVAL;
  if (a->precision != MVP_PREC_BF16 && a->precision != MVP_PREC_BF16X3 && a->precision != MVP_PREC_F16X2) return MVP_EINVAL;
  if (a->precision == MVP_PREC_F16X2) {
    // the two-product mode (opt-in): plain linear GEMMs of the ViT blocks, K % 64 == 0; a reduced tile rule (three shapes)
    if (a->conv || a->splitk > 1 || (a->K & 63) || a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi) return MVP_EINVAL;
    if (!a->out_f32 && !a->out_hi) return MVP_EINVAL;
    hipStream_t s2 = (hipStream_t)stream;
    if (a->N >= 1024) return launch_gemm<128, 128, 64, 2, 1, false, 8>(a, s2);
    const long t64 = (long)((a->M + 63) / 64) * ((a->N + 63) / 64);
    return t64 <= 1280 ? launch_gemm<64, 64, 64, 2, 1>(a, s2) : launch_gemm<128, 64, 64, 2, 1>(a, s2);
  }
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
  const int big = big_env >= 0 ? big_env : ((a->tile_policy & MVP_TILES_SHARED) ? 1 : 0);
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
    // long-K, mid-N GEMMs with many tiles (the DPT probe's coarse-grid input gradients: M = 12544, N = 512, K = 4608): the tile the
    // convolutions of the same shape use (165 us against 244 on 128x64)
    if (a->K >= 4096 && t128 >= 300) return launch_gemm<128, 128, 64, 3, 1, false, 8>(a, s);
    const long t64 = (long)((a->M + 63) / 64) * ((a->N + 63) / 64);
    if (t64 <= 1280) return launch_gemm<64, 64, 64, 3, 1>(a, s);  // 5 resident per CU
    return launch_gemm<128, 64, 64, 3, 1>(a, s);
  }
  if (t128 >= 400) return launch_gemm<128, 128, 64, 1, 2>(a, s);
  return launch_gemm<64, 64, 64, 1, 2>(a, s);
#endif
}
