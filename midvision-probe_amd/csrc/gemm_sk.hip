// Stream-K bf16 / split-bf16 MFMA GEMM for the frozen-backbone linears (qkv, proj, fc1, fc2, patch-embed, probe head):
//
//   Y[M,N] = act(A[M,K] · W[N,K]ᵀ + bias) + residual            (same contract as gemm.hip, plain linear form only)
//
// Why a second kernel.  What the round-1 counters and tools/micro/fill_bench.hip say about the tile-per-workgroup kernel
// at M ~ 3k rows:  (1) operand fill is ISSUE-bound: one global_load_lds b128 moves 1 KiB and costs the issuing wave
// ~60-130 cycles, a CU's four SIMDs together sustain ~54 B/clk from L2 (113 GB/s/CU measured with 16 waves loading, 57
// with 4), so a 64x64 tile in bf16x3 (85 B/clk of operands at full MFMA rate) can never feed the matrix pipe, a 128x128
// tile (43 B/clk) can;  (2) at 128x128 the four hot GEMMs have 150-600 tiles for 256 CUs: whole-tile scheduling leaves
// 41 % of the chip idle (150 tiles) or runs 2.34 rounds;  (3) with only ~2 waves per SIMD the loop must hide LDS and DMA
// latency inside the wave, not by occupancy.
//
// Design.  One workgroup of 8 waves per CU (grid = #CUs), tile 128x128x64, TWO LDS stages of 64 KiB (bf16x3).
//   * stream-K: the tiles x k-iterations space is cut into #CUs equal contiguous ranges, so every CU carries the same
//     number of k-iterations whatever the tile count; a workgroup whose range covers a whole tile runs the fused epilogue
//     directly, partial tiles leave fp32 partials in a workspace and the LAST arriver (one agent-scope counter per tile,
//     nobody waits) sums them in the fixed order of the contributing workgroups (bit-reproducible) and runs the epilogue.
//   * software pipeline per k-tile:  [ds_read ks=1 | MFMA ks=0] -> vmcnt(0)+lgkmcnt(0)+s_barrier -> [LDS-DMA of tile
//     kt+2, one piece between MFMA groups | ds_read (kt+1, ks=0) | MFMA ks=1].  Fragments are double-buffered in
//     registers, the DMA of a tile gets a whole k-tile of MFMA time (~1500 cycles) to land, one barrier per k-tile.
//   * two waves per SIMD: one wave's DMA / ds_read issue slots sit under its partner's MFMAs.
// LDS image, swizzle, swapped MFMA operand order and the LDS-staged full-line epilogue are those of gemm.hip.
#include "mvp_common.h"

#ifndef MVP_SK_DMA_INTERLEAVE
#define MVP_SK_DMA_INTERLEAVE 1
#endif

namespace {

#ifndef MVP_SK_SPEC
#define MVP_SK_SPEC 1
#endif
constexpr bool SK_SPEC = MVP_SK_SPEC != 0;
constexpr int SK_BM = 128, SK_BN = 128, SK_BK = 64, SK_NW = 8;
constexpr int SK_CTR_BYTES = 65536;
constexpr int SK_PART_BYTES = SK_BM * SK_BN * 4;  // one fp32 partial tile

__device__ __forceinline__ float sk_gelu_erf(float x) {  // Abramowitz-Stegun 7.1.26: the same operations, in the same order, as gemm_epilogue.h's gelu_erf
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

// logical tile index -> (tm, tn): the 8 XCD slices of the (contiguous) iteration space map to XR x XC rectangles of the
// output (see gemm.hip region_tile); inside a rectangle tiles run row-major so that consecutive workgroups share A rows.
__device__ __forceinline__ void sk_tile(int L, int TM, int TN, int M, int N, int& tm, int& tn) {
  int XR = 1;
  long best = (long)M * 8 + N;
#pragma unroll
  for (int c = 2; c <= 8; c <<= 1) {
    const long cost = (long)M * 8 / c + (long)N * c;
    if (cost < best) { best = cost; XR = c; }
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
  tm = TM - 1; tn = TN - 1;
}

// SPEC = false: 8 identical waves (4 x 2 grid of 32 x 64 wave tiles), each issues its share of the LDS-DMA between its MFMAs.
// SPEC = true : wave specialisation — waves 0-3 (one per SIMD) are MFMA waves with 64 x 64 tiles (2 x 2 grid), waves 4-7 (their
//               SIMD partners) are loader waves that issue ALL LDS-DMA: a DMA piece costs its issuing wave ~60-130 cycles in
//               which that wave issues nothing else, so the MFMA stream of a SIMD is never interrupted by operand fill.
template <int SPLIT, bool SPEC>
__global__ __launch_bounds__(SK_NW * 64) void gemm_sk_kernel(const mvp_gemm_args p, const int tiles, const int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = SK_BM, BN = SK_BN, BK = SK_BK;
  constexpr int CW = SPEC ? 4 : 8;                    // MFMA ("compute") waves
  constexpr int LW = SPEC ? 4 : 8;                    // waves that issue LDS-DMA
  constexpr int WNW = 2;
  constexpr int NARR = (SPLIT == 3) ? 2 : 1;
  constexpr int ROWB = BK * 2, RPP = 1024 / ROWB;     // 128-byte rows, 8 rows per 1-KiB LDS-DMA piece
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
  constexpr int STAGE = (A_BYTES + W_BYTES) * NARR;
  constexpr int WM = BM / (CW / WNW), WN = BN / WNW;  // 32 x 64 (8 compute waves) or 64 x 64 (4)
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int NT_THREADS = CW * 64;                 // threads that hold accumulators
  constexpr int APASS = BM / (LW * RPP), WPASS = BN / (LW * RPP);
  constexpr int NPIECE = (APASS + WPASS) * NARR;      // LDS-DMA instructions per loading wave per k-tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_mma = !SPEC || wave < CW;
  const bool is_ldr = !SPEC || wave >= CW;
  const int lw = SPEC ? (wave - CW) : wave;           // index among the loading waves
  const long T = (long)tiles * nk;
  const int G = (int)min((long)gridDim.x, T);  // never more ranges than k-iterations: every range is non-empty
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  if (wg >= G) return;
  long it = (long)wg * T / G;
  const long it_end = (long)(wg + 1) * T / G;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;

  const int cwv = is_mma ? wave : 0;
  const int wm0 = (cwv / WNW) * WM, wn0 = (cwv % WNW) * WN;
  const int rsub = lane >> 3;                         // row inside a piece
  const int csrc = ((lane & 7) ^ (rsub & 7)) << 3;    // swizzled source chunk (elements)
  const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;

  int* ctr = (int*)p.splitk_ws;
  char* part = (char*)p.splitk_ws + SK_CTR_BYTES;
  __shared__ int s_last;

  while (it < it_end) {
    const int t = (int)(it / nk);
    const int kb = (int)(it - (long)t * nk);
    const int ke = (int)min((long)nk, (long)kb + (it_end - it));
    const int n = ke - kb;
    it += n;
    int tm, tn;
    sk_tile(t, tiles_m, tiles_n, p.M, p.N, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // one LDS-DMA piece of tile kt (q: 0..NPIECE-1 = [A pass0 hi, lo, A pass1 hi, lo, W pass0 hi, lo, W pass1 hi, lo])
    auto issue_piece = [&](int q, int kt) {
      char* base = smem + (kt & 1) * STAGE;
      const int k0 = (kb + kt) * BK;
      const int arr = (NARR == 2) ? (q & 1) : 0, ps = (NARR == 2) ? (q >> 1) : q;
      if (ps < APASS) {
        const int r = ps * LW * RPP + lw * RPP;
        const int grow = min(m0 + r + rsub, p.M - 1);
        const size_t off = (size_t)grow * p.lda + k0 + csrc;
        __builtin_amdgcn_global_load_lds(GLB_PTR((arr ? p.a_lo : p.a_hi) + off), LDS_PTR(base + arr * A_BYTES + r * ROWB), 16, 0, 0);
      } else {
        const int r = (ps - APASS) * LW * RPP + lw * RPP;
        const int grow = min(n0 + r + rsub, p.N - 1);
        const size_t off = (size_t)grow * p.ldw + k0 + csrc;
        __builtin_amdgcn_global_load_lds(GLB_PTR((arr ? p.w_lo : p.w_hi) + off), LDS_PTR(base + A_BYTES * NARR + arr * W_BYTES + r * ROWB), 16, 0, 0);
      }
    };

    struct Frag { bf16x8_t a_hi[MT], a_lo[MT], w_hi[NT], w_lo[NT]; };
    auto read_frag = [&](Frag& f, int kt, int ks) {
      const char* base = smem + (kt & 1) * STAGE;
      const char* ab = base + (wm0 + frow) * ROWB;
      const char* wb = base + A_BYTES * NARR + (wn0 + frow) * ROWB;
      const int coff = (((ks << 2) + fq) ^ fsw) << 4;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        f.a_hi[j] = *(const bf16x8_t*)(ab + j * 16 * ROWB + coff);
        if (SPLIT == 3) f.a_lo[j] = *(const bf16x8_t*)(ab + A_BYTES + j * 16 * ROWB + coff);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f.w_hi[i] = *(const bf16x8_t*)(wb + i * 16 * ROWB + coff);
        if (SPLIT == 3) f.w_lo[i] = *(const bf16x8_t*)(wb + W_BYTES + i * 16 * ROWB + coff);
      }
    };

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // MFMAs of one k-step; `dma_kt` >= 0: the NPIECE LDS-DMA pieces of that tile are issued between the MFMA groups
    auto mma = [&](const Frag& f, int dma_kt) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if (!SPEC && MVP_SK_DMA_INTERLEAVE && dma_kt >= 0 && q < NPIECE) {
            issue_piece(q, dma_kt);
            ++q;
          }
          if (SPLIT == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w_lo[i], f.a_hi[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w_hi[i], f.a_lo[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w_hi[i], f.a_hi[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- prologue: tile 0 (and 1) in flight, fragments (0, 0) in registers
    __syncthreads();  // the previous segment's epilogue / partial hand-over is done with the staging buffers
    if (is_ldr) {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) issue_piece(q, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (is_ldr && n > 1) {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) issue_piece(q, 1);
    }
    if (SPEC && !is_mma) {
      // loader wave: one barrier per k-tile, like the MFMA waves.  At barrier kt its pieces of tile kt+1 have landed and (the
      // MFMA waves waited lgkmcnt(0)) every read of tile kt is complete, so tile kt+2 goes into tile kt's buffer at once.
      for (int kt = 0; kt < n; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < n) {
#pragma unroll
          for (int q = 0; q < NPIECE; ++q) issue_piece(q, kt + 2);
        }
      }
    } else {
      Frag cur, nxt;
      read_frag(cur, 0, 0);
      for (int kt = 0; kt < n; ++kt) {
        read_frag(nxt, kt, 1);
        mma(cur, -1);
        // tile kt+1 has landed (this wave's pieces; the barrier extends it to everybody's), and every read of tile kt has
        // completed: its buffer may be refilled
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int dk = (kt + 2 < n) ? kt + 2 : -1;
        if (!SPEC && !MVP_SK_DMA_INTERLEAVE && dk >= 0) {  // burst, BEFORE the fragment reads (an LDS-DMA issued after them waits for them)
#pragma unroll
          for (int q = 0; q < NPIECE; ++q) issue_piece(q, dk);
        }
        if (kt + 1 < n) read_frag(cur, kt + 1, 0);
        mma(nxt, dk);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();  // staging buffers are free: the epilogue reuses them as scratch

    // ---- partial tile: hand over through the workspace; the last arriver reduces in workgroup order
    if (n != nk) {
      constexpr int AUX_SC1 = 16;
      const int tail = (kb != 0) ? 1 : 0;  // segment that ends at the tile's end (and does not start it)
      if (is_mma) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part + ((size_t)wg * 2 + (tail && ke == nk ? 0 : 1)) * SK_PART_BYTES, 0, SK_PART_BYTES, 0x00020000);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, ((i * MT + j) * NT_THREADS + tid) * 16, 0, AUX_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // contributing workgroups: those whose iteration range meets [t*nk, (t+1)*nk);  w(i) = ceil((i+1) G / T) - 1
      const long i0 = (long)t * nk, i1 = (long)(t + 1) * nk - 1;
      const int w_first = (int)(((i0 + 1) * G + T - 1) / T) - 1, w_last = (int)(((i1 + 1) * G + T - 1) / T) - 1;
      const int S = w_last - w_first + 1;
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // see gemm.hip: the sc1 stores' vmcnt(0) alone is not enough on a busy chip
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = (__hip_atomic_fetch_add(ctr + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1) ? 1 : 0;
      }
      __syncthreads();
      if (!s_last) continue;
      // last arriver: one agent-scope acquire -> vmcnt(0) -> barrier -> sc1 loads (same protocol as gemm.hip's split-K)
      if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (is_mma)
      for (int w = w_first; w <= w_last; ++w) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part + ((size_t)w * 2 + (w == w_last ? 0 : 1)) * SK_PART_BYTES, 0, SK_PART_BYTES, 0x00020000);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * MT + j) * NT_THREADS + tid) * 16, 0, AUX_SC1));
            acc[i][j] = (w == w_first) ? v : acc[i][j] + v;
          }
      }
      if (tid == 0) __hip_atomic_store(ctr + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- fused epilogue through LDS (per-wave scratch [32][WN + 4] fp32), whole-line stores
    if (!is_mma) continue;  // (loader waves rejoin at the next segment's prologue barrier)
    constexpr int EPW = WN + 4, EP_BYTES = 32 * EPW * 4, LPR = WN / 4, RPI = 64 / LPR;
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
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < NT; ++i) *(f32x4_t*)(ep + (jj * 16 + frow) * EPW + i * 16 + fq * 4) = acc[i][h * 2 + jj];
#pragma unroll
      for (int itr = 0; itr < 32 / RPI; ++itr) {
        const int lr = itr * RPI + er;
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
          for (int e = 0; e < 4; ++e) v[e] = sk_gelu_erf(v[e]);
        } else if (p.act == MVP_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
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
        if (p.out_f32) {
          float* op = p.out_f32 + (size_t)orow * p.ldo + ncol;
          if (vec_ok && ((p.ldo & 3) == 0)) {
            *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            for (int e = 0; e < 4; ++e) if (ncol + e < p.N) op[e] = v[e];
          }
        }
        if (p.out_hi) {
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
}

int sk_cu_count() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
  }();
  return cus;
}

}  // namespace

extern "C" int64_t mvp_gemm_streamk_workspace_bytes(void) {
  // tile counters (fixed 64 KiB region, self-resetting) + two fp32 partial tiles per workgroup (one workgroup per CU)
  return SK_CTR_BYTES + (int64_t)1024 * 2 * SK_PART_BYTES / 2;  // sized for up to 512 CUs
}

// Called by mvp_gemm_bias_act_res when splitk == MVP_GEMM_STREAMK.
extern "C" int mvp_gemm_streamk(const mvp_gemm_args* a, void* stream) {
  if (a && (a->pair_layout != MVP_PAIR_SEPARATE || a->out_pair_layout != MVP_PAIR_SEPARATE)) return MVP_EINVAL;  // separate hi / lo arrays only
  if (!a || !a->a_hi || !a->w_hi || a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K & 63)) return MVP_EINVAL;
  if (a->conv || a->relu_mask || a->out_mask || a->residual2 || a->act_after_res || a->residual_hi) return MVP_EINVAL;
  if ((a->lda & 7) || (a->ldw & 7) || (!a->out_f32 && !a->out_hi)) return MVP_EINVAL;
  if (a->precision != MVP_PREC_BF16 && a->precision != MVP_PREC_BF16X3) return MVP_EINVAL;
  if (a->precision == MVP_PREC_BF16X3 && (!a->a_lo || !a->w_lo)) return MVP_EINVAL;
  const int G = sk_cu_count();
  const int tiles = ((a->M + SK_BM - 1) / SK_BM) * ((a->N + SK_BN - 1) / SK_BN);
  if (tiles > SK_CTR_BYTES / 4 || G > 512) return MVP_EINVAL;
  if (!a->splitk_ws || a->splitk_ws_bytes < mvp_gemm_streamk_workspace_bytes() || ((uintptr_t)a->splitk_ws & 15)) return MVP_EINVAL;
  const int nk = a->K / SK_BK;
  hipStream_t s = (hipStream_t)stream;
  if (a->precision == MVP_PREC_BF16X3) {
    constexpr int SMEM = 2 * (SK_BM + SK_BN) * SK_BK * 2 * 2;
    static int cfg = (int)hipFuncSetAttribute((const void*)gemm_sk_kernel<3, SK_SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (cfg != 0) return MVP_ELAUNCH;
    hipLaunchKernelGGL((gemm_sk_kernel<3, SK_SPEC>), dim3(G), dim3(SK_NW * 64), SMEM, s, *a, tiles, nk);
  } else {
    constexpr int SMEM_OP = 2 * (SK_BM + SK_BN) * SK_BK * 2;
    constexpr int SMEM_EP = SK_NW * 32 * (SK_BN / 2 + 4) * 4;
    constexpr int SMEM = SMEM_OP > SMEM_EP ? SMEM_OP : SMEM_EP;
    static int cfg = (int)hipFuncSetAttribute((const void*)gemm_sk_kernel<1, SK_SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (cfg != 0) return MVP_ELAUNCH;
    hipLaunchKernelGGL((gemm_sk_kernel<1, SK_SPEC>), dim3(G), dim3(SK_NW * 64), SMEM, s, *a, tiles, nk);
  }
  MVP_LAUNCH_CHECK();
  return MVP_OK;
}
