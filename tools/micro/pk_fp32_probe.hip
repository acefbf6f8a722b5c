// Diagnostic: do packed fp32 VALU instructions return wrong results while another stream runs MFMA-heavy work?
// Kernel `pk` computes the bilinear blend of linear_bins_fwd_cells two ways per iteration — with v_pk_mul_f32 / v_pk_fma_f32 (op_sel
// lane selects, inline asm: exactly the instruction forms the SLP-vectorised build contained) and with scalar v_mul / v_fma — and
// counts bit mismatches.  It runs (a) alone, (b) beside an MFMA loop kernel on a second stream, (c) beside the library's GEMM.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/micro/pk_fp32_probe.hip -o tools/micro/pk_fp32_probe.bin -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mvp_hip.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) {
  f2 d;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// d = {a.x * b.y + c.x, a.y * b.x + c.y}: the crossed lane select the compiler emitted (op_sel:[0,1,0] op_sel_hi:[1,0,1])
__device__ __forceinline__ f2 pk_fma_bcast_hi(f2 a, f2 b, f2 c) {
  f2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__global__ __launch_bounds__(256) void pk(unsigned long long* bad, unsigned long long* total, int iters, unsigned seed) {
  unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; };
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const f2 a = {rnd(), rnd()}, bq = {rnd(), rnd()}, c = {rnd(), rnd()}, e = {rnd(), rnd()};
    const float wx1 = rnd() + 0.5f, wy1 = rnd() + 0.5f;
    const f2 wx = {1.f - wx1, wx1}, wy = {1.f - wy1, wy1};
    // packed: t = a * wx.x (both lanes by wx.x via plain pk_mul with a splat), then fma with the .y broadcast
    const f2 t0 = pk_fma_bcast_hi(bq, wx, pk_mul(a, f2{wx.x, wx.x}));   // wx0 * a + wx1 * bq
    const f2 t1 = pk_fma_bcast_hi(e, wx, pk_mul(c, f2{wx.x, wx.x}));    // wx0 * c + wx1 * e
    const f2 l = pk_fma_bcast_hi(t1, wy, pk_mul(t0, f2{wy.x, wy.x}));   // wy0 * t0 + wy1 * t1
    // scalar reference, same operation order and roundings
    float r[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float wxs = q == 0 ? wx.y : wx.x, wys = q == 0 ? wy.y : wy.x;  // lane 0 takes the second operand's high half, lane 1 its low half
      const float u0 = __builtin_fmaf(bq[q], wxs, a[q] * wx.x);
      const float u1 = __builtin_fmaf(e[q], wxs, c[q] * wx.x);
      r[q] = __builtin_fmaf(u1, wys, u0 * wy.x);
    }
    nbad += (__float_as_uint(l.x) != __float_as_uint(r[0])) + (__float_as_uint(l.y) != __float_as_uint(r[1]));
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * 512ull);
}

// v_pk_add_f32 (no lane selects): what a packed fp32 SUM reduction uses (RCCL's gfx950 code object contains 108 of them)
__global__ __launch_bounds__(256) void pkadd(unsigned long long* bad, unsigned long long* total, int iters, unsigned seed) {
  unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; };
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const f2 a = {rnd(), rnd()}, b = {rnd(), rnd()}, c = {rnd(), rnd()};
    f2 d, e2;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(e2) : "v"(d), "v"(c));
    const float r0 = (a.x + b.x) + c.x, r1 = (a.y + b.y) + c.y;
    nbad += (__float_as_uint(e2.x) != __float_as_uint(r0)) + (__float_as_uint(e2.y) != __float_as_uint(r1));
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * 512ull);
}

// One packed form at a time: v_pk_mul_f32, v_pk_fma_f32 without lane selects, v_pk_fma_f32 with the crossed lane select
template <int FORM>
__global__ __launch_bounds__(256) void pkform(unsigned long long* bad, unsigned long long* total, int iters, unsigned seed) {
  unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; };
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const f2 a = {rnd(), rnd()}, b = {rnd(), rnd()}, c = {rnd(), rnd()};
    f2 d;
    float r0, r1;
    if (FORM == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      r0 = a.x * b.x; r1 = a.y * b.y;
    } else if (FORM == 1) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      r0 = __builtin_fmaf(a.x, b.x, c.x); r1 = __builtin_fmaf(a.y, b.y, c.y);
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      r0 = __builtin_fmaf(a.x, b.y, c.x); r1 = __builtin_fmaf(a.y, b.x, c.y);
    }
    nbad += (__float_as_uint(d.x) != __float_as_uint(r0)) + (__float_as_uint(d.y) != __float_as_uint(r1));
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * 512ull);
}

// Controls: the same blend computed twice with SCALAR v_mul / v_fma (an opaque asm barrier keeps the compiler from merging the two),
// and v_cvt_pk_bf16_f32 (the conversion every epilogue of the library uses) against integer round-to-nearest-even.
__global__ __launch_bounds__(256) void ctl(unsigned long long* bad, unsigned long long* total, int iters, unsigned seed) {
  unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; };
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float a = rnd(), bq = rnd(), c = rnd(), e = rnd(), wx1 = rnd() + 0.5f, wy1 = rnd() + 0.5f;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float r1 = __builtin_fmaf(__builtin_fmaf(e, wx1, c * wx0), wy1, __builtin_fmaf(bq, wx1, a * wx0) * wy0);
    asm volatile("" : "+v"(a), "+v"(bq), "+v"(c), "+v"(e));
    const float r2 = __builtin_fmaf(__builtin_fmaf(e, wx1, c * wx0), wy1, __builtin_fmaf(bq, wx1, a * wx0) * wy0);
    nbad += __float_as_uint(r1) != __float_as_uint(r2);
    unsigned pk;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(r1), "v"(a));
    auto rne = [](float f) { const unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    nbad += pk != (rne(r1) | (rne(a) << 16));
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * 512ull);
}

__global__ __launch_bounds__(256) void mfma_load(float* out, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * (i + 1)); }
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

// MFMA loop whose accumulators live in AccVGPRs (as the library's GEMM instantiations do), with accvgpr moves every iteration
__global__ __launch_bounds__(256) void mfma_agpr_load(float* out, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); b[i] = (__bf16)(0.02f * (i + 1)); }
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float carry = 0.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+a"(acc0), "+a"(acc1) : "v"(a), "v"(b));
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc0[0]));
    carry += t;
  }
  out[blockIdx.x * 256 + threadIdx.x] = carry + acc1[1];
}

// MFMA loop fed by LDS fragment reads (ds_read_b128 returns land in the VGPR file while MFMAs read it)
__global__ __launch_bounds__(256) void mfma_lds_load(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) bf8 frag[1024];
  for (int i = threadIdx.x; i < 1024; i += 256)
    for (int k = 0; k < 8; ++k) frag[i][k] = (__bf16)(0.001f * (i + k));
  __syncthreads();
  f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const bf8 a0 = frag[(threadIdx.x + it * 7) & 1023], a1 = frag[(threadIdx.x + it * 13 + 256) & 1023];
    const bf8 b0 = frag[(threadIdx.x + it * 5 + 512) & 1023], b1 = frag[(threadIdx.x + it * 3 + 768) & 1023];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

// LDS-DMA load: every wave streams 1-KiB pieces of a buffer into LDS (global_load_lds_dwordx4), the GEMM's operand staging without its MFMAs
__global__ __launch_bounds__(256) void dma_load(const char* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const char* g = src + ((size_t)((blockIdx.x * 4 + wave + it) & 1023) << 10) + lane * 16;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)(lds + wave * 8192 + (it & 7) * 1024), 16, 0, 0);
    if ((it & 7) == 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      acc += *(const float*)(lds + wave * 8192 + lane * 4);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef int (*gemm_fn)(const mvp_gemm_args*, void*);

int main(int argc, char** argv) {
  unsigned long long *bad, *total;
  hipMalloc(&bad, 8); hipMalloc(&total, 8);
  float* sink; hipMalloc(&sink, 4096 * 256 * 4);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  gemm_fn gemm = nullptr;
  mvp_gemm_args g;
  memset(&g, 0, sizeof(g));
  if (argc > 1) {
    void* so = dlopen(argv[1], RTLD_NOW);
    if (so) gemm = (gemm_fn)dlsym(so, "mvp_gemm_bias_act_res");
    const int M = 128, N = 768, K = 768;
    std::vector<uint16_t> h((size_t)N * K);
    for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x1ff));
    uint16_t *ah, *al, *wh, *wl; float* out;
    hipMalloc(&ah, M * K * 2); hipMalloc(&al, M * K * 2); hipMalloc(&wh, N * K * 2); hipMalloc(&wl, N * K * 2); hipMalloc(&out, M * N * 4);
    hipMemcpy(ah, h.data(), M * K * 2, hipMemcpyHostToDevice); hipMemcpy(al, h.data(), M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(wh, h.data(), N * K * 2, hipMemcpyHostToDevice); hipMemcpy(wl, h.data(), N * K * 2, hipMemcpyHostToDevice);
    g.a_hi = ah; g.a_lo = al; g.w_hi = wh; g.w_lo = wl; g.out_f32 = out;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N; g.ldob = N; g.precision = MVP_PREC_BF16X3;
  }
  char* dsrc; hipMalloc(&dsrc, 1 << 20); hipMemset(dsrc, 1, 1 << 20);
  hipFuncSetAttribute((const void*)dma_load, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  for (int which = 0; which < 6; ++which)
  for (int mode = 0; mode < (gemm ? 6 : 2); ++mode) {
    hipMemset(bad, 0, 8); hipMemset(total, 0, 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 300; ++rep) {
      if (mode == 1) hipLaunchKernelGGL(mfma_load, dim3(2048), dim3(256), 0, s2, sink, 2000);
      if (mode == 2) for (int i = 0; i < 12; ++i) gemm(&g, s2);
      if (mode == 3) hipLaunchKernelGGL(dma_load, dim3(1280), dim3(256), 32768, s2, dsrc, sink, 4000);
      if (mode == 4) hipLaunchKernelGGL(mfma_agpr_load, dim3(2048), dim3(256), 0, s2, sink, 2000);
      if (mode == 5) hipLaunchKernelGGL(mfma_lds_load, dim3(2048), dim3(256), 0, s2, sink, 1000);
      if (which == 0) hipLaunchKernelGGL(pk, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
      else if (which == 1) hipLaunchKernelGGL(ctl, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
      else if (which == 2) hipLaunchKernelGGL(pkadd, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
      else if (which == 3) hipLaunchKernelGGL(pkform<0>, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
      else if (which == 4) hipLaunchKernelGGL(pkform<1>, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
      else hipLaunchKernelGGL(pkform<2>, dim3(30), dim3(256), 0, s1, bad, total, 400, (unsigned)rep * 7919u + 1u);
    }
    hipDeviceSynchronize();
    unsigned long long hb = 0, ht = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, total, 8, hipMemcpyDeviceToHost);
    printf("%s, mode %d (%s): %llu results differ out of %llu\n", which == 0 ? "packed fp32 vs scalar" : which == 1 ? "control: scalar fp32 twice + v_cvt_pk_bf16_f32" : which == 2 ? "v_pk_add_f32 vs scalar add" : which == 3 ? "v_pk_mul_f32 alone" : which == 4 ? "v_pk_fma_f32 (no op_sel) alone" : "v_pk_fma_f32 op_sel alone", mode,
           mode == 0 ? "packed-fp32 kernel alone" : mode == 1 ? "beside an MFMA loop kernel" : mode == 2 ? "beside the library's 64x64 GEMM" : mode == 3 ? "beside an LDS-DMA loop kernel" : mode == 4 ? "beside an MFMA loop with AccVGPR accumulators" : "beside an MFMA loop fed by ds_read_b128", hb, ht);
  }
  return 0;
}
