// Diagnostic: is a co-resident workgroup's LDS disturbed while the library's GEMM kernels run on another stream?
// Canary workgroups fill a small static LDS array with a pattern, re-read it for ~CANARY_US microseconds and count words that changed;
// meanwhile mvp_gemm_bias_act_res (64x64 tile, LDS-DMA staged) runs in a loop on a second stream.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/micro/lds_canary.hip -o tools/micro/lds_canary.bin -ldl
//   tools/micro/lds_canary.bin midvision-probe_amd/csrc/libmvp_hip.so [N=768] [K=3072] [M=84]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mvp_hip.h"

__global__ __launch_bounds__(256) void canary(unsigned long long* bad, unsigned long long* checks, long long cycles, int use_shfl) {
  __shared__ unsigned words[704];  // 2816 bytes, like linear_bins_fwd_cells
  const unsigned tag = 0x9E3779B9u * (blockIdx.x + 1);
  for (int i = threadIdx.x; i < 704; i += 256) words[i] = tag ^ (unsigned)i;
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned long long nbad = 0, n = 0;
  float acc = (float)threadIdx.x;
  while (wall_clock64() - t0 < cycles) {
    for (int i = threadIdx.x; i < 704; i += 256) {
      if (words[i] != (tag ^ (unsigned)i)) ++nbad;
      ++n;
    }
    if (use_shfl) {  // the same cross-lane traffic the bins kernel has (ds_bpermute) — and a check that it is exact
      float v = acc;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int w = threadIdx.x >> 6;
      const float want = (float)(64 * (64 * w) + 63 * 32);  // sum of threadIdx over the wave
      if (v != want) ++nbad;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (threadIdx.x == 0) atomicAdd(checks, n);
}

typedef int (*gemm_fn)(const mvp_gemm_args*, void*);

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s libmvp_hip.so [N] [K] [M]\n", argv[0]); return 2; }
  void* so = dlopen(argv[1], RTLD_NOW);
  if (!so) { printf("dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(so, "mvp_gemm_bias_act_res");
  const int N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 3072, M = argc > 4 ? atoi(argv[4]) : 84;
  const int Mp = (M + 127) / 128 * 128;
  std::vector<uint16_t> h((size_t)(Mp > N ? Mp : N) * K);
  for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x1ff));  // bf16 values near 1
  uint16_t *a_hi, *a_lo, *w_hi, *w_lo; float* out;
  hipMalloc(&a_hi, (size_t)Mp * K * 2); hipMalloc(&a_lo, (size_t)Mp * K * 2); hipMalloc(&w_hi, (size_t)N * K * 2); hipMalloc(&w_lo, (size_t)N * K * 2);
  hipMalloc(&out, (size_t)M * N * 4);
  hipMemcpy(a_hi, h.data(), (size_t)Mp * K * 2, hipMemcpyHostToDevice); hipMemcpy(a_lo, h.data(), (size_t)Mp * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(w_hi, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemcpy(w_lo, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  unsigned long long *bad, *checks;
  hipMalloc(&bad, 8); hipMalloc(&checks, 8);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  mvp_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a_hi; g.a_lo = a_lo; g.w_hi = w_hi; g.w_lo = w_lo; g.out_f32 = out;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldr = N; g.ldo = N; g.ldob = N; g.precision = MVP_PREC_BF16X3;
  for (int mode = 0; mode < 3; ++mode) {  // 0: canary alone, 1: canary + GEMM load, 2: canary with shuffles + GEMM load
    hipMemset(bad, 0, 8); hipMemset(checks, 0, 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 200; ++rep) {
      if (mode >= 1)
        for (int i = 0; i < 4; ++i)
          if (gemm(&g, s2) != 0) { printf("gemm launch failed\n"); return 1; }
      hipLaunchKernelGGL(canary, dim3(64), dim3(256), 0, s1, bad, checks, (long long)2000, mode == 2);  // 100 MHz wall clock: 20 us
    }
    hipDeviceSynchronize();
    unsigned long long hb = 0, hc = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, checks, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %llu corrupted observations in %llu LDS word checks\n", mode,
           mode == 0 ? "canary alone" : mode == 1 ? "canary + GEMM on another stream" : "canary with wave shuffles + GEMM", hb, hc);
  }
  return 0;
}
