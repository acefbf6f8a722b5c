// L2 -> CU operand-fill microbenchmark (diagnostic, not part of the library): what rate can one CU / the chip pull
// GEMM operand tiles at, by instruction form and by memory layout?  All variants move the same bytes:
// per "k-tile" a workgroup of 256 threads fetches ROWS rows of 128 B (two arrays: hi, lo).
//   V0  global_load_lds b128, 8 rows x 128 B per wave-instruction, row stride LD bytes   (the GEMM's current pattern)
//   V1  global_load_lds b128, 1 KiB contiguous per wave-instruction                       (pre-tiled layout)
//   V2  global_load_dwordx4 -> VGPR (no LDS), strided rows
//   V3  global_load_dwordx4 -> VGPR -> ds_write_b128, strided rows
//   V4  global_load_dwordx4 -> VGPR, contiguous
// Footprint is sized to stay L2-resident per XCD (every workgroup walks the same FOOT bytes at its own phase).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
#ifndef NMFMA
#define NMFMA 0   // MFMAs issued by the same wave after every LDS-DMA piece (variant 6)
#endif

template <int V, int ROWS, int DEPTH>
__global__ __launch_bounds__(256) void fill(const char* __restrict__ src, size_t foot, int ld, int nk, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PIECES = ROWS * 2 / 8 / 4;  // 1-KiB pieces per wave per k-tile (hi + lo arrays)
  const size_t half = foot / 2;
  f4 acc = {0, 0, 0, 0};
  f4 macc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf8 ma, mb;
  for (int e = 0; e < 8; ++e) { ma[e] = (__bf16)(0.001f * (lane + e)); mb[e] = (__bf16)(0.002f * (lane - e)); }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)foot, 0x00020000);
  // every workgroup starts at its own offset and walks forward
  size_t base = ((size_t)blockIdx.x * 7919u * 1024u) % (half / 2);
  for (int kt = 0; kt < nk; ++kt) {
    char* buf = smem + (kt % DEPTH) * (ROWS * 128 * 2);
#pragma unroll
    for (int ps = 0; ps < PIECES; ++ps) {
      const int arr = ps & 1, piece = (ps >> 1) * 4 + wave;   // piece index inside the array's tile
      size_t off;
      if (V == 0 || V == 2 || V == 3 || V == 5 || V == 6 || V == 7) {   // strided rows: piece = 8 rows x 128 B
        const int row = piece * 8 + (lane >> 3);
        off = ((size_t)row * ld + (size_t)kt * 128 + (lane & 7) * 16);
      } else {                            // contiguous: tile kt of this workgroup is one block
        off = ((size_t)kt * ROWS * 128 + (size_t)piece * 1024 + lane * 16);
      }
      off = (base + off) % (half - 4096);
      off &= ~(size_t)15;
      const char* g = src + arr * half + off;
      if (V == 0 || V == 1) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(buf + arr * ROWS * 128 + piece * 1024), 16, 0, 0);
      } else if (V == 6) {  // the GEMM situation: operand fill and MFMAs from the same SIMD
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(buf + arr * ROWS * 128 + piece * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) macc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc[q & 3], 0, 0, 0);
      } else if (V == 7) {  // buffer form + MFMAs
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(buf + arr * ROWS * 128 + piece * 1024), 16, (int)(arr * half + off), 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) macc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc[q & 3], 0, 0, 0);
      } else if (V == 5) {  // buffer form: SGPR resource + 32-bit per-lane offset
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(buf + arr * ROWS * 128 + piece * 1024), 16, (int)(arr * half + off), 0, 0, 0);
      } else {
        const f4 v = *(const f4*)g;
        if (V == 3) *(f4*)(buf + arr * ROWS * 128 + piece * 1024 + lane * 16) = v;
        else acc += v;
      }
    }
    if (DEPTH == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // touch LDS so the data dependency is real
      if (V != 2 && V != 4) acc += *(const f4*)(buf + tid * 16);
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
      __builtin_amdgcn_s_barrier();
      if (V != 2 && V != 4) acc += *(const f4*)(smem + ((kt + 1) % DEPTH) * (ROWS * 128 * 2) + tid * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  acc += macc[0] + macc[1] + macc[2] + macc[3];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int V, int ROWS, int DEPTH>
void run(const char* name, const char* src, size_t foot, float* sink, int grid, int nk, int lds_pad) {
  const int smem = ROWS * 128 * 2 * DEPTH + lds_pad;
  hipFuncSetAttribute((const void*)fill<V, ROWS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fill<V, ROWS, DEPTH>), dim3(grid), dim3(256), smem, 0, src, foot, 6144, nk, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((fill<V, ROWS, DEPTH>), dim3(grid), dim3(256), smem, 0, src, foot, 6144, nk, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double bytes = (double)grid * nk * ROWS * 128 * 2;
  printf("%-34s rows=%3d depth=%d grid=%5d lds=%6d : %8.1f us  %6.2f TB/s  %6.1f GB/s/CU\n", name, ROWS, DEPTH, grid, smem, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / 256);
}

int main(int argc, char** argv) {
  const size_t foot = (argc > 1 ? atol(argv[1]) : 2) * (size_t)1 << 20;  // MiB walked by every workgroup
  char* src; float* sink;
  hipMalloc(&src, foot + (1 << 20)); hipMalloc(&sink, 64);
  hipMemset(src, 1, foot + (1 << 20));
  printf("footprint %zu MiB (every workgroup walks it; <= 4 MiB stays in each XCD's L2)\n", foot >> 20);
  const int nk = 96;
  for (int occ = 1; occ <= 4; occ *= 2) {
    const int grid = 256 * occ * 4;                    // 4 rounds of full residency
    const int pad = 160 * 1024 / occ - 128 * 128 * 2 * 2 - 1024;  // LDS padding so that exactly `occ` workgroups fit per CU (depth 2, 128 rows)
    const int p = pad > 0 ? pad : 0;
    printf("-- %d workgroup(s) per CU\n", occ);
    run<0, 128, 1>("V0 lds-dma strided  (single)", src, foot, sink, grid, nk, p + 128 * 128 * 2);
    run<0, 128, 2>("V0 lds-dma strided  (2-stage)", src, foot, sink, grid, nk, p);
    run<5, 128, 2>("V5 buffer_load-lds strided (2-stage)", src, foot, sink, grid, nk, p);
    run<6, 128, 2>("V6 lds-dma strided + NMFMA mfma/piece", src, foot, sink, grid, nk, p);
    run<7, 128, 2>("V7 buffer-lds strided + NMFMA mfma/piece", src, foot, sink, grid, nk, p);
    run<1, 128, 2>("V1 lds-dma contiguous (2-stage)", src, foot, sink, grid, nk, p);
    run<2, 128, 2>("V2 regs strided", src, foot, sink, grid, nk, p);
    run<4, 128, 2>("V4 regs contiguous", src, foot, sink, grid, nk, p);
    run<3, 128, 2>("V3 regs+ds_write strided", src, foot, sink, grid, nk, p);
  }
  printf("-- 1 workgroup per CU, 256 rows (128x128 tile), depth 2 = 128 KiB\n");
  run<0, 256, 2>("V0 lds-dma strided", src, foot, sink, 1024, nk, 0);
  run<5, 256, 2>("V5 buffer_load-lds strided", src, foot, sink, 1024, nk, 0);
  run<1, 256, 2>("V1 lds-dma contiguous", src, foot, sink, 1024, nk, 0);
  run<3, 256, 2>("V3 regs+ds_write strided", src, foot, sink, 1024, nk, 0);
  return 0;
}
