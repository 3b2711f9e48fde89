// LDS-DMA rate of the dense kernel's ACTIVATION pattern from cache-resident rows: a piece = 1 KiB = (1024 / SEG) rows x SEG bytes,
// rows `stride` bytes apart; every work-group sweeps the k-steps of ITS 256-row panel (8 waves x 32 rows), panels taken from a set of
// `npanels` resident panels (npanels x 256 x stride bytes: keep it inside L2 / MALL).  r06 probe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attic/dma_pattern_probe.hip -o tools/dma_pattern_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int MODE>   // MODE 0: LDS-DMA, 1: -> VGPR
__global__ __launch_bounds__(512) void k_pat(const char* __restrict__ src, int stride, int npanels, int ksteps, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int LPR = SEG / 16, RPP = 64 / LPR, NP = 2048 / 1024 * (SEG / 64);   // lanes per row, rows per piece; a wave's 32 rows x SEG bytes = NP pieces... (32 * SEG / 1024)
  constexpr int PIECES = 32 * SEG / 1024;
  const uint64_t a = (uint64_t)(src + (size_t)(blockIdx.x % npanels) * 256 * stride);
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
  r.z = __builtin_amdgcn_readfirstlane(256 * stride);
  r.w = 0x00020000;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  int voff[PIECES];
#pragma unroll
  for (int s = 0; s < PIECES; s++) voff[s] = (wave * 32 + s * RPP + lane / LPR) * stride + (lane % LPR) * 16;
  float acc = 0.f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 v[PIECES];
  for (int it = 0; it < iters; it++)
    for (int ks = 0; ks < ksteps; ks++) {
      const int soff = __builtin_amdgcn_readfirstlane(ks * SEG);
#pragma unroll
      for (int s = 0; s < PIECES; s++) {
        if (MODE == 0) {
          const unsigned lb = lds0 + (unsigned)((((ks & (SEG == 64 ? 3 : 1)) * 8 + wave) * PIECES + s) * 1024);
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff[s]), "s"(r), "s"(soff), "s"(lb) : "memory");
        } else {
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v[s]) : "v"(voff[s]), "s"(r), "s"(soff) : "memory");
        }
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(SEG == 64 ? 2 * PIECES : PIECES / 2) : "memory");
      else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < PIECES; s++) asm volatile("" :: "v"(v[s]));
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE == 0) acc = *(float*)(lds + t * 4);
  if (acc == 123.456f) sink[0] = acc;
}

template <int SEG, int MODE>
double run(const char* src, int stride, int npanels, int iters, float* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ksteps = stride / SEG;
  const size_t ldsb = (SEG == 64 ? 4 : 2) * 8 * (32 * SEG);
  CK(hipFuncSetAttribute((const void*)k_pat<SEG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipLaunchKernelGGL((k_pat<SEG, MODE>), dim3(256), dim3(512), ldsb, 0, src, stride, npanels, ksteps, iters, sink);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_pat<SEG, MODE>), dim3(256), dim3(512), ldsb, 0, src, stride, npanels, ksteps, iters, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return 256.0 * iters * ksteps * 256.0 * SEG / (ms * 1e-3) / 1e9;
}

int main() {
  char* buf; float* sink;
  CK(hipMalloc(&buf, (size_t)1 << 30)); CK(hipMemset(buf, 1, (size_t)1 << 30)); CK(hipMalloc(&sink, 16));
  printf("LDS-DMA / ->VGPR of (1024 / SEG) rows x SEG bytes per piece, GB/s chip-wide (B/clk/CU at 2.1 GHz)\n");
  const int strides[] = {896, 2752, 1024, 4096};
  for (int stride : strides)
    for (int npanels : {4, 64, 4096}) {
      if ((size_t)npanels * 256 * stride > ((size_t)1 << 30)) continue;
      const int iters = npanels == 4096 ? 4 : 40;
      double a = run<64, 0>(buf, stride, npanels, iters, sink), b = run<128, 0>(buf, stride, npanels, iters, sink), c = run<256, 0>(buf, stride, npanels, iters, sink);
      double d = run<64, 1>(buf, stride, npanels, iters, sink), e = run<128, 1>(buf, stride, npanels, iters, sink);
      printf("row stride %4d B, %4d resident panels (%6.1f MB): DMA SEG 64: %6.0f (%4.1f)  128: %6.0f (%4.1f)  256: %6.0f (%4.1f) | VGPR SEG 64: %6.0f (%4.1f) 128: %6.0f (%4.1f)\n",
             stride, npanels, npanels * 256.0 * stride / 1e6, a, a / 256 / 2.1, b, b / 256 / 2.1, c, c / 256 / 2.1, d, d / 256 / 2.1, e, e / 256 / 2.1);
    }
  return 0;
}
