// How fast can ONE CU pull operand bytes, by path?  (r06 probe behind the dense kernel's k-loop: 26-32 KB per k-step per CU.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attic/dma_rate_probe.hip -o tools/dma_rate_probe.bin
// Modes: 0 LDS-DMA (buffer_load_dwordx4 ... lds), 1 global_load_dwordx4 -> VGPR, 2 global_load -> VGPR -> ds_write_b128.
// Source sets: "shared" = every work-group reads the same S KB (L2-resident, beyond L1), "private" = each work-group streams its own
// slab of a 2 GB buffer (HBM).  256 work-groups x 512 threads, PIECES 1-KiB pieces per wave in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k_probe(const char* __restrict__ src, size_t wg_stride, unsigned span, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const char* base = src + (size_t)blockIdx.x * wg_stride;
  const uint64_t a = (uint64_t)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
  r.z = __builtin_amdgcn_readfirstlane((int)span);
  r.w = 0x00020000;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned off = (unsigned)(wave * 1024 + lane * 16);           // a wave's piece: 1 KiB contiguous; the 8 waves cover 8 KiB per round
  if (MODE == 0) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        const unsigned lb = lds0 + (unsigned)((d * 8 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(off), "s"(r), "s"(lb) : "memory");
        off += 8192; if (off >= span) off -= span;
      }
      asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x = *(float*)(lds + t * 4);
  } else {
    f32x4 v[DEPTH];
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[d]) : "v"(off), "s"(r) : "memory");
        off += 8192; if (off >= span) off -= span;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        if (MODE == 2) *(f32x4*)(lds + (d * 8 + wave) * 1024 + lane * 16) = v[d];
        else { asm volatile("" :: "v"(v[d])); }
      }
    }
    if (MODE == 2) { __syncthreads(); acc.x = *(float*)(lds + t * 4); }
    else acc = v[0];
  }
  if (acc.x == 123.456f) sink[0] = acc.x + acc.y;
}

template <int MODE, int DEPTH>
double run(const char* src, size_t wg_stride, unsigned span, int iters, float* sink, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t ldsb = DEPTH * 8 * 1024;
  CK(hipFuncSetAttribute((const void*)k_probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipLaunchKernelGGL((k_probe<MODE, DEPTH>), dim3(grid), dim3(512), ldsb, 0, src, wg_stride, span, iters, sink);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_probe<MODE, DEPTH>), dim3(grid), dim3(512), ldsb, 0, src, wg_stride, span, iters, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * iters * DEPTH * 8192.0;
  return bytes / (ms * 1e-3) / 1e9;   // GB/s chip-wide
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* buf; float* sink;
  CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big)); CK(hipMalloc(&sink, 16));
  const int grid = 256;
  printf("%-34s %10s %10s %10s   (GB/s chip-wide; / 256 CUs / ~2.1 GHz = B/clk/CU)\n", "source", "LDS-DMA", "->VGPR", "->VGPR->LDS");
  struct Case { const char* name; size_t stride; unsigned span; int iters; } cases[] = {
    {"shared 16 KB (L1-resident)", 0, 16u << 10, 4000},
    {"shared 512 KB (L2, beyond L1)", 0, 512u << 10, 4000},
    {"shared 8 MB (L2 of 8 XCDs / MALL)", 0, 8u << 20, 2000},
    {"private 8 MB slabs (HBM)", (size_t)8 << 20, 8u << 20, 250},
  };
  for (auto& c : cases) {
    double a4 = run<0, 4>(buf, c.stride, c.span, c.iters, sink, grid), a8 = run<0, 8>(buf, c.stride, c.span, c.iters / 2, sink, grid);
    double b4 = run<1, 4>(buf, c.stride, c.span, c.iters, sink, grid), b8 = run<1, 8>(buf, c.stride, c.span, c.iters / 2, sink, grid);
    double c4 = run<2, 4>(buf, c.stride, c.span, c.iters, sink, grid), c8 = run<2, 8>(buf, c.stride, c.span, c.iters / 2, sink, grid);
    printf("%-34s depth4 %8.0f %10.0f %10.0f\n", c.name, a4, b4, c4);
    printf("%-34s depth8 %8.0f %10.0f %10.0f   -> %.1f / %.1f / %.1f B/clk/CU at 2.1 GHz\n", "", a8, b8, c8, a8 / 256 / 2.1, b8 / 256 / 2.1, c8 / 256 / 2.1);
  }
  return 0;
}
