// What HBM delivers to the ACCESS PATTERN of the dense kernel's activation stream: 256-row panels of a row-major fp32 matrix
// [M x K], each k-step fetching SEG bytes of every row of the panel (the kernel: SEG = 64 = one MFMA k-extent of fp32), DEPTH
// k-steps in flight per thread.  hipcc --offload-arch=gfx950 -O3 tools/attic/stride_read_probe.hip -o tools/stride_read_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SEG, int DEPTH>
__global__ __launch_bounds__(512) void k_read(const float* __restrict__ a, long m, int k, float* __restrict__ sink) {
  constexpr int LPR = SEG / 16;                    // lanes per row
  constexpr int RPP = 512 / LPR;                   // rows per pass of the work-group
  constexpr int PASSES = 256 / RPP;
  const int t = threadIdx.x;
  const long panels = (m + 255) / 256;
  const int steps = k * 4 / SEG;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long p = blockIdx.x; p < panels; p += gridDim.x) {
    const float* base[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; q++) {
      long row = p * 256 + q * RPP + t / LPR;
      if (row >= m) row = m - 1;
      base[q] = a + row * (long)k + (t % LPR) * 4;
    }
    for (int s0 = 0; s0 < steps; s0 += DEPTH) {
      float4 v[DEPTH][PASSES];
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int q = 0; q < PASSES; q++)
          v[d][q] = (s0 + d < steps) ? *(const float4*)(base[q] + (long)(s0 + d) * (SEG / 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int q = 0; q < PASSES; q++) { acc.x += v[d][q].x; acc.y += v[d][q].y; acc.z += v[d][q].z; acc.w += v[d][q].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int SEG, int DEPTH>
void run(const char* name, const float* a, long m, int k, float* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < 12; it++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_read<SEG, DEPTH>), dim3(256), dim3(512), 0, 0, a, m, k, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)m * k * 4;
  printf("  %-34s median %7.1f us  %6.0f GB/s   (best %6.0f GB/s)\n", name, ts[ts.size() / 2] * 1e3, bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6);
}

int main() {
  const long m = 192000 * 4;                       // 4 x the C2 batch: 2.1 GB at K = 688, beyond L2 + MALL
  float* sink; CK(hipMalloc(&sink, 16));
  for (int k : {224, 688}) {
    float* a; CK(hipMalloc(&a, (size_t)m * k * 4)); CK(hipMemset(a, 0, (size_t)m * k * 4));
    printf("M = %ld rows, K = %d fp32 (row = %d B), 256 work-groups x 512 threads, 256-row panels:\n", m, k, k * 4);
    run<64, 2>("64 B per row and step, 2 in flight", a, m, k, sink);
    run<64, 4>("64 B per row and step, 4 in flight", a, m, k, sink);
    run<64, 7>("64 B per row and step, 7 in flight", a, m, k, sink);
    run<128, 2>("128 B per row and step, 2 in flight", a, m, k, sink);
    run<128, 4>("128 B per row and step, 4 in flight", a, m, k, sink);
    run<256, 2>("256 B per row and step, 2 in flight", a, m, k, sink);
    CK(hipFree(a));
  }
  return 0;
}
