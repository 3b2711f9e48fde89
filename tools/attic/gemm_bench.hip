// Stand-alone micro-benchmark of the dense-layer kernel (tools only, not part of librgnn.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/attic/gemm_bench.hip -o /tmp/gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/rgnn.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  struct Shape { int64_t m; int k1, k2, n; };
  std::vector<Shape> shapes = {{192000, 224, 0, 512}, {192000, 128, 0, 512}, {192000, 448, 0, 512}, {192000, 896, 0, 512}, {192000, 224, 0, 464}, {192000, 224, 0, 928}, {192000, 224, 464, 224}, {192000, 224, 464, 128},
                               {192000, 128, 0, 544}, {192000, 128, 272, 64}, {192000, 128, 0, 224}, {192000, 64, 0, 128},
                               {8192, 4096, 0, 4096}};
  for (auto s : shapes) {
    const int K = s.k1 + s.k2;
    float *A1, *A2 = nullptr, *W, *b, *out, *stats;
    CK(hipMalloc(&A1, s.m * (size_t)s.k1 * 4));
    if (s.k2) CK(hipMalloc(&A2, s.m * (size_t)s.k2 * 4));
    CK(hipMalloc(&W, (size_t)s.n * K * 4));
    CK(hipMalloc(&b, s.n * 4));
    CK(hipMalloc(&out, s.m * (size_t)s.n * 4));
    CK(hipMalloc(&stats, rgnn_linear_stat_panels(s.m) * RGNN_STAT_ROWS * (size_t)s.n * 4));
    std::vector<float> h(s.m * (size_t)(s.k1 > s.k2 ? s.k1 : s.k2));
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(A1, h.data(), s.m * (size_t)s.k1 * 4, hipMemcpyHostToDevice));
    if (s.k2) CK(hipMemcpy(A2, h.data(), s.m * (size_t)s.k2 * 4, hipMemcpyHostToDevice));
    std::vector<float> hw((size_t)s.n * K);
    for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b, 0, s.n * 4));
    rgnn_linear_args a = {};
    a.A1 = A1; a.lda1 = s.k1; a.k1 = s.k1; a.A2 = A2; a.lda2 = s.k2; a.k2 = s.k2;
    a.W1 = W; a.W2 = nullptr; a.ldw = K; a.w_split = s.n; a.bias1 = b; a.out = out; a.ldo = s.n; a.m = s.m; a.n = s.n;
    a.col_stats = (argc > 1) ? stats : nullptr;
    void* planes = nullptr;
    if (getenv("GB_X3")) {
      const int kp = rgnn_linear_planes_kp(K);
      CK(hipMalloc(&planes, (size_t)3 * s.n * kp * 2));
      rgnn_linear_split_weights(W, nullptr, K, s.n, s.n, K, planes, nullptr);
      a.W_planes = planes; a.w_planes_kp = kp;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) rgnn_linear_fwd(&a, nullptr);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++) rgnn_linear_fwd(&a, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double fl = 2.0 * s.m * K * s.n;
    printf("M=%ld K=%d(+%d) N=%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", (long)s.m, s.k1, s.k2, s.n, ms, fl / ms / 1e9,
           100 * fl / ms / 1e9 / 157.3);
    hipFree(A1); hipFree(A2); hipFree(W); hipFree(b); hipFree(out); hipFree(stats);
  }
  return 0;
}
