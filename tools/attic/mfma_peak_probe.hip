// What the bf16 matrix pipe SUSTAINS on this part (tools only): a kernel of nothing but independent v_mfma_f32_32x32x16_bf16,
// 8 waves per CU (two per SIMD, like k_linear_dma) or 4, for ~2 ms.  Prints TFLOP/s against the 2 500 quoted as dense peak --
// the clock under this load is what it is (power), so this is the ceiling any GEMM on the part can be measured against.
//   hipcc --offload-arch=gfx950 -O3 tools/attic/mfma_peak_probe.hip -o tools/mfma_peak_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = seed * a;
  bf16x8 fa, fb;
  for (int e = 0; e < 8; e++) { fa[e] = (__bf16)(seed + e + threadIdx.x); fb[e] = (__bf16)(seed * 2 + e); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 8; waves >= 4; waves -= 4) {
    const int iters = 20000;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe<4>, dim3(256), dim3(64 * waves), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 256.0 * waves * iters * 4 * (2.0 * 32 * 32 * 16);
      printf("%d waves per CU: %.2f ms, %.0f TFLOP/s bf16 = %.2f of 2500 (implied clock if the pipe never idles: %.2f GHz)\n", waves, ms,
             flops / ms / 1e9, flops / ms / 1e9 / 2500.0, flops / ms / 1e9 / 2500.0 * 2.4);
    }
  }
  return 0;
}
