// Micro-probe: issue rate of v_mfma_f32_32x32x2_f32 under different co-issue conditions (tools only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int FILL_VALU, int FILL_LDS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
  float v0 = a, v1 = b;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int k = 0; k < NACC; k++) {
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FILL_VALU; f++) { v0 = v0 * 1.0001f + v1; v1 = v1 * 0.9999f + v0; }
        if (FILL_LDS && ((u * NACC + k) % FILL_LDS == 0)) {
          float4 t = *(const float4*)(lp + ((it + u) & 63) * 4);
          a += t.x * 1e-30f; b += t.y * 1e-30f;
        }
      }
    }
  }
  float s = v0 + v1;
  for (int k = 0; k < NACC; k++) for (int r = 0; r < 16; r++) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FV, int FL>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC, FV, FL><<<256 * blocks_per_cu, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC, FV, FL><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 32 * 32 * 2 * 16.0 * NACC * iters * 4.0 * 256 * blocks_per_cu;
  printf("%-28s waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; w++) {
    run<4, 0, 0>("4acc pure", w);
    run<2, 0, 0>("2acc pure", w);
    run<4, 1, 0>("4acc +2valu/mfma", w);
    run<4, 3, 0>("4acc +6valu/mfma", w);
    run<4, 0, 4>("4acc +lds b128 /4mfma", w);
    run<4, 2, 4>("4acc +4valu +lds/4", w);
  }
  return 0;
}
