// Micro-probe (tools only): can six v_mfma_f32_32x32x16_bf16 on a 3-way bf16 split of fp32 operands beat
// v_mfma_f32_32x32x2_f32 once the in-register split of the A operand (VALU) and the fragment reads (LDS) are paid?
// One iteration = one k32 step of a 64x64 wave tile: 2 k16 halves x 4 tiles x 6 products = 48 MFMAs, SPLIT_VALU extra
// VALU instructions, 24 ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SPLIT_VALU, int LDS_READS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed + i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  bf16x8 fa[2][3], fb[2][3];
  for (int i = 0; i < 2; i++) for (int p = 0; p < 3; p++) for (int e = 0; e < 8; e++) {
    fa[i][p][e] = (__bf16)(seed + i + p + e + threadIdx.x); fb[i][p][e] = (__bf16)(seed * 2 + i + p + e);
  }
  float v0 = seed + threadIdx.x, v1 = seed * 3;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (LDS_READS) {
#pragma unroll
        for (int q = 0; q < LDS_READS / 2; q++) {
          const float4 t = *(const float4*)(lp + ((it * 2 + h + q * 7) & 31) * 256);
          fa[q & 1][q % 3][0] = (__bf16)t.x; fb[q & 1][(q + 1) % 3][1] = (__bf16)t.y;
        }
      }
      // products: hh, hm, mh, hl, lh, mm
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          f32x16 c = acc[i * 2 + j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
          acc[i * 2 + j] = c;
        }
#pragma unroll
      for (int f = 0; f < SPLIT_VALU / 4; f++) { v0 = v0 * 1.0001f + v1; v1 = v1 * 0.9999f + v0; }
    }
  }
  float s = v0 + v1;
  for (int k = 0; k < 4; k++) for (int r = 0; r < 16; r++) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SV, int LR>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<SV, LR><<<256 * blocks_per_cu, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<SV, LR><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // fp32-equivalent flops: a 64x64 wave tile, k = 32 per iteration
  double flops = 2.0 * 64 * 64 * 32 * (double)iters * 4.0 * 256 * blocks_per_cu;
  printf("%-34s waves/SIMD=%d : %.3f ms  %.1f fp32-equivalent TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; w++) {
    run<0, 0>("pure 48 mfma / k32", w);
    run<88, 0>("+88 valu / k32 (split A)", w);
    run<176, 0>("+176 valu / k32 (split A and W)", w);
    run<88, 24>("+88 valu +24 ds_read_b128", w);
  }
  return 0;
}
