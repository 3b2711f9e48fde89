// Node-embedding front in one pass (r03): out = act3(W3 act2(W2 act1(W1 x + b1) + b2) + b3) for the shipped shape
// x [M, <= 8] -> 32 -> 64 -> 128 (gnn_models.py:137-178 with node_feature_embedding_layer_dimensions [32, 64, 128, 224]; the last
// Linear, 128 -> 224, is folded into the first conv layer, mpnn_layers.py).  Three launches that wrote and re-read [M, 32] and
// [M, 64] become one that reads 4 k0 bytes and writes 512 per node: 0.07 -> 0.03 ms on the C2 batch.
//
// One work-group = 8 waves x 32 rows.  Layer 1 (k0 <= 8 inputs) is plain fp32 FMAs, every lane producing exactly the 16 hidden
// values of ITS row that its MFMA A-fragments of layer 2 hold (k = 8 (lane >> 5) .. + 7 of each 16-wide k-step).  Layers 2 and 3
// run in the f16x2 form of linear_dma.hip (two f16 terms per operand after an exact power-of-two pre-scale, products l h', h l',
// h h' on v_mfma_f32_32x32x16_f16, fp32 accumulate) with their weight planes (rgnn_linear_split_weights_f16) resident in LDS; the
// pre-scale of an activation fragment comes from the maximum the wave itself just computed, so no device-wide bound is needed.
// Between layers 2 and 3 a wave turns its 32 x 64 accumulator tile (one column per lane) into row fragments through a private
// LDS block.  The output's bound (max |out|) is raised like every producer's (rgnn.h).
#include "linear_common.h"

namespace {

typedef short raw16x8 __attribute__((ext_vector_type(8)));
constexpr int EM_THREADS = 512, EM_ROWS = 256;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// 2^sa with bound 2^sa < 2^15 (the rule of linear_dma.hip), and its inverse
__device__ __forceinline__ void prescale_of(float bound, float& a_mul, float& inv) {
  const int be = (int)((__float_as_uint(bound) >> 23) & 255u);
  int se = 268 - be;
  se = se > 253 ? 253 : se;
  a_mul = __uint_as_float((unsigned)se << 23);
  inv = __uint_as_float((unsigned)(254 - se) << 23);
}
__device__ __forceinline__ void split8(const float (&x)[8], float mul, raw16x8& h, raw16x8& l) {
  f16x8_t hh, ll;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float v = x[i] * mul;
    const _Float16 a = (_Float16)v;
    hh[i] = a;
    ll[i] = (_Float16)(v - (float)a);
  }
  h = __builtin_bit_cast(raw16x8, hh);
  l = __builtin_bit_cast(raw16x8, ll);
}
__device__ __forceinline__ f32x16 mm3(const raw16x8& ah, const raw16x8& al, const raw16x8& bh, const raw16x8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, al), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bl), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, ah), __builtin_bit_cast(f16x8_t, bh), c, 0, 0, 0);
  return c;
}

template <int N1, int N2, int N3>
__global__ __launch_bounds__(EM_THREADS) void k_embed3(const float* __restrict__ x, int64_t ldx, int k0, const float* __restrict__ W1,
                                                      int64_t ldw1, const float* __restrict__ b1, const _Float16* __restrict__ P2,
                                                      const float* __restrict__ b2, const _Float16* __restrict__ P3,
                                                      const float* __restrict__ b3, int relu3, int64_t m, float* __restrict__ out,
                                                      int64_t ldo, float* __restrict__ out_absmax) {
  constexpr int KS2 = N1 / 16, KS3 = N2 / 16, T2 = N2 / 32, T3 = N3 / 32, LDS2 = N2 + 4;
  constexpr int P2_HALFS = KS2 * 2 * N2 * 16, P3_HALFS = KS3 * 2 * N3 * 16;
  extern __shared__ __attribute__((aligned(16))) char em_lds[];
  float* const w1s = (float*)em_lds;                                    // [N1][8] + [N1] bias
  float* const b2s = w1s + N1 * 9;                                      // [N2]
  float* const b3s = b2s + N2;                                          // [N3]
  _Float16* const p2s = (_Float16*)(b3s + N3);
  _Float16* const p3s = p2s + P2_HALFS;
  float* const stage = (float*)(p3s + P3_HALFS);                        // [8 waves][32][LDS2]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5;
  for (int i = t; i < N1 * 8; i += EM_THREADS) w1s[i] = ((i & 7) < k0) ? W1[(int64_t)(i >> 3) * ldw1 + (i & 7)] : 0.f;
  for (int i = t; i < N1; i += EM_THREADS) w1s[N1 * 8 + i] = b1 ? b1[i] : 0.f;
  for (int i = t; i < N2; i += EM_THREADS) b2s[i] = b2 ? b2[i] : 0.f;
  for (int i = t; i < N3; i += EM_THREADS) b3s[i] = b3 ? b3[i] : 0.f;
  for (int i = t; i < P2_HALFS / 8; i += EM_THREADS) ((uint4*)p2s)[i] = ((const uint4*)P2)[i];
  for (int i = t; i < P3_HALFS / 8; i += EM_THREADS) ((uint4*)p3s)[i] = ((const uint4*)P3)[i];
  const float w2_inv = ((const float*)(P2 + P2_HALFS))[2], w3_inv = ((const float*)(P3 + P3_HALFS))[2];   // footer: 1 / 2^sw
  __syncthreads();
  float* const my_stage = stage + wave * 32 * LDS2;
  float amax = 0.f;
  const int64_t tiles = (m + EM_ROWS - 1) / EM_ROWS;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = tile * EM_ROWS + wave * 32;
    const int64_t row = row0 + (lane & 31);
    float xin[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xin[q] = (q < k0 && row < m) ? x[row * ldx + q] : 0.f;
    // ---- layer 1: K FMAs in k order, then the bias, then the clamp (k_linear_tiny's arithmetic)
    float h1[KS2][8];
    float m1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS2; ks++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int c = 16 * ks + 8 * half + i;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) acc = fmaf(xin[q], w1s[c * 8 + q], acc);
        acc += w1s[N1 * 8 + c];
        acc = fmaxf(acc, 0.f);
        h1[ks][i] = acc;
        m1 = fmaxf(m1, acc);
      }
    m1 = wave_max(m1);
    float mul1, inv1;
    prescale_of(m1, mul1, inv1);
    // ---- layer 2
    f32x16 acc2[T2];
#pragma unroll
    for (int j = 0; j < T2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc2[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS2; ks++) {
      raw16x8 ah, al;
      split8(h1[ks], mul1, ah, al);
#pragma unroll
      for (int j = 0; j < T2; j++) {
        const int col = j * 32 + (lane & 31);
        const raw16x8 bh = *(const raw16x8*)(p2s + ((ks * 2 + 0) * N2 + col) * 16 + 8 * half);
        const raw16x8 bl = *(const raw16x8*)(p2s + ((ks * 2 + 1) * N2 + col) * 16 + 8 * half);
        acc2[j] = mm3(ah, al, bh, bl, acc2[j]);
      }
    }
    const float o2 = inv1 * w2_inv;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < T2; j++) {
      const int col = j * 32 + (lane & 31);
      const float bb = b2s[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = fmaxf(acc2[j][r] * o2 + bb, 0.f);
        m2 = fmaxf(m2, v);
        my_stage[((r & 3) + 8 * (r >> 2) + 4 * half) * LDS2 + col] = v;
      }
    }
    m2 = wave_max(m2);
    float mul2, inv2;
    prescale_of(m2, mul2, inv2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- layer 3
    f32x16 acc3[T3];
#pragma unroll
    for (int j = 0; j < T3; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc3[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS3; ks++) {
      const float4 x0 = *(const float4*)(my_stage + (lane & 31) * LDS2 + 16 * ks + 8 * half);
      const float4 x1 = *(const float4*)(my_stage + (lane & 31) * LDS2 + 16 * ks + 8 * half + 4);
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      raw16x8 ah, al;
      split8(xs, mul2, ah, al);
#pragma unroll
      for (int j = 0; j < T3; j++) {
        const int col = j * 32 + (lane & 31);
        const raw16x8 bh = *(const raw16x8*)(p3s + ((ks * 2 + 0) * N3 + col) * 16 + 8 * half);
        const raw16x8 bl = *(const raw16x8*)(p3s + ((ks * 2 + 1) * N3 + col) * 16 + 8 * half);
        acc3[j] = mm3(ah, al, bh, bl, acc3[j]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                     // (the staging block is free for the next tile)
    const float o3 = inv2 * w3_inv;
#pragma unroll
    for (int j = 0; j < T3; j++) {
      const int col = j * 32 + (lane & 31);
      const float bb = b3s[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = acc3[j][r] * o3 + bb;
        if (relu3) v = fmaxf(v, 0.f);
        const int64_t orow = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (orow < m) {
          out[orow * ldo + col] = v;
          amax = fmaxf(amax, fabsf(v));
        }
      }
    }
  }
  if (out_absmax != nullptr) {
    amax = wave_max(amax);
    if (lane == 0) bound_raise(out_absmax, blockIdx.x * 8 + wave, amax);
  }
}

}  // namespace

extern "C" int32_t rgnn_embed3_supported(int32_t k0, int32_t n1, int32_t n2, int32_t n3) {
  return (k0 >= 1 && k0 <= 8 && n1 == 32 && n2 == 64 && n3 == 128) ? 1 : 0;
}

extern "C" int rgnn_embed3(const float* x, int64_t ldx, int32_t k0, const float* W1, int64_t ldw1, const float* b1, int32_t n1,
                           const void* W2_planes_f16, const float* b2, int32_t n2, const void* W3_planes_f16, const float* b3,
                           int32_t n3, int32_t relu3, int64_t m, float* out, int64_t ldo, float* out_absmax, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0, "negative m");
  if (m == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rgnn_embed3_supported(k0, n1, n2, n3), "rgnn_embed3 takes <= 8 inputs and layers of 32, 64 and 128 columns");
  RGNN_CHECK_ARG(x && W1 && W2_planes_f16 && W3_planes_f16 && out, "null pointers");
  constexpr int N1 = 32, N2 = 64, N3 = 128;
  const size_t lds = (size_t)(N1 * 9 + N2 + N3) * 4 + (size_t)(N1 / 16 * 2 * N2 * 16 + N2 / 16 * 2 * N3 * 16) * 2 + (size_t)8 * 32 * (N2 + 4) * 4;
  static RgnnOncePerDevice attr_once;                       // (per kernel and device: common.h)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_embed3<N1, N2, N3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int64_t tiles = (m + EM_ROWS - 1) / EM_ROWS;
  const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
  hipLaunchKernelGGL((k_embed3<N1, N2, N3>), dim3(grid), dim3(EM_THREADS), lds, (hipStream_t)stream, x, ldx, k0, W1, ldw1, b1,
                     (const _Float16*)W2_planes_f16, b2, (const _Float16*)W3_planes_f16, b3, relu3, m, out, ldo, out_absmax);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
