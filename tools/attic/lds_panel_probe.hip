// Prototype for VERDICT r02 item 7: the K = 464 part of a conv layer's update GEMM run from AGGREGATED ROWS PARKED IN
// LDS instead of rows written to HBM by the edge kernel and read back by the dense kernel.  One work-group per CU owns 64-target
// panels: (a) the panel's 64 x 464 fp32 rows arrive in LDS (here: copied from a global matrix, standing in for the edge phase
// that would produce them in place), (b) eight waves multiply them with W_pm [224 x 464] in the f16x2 form (two f16 terms per
// operand, three v_mfma_f32_32x32x16_f16 per fp32 product; weight fragments straight from L2 into registers, activation
// fragments from LDS, split on the fly), (c) the 64 x 224 result is stored.  Timed: (a)+(b)+(c), (b)+(c) alone (LDS contents
// left as they are), and (b) alone.  Build: hipcc --offload-arch=gfx950 -O3 tools/attic/lds_panel_probe.hip -o tools/lds_panel_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KD = 464, ND = 224, KS = KD / 16, NT = ND / 32, PR = 64, LDSW = 484;   // LDS row stride in words: 4 mod 32 (b128 reads)

template <int MODE>   // 0: fill + multiply + store, 1: multiply + store, 2: multiply only
__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ m_rows, long rows, const _Float16* __restrict__ wp,
                                               float* __restrict__ out) {
  extern __shared__ float panel[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rb = wave & 1, cg = wave >> 1;
  const long panels = (rows + PR - 1) / PR;
  float sink = 0.f;
  for (long p = blockIdx.x; p < panels; p += gridDim.x) {
    if (MODE == 0) {
      __syncthreads();
      for (int i = t; i < PR * (KD / 4); i += 512) {
        const int r = i / (KD / 4), c4 = i % (KD / 4);
        long gr = p * PR + r; if (gr >= rows) gr = rows - 1;
        *(float4*)(panel + r * LDSW + c4 * 4) = *(const float4*)(m_rows + gr * KD + c4 * 4);
      }
      __syncthreads();
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float* arow = panel + (rb * 32 + (lane & 31)) * LDSW + 8 * (lane >> 5);
    const int ntile = (cg + 4 < NT) ? 2 : 1;
    // weight fragments two k-steps ahead of the MFMAs that use them (they come from L2: ~1 us away), activation fragment one ahead
    auto load_b = [&](int ks, f16x8 (&bh)[2], f16x8 (&bl)[2]) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int col = (cg + 4 * (j < ntile ? j : 0)) * 32 + (lane & 31);
        const int kk = ks < KS ? ks : KS - 1;
        bh[j] = *(const f16x8*)(wp + ((long)(0 * KS + kk) * ND + col) * 16 + 8 * (lane >> 5));
        bl[j] = *(const f16x8*)(wp + ((long)(1 * KS + kk) * ND + col) * 16 + 8 * (lane >> 5));
      }
    };
    auto load_a = [&](int ks, f16x8& ah, f16x8& al) {
      const int kk = ks < KS ? ks : KS - 1;
      const float4 x0 = *(const float4*)(arow + kk * 16), x1 = *(const float4*)(arow + kk * 16 + 4);
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int i = 0; i < 8; i++) { const _Float16 h = (_Float16)xs[i]; ah[i] = h; al[i] = (_Float16)(xs[i] - (float)h); }
    };
    f16x8 bh0[2], bl0[2], bh1[2], bl1[2], bh2[2], bl2[2], ah0, al0, ah1, al1;
    load_b(0, bh0, bl0); load_b(1, bh1, bl1); load_a(0, ah0, al0);
    auto step = [&](int ks, f16x8 (&ch)[2], f16x8 (&cl)[2], f16x8 (&fh)[2], f16x8 (&fl)[2], f16x8& ach, f16x8& acl, f16x8& afh, f16x8& afl) {
      if (ks >= KS) return;
      load_b(ks + 2, fh, fl);
      load_a(ks + 1, afh, afl);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= ntile) break;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(acl, ch[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach, cl[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ach, ch[j], acc[j], 0, 0, 0);
      }
    };
    for (int ks = 0; ks < KS; ks += 6) {              // (rings of three weight and two activation slots, named: no indexed registers)
      step(ks + 0, bh0, bl0, bh2, bl2, ah0, al0, ah1, al1);
      step(ks + 1, bh1, bl1, bh0, bl0, ah1, al1, ah0, al0);
      step(ks + 2, bh2, bl2, bh1, bl1, ah0, al0, ah1, al1);
      step(ks + 3, bh0, bl0, bh2, bl2, ah1, al1, ah0, al0);
      step(ks + 4, bh1, bl1, bh0, bl0, ah0, al0, ah1, al1);
      step(ks + 5, bh2, bl2, bh1, bl1, ah1, al1, ah0, al0);
    }
    if (MODE <= 1) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j >= ntile) break;
        const int col = (cg + 4 * j) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long gr = p * PR + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gr < rows) out[gr * ND + col] = acc[j][r];
        }
      }
    } else {
      sink += acc[0][0] + acc[1][5];
    }
  }
  if (MODE == 2 && sink == 123.456f) out[0] = sink;
}

template <int MODE>
double run(const char* name, const float* m, long rows, const _Float16* wp, float* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = (size_t)PR * LDSW * 4;
  CK(hipFuncSetAttribute((const void*)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<float> ts;
  for (int it = 0; it < 12; it++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<MODE>), dim3(256), dim3(512), lds, 0, m, rows, wp, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("  %-58s median %7.1f us (best %7.1f)\n", name, ts[ts.size() / 2] * 1e3, ts[0] * 1e3);
  return ts[ts.size() / 2] * 1e3;
}

int main() {
  const long rows = 133516;                        // targets with edges of the C2 batch
  std::vector<float> hm((size_t)rows * KD), hw((size_t)ND * KD);
  srand(1);
  for (auto& v : hm) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  std::vector<_Float16> hp((size_t)2 * KS * ND * 16);
  for (int ks = 0; ks < KS; ks++)
    for (int c = 0; c < ND; c++)
      for (int i = 0; i < 16; i++) {
        const float w = hw[(size_t)c * KD + ks * 16 + i];
        const _Float16 h = (_Float16)w;
        hp[((size_t)(0 * KS + ks) * ND + c) * 16 + i] = h;
        hp[((size_t)(1 * KS + ks) * ND + c) * 16 + i] = (_Float16)(w - (float)h);
      }
  float *dm, *dout; _Float16* dp;
  CK(hipMalloc(&dm, hm.size() * 4)); CK(hipMalloc(&dout, (size_t)rows * ND * 4)); CK(hipMalloc(&dp, hp.size() * 2));
  CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
  printf("aggregated rows [%ld x %d] x W_pm^T [%d x %d], 64-row panels parked in LDS, 256 work-groups x 8 waves, f16x2 form:\n", rows, KD, KD, ND);
  run<0>("(a) rows global -> LDS, (b) multiply from LDS, (c) store", dm, rows, dp, dout);
  // correctness of (a)+(b)+(c) against float64 on a few rows
  std::vector<float> ho((size_t)rows * ND);
  CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, big = 0;
  for (long r : {0L, 63L, 64L, 70001L, rows - 1})
    for (int c = 0; c < ND; c++) {
      double s = 0;
      for (int k = 0; k < KD; k++) s += (double)hm[(size_t)r * KD + k] * (double)hw[(size_t)c * KD + k];
      worst = std::max(worst, fabs(s - (double)ho[(size_t)r * ND + c])); big = std::max(big, fabs(s));
    }
  printf("  max |error| / max |value| on sampled rows against float64: %.2e\n", worst / big);
  run<1>("(b) + (c): LDS contents as they are", dm, rows, dp, dout);
  run<2>("(b) alone", dm, rows, dp, dout);
  return 0;
}
