// Column statistics shared by the dense epilogues (linear_common.h) and the BatchNorm kernels (norm.hip).
#pragma once
#include "common.h"

namespace {

// Column statistics of a 128-row panel are kept as {count, pivot, s1 = sum (v - pivot), s2 = sum (v - pivot)^2}
// (RGNN_STAT_ROWS floats per column, rgnn.h): every lane sums its rows about a PIVOT taken from the data (the first value it
// produces), so a column whose spread is tiny against its mean loses nothing to cancellation or to the rounding of a mean --
// a constant column has s1 == s2 == 0 exactly, and sums of small differences are exact or nearly so.  Two partial results are
// merged by moving the second onto the pivot of the first.  Mean and variance are only formed in float64 (norm.hip).
struct ColStat { float n, piv, s1, s2; };
__device__ __forceinline__ ColStat stat_make(float cnt, float piv, float s1, float s2) {
  ColStat r;
  r.n = cnt; r.piv = cnt > 0.f ? piv : 0.f; r.s1 = cnt > 0.f ? s1 : 0.f; r.s2 = cnt > 0.f ? s2 : 0.f;
  return r;
}
__device__ __forceinline__ ColStat stat_merge(const ColStat a, const ColStat b) {
  if (a.n <= 0.f) return b;                           // (keeps the pivot next to the data)
  const float d = b.piv - a.piv;                      // exact when the pivots are close (the case that matters)
  ColStat r;
  r.n = a.n + b.n;
  r.piv = a.piv;
  r.s1 = a.s1 + (b.s1 + b.n * d);
  r.s2 = a.s2 + (b.s2 + 2.f * d * b.s1 + b.n * d * d);
  return r;
}

__device__ __forceinline__ ColStat stat_merge_xor(const ColStat a, int off) {                    // merge with lane ^ off
  ColStat b;
  b.n = __shfl_xor(a.n, off, 64); b.piv = __shfl_xor(a.piv, off, 64); b.s1 = __shfl_xor(a.s1, off, 64); b.s2 = __shfl_xor(a.s2, off, 64);
  return stat_merge(a, b);
}
__device__ __forceinline__ void stat_store(float* __restrict__ dst, int64_t stride, const ColStat a) {   // rows of one column
  dst[0] = a.n; dst[stride] = a.piv; dst[2 * stride] = a.s1; dst[3 * stride] = a.s2;
}
__device__ __forceinline__ ColStat stat_load(const float* __restrict__ src, int64_t stride) {
  ColStat a;
  a.n = src[0]; a.piv = src[stride]; a.s1 = src[2 * stride]; a.s2 = src[3 * stride];
  return a;
}
// LDS exchange between the waves of a work-group: {pivot, s1, s2} per (wave, column) + ONE count per wave (a wave counts the
// same rows for every column) -- the LDS-DMA kernel's budget at 224-column tiles has no room for a fourth float per column.
constexpr int STAT_LDS_ROWS = 3;
__host__ __device__ constexpr int stat_lds_floats(int waves_m, int bn) { return waves_m * bn * STAT_LDS_ROWS + waves_m; }

// float64 sums about a common pivot K: what a panel {n, piv, s1, s2} adds to (rows, sum (v - K), sum (v - K)^2)
__device__ __forceinline__ void stat_accumulate(double n, double piv, double s1, double s2, double K, double& S0, double& S1, double& S2) {
  const double d = piv - K;
  S0 += n;
  S1 += s1 + n * d;
  S2 += s2 + 2.0 * d * s1 + n * d * d;
}

// Per-segment statistics travel between the segment kernels of norm.hip as float64 {mean, M2 = sum (v - mean)^2}, formed from
// sums about a pivot K: rows m, t1 = sum (v - K), t2 = sum (v - K)^2.
__device__ __forceinline__ void seg_moments_store(double* __restrict__ seg_sums, int64_t f, int n, int c, int64_t m, double K, double t1, double t2) {
  double mean = 0.0, m2 = 0.0;
  if (m > 0) {
    const double dk = t1 / (double)m;
    mean = K + dk;
    m2 = t2 - t1 * dk;
    if (m2 < 0.0) m2 = 0.0;
  }
  seg_sums[(f * 2 + 0) * n + c] = mean;
  seg_sums[(f * 2 + 1) * n + c] = m2;
}

// One entry of the BatchNorm-apply table (RGNN_AFFINE_ROWS rows, rgnn.h): y = (x - mean_hi) g + t with mean_hi = fl(mean),
// g = fl(gamma / sqrt(var + eps)) and t = fl(beta - (mean - mean_hi) g) -- the subtraction of a value next to the data is exact
// or nearly so, where the single multiply-add x scale + (beta - mean scale) of r03 (and of ATen's CPU kernel) rounds
// mean * scale: 6e-8 |mean| / std of the normalised value, 1e-4 on a column that is constant up to rounding.
__device__ __forceinline__ void bn_table_entry(double mean, double var, double gm, double bt, double eps, float& mean_hi, float& g, float& t) {
  mean_hi = (float)mean;
  g = (float)(gm / sqrt(var + eps));
  t = (float)(bt - (mean - (double)mean_hi) * (double)g);
}

}  // namespace
