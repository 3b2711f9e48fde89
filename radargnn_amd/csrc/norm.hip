// Train-mode BatchNorm1d (gnn/gnn_models.py:71-73,126-128 via torch_geometric.nn.BatchNorm), the activation
// that follows it, and the row softmax of postprocessor/inference.py:46,62.
//
// The column statistics arrive as {count, pivot, sum (v - pivot), sum (v - pivot)^2} per 128-row panel (RGNN_STAT_ROWS, rgnn.h;
// stats.h) from the epilogue of the dense layer (linear.hip); they are combined here in float64 about the first panel's
// pivot -- deterministic, no atomics.
#include "common.h"
#include "stats.h"
#include <math.h>
#include <stdlib.h>

namespace {

// Upper bound of |(x - mean_hi) g + t| over a column, for the f16x2 dense form that applies the table to its A1 operand (it derives
// its exact power-of-two pre-scale from the bound; a bound 2^18 above the tensor's magnitude still costs nothing, 2^22 costs 4e-5).
//   any table:        |g| (B + |mean_hi|) + |t|, B the bound of the WHOLE input tensor.  Loose when one column is huge and
//                     another nearly constant (g = gamma / sqrt(eps) = 316 gamma): 1e6-sized outliers beside a constant column put
//                     the bound 2^28 above what the normalised tensor holds.
//   batch statistics: every row the statistics counted satisfies (x - mean)^2 <= sum_i (x_i - mean)^2 (rows - 1) / rows =
//                     (rows - 1) var, so |x - mean_hi| <= sqrt((rows - 1) var) + |mean - mean_hi|.  That is a statement about the
//                     data the table was made from, and the consumers of a training-mode table read exactly those rows.  What the
//                     fp32 partial sums may have missed is covered twice: 2^-10 relative on the deviation (their error is relative
//                     to the spread: sums about panel pivots) and 2^-18 (B + |mean_hi|) absolute (a column that is constant up
//                     to rounding has var = 0 here and deviations of a few ulp there).  The bound is the SMALLER of the two.
__device__ __forceinline__ double bn_bound(double gg, double mh, double tt, double in_bound, bool from_batch, double rows, double var) {
  const double loose = fabs(gg) * (in_bound + fabs(mh)) + fabs(tt);
  if (!from_batch || !(rows > 0.0)) return loose * 1.0001;
  const double dev = sqrt((rows > 1.0 ? rows - 1.0 : 0.0) * (var > 0.0 ? var : 0.0)) * (1.0 + 0x1p-10) + 0x1p-18 * (in_bound + fabs(mh));
  const double tight = fabs(gg) * dev + fabs(tt);
  return (tight < loose ? tight : loose) * 1.0001;
}

// From a channel's combined sums (rows, dk = mean - K, second moment about K) to its table entry, running statistics and bound.
struct BnChannel { float gamma, beta, rmean, rvar; };       // a channel's parameters, requested before the panels are (k_bn_finalize4)
__device__ __forceinline__ BnChannel bn_channel_load(int c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ running_mean, const float* __restrict__ running_var) {
  BnChannel p;
  p.gamma = gamma ? gamma[c] : 1.f; p.beta = beta ? beta[c] : 0.f;
  p.rmean = running_mean ? running_mean[c] : 0.f; p.rvar = running_var ? running_var[c] : 1.f;
  return p;
}
__device__ __forceinline__ void bn_finish_channel(int c, int n, int training, double K, double s0, double s1, double s2, int64_t m,
                                                  const BnChannel ch, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  float momentum, float eps, float* __restrict__ scale_shift,
                                                  float* __restrict__ out_bound, float in_bound_max) {
  double mean, var, rows_counted = 0.0;
  if (training) {
    const double rows = s0 > 0.0 ? s0 : (double)m;        // (the panels' own count; m only when nothing was counted)
    rows_counted = rows;
    const bool none = !(rows > 0.0);                      // (no row at all: 0 / 0 below -- mean 0, variance 0, running statistics untouched)
    const double dk = none ? 0.0 : s1 / rows;
    mean = none ? 0.0 : K + dk;
    var = none ? 0.0 : s2 / rows - dk * dk;  // biased variance, as F.batch_norm normalises with
    if (var < 0.0) var = 0.0;
    if (running_mean && !none) {
      const double unbiased = (rows > 1.0) ? var * rows / (rows - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)ch.rmean + (double)momentum * mean);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)ch.rvar + (double)momentum * unbiased);
    }
  } else {
    mean = (double)ch.rmean;
    var = (double)ch.rvar;
  }
  const double gm = (double)ch.gamma, bt = (double)ch.beta;
  float mh, gg, tt;
  bn_table_entry(mean, var, gm, bt, (double)eps, mh, gg, tt);
  scale_shift[c] = mh;
  scale_shift[n + c] = gg;
  scale_shift[2 * n + c] = tt;
  if (out_bound != nullptr) {
    const double b = bn_bound((double)gg, (double)mh, (double)tt, (double)in_bound_max, training != 0, rows_counted, var);
    atomicMax((unsigned int*)out_bound + (c & (RGNN_BOUND_SLOTS - 1)), __float_as_uint((float)b));
  }
}

// one block = 16 channels x 64 panel groups (14 blocks at C = 224 instead of 4: the reduction is latency-bound, 1500
// panels at N = 192 000): 64-B reads of the partials, float64 accumulation, LDS combine
//
// Two parts (col_stats_b != NULL): the two row-subset launches of one layer (targets with / without incoming edges) each
// leave their partial sums in a buffer of their own; live_a / live_b ([dev], optional) hold the ROW COUNT of the launch
// (its m_dev), so only the panels it really wrote are read -- the buffers need no zero fill and the half of the rows that
// no launch reaches is never fetched.
__global__ __launch_bounds__(1024) void k_bn_finalize(const float* __restrict__ col_stats, int64_t panels,
                                                     const int64_t* __restrict__ live_a,
                                                     const float* __restrict__ col_stats_b, int64_t panels_b,
                                                     const int64_t* __restrict__ live_b, int64_t m,
                                                     int n, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var,
                                                     int64_t* __restrict__ num_batches_tracked, int training,
                                                     float momentum, float eps, float* __restrict__ scale_shift,
                                                     const float* __restrict__ in_bound, float* __restrict__ out_bound) {
  constexpr int CH = 16, GR = 64, R = RGNN_STAT_ROWS;
  __shared__ double red[4][GR][CH];
  __shared__ float in_b[4];
  if (out_bound != nullptr && threadIdx.x < RGNN_BOUND_SLOTS) {     // maximum over the slots of the input's bound (rgnn.h)
    float v = in_bound[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) in_b[threadIdx.x >> 6] = v;
  }
  const int lc = threadIdx.x & (CH - 1), g = threadIdx.x / CH;
  const int c = blockIdx.x * CH + lc;
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && num_batches_tracked) *num_batches_tracked += 1;
  // Per thread: rows s0 and the sums s1 = sum (v - K), s2 = sum (v - K)^2 about K = the pivot of the first panel the thread
  // meets (no load of its own: a shared pivot read up front was one more dependent memory latency in a kernel that is a chain
  // of four of them); the 64 partials of a channel are merged in float64 by moving each onto the pivot of the first.
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, K = 0.0;
  bool have = false;
  if (training && c < n) {
    // (both row counts are requested before either is looked at: one latency, not two -- a NULL count reads a valid dummy word)
    const int64_t la = *(live_a ? live_a : (const int64_t*)col_stats);
    const int64_t lb = *(live_b ? live_b : (const int64_t*)col_stats);
    int64_t npp[2];
    npp[0] = panels;
    if (live_a) { const int64_t lp = (la + RGNN_STAT_PANEL_ROWS - 1) / RGNN_STAT_PANEL_ROWS; npp[0] = lp < panels ? lp : panels; }
    npp[1] = col_stats_b ? panels_b : 0;
    if (col_stats_b && live_b) { const int64_t lp = (lb + RGNN_STAT_PANEL_ROWS - 1) / RGNN_STAT_PANEL_ROWS; npp[1] = lp < panels_b ? lp : panels_b; }
    for (int part = 0; part < 2; part++) {
      const float* st = part ? col_stats_b : col_stats;
      const int64_t np = npp[part];
      // 8 panels (32 independent loads) in flight per thread (the loop is latency bound: 1500 panels / 64 groups).  The last
      // round is predicated, not a loop of its own; loads past the end are CLAMPED to the last panel and their values dropped
      // afterwards (a conditional load would be a branch per load and serialise the round).  Branch-free accumulation: a panel
      // that counts nothing adds exact zeros.
      for (int64_t p = g; p < np; p += 8 * GR) {
        float a[8], b[8], e[8], f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int64_t pp = p + u * GR;
          const int64_t pc = pp < np ? pp : np - 1;
          a[u] = st[(pc * R + 0) * n + c];
          b[u] = st[(pc * R + 1) * n + c];
          e[u] = st[(pc * R + 2) * n + c];
          f[u] = st[(pc * R + 3) * n + c];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool okp = p + u * GR < np && a[u] > 0.f;
          K = (okp && !have) ? (double)b[u] : K;
          have = have || okp;
          const double w = okp ? (double)a[u] : 0.0, t1 = okp ? (double)e[u] : 0.0, t2 = okp ? (double)f[u] : 0.0;
          const double d = okp ? (double)b[u] - K : 0.0;
          s0 += w; s1 += t1 + w * d; s2 += t2 + 2.0 * d * t1 + w * d * d;
        }
      }
    }
  }
  red[0][g][lc] = s0;
  red[1][g][lc] = s1;
  red[2][g][lc] = s2;
  red[3][g][lc] = K;
  __syncthreads();
  if (g != 0 || c >= n) return;
  s0 = 0.0; s1 = 0.0; s2 = 0.0; K = 0.0;
  if (training)
    for (int i = 0; i < GR; i++) {                        // (fixed order: deterministic)
      const double nb = red[0][i][lc];
      if (nb <= 0.0) continue;
      if (s0 <= 0.0) { s0 = nb; s1 = red[1][i][lc]; s2 = red[2][i][lc]; K = red[3][i][lc]; continue; }
      const double d = red[3][i][lc] - K, b1 = red[1][i][lc];
      s0 += nb; s1 += b1 + nb * d; s2 += red[2][i][lc] + 2.0 * d * b1 + nb * d * d;
    }
  bn_finish_channel(c, n, training, K, s0, s1, s2, m, bn_channel_load(c, gamma, beta, running_mean, running_var), running_mean,
                    running_var, momentum, eps, scale_shift, out_bound,
                    out_bound ? fmaxf(fmaxf(in_b[0], in_b[1]), fmaxf(in_b[2], in_b[3])) : 0.f);
}

// The same for n % 4 == 0 (every layer of the shipped models), 16-byte loads: one block = 8 channels (two quads) x 512 panel
// groups -- four times fewer load instructions for the same bytes and twice the work-groups.  (The scalar form above issued
// 1 536 four-byte wave loads per CU on 14 CUs: bound by the address rate of those CUs, and the {count, pivot, s1, s2} panels
// doubled it -- 11 -> 19 us per call.)  All sums are taken about ONE pivot per channel, the first written panel's (requested
// together with the launches' row counts: one memory latency), so partial sums simply add: wave shuffles, then 16 LDS slots.
template <int ABL>
__global__ __launch_bounds__(1024) void k_bn_finalize4(const float* __restrict__ col_stats, int64_t panels,
                                                      const int64_t* __restrict__ live_a,
                                                      const float* __restrict__ col_stats_b, int64_t panels_b,
                                                      const int64_t* __restrict__ live_b, int64_t m,
                                                      int n, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var,
                                                      int64_t* __restrict__ num_batches_tracked, int training,
                                                      float momentum, float eps, float* __restrict__ scale_shift,
                                                      const float* __restrict__ in_bound, float* __restrict__ out_bound) {
  constexpr int GR = 512, R = RGNN_STAT_ROWS;
  __shared__ double red[16][2][9];
  __shared__ double kpiv[2][4];
  __shared__ float in_b[4];
  if (out_bound != nullptr && threadIdx.x < RGNN_BOUND_SLOTS) {     // maximum over the slots of the input's bound (rgnn.h)
    float v = in_bound[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) in_b[threadIdx.x >> 6] = v;
  }
  const int q = threadIdx.x & 1, g = threadIdx.x >> 1;
  const int c0 = blockIdx.x * 8 + q * 4;
  BnChannel chp{1.f, 0.f, 0.f, 1.f};                        // (threads 0 .. 7 finish a channel each: their parameters are requested NOW)
  if (threadIdx.x < 8 && blockIdx.x * 8 + threadIdx.x < n) chp = bn_channel_load(blockIdx.x * 8 + threadIdx.x, gamma, beta, running_mean, running_var);
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && num_batches_tracked) *num_batches_tracked += 1;
  // (per thread and inside a wave the sums stay in float32: they are sums of SMALL terms -- distances to the pivot -- whose float32
  //  rounding is 1e-7 of the column's spread, and the float64 form of this part cost 3.6 of the kernel's 19 us; the waves'
  //  partial sums are added in float64)
  float s0 = 0.f, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, K[4] = {0.f, 0.f, 0.f, 0.f};
  if (training && c0 < n) {
    // (row counts and both candidate pivots are requested together: one latency; a NULL pointer reads a valid dummy word)
    const int64_t la = *(live_a ? live_a : (const int64_t*)col_stats);
    const int64_t lb = *(live_b ? live_b : (const int64_t*)col_stats);
    const float4 ka = *(const float4*)(col_stats + (int64_t)1 * n + c0);
    const float4 kb = *(const float4*)((col_stats_b ? col_stats_b : col_stats) + (int64_t)1 * n + c0);
    int64_t npp[2];
    npp[0] = panels;
    if (live_a) { const int64_t lp = (la + RGNN_STAT_PANEL_ROWS - 1) / RGNN_STAT_PANEL_ROWS; npp[0] = lp < panels ? lp : panels; }
    npp[1] = col_stats_b ? panels_b : 0;
    if (col_stats_b && live_b) { const int64_t lp = (lb + RGNN_STAT_PANEL_ROWS - 1) / RGNN_STAT_PANEL_ROWS; npp[1] = lp < panels_b ? lp : panels_b; }
    // (the pivot of a panel that counted nothing is whatever the uninitialised panel held: only a counted panel 0 lends its pivot --
    //  ADVICE r04; with neither, the sums are taken about 0: finite, merely less accurate, and only for row lists whose first
    //  128 rows are all padding)
    const float ca = *(col_stats + c0), cb = *((col_stats_b ? col_stats_b : col_stats) + c0);
    // ... and not that panel's pivot itself (the value of its first row) but its MEAN, pivot + s1 / count: every panel is moved onto
    // K in float32 below, which costs (K - column mean)^2 / variance of the accuracy -- a first row 20 spreads away from the rest
    // (a tiny frame of very different points at the head of the batch) made that 4e-5 (r05, fuzz seed 202 case 37).  The first
    // panel's mean is off only by its own outliers' share.  (Requested with the pivots: no further latency.)
    const float4 ea1 = *(const float4*)(col_stats + (int64_t)2 * n + c0);
    const float4 eb1 = *(const float4*)((col_stats_b ? col_stats_b : col_stats) + (int64_t)2 * n + c0);
    const bool use_a = npp[0] > 0 && ca > 0.f, use_b = !use_a && npp[1] > 0 && cb > 0.f;
    const float4 kk = use_a ? ka : (use_b ? kb : make_float4(0.f, 0.f, 0.f, 0.f));
    const float4 e1 = use_a ? ea1 : (use_b ? eb1 : make_float4(0.f, 0.f, 0.f, 0.f));
    const float inv = use_a ? 1.f / ca : (use_b ? 1.f / cb : 0.f);
    K[0] = kk.x + e1.x * inv; K[1] = kk.y + e1.y * inv; K[2] = kk.z + e1.z * inv; K[3] = kk.w + e1.w * inv;
    for (int part = 0; part < 2; part++) {
      const float* st = part ? col_stats_b : col_stats;
      const int64_t np = npp[part];
      // 4 panels (16 sixteen-byte loads) in flight per thread; the last round is predicated, loads past the end are CLAMPED to the
      // last panel and dropped; a panel that counts nothing adds exact zeros (branch-free)
      for (int64_t p = g; p < np; p += 4 * GR) {
        float4 a[4], b[4], e[4], f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int64_t pp = p + u * GR;
          const int64_t pc = pp < np ? pp : np - 1;
          a[u] = *(const float4*)(st + (pc * R + 0) * n + c0);
          b[u] = *(const float4*)(st + (pc * R + 1) * n + c0);
          if (ABL & 1) { e[u] = a[u]; f[u] = b[u]; continue; }      // (timing experiment: half the bytes)
          e[u] = *(const float4*)(st + (pc * R + 2) * n + c0);
          f[u] = *(const float4*)(st + (pc * R + 3) * n + c0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool okp = p + u * GR < np && a[u].x > 0.f;
          const float w = okp ? a[u].x : 0.f;
          const float bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w}, ev[4] = {e[u].x, e[u].y, e[u].z, e[u].w}, fv[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
          s0 += w;
          if (ABL & 2) { s1[0] += ev[0] + fv[1]; continue; }   // (timing experiment: no accumulation)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float t1 = okp ? ev[i] : 0.f, t2 = okp ? fv[i] : 0.f, d = okp ? bv[i] - K[i] : 0.f;   // (exact when the pivots are close)
            s1[i] += t1 + w * d;
            s2[i] += t2 + 2.f * d * t1 + w * d * d;
          }
        }
      }
    }
  }
  // lanes of a wave with the same quad (lane & 1), then the 16 waves through LDS in float64; fixed order: deterministic
#pragma unroll
  for (int o = 2; o < ((ABL & 4) ? 4 : 64); o <<= 1) {   // (4: timing experiment)
    s0 += __shfl_xor(s0, o, 64);
#pragma unroll
    for (int i = 0; i < 4; i++) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 2) {
    red[wave][lane][0] = (double)s0;
#pragma unroll
    for (int i = 0; i < 4; i++) { red[wave][lane][1 + i] = (double)s1[i]; red[wave][lane][5 + i] = (double)s2[i]; }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) kpiv[lane][i] = (double)K[i];
    }
  }
  __syncthreads();
  if (threadIdx.x >= 8) return;
  const int qq = threadIdx.x >> 2, ii = threadIdx.x & 3, c = blockIdx.x * 8 + threadIdx.x;
  if (c >= n) return;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int w = 0; w < 16; w++) { t0 += red[w][qq][0]; t1 += red[w][qq][1 + ii]; t2 += red[w][qq][5 + ii]; }
  bn_finish_channel(c, n, training, kpiv[qq][ii], t0, t1, t2, m, chp, running_mean, running_var, momentum, eps, scale_shift,
                    out_bound, out_bound ? fmaxf(fmaxf(in_b[0], in_b[1]), fmaxf(in_b[2], in_b[3])) : 0.f);
}

// Column statistics of an existing [m, n] matrix in the per-128-row-panel layout the dense layer's epilogue writes
// (for a BatchNorm whose input was not produced by rgnn_linear_fwd).
__global__ __launch_bounds__(256) void k_column_stats(const float* __restrict__ x, int64_t ldx, int64_t m, int n,
                                                     float* __restrict__ col_stats) {
  __shared__ float red[4][4][64];
  const int panel = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f, piv = 0.f, cn = 0.f;
  if (c < n) {
    const int64_t r0 = (int64_t)panel * 128 + g * 32;
    // pivot = the MEAN of the group's (up to) 32 rows, not its first row: sums about a pivot lose (distance of the pivot from the
    // mean / spread)^2 of their float32 accuracy, and a first row 20 spreads out -- a tiny frame of very different points at the head
    // of the batch -- cost 4e-5 (r05, tools/fuzz_hot_path.py seed 202 case 37; the rows are read twice, from L1 the second time)
    float sum = 0.f;
    for (int i = 0; i < 32; i++) {
      const int64_t r = r0 + i;
      if (r < m) { sum += x[r * ldx + c]; cn += 1.f; }
    }
    if (cn > 0.f) piv = sum / cn;
    for (int i = 0; i < 32; i++) {
      const int64_t r = r0 + i;
      if (r < m) {
        const float d = x[r * ldx + c] - piv;
        s1 += d;
        s2 += d * d;
      }
    }
  }
  const ColStat mine = stat_make(cn, piv, s1, s2);
  const int l = threadIdx.x & 63;
  red[0][g][l] = mine.n; red[1][g][l] = mine.piv; red[2][g][l] = mine.s1; red[3][g][l] = mine.s2;
  __syncthreads();
  if (g == 0 && c < n) {
    ColStat a = mine;
#pragma unroll
    for (int w = 1; w < 4; w++) {
      ColStat b;
      b.n = red[0][w][l]; b.piv = red[1][w][l]; b.s1 = red[2][w][l]; b.s2 = red[3][w][l];
      a = stat_merge(a, b);
    }
    stat_store(col_stats + ((int64_t)panel * RGNN_STAT_ROWS) * n + c, n, a);
  }
}

__global__ __launch_bounds__(256) void k_scale_shift_act(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ ss, int64_t m, int n, int relu,
                                                        float* __restrict__ y, int64_t ldy) {
  // one thread per (row, 4-channel group) when n % 4 == 0, else per element
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((n & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0) {
    const int g = n >> 2;
    if (idx >= m * g) return;
    const int64_t r = idx / g;
    const int c = (int)(idx - r * g) * 4;
    const float4 v = *(const float4*)(x + r * ldx + c);
    const float4 mu = *(const float4*)(ss + c);
    const float4 sc = *(const float4*)(ss + n + c);
    const float4 sh = *(const float4*)(ss + 2 * n + c);
    // (subtract, then fmaf: like the A-operand path of k_linear_dma that applies the same table inside the consumer layer -- the
    //  two ways of running a model give the same bits)
    float4 o = make_float4(fmaf(v.x - mu.x, sc.x, sh.x), fmaf(v.y - mu.y, sc.y, sh.y), fmaf(v.z - mu.z, sc.z, sh.z), fmaf(v.w - mu.w, sc.w, sh.w));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *(float4*)(y + r * ldy + c) = o;
  } else {
    if (idx >= m * n) return;
    const int64_t r = idx / n;
    const int c = (int)(idx - r * n);
    float o = fmaf(x[r * ldx + c] - ss[c], ss[n + c], ss[2 * n + c]);
    if (relu) o = fmaxf(o, 0.f);
    y[r * ldy + c] = o;
  }
}

// ---- BatchNorm with PER-FRAME statistics in a batched launch.  The reference runs inference with batch_size = 1 and never
// calls .eval() (evaluate.py:40, postprocessor/inference.py:57-62, gnn/gnn_models.py:124-128), so every frame is normalised
// with its OWN batch statistics; a batch of frames laid back to back reproduces that when the statistics are taken per
// segment [seg_ptr[f], seg_ptr[f + 1]) of rows.  Three launches: partial sums per (segment, 64-channel slab), a per-channel
// finish that also walks the running statistics through the segments IN ORDER (what a loop of single-frame forwards does to
// them), and the apply pass with a per-segment scale / shift table.
__global__ __launch_bounds__(256) void k_bn_seg_stats(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ seg_ptr,
                                                     int n, double* __restrict__ seg_sums /*[F][2][n]: mean, M2*/) {
  __shared__ double red[2][4][64];
  const int f = blockIdx.x;
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  const int64_t r0 = seg_ptr[f], r1 = seg_ptr[f + 1];
  double s1 = 0.0, s2 = 0.0;
  const double K = (c < n && r1 > r0) ? (double)x[r0 * ldx + c] : 0.0;    // pivot: the segment's first row (stats.h)
  if (c < n) {
    int64_t r = r0 + g;
    for (; r + 12 < r1; r += 16) {                       // four independent loads in flight per thread
      const double a = (double)x[r * ldx + c] - K, b = (double)x[(r + 4) * ldx + c] - K, d = (double)x[(r + 8) * ldx + c] - K,
                   e = (double)x[(r + 12) * ldx + c] - K;
      s1 += a + b + d + e;
      s2 += a * a + b * b + d * d + e * e;
    }
    for (; r < r1; r += 4) { const double a = (double)x[r * ldx + c] - K; s1 += a; s2 += a * a; }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    const double t1 = red[0][0][lc] + red[0][1][lc] + red[0][2][lc] + red[0][3][lc];
    const double t2 = red[1][0][lc] + red[1][1][lc] + red[1][2][lc] + red[1][3][lc];
    seg_moments_store(seg_sums, f, n, c, r1 - r0, K, t1, t2);
  }
}

// Statistics AND apply of one (segment, 64-channel slab) in one block (r03): the statistics of a channel need nothing from other
// slabs, so the block that summed a slab can normalise it -- pass 1 reads the slab (float64 sums, 16 row groups), pass 2 reads it
// again (3 000 rows x 64 channels = 768 KB: L2 hits) and writes act(x scale + shift).  Replaces k_bn_seg_stats +
// k_scale_shift_act_seg (read, read, write -> read, write); k_bn_seg_finalize still walks the running statistics and the bound
// from the sums this kernel leaves behind.  scale / shift are formed exactly as k_bn_seg_finalize forms them (same float64
// expressions, rounded to float once), so the fused and the split path give the same bits for equal sums.
__global__ __launch_bounds__(1024) void k_bn_seg_fused(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ seg_ptr,
                                                      int n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int relu, double* __restrict__ seg_sums, float* __restrict__ y,
                                                      int64_t ldy) {
  __shared__ double red[2][16][64];
  __shared__ float ss[3][64];
  const int f = blockIdx.x;
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  const int64_t r0 = seg_ptr[f], r1 = seg_ptr[f + 1];
  double s1 = 0.0, s2 = 0.0;
  const double K = (c < n && r1 > r0) ? (double)x[r0 * ldx + c] : 0.0;    // pivot: the segment's first row (stats.h)
  if (c < n) {
    int64_t r = r0 + g;
    for (; r + 48 < r1; r += 64) {                       // four independent loads in flight per thread
      const double a = (double)x[r * ldx + c] - K, b = (double)x[(r + 16) * ldx + c] - K, d = (double)x[(r + 32) * ldx + c] - K,
                   e = (double)x[(r + 48) * ldx + c] - K;
      s1 += a + b + d + e;
      s2 += a * a + b * b + d * d + e * e;
    }
    for (; r < r1; r += 16) { const double a = (double)x[r * ldx + c] - K; s1 += a; s2 += a * a; }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) { t1 += red[0][i][lc]; t2 += red[1][i][lc]; }
    const int64_t m = r1 - r0;
    seg_moments_store(seg_sums, f, n, c, m, K, t1, t2);
    float mh = 0.f, gg = 0.f, tt = 0.f;
    if (m > 0) {
      const double mean = seg_sums[((int64_t)f * 2 + 0) * n + c];
      const double var = seg_sums[((int64_t)f * 2 + 1) * n + c] / (double)m;
      const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
      bn_table_entry(mean, var, gm, bt, (double)eps, mh, gg, tt);
    }
    ss[0][lc] = mh; ss[1][lc] = gg; ss[2][lc] = tt;
  }
  __syncthreads();
  if (c >= n) return;
  const float mu = ss[0][lc], sc = ss[1][lc], sh = ss[2][lc];
  for (int64_t r = r0 + g; r < r1; r += 16) {
    float o = fmaf(x[r * ldx + c] - mu, sc, sh);
    if (relu) o = fmaxf(o, 0.f);
    y[r * ldy + c] = o;
  }
}

// (r03: one block = 64 channels x 16 frame groups.  The scale / shift of a (frame, channel) needs nothing from other frames, so
//  that part -- two float64 divisions and a square root each -- runs in parallel over the frames; only the running statistics
//  are a recurrence over the frames, and walking it costs two fused multiply-adds per frame once the means and unbiased
//  variances lie ready in the scratch the sums came in.  Same expressions, same order, same bits as the sequential form, which
//  took 44 us per call on 64 frames x 224 channels.)
__global__ __launch_bounds__(1024) void k_bn_seg_finalize(double* __restrict__ seg_sums, const int64_t* __restrict__ seg_ptr,
                                                         int64_t n_seg, int n, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked,
                                                         float momentum, float eps, float* __restrict__ table /*[F][RGNN_AFFINE_ROWS][n]*/,
                                                         const float* __restrict__ in_bound, float* __restrict__ out_bound) {
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  __shared__ float in_b;
  __shared__ float bmax[16][64];
  __shared__ int live_s[16];
  if (out_bound != nullptr && threadIdx.x < 64) {       // maximum over the slots of the input's bound (rgnn.h)
    float v = fmaxf(fmaxf(in_bound[threadIdx.x], in_bound[threadIdx.x + 64]), fmaxf(in_bound[threadIdx.x + 128], in_bound[threadIdx.x + 192]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (threadIdx.x == 0) in_b = v;
  }
  __syncthreads();
  const bool okc = c < n;
  const double gm = (okc && gamma) ? (double)gamma[c] : 1.0, bt = (okc && beta) ? (double)beta[c] : 0.0;
  double bound = 0.0;
  int live = 0;
  for (int64_t f = g; f < n_seg; f += 16) {
    const int64_t m = seg_ptr[f + 1] - seg_ptr[f];
    live += m > 0 ? 1 : 0;
    if (!okc) continue;
    double mean = 0.0, unbiased = 0.0, var_b = 0.0;
    float mh = 0.f, gg = 0.f, tt = 0.f;
    if (m > 0) {
      mean = seg_sums[(f * 2 + 0) * n + c];
      const double var = seg_sums[(f * 2 + 1) * n + c] / (double)m;              // biased, as F.batch_norm normalises with
      var_b = var;
      bn_table_entry(mean, var, gm, bt, (double)eps, mh, gg, tt);
      unbiased = (m > 1) ? var * (double)m / (double)(m - 1) : var;
    }
    seg_sums[(f * 2 + 0) * n + c] = mean;                 // (the scratch now holds what the recurrence below reads)
    seg_sums[(f * 2 + 1) * n + c] = unbiased;
    table[(f * RGNN_AFFINE_ROWS + 0) * n + c] = mh;
    table[(f * RGNN_AFFINE_ROWS + 1) * n + c] = gg;
    table[(f * RGNN_AFFINE_ROWS + 2) * n + c] = tt;
    if (out_bound != nullptr) {                           // (a frame's table normalises that frame's rows: its own statistics)
      const double b = bn_bound((double)gg, (double)mh, (double)tt, (double)in_b, m > 0, (double)m, var_b);
      bound = b > bound ? b : bound;
    }
  }
  bmax[g][lc] = (float)bound;
  if (lc == 0) live_s[g] = live;
  __syncthreads();
  if (g != 0) return;
  if (blockIdx.x == 0 && lc == 0 && num_batches_tracked) {
    int tot = 0;
    for (int i = 0; i < 16; i++) tot += live_s[i];
    *num_batches_tracked += tot;
  }
  if (!okc) return;
  if (out_bound != nullptr) {
    float b = bmax[0][lc];
#pragma unroll
    for (int i = 1; i < 16; i++) b = fmaxf(b, bmax[i][lc]);
    atomicMax((unsigned int*)out_bound + (c & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(b));
  }
  if (running_mean) {                                    // one single-frame forward after the other (float storage each time)
    double rm = (double)running_mean[c], rv = (double)running_var[c];
    for (int64_t f = 0; f < n_seg; f++) {
      if (seg_ptr[f + 1] - seg_ptr[f] <= 0) continue;
      rm = (double)(float)((1.0 - (double)momentum) * rm + (double)momentum * seg_sums[(f * 2 + 0) * n + c]);
      rv = (double)(float)((1.0 - (double)momentum) * rv + (double)momentum * seg_sums[(f * 2 + 1) * n + c]);
    }
    running_mean[c] = (float)rm; running_var[c] = (float)rv;
  }
}

// y[r] = act((x[r] - mean[seg(r)]) g[seg(r)] + t[seg(r)]): one block per 64 rows x all channels; rows are sorted by segment, so a
// block finds the segment of its first row by binary search and steps forward from there.
__global__ __launch_bounds__(256) void k_scale_shift_act_seg(const float* __restrict__ x, int64_t ldx, const float* __restrict__ table,
                                                            const int64_t* __restrict__ seg_ptr, int64_t n_seg, int64_t m, int n,
                                                            int relu, float* __restrict__ y, int64_t ldy) {
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  int64_t lo = 0, hi = n_seg;                            // last segment whose start is <= row0
  while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (seg_ptr[mid] <= row0) lo = mid; else hi = mid; }
  const int g4 = (n + 3) >> 2;
  const bool vec = (n & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0;
  for (int64_t i = threadIdx.x; i < (int64_t)64 * g4; i += blockDim.x) {
    const int64_t r = row0 + i / g4;
    if (r >= m) break;
    const int c = (int)(i % g4) * 4;
    int64_t f = lo;
    while (f + 1 < n_seg && seg_ptr[f + 1] <= r) f++;    // (at most the few segments a 64-row block touches)
    const float* mu = table + (f * RGNN_AFFINE_ROWS + 0) * n, *sc = mu + n, *sh = sc + n;
    if (vec) {
      const float4 v = *(const float4*)(x + r * ldx + c), u = *(const float4*)(mu + c), a = *(const float4*)(sc + c), b = *(const float4*)(sh + c);
      float4 o = make_float4(fmaf(v.x - u.x, a.x, b.x), fmaf(v.y - u.y, a.y, b.y), fmaf(v.z - u.z, a.z, b.z), fmaf(v.w - u.w, a.w, b.w));
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *(float4*)(y + r * ldy + c) = o;
    } else {
      for (int j = 0; j < 4 && c + j < n; j++) {
        float o = fmaf(x[r * ldx + c + j] - mu[c + j], sc[c + j], sh[c + j]);
        if (relu) o = fmaxf(o, 0.f);
        y[r * ldy + c + j] = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ x, int64_t ldx, int64_t m, int n,
                                                     float* __restrict__ y, int64_t ldy) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  const float* xr = x + r * ldx;
  float mx = -INFINITY;
  for (int c = 0; c < n; c++) mx = fmaxf(mx, xr[c]);
  float s = 0.f;
  for (int c = 0; c < n; c++) s += expf(xr[c] - mx);
  for (int c = 0; c < n; c++) y[r * ldy + c] = expf(xr[c] - mx) / s;
}

// Backward coefficients of BatchNorm1d (+ fused ReLU) for rgnn_bn_bwd_apply, one launch instead of a dozen [C]-sized
// float64 tensor operations: from the forward column statistics (or the running statistics in eval mode) and the backward
// partial sums (sum g, sum g h per panel; g = relu'(y) dy):
//     dx = A g + B h + C,   A = gamma rstd,   B = -gamma rstd^2 S / m,   C = -gamma rstd sum(g) / m + gamma rstd^2 mean S / m,
//     S = sum g xhat = (sum g h - mean sum g) rstd;   d gamma = S,   d beta = sum g      (eval mode: B = C = 0)
__global__ __launch_bounds__(1024) void k_bn_bwd_coef(const float* __restrict__ fwd_stats, int64_t panels_f,
                                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                     const float* __restrict__ bwd_part, int64_t panels_b, int64_t m, int n,
                                                     const float* __restrict__ gamma, float eps, int use_batch,
                                                     float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  constexpr int CH = 16, GR = 64;
  __shared__ double red[5][GR][CH];
  const int lc = threadIdx.x & (CH - 1), g = threadIdx.x / CH;
  const int c = blockIdx.x * CH + lc;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, b1 = 0.0, b2 = 0.0, K = 0.0;
  if (c < n) {
    if (use_batch) {                                     // forward statistics: panels -> sums about K
      K = (double)fwd_stats[(int64_t)1 * n + c];
      for (int64_t p = g; p < panels_f; p += GR) {
        const ColStat a = stat_load(fwd_stats + (p * RGNN_STAT_ROWS) * n + c, n);
        if (a.n > 0.f) stat_accumulate((double)a.n, (double)a.piv, (double)a.s1, (double)a.s2, K, s0, s1, s2);
      }
    }
    for (int64_t p = g; p < panels_b; p += GR) {
      b1 += (double)bwd_part[(p * 2 + 0) * n + c];
      b2 += (double)bwd_part[(p * 2 + 1) * n + c];
    }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2; red[2][g][lc] = b1; red[3][g][lc] = b2; red[4][g][lc] = s0;
  __syncthreads();
  if (g != 0 || c >= n) return;
  s0 = s1 = s2 = b1 = b2 = 0.0;
#pragma unroll
  for (int i = 0; i < GR; i++) { s1 += red[0][i][lc]; s2 += red[1][i][lc]; b1 += red[2][i][lc]; b2 += red[3][i][lc]; s0 += red[4][i][lc]; }
  double mean, var;
  if (use_batch) {
    const double rows = s0 > 0.0 ? s0 : (double)m;
    const double dk = s1 / rows;
    mean = K + dk;
    var = s2 / rows - dk * dk;
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const double sg = b1, sxhat = (b2 - mean * b1) * rstd;
  double A = gm * rstd, B = 0.0, C = 0.0;
  if (use_batch) {
    B = -gm * rstd * rstd * sxhat / (double)m;
    C = -gm * rstd * sg / (double)m + gm * rstd * rstd * mean * sxhat / (double)m;
  }
  coef[c] = (float)A; coef[n + c] = (float)B; coef[2 * n + c] = (float)C;
  if (dgamma) dgamma[c] = (float)sxhat;
  if (dbeta) dbeta[c] = (float)sg;
}

// ... four channels per work-group with 16-byte loads, 256 panel groups: 116 work-groups on a 464-column layer instead of 29 whose
// lanes read 4 bytes of every 1 856-byte statistics row (63 -> see MEASUREMENTS.md section 8)
__global__ __launch_bounds__(256) void k_bn_bwd_coef4(const float* __restrict__ fwd_stats, int64_t panels_f,
                                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                     const float* __restrict__ bwd_part, int64_t panels_b, int64_t m, int n,
                                                     const float* __restrict__ gamma, float eps, int use_batch,
                                                     float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double red[4][20];
  const int c0 = blockIdx.x * 4, t = threadIdx.x;
  double acc[20];                                      // per channel j: [5 j + 0..4] = rows, sum (v - K), sum (v - K)^2, sum g, sum g h
#pragma unroll
  for (int i = 0; i < 20; i++) acc[i] = 0.0;
  float4 Kf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (use_batch) {
    Kf = *(const float4*)(fwd_stats + (int64_t)1 * n + c0);
    const double K[4] = {(double)Kf.x, (double)Kf.y, (double)Kf.z, (double)Kf.w};
    for (int64_t p = t; p < panels_f; p += 256) {
      const float* base = fwd_stats + (p * RGNN_STAT_ROWS) * n + c0;
      const float4 cn = *(const float4*)base, pv = *(const float4*)(base + n), a1 = *(const float4*)(base + 2 * (int64_t)n),
                   a2 = *(const float4*)(base + 3 * (int64_t)n);
      if (cn.x > 0.f) stat_accumulate((double)cn.x, (double)pv.x, (double)a1.x, (double)a2.x, K[0], acc[0], acc[1], acc[2]);
      if (cn.y > 0.f) stat_accumulate((double)cn.y, (double)pv.y, (double)a1.y, (double)a2.y, K[1], acc[5], acc[6], acc[7]);
      if (cn.z > 0.f) stat_accumulate((double)cn.z, (double)pv.z, (double)a1.z, (double)a2.z, K[2], acc[10], acc[11], acc[12]);
      if (cn.w > 0.f) stat_accumulate((double)cn.w, (double)pv.w, (double)a1.w, (double)a2.w, K[3], acc[15], acc[16], acc[17]);
    }
  }
  for (int64_t p = t; p < panels_b; p += 256) {
    const float4 g1 = *(const float4*)(bwd_part + (p * 2 + 0) * n + c0), g2 = *(const float4*)(bwd_part + (p * 2 + 1) * n + c0);
    acc[3] += (double)g1.x; acc[4] += (double)g2.x; acc[8] += (double)g1.y; acc[9] += (double)g2.y;
    acc[13] += (double)g1.z; acc[14] += (double)g2.z; acc[18] += (double)g1.w; acc[19] += (double)g2.w;
  }
#pragma unroll
  for (int i = 0; i < 20; i++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o, 64);
  }
  if ((t & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 20; i++) red[t >> 6][i] = acc[i];
  }
  __syncthreads();
  if (t >= 4) return;
  const int c = c0 + t;
  const double s0 = red[0][5 * t] + red[1][5 * t] + red[2][5 * t] + red[3][5 * t];
  const double s1 = red[0][5 * t + 1] + red[1][5 * t + 1] + red[2][5 * t + 1] + red[3][5 * t + 1];
  const double s2 = red[0][5 * t + 2] + red[1][5 * t + 2] + red[2][5 * t + 2] + red[3][5 * t + 2];
  const double b1 = red[0][5 * t + 3] + red[1][5 * t + 3] + red[2][5 * t + 3] + red[3][5 * t + 3];
  const double b2 = red[0][5 * t + 4] + red[1][5 * t + 4] + red[2][5 * t + 4] + red[3][5 * t + 4];
  double mean, var;
  if (use_batch) {
    const double K = (double)(t == 0 ? Kf.x : t == 1 ? Kf.y : t == 2 ? Kf.z : Kf.w);
    const double rows = s0 > 0.0 ? s0 : (double)m;
    const double dk = s1 / rows;
    mean = K + dk;
    var = s2 / rows - dk * dk;
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const double sg = b1, sxhat = (b2 - mean * b1) * rstd;
  double A = gm * rstd, B = 0.0, C = 0.0;
  if (use_batch) {
    B = -gm * rstd * rstd * sxhat / (double)m;
    C = -gm * rstd * sg / (double)m + gm * rstd * rstd * mean * sxhat / (double)m;
  }
  coef[c] = (float)A; coef[n + c] = (float)B; coef[2 * n + c] = (float)C;
  if (dgamma) dgamma[c] = (float)sxhat;
  if (dbeta) dbeta[c] = (float)sg;
}

}  // namespace

extern "C" int rgnn_bn_bwd_coef(const float* fwd_stats, int64_t panels_f, const float* running_mean, const float* running_var,
                                const float* bwd_part, int64_t panels_b, int64_t m, int32_t n, const float* gamma, float eps,
                                int32_t use_batch, float* coef, float* dgamma, float* dbeta, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && m >= 1 && coef && bwd_part && panels_b >= 1, "bad arguments");
  RGNN_CHECK_ARG(!use_batch || (fwd_stats && panels_f >= 1), "batch statistics need the forward column sums");
  RGNN_CHECK_ARG(use_batch || (running_mean && running_var), "eval mode needs running statistics");
  if (n % 4 == 0 && (((uintptr_t)bwd_part | (uintptr_t)(fwd_stats ? fwd_stats : bwd_part)) & 15) == 0)
    hipLaunchKernelGGL(k_bn_bwd_coef4, dim3(n / 4), dim3(256), 0, (hipStream_t)stream, fwd_stats, panels_f, running_mean,
                       running_var, bwd_part, panels_b, m, n, gamma, eps, use_batch, coef, dgamma, dbeta);
  else
    hipLaunchKernelGGL(k_bn_bwd_coef, dim3(rgnn_blocks(n, 16)), dim3(1024), 0, (hipStream_t)stream, fwd_stats, panels_f, running_mean,
                       running_var, bwd_part, panels_b, m, n, gamma, eps, use_batch, coef, dgamma, dbeta);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_batchnorm_finalize(const float* col_stats, int64_t panels, int64_t m, int32_t n, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, int32_t training, float momentum, float eps,
                                       float* scale_shift, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && scale_shift, "bad arguments");
  RGNN_CHECK_ARG(!training || (col_stats && m >= 1 && panels >= 1), "training mode needs column statistics");
  RGNN_CHECK_ARG(training || (running_mean && running_var), "eval mode needs running statistics");
  return rgnn_batchnorm_finalize_bound(col_stats, panels, nullptr, nullptr, 0, nullptr, m, n, gamma, beta, running_mean, running_var,
                                       num_batches_tracked, training, momentum, eps, scale_shift, nullptr, nullptr, stream);
}

extern "C" int rgnn_batchnorm_finalize_parts(const float* stats_a, int64_t panels_a, const int64_t* rows_a,
                                             const float* stats_b, int64_t panels_b, const int64_t* rows_b, int64_t m,
                                             int32_t n, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, int64_t* num_batches_tracked, int32_t training,
                                             float momentum, float eps, float* scale_shift, rgnn_stream_t stream) {
  return rgnn_batchnorm_finalize_bound(stats_a, panels_a, rows_a, stats_b, panels_b, rows_b, m, n, gamma, beta, running_mean,
                                       running_var, num_batches_tracked, training, momentum, eps, scale_shift, nullptr, nullptr,
                                       stream);
}

extern "C" int rgnn_batchnorm_finalize_bound(const float* stats_a, int64_t panels_a, const int64_t* rows_a,
                                             const float* stats_b, int64_t panels_b, const int64_t* rows_b, int64_t m,
                                             int32_t n, const float* gamma, const float* beta, float* running_mean,
                                             float* running_var, int64_t* num_batches_tracked, int32_t training,
                                             float momentum, float eps, float* scale_shift, const float* in_bound,
                                             float* out_bound, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && scale_shift, "bad arguments");
  RGNN_CHECK_ARG(out_bound == nullptr || in_bound != nullptr, "out_bound needs in_bound");
  RGNN_CHECK_ARG(!training || (stats_a && m >= 1 && panels_a >= 1 && (stats_b == nullptr || panels_b >= 1)),
                 "training mode needs column statistics");
  RGNN_CHECK_ARG(training || (running_mean && running_var), "eval mode needs running statistics");
  const char* abl_e = RGNN_ENV("RGNN_BN_FIN_ABL");          // (experiments only: wrong results)
  const int abl = abl_e ? atoi(abl_e) : 0;
#define RGNN_FIN4(A) hipLaunchKernelGGL(k_bn_finalize4<A>, dim3(rgnn_blocks(n, 8)), dim3(1024), 0, (hipStream_t)stream, stats_a, panels_a, rows_a, \
                       stats_b, panels_b, rows_b, m, n, gamma, beta, running_mean, running_var, num_batches_tracked, training,   \
                       momentum, eps, scale_shift, in_bound, out_bound)
  if ((n & 3) == 0 && training && (((uintptr_t)stats_a | (uintptr_t)stats_b) & 15) == 0 && abl >= 0) {
    switch (abl) { case 1: RGNN_FIN4(1); break; case 2: RGNN_FIN4(2); break; case 4: RGNN_FIN4(4); break; case 7: RGNN_FIN4(7); break; default: RGNN_FIN4(0); }
  } else
    hipLaunchKernelGGL(k_bn_finalize, dim3(rgnn_blocks(n, 16)), dim3(1024), 0, (hipStream_t)stream, stats_a, panels_a, rows_a,
                       stats_b, panels_b, rows_b, m, n, gamma, beta, running_mean, running_var, num_batches_tracked, training,
                       momentum, eps, scale_shift, in_bound, out_bound);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_batchnorm_segments(const float* x, int64_t ldx, const int64_t* seg_ptr, int64_t n_seg, int32_t n,
                                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float momentum, float eps, double* seg_sums, float* table,
                                       const float* in_bound, float* out_bound, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && n_seg >= 0, "bad sizes");
  if (n_seg == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && seg_ptr && seg_sums && table, "null pointers");
  RGNN_CHECK_ARG(out_bound == nullptr || in_bound != nullptr, "out_bound needs in_bound");
  RGNN_CHECK_ARG(n_seg < 65536 * 32, "too many segments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_seg_stats, dim3((unsigned)n_seg, (unsigned)((n + 63) / 64)), dim3(256), 0, s, x, ldx, seg_ptr, n, seg_sums);
  hipLaunchKernelGGL(k_bn_seg_finalize, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, seg_sums, seg_ptr, n_seg, n,
                     gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, table, in_bound, out_bound);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_batchnorm_act_segments(const float* x, int64_t ldx, const int64_t* seg_ptr, int64_t n_seg, int32_t n,
                                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, float momentum, float eps, int32_t relu,
                                           double* seg_sums, float* table, float* y, int64_t ldy, const float* in_bound,
                                           float* out_bound, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && n_seg >= 0, "bad sizes");
  if (n_seg == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && seg_ptr && seg_sums && table && y, "null pointers");
  RGNN_CHECK_ARG(out_bound == nullptr || in_bound != nullptr, "out_bound needs in_bound");
  RGNN_CHECK_ARG(n_seg < 65536 * 32, "too many segments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_seg_fused, dim3((unsigned)n_seg, (unsigned)((n + 63) / 64)), dim3(1024), 0, s, x, ldx, seg_ptr, n, gamma, beta,
                     eps, relu, seg_sums, y, ldy);
  hipLaunchKernelGGL(k_bn_seg_finalize, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, seg_sums, seg_ptr, n_seg, n,
                     gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, table, in_bound, out_bound);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

namespace {
// Mean and M2 per segment from the statistics the dense launches left per 128-row panel of their (segment-padded) row lists:
// segment f owns panels [start_a[f], start_a[f + 1]) of list a and [start_b[f], start_b[f + 1]) of list b.  float64, fixed order.
__global__ __launch_bounds__(256) void k_bn_seg_from_panels(const float* __restrict__ stats_a, const int32_t* __restrict__ start_a,
                                                           const float* __restrict__ stats_b, const int32_t* __restrict__ start_b,
                                                           int n, double* __restrict__ seg_sums) {
  __shared__ double red[2][4][64];
  const int f = blockIdx.x, lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  __shared__ double red0[4][64];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, K = 0.0;
  if (c < n) {
    // pivot: the pivot of the segment's first panel (list a if it has one, else list b)
    if (start_a[f + 1] > start_a[f]) K = (double)stats_a[((int64_t)start_a[f] * RGNN_STAT_ROWS + 1) * n + c];
    else if (stats_b != nullptr && start_b[f + 1] > start_b[f]) K = (double)stats_b[((int64_t)start_b[f] * RGNN_STAT_ROWS + 1) * n + c];
    for (int part = 0; part < 2; part++) {
      const float* st = part ? stats_b : stats_a;
      const int32_t* sp = part ? start_b : start_a;
      if (st == nullptr) continue;
      for (int p = sp[f] + g; p < sp[f + 1]; p += 4) {
        const ColStat a = stat_load(st + ((int64_t)p * RGNN_STAT_ROWS) * n + c, n);
        if (a.n > 0.f) stat_accumulate((double)a.n, (double)a.piv, (double)a.s1, (double)a.s2, K, s0, s1, s2);
      }
    }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2; red0[g][lc] = s0;
  __syncthreads();
  if (g == 0 && c < n) {
    const double t0 = red0[0][lc] + red0[1][lc] + red0[2][lc] + red0[3][lc];
    const double t1 = red[0][0][lc] + red[0][1][lc] + red[0][2][lc] + red[0][3][lc];
    const double t2 = red[1][0][lc] + red[1][1][lc] + red[1][2][lc] + red[1][3][lc];
    seg_moments_store(seg_sums, f, n, c, (int64_t)t0, K, t1, t2);
  }
}
}  // namespace

extern "C" int rgnn_batchnorm_segments_from_panels(const float* stats_a, const int32_t* panel_start_a, const float* stats_b,
                                                   const int32_t* panel_start_b, const int64_t* seg_ptr, int64_t n_seg, int32_t n,
                                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                                   int64_t* num_batches_tracked, float momentum, float eps, double* seg_sums,
                                                   float* table, const float* in_bound, float* out_bound, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 1 && n_seg >= 0, "bad sizes");
  if (n_seg == 0) return RGNN_OK;
  RGNN_CHECK_ARG(stats_a && panel_start_a && seg_ptr && seg_sums && table, "null pointers");
  RGNN_CHECK_ARG((stats_b == nullptr) == (panel_start_b == nullptr), "stats_b and panel_start_b go together");
  RGNN_CHECK_ARG(out_bound == nullptr || in_bound != nullptr, "out_bound needs in_bound");
  RGNN_CHECK_ARG(n_seg < 65536 * 32, "too many segments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bn_seg_from_panels, dim3((unsigned)n_seg, (unsigned)((n + 63) / 64)), dim3(256), 0, s, stats_a, panel_start_a,
                     stats_b, panel_start_b, n, seg_sums);
  hipLaunchKernelGGL(k_bn_seg_finalize, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, seg_sums, seg_ptr, n_seg, n,
                     gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, table, in_bound, out_bound);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_scale_shift_act_segments(const float* x, int64_t ldx, const float* table, const int64_t* seg_ptr,
                                             int64_t n_seg, int64_t m, int32_t n, int32_t relu, float* y, int64_t ldy,
                                             rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && table && seg_ptr && y && n_seg >= 1, "null pointers");
  hipLaunchKernelGGL(k_scale_shift_act_seg, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, table,
                     seg_ptr, n_seg, m, n, relu, y, ldy);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_scale_shift_act(const float* x, int64_t ldx, const float* scale_shift, int64_t m, int32_t n,
                                    int32_t relu, float* y, int64_t ldy, rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && scale_shift && y, "null pointers");
  const int64_t work = ((n & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0) ? m * (n >> 2) : m * n;
  hipLaunchKernelGGL(k_scale_shift_act, dim3(rgnn_blocks(work, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     scale_shift, m, n, relu, y, ldy);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_softmax_rows(const float* x, int64_t ldx, int64_t m, int32_t n, float* y, int64_t ldy,
                                 rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && y, "null pointers");
  hipLaunchKernelGGL(k_softmax_rows, dim3(rgnn_blocks(m, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, m, n, y, ldy);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_column_stats(const float* x, int64_t ldx, int64_t m, int32_t n, float* col_stats,
                                 rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(x && col_stats, "null pointers");
  const unsigned panels = (unsigned)((m + 127) / 128);
  hipLaunchKernelGGL(k_column_stats, dim3(panels, (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, m,
                     n, col_stats);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
