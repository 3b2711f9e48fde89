// Train-mode BatchNorm1d (gnn/gnn_models.py:71-73,126-128 via torch_geometric.nn.BatchNorm), the activation
// that follows it, and the row softmax of postprocessor/inference.py:46,62.
//
// The column sums / sums of squares arrive as fp32 partials per 128-row panel from the epilogue of the dense
// layer (linear.hip); they are combined here in float64, one thread per channel -- deterministic, no atomics.
#include "common.h"
#include <math.h>

namespace {

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
  constexpr int CH = 16, GR = 64;
  __shared__ double red[2][GR][CH];
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
  double s1 = 0.0, s2 = 0.0;
  if (training && c < n) {
    for (int part = 0; part < 2; part++) {
      const float* st = part ? col_stats_b : col_stats;
      if (st == nullptr) continue;
      int64_t np = part ? panels_b : panels;
      const int64_t* live = part ? live_b : live_a;
      if (live) { const int64_t lp = (*live + RGNN_STAT_PANEL_ROWS - 1) / RGNN_STAT_PANEL_ROWS; np = lp < np ? lp : np; }
      // 8 independent loads in flight per thread (the loop is latency bound otherwise: 1500 panels / 64 groups).  The last
      // round is predicated, not a loop of its own: a tail of up to seven panels, one load latency each, was half of the
      // kernel's 13 us (r03).  Panels beyond the end add exact zeros: same sums.
      // (loads past the end are CLAMPED to the last panel and their values dropped afterwards: a conditional load would be a
      //  branch per load and serialise the round)
      for (int64_t p = g; p < np; p += 8 * GR) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int64_t pp = p + u * GR;
          const int64_t pc = pp < np ? pp : np - 1;
          a[u] = st[(pc * 2 + 0) * n + c];
          b[u] = st[(pc * 2 + 1) * n + c];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool okp = p + u * GR < np;
          s1 += okp ? (double)a[u] : 0.0; s2 += okp ? (double)b[u] : 0.0;
        }
      }
    }
  }
  red[0][g][lc] = s1;
  red[1][g][lc] = s2;
  __syncthreads();
  if (g != 0 || c >= n) return;
  double mean, var;
  if (training) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int i = 0; i < GR; i++) { s1 += red[0][i][lc]; s2 += red[1][i][lc]; }
    mean = s1 / (double)m;
    var = s2 / (double)m - mean * mean;  // biased variance, as F.batch_norm normalises with
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unbiased = (m > 1) ? var * (double)m / (double)(m - 1) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
  } else {
    mean = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  const double sc = gm / sqrt(var + (double)eps);
  scale_shift[c] = (float)sc;
  scale_shift[n + c] = (float)(bt - mean * sc);
  if (out_bound != nullptr) {
    // upper bound of |x scale + shift| over the column, for the f16x2 dense form that applies this table to its A1 operand:
    // |scale| B + |shift| with B the bound of the input, from the ROUNDED table entries and slightly widened (the consumer
    // evaluates fma(x, scale, shift) in fp32).  Loose by design -- a near-constant column has scale = gamma / sqrt(eps) -- and
    // harmless: the form keeps full accuracy up to 2^19 between bound and typical magnitude.  (The tighter batch-statistics
    // bound |gamma| sqrt(m - 1) + |beta| is NOT used: it assumes exact statistics, and a bound must hold for the table as it is.)
    const double fs = fabs((double)(float)sc), fh = fabs((double)(float)(bt - mean * sc));
    const double b = fs * (double)fmaxf(fmaxf(in_b[0], in_b[1]), fmaxf(in_b[2], in_b[3])) + fh;
    atomicMax((unsigned int*)out_bound + (c & (RGNN_BOUND_SLOTS - 1)), __float_as_uint((float)(b * 1.0001)));
  }
}

// Column sums / sums of squares of an existing [m, n] matrix in the same per-128-row-panel layout the dense
// layer's epilogue writes (for a BatchNorm whose input was not produced by rgnn_linear_fwd).
__global__ __launch_bounds__(256) void k_column_stats(const float* __restrict__ x, int64_t ldx, int64_t m, int n,
                                                     float* __restrict__ col_stats) {
  __shared__ float red[2][4][64];
  const int panel = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < n) {
    const int64_t r0 = (int64_t)panel * 128 + g * 32;
    for (int i = 0; i < 32; i++) {
      const int64_t r = r0 + i;
      if (r < m) {
        const float v = x[r * ldx + c];
        s1 += v;
        s2 += v * v;
      }
    }
  }
  red[0][g][threadIdx.x & 63] = s1;
  red[1][g][threadIdx.x & 63] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    const int l = threadIdx.x;
    col_stats[((int64_t)panel * 2 + 0) * n + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    col_stats[((int64_t)panel * 2 + 1) * n + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
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
    const float4 sc = *(const float4*)(ss + c);
    const float4 sh = *(const float4*)(ss + n + c);
    // (fmaf, like the A-operand path of k_linear_dma that applies the same scale / shift inside the consumer layer: the two
    //  ways of running a model give the same bits)
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *(float4*)(y + r * ldy + c) = o;
  } else {
    if (idx >= m * n) return;
    const int64_t r = idx / n;
    const int c = (int)(idx - r * n);
    float o = fmaf(x[r * ldx + c], ss[c], ss[n + c]);
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
                                                     int n, double* __restrict__ seg_sums /*[F][2][n]*/) {
  __shared__ double red[2][4][64];
  const int f = blockIdx.x;
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  const int64_t r0 = seg_ptr[f], r1 = seg_ptr[f + 1];
  double s1 = 0.0, s2 = 0.0;
  if (c < n) {
    int64_t r = r0 + g;
    for (; r + 12 < r1; r += 16) {                       // four independent loads in flight per thread
      const float a = x[r * ldx + c], b = x[(r + 4) * ldx + c], d = x[(r + 8) * ldx + c], e = x[(r + 12) * ldx + c];
      s1 += (double)a + (double)b + (double)d + (double)e;
      s2 += (double)a * a + (double)b * b + (double)d * d + (double)e * e;
    }
    for (; r < r1; r += 4) { const float a = x[r * ldx + c]; s1 += (double)a; s2 += (double)a * a; }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    seg_sums[((int64_t)f * 2 + 0) * n + c] = red[0][0][lc] + red[0][1][lc] + red[0][2][lc] + red[0][3][lc];
    seg_sums[((int64_t)f * 2 + 1) * n + c] = red[1][0][lc] + red[1][1][lc] + red[1][2][lc] + red[1][3][lc];
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
  __shared__ float ss[2][64];
  const int f = blockIdx.x;
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  const int64_t r0 = seg_ptr[f], r1 = seg_ptr[f + 1];
  double s1 = 0.0, s2 = 0.0;
  if (c < n) {
    int64_t r = r0 + g;
    for (; r + 48 < r1; r += 64) {                       // four independent loads in flight per thread
      const float a = x[r * ldx + c], b = x[(r + 16) * ldx + c], d = x[(r + 32) * ldx + c], e = x[(r + 48) * ldx + c];
      s1 += (double)a + (double)b + (double)d + (double)e;
      s2 += (double)a * a + (double)b * b + (double)d * d + (double)e * e;
    }
    for (; r < r1; r += 16) { const float a = x[r * ldx + c]; s1 += (double)a; s2 += (double)a * a; }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) { t1 += red[0][i][lc]; t2 += red[1][i][lc]; }
    seg_sums[((int64_t)f * 2 + 0) * n + c] = t1;
    seg_sums[((int64_t)f * 2 + 1) * n + c] = t2;
    const int64_t m = r1 - r0;
    double sc = 0.0, sh = 0.0;
    if (m > 0) {
      const double mean = t1 / (double)m;
      double var = t2 / (double)m - mean * mean;
      if (var < 0.0) var = 0.0;
      const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
      sc = gm / sqrt(var + (double)eps);
      sh = bt - mean * sc;
    }
    ss[0][lc] = (float)sc; ss[1][lc] = (float)sh;
  }
  __syncthreads();
  if (c >= n) return;
  const float sc = ss[0][lc], sh = ss[1][lc];
  for (int64_t r = r0 + g; r < r1; r += 16) {
    float o = fmaf(x[r * ldx + c], sc, sh);
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
                                                         float momentum, float eps, float* __restrict__ table /*[F][2][n]*/,
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
    double sc = 0.0, sh = 0.0, mean = 0.0, unbiased = 0.0;
    if (m > 0) {
      mean = seg_sums[(f * 2 + 0) * n + c] / (double)m;
      double var = seg_sums[(f * 2 + 1) * n + c] / (double)m - mean * mean;      // biased, as F.batch_norm normalises with
      if (var < 0.0) var = 0.0;
      sc = gm / sqrt(var + (double)eps);
      sh = bt - mean * sc;
      unbiased = (m > 1) ? var * (double)m / (double)(m - 1) : var;
    }
    seg_sums[(f * 2 + 0) * n + c] = mean;                 // (the scratch now holds what the recurrence below reads)
    seg_sums[(f * 2 + 1) * n + c] = unbiased;
    table[(f * 2 + 0) * n + c] = (float)sc;
    table[(f * 2 + 1) * n + c] = (float)sh;
    if (out_bound != nullptr) {
      const double b = fabs((double)(float)sc) * (double)in_b + fabs((double)(float)sh);
      bound = b > bound ? b : bound;
    }
  }
  bmax[g][lc] = (float)(bound * 1.0001);
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

// y[r] = act(x[r] * scale[seg(r)] + shift[seg(r)]): one block per 64 rows x all channels; rows are sorted by segment, so a
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
    const float* sc = table + (f * 2 + 0) * n, *sh = table + (f * 2 + 1) * n;
    if (vec) {
      const float4 v = *(const float4*)(x + r * ldx + c), a = *(const float4*)(sc + c), b = *(const float4*)(sh + c);
      float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *(float4*)(y + r * ldy + c) = o;
    } else {
      for (int j = 0; j < 4 && c + j < n; j++) {
        float o = fmaf(x[r * ldx + c + j], sc[c + j], sh[c + j]);
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
  __shared__ double red[4][GR][CH];
  const int lc = threadIdx.x & (CH - 1), g = threadIdx.x / CH;
  const int c = blockIdx.x * CH + lc;
  double s1 = 0.0, s2 = 0.0, b1 = 0.0, b2 = 0.0;
  if (c < n) {
    if (use_batch)
      for (int64_t p = g; p < panels_f; p += GR) {
        s1 += (double)fwd_stats[(p * 2 + 0) * n + c];
        s2 += (double)fwd_stats[(p * 2 + 1) * n + c];
      }
    for (int64_t p = g; p < panels_b; p += GR) {
      b1 += (double)bwd_part[(p * 2 + 0) * n + c];
      b2 += (double)bwd_part[(p * 2 + 1) * n + c];
    }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2; red[2][g][lc] = b1; red[3][g][lc] = b2;
  __syncthreads();
  if (g != 0 || c >= n) return;
  s1 = s2 = b1 = b2 = 0.0;
#pragma unroll
  for (int i = 0; i < GR; i++) { s1 += red[0][i][lc]; s2 += red[1][i][lc]; b1 += red[2][i][lc]; b2 += red[3][i][lc]; }
  double mean, var;
  if (use_batch) {
    mean = s1 / (double)m;
    var = s2 / (double)m - mean * mean;
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
  hipLaunchKernelGGL(k_bn_finalize, dim3(rgnn_blocks(n, 16)), dim3(1024), 0, (hipStream_t)stream, col_stats, panels,
                     (const int64_t*)nullptr, (const float*)nullptr, (int64_t)0, (const int64_t*)nullptr, m, n,
                     gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, scale_shift,
                     (const float*)nullptr, (float*)nullptr);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
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
// Column sums per segment from the partial sums the dense launches left per 128-row panel of their (segment-padded) row lists:
// segment f owns panels [start_a[f], start_a[f + 1]) of list a and [start_b[f], start_b[f + 1]) of list b.  float64, fixed order.
__global__ __launch_bounds__(256) void k_bn_seg_from_panels(const float* __restrict__ stats_a, const int32_t* __restrict__ start_a,
                                                           const float* __restrict__ stats_b, const int32_t* __restrict__ start_b,
                                                           int n, double* __restrict__ seg_sums) {
  __shared__ double red[2][4][64];
  const int f = blockIdx.x, lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lc;
  double s1 = 0.0, s2 = 0.0;
  if (c < n) {
    for (int part = 0; part < 2; part++) {
      const float* st = part ? stats_b : stats_a;
      const int32_t* sp = part ? start_b : start_a;
      if (st == nullptr) continue;
      for (int p = sp[f] + g; p < sp[f + 1]; p += 4) {
        s1 += (double)st[((int64_t)p * 2 + 0) * n + c];
        s2 += (double)st[((int64_t)p * 2 + 1) * n + c];
      }
    }
  }
  red[0][g][lc] = s1; red[1][g][lc] = s2;
  __syncthreads();
  if (g == 0 && c < n) {
    seg_sums[((int64_t)f * 2 + 0) * n + c] = red[0][0][lc] + red[0][1][lc] + red[0][2][lc] + red[0][3][lc];
    seg_sums[((int64_t)f * 2 + 1) * n + c] = red[1][0][lc] + red[1][1][lc] + red[1][2][lc] + red[1][3][lc];
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
