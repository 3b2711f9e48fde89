// Detection loss of the reference's trainer on the device (second half): gnn/trainer.py:181-222 computes
//   loss = alpha * CrossEntropyLoss(weight)(cls, label) + beta * (1 / num_bb) * sum_{i : label_i != bg} HuberLoss()(bb_true_i, bb_i)
// with a Python loop over the nodes for the second term (one HuberLoss call per object node: 192 000 iterations per C2
// batch).  Here: one pass over the nodes for the four global sums (sum w_y nll, sum w_y, sum huber_i, num_bb), a one-block
// finish, and -- for backward -- one pass writing d loss / d cls and d loss / d bb.  torch semantics kept: weighted mean for
// the cross entropy (sum w nll / sum w), HuberLoss(delta) averaged over the box dimensions of a node, loss_bb = 0 when the
// batch holds no object or the sum is NaN (trainer.py:203-217: the batch's box loss is then ignored).
#include "common.h"
#include <math.h>

namespace {

constexpr int LOSS_THREADS = 256;

struct LossParams {
  const float* cls; int64_t ldc; int K;
  const float* bb; int64_t ldb; int W;
  const float* y; int64_t ldy;          // [n, 1 + W]: label | box
  const float* weight;                  // [K] or null (all ones)
  int64_t n; int bg; float delta;
};

__device__ __forceinline__ double block_sum(double v, double* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < LOSS_THREADS / 64; i++) t += lds[i];
  return t;   // valid in thread 0
}

// per node: nll, weight, huber (mean over the box dimensions), is-object
__device__ __forceinline__ void node_terms(const LossParams& p, int64_t i, float& wnll, float& w, float& hub, float& obj,
                                           float& lse_out, int& label_out) {
  const float* c = p.cls + i * p.ldc;
  float m = c[0];
  for (int k = 1; k < p.K; k++) m = fmaxf(m, c[k]);
  float se = 0.f;
  for (int k = 0; k < p.K; k++) se += expf(c[k] - m);
  const float lse = m + logf(se);
  const int label = (int)p.y[i * p.ldy];                 // .long() of the float label
  const bool ok = label >= 0 && label < p.K;
  w = ok ? (p.weight ? p.weight[label] : 1.f) : 0.f;
  wnll = ok ? w * (lse - c[label]) : 0.f;
  lse_out = lse; label_out = label;
  hub = 0.f; obj = 0.f;
  if (label != p.bg) {
    obj = 1.f;
    float s = 0.f;
    for (int d = 0; d < p.W; d++) {
      const float e = p.bb[i * p.ldb + d] - p.y[i * p.ldy + 1 + d];
      const float a = fabsf(e);
      s += (a <= p.delta) ? 0.5f * e * e : p.delta * (a - 0.5f * p.delta);
    }
    hub = s / (float)p.W;
  }
}

__global__ __launch_bounds__(LOSS_THREADS) void k_loss_partial(const LossParams p, double* __restrict__ partial) {
  __shared__ double lds[LOSS_THREADS / 64];
  double a = 0, b = 0, c = 0, d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
    float wnll, w, hub, obj, lse; int label;
    node_terms(p, i, wnll, w, hub, obj, lse, label);
    a += wnll; b += w; c += hub; d += obj;
  }
  a = block_sum(a, lds); b = block_sum(b, lds); c = block_sum(c, lds); d = block_sum(d, lds);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 4 + 0] = a; partial[blockIdx.x * 4 + 1] = b;
    partial[blockIdx.x * 4 + 2] = c; partial[blockIdx.x * 4 + 3] = d;
  }
}

// sums[0..3] = the four totals; out[0] = loss, out[1] = loss_cls, out[2] = loss_bb
__global__ __launch_bounds__(LOSS_THREADS) void k_loss_finish(const double* __restrict__ partial, int nb, float alpha, float beta,
                                                             double* __restrict__ sums, float* __restrict__ out) {
  __shared__ double lds[LOSS_THREADS / 64];
  double v[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    for (int q = 0; q < 4; q++) v[q] += partial[i * 4 + q];
  for (int q = 0; q < 4; q++) v[q] = block_sum(v[q], lds);
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; q++) sums[q] = v[q];
    const double lc = v[0] / v[1];                                  // (0 / 0 = NaN like torch when every weight is 0)
    double lb = (v[3] > 0) ? v[2] / v[3] : 0.0;
    if (lb != lb) lb = 0.0;                                          // NaN box loss: ignored (trainer.py:210-217)
    out[0] = (float)(alpha * lc + beta * lb); out[1] = (float)lc; out[2] = (float)lb;
  }
}

__global__ __launch_bounds__(LOSS_THREADS) void k_loss_bwd(const LossParams p, const double* __restrict__ sums, float alpha,
                                                          float beta, const float* __restrict__ gout,
                                                          float* __restrict__ dcls, int64_t lddc, float* __restrict__ dbb,
                                                          int64_t lddb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const float g = gout ? gout[0] : 1.f;
  float wnll, w, hub, obj, lse; int label;
  node_terms(p, i, wnll, w, hub, obj, lse, label);
  const float sc = g * alpha * w / (float)sums[1];
  const float* c = p.cls + i * p.ldc;
  for (int k = 0; k < p.K; k++) dcls[i * lddc + k] = sc * (expf(c[k] - lse) - (k == label ? 1.f : 0.f));
  const double lb = (sums[3] > 0) ? sums[2] / sums[3] : 0.0;
  const bool live = obj != 0.f && sums[3] > 0 && lb == lb;
  const float sb = live ? g * beta / ((float)sums[3] * (float)p.W) : 0.f;
  for (int d = 0; d < p.W; d++) {
    float e = p.bb[i * p.ldb + d] - p.y[i * p.ldy + 1 + d];
    e = fminf(fmaxf(e, -p.delta), p.delta);                          // d huber / d pred
    dbb[i * lddb + d] = live ? sb * e : 0.f;
  }
}

}  // namespace

extern "C" int64_t rgnn_detection_loss_blocks(int64_t n) {
  int64_t nb = (n + LOSS_THREADS - 1) / LOSS_THREADS;
  return nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
}

extern "C" int rgnn_detection_loss(const float* cls, int64_t ldc, int32_t n_classes, const float* boxes, int64_t ldb,
                                   int32_t box_width, const float* y, int64_t ldy, const float* class_weight, int64_t n,
                                   int32_t bg_index, float delta, float cls_loss_weight, float bb_loss_weight,
                                   double* partial_tmp, double* sums, float* loss_out, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_classes >= 1 && box_width >= 1 && delta > 0, "bad sizes");
  RGNN_CHECK_ARG(partial_tmp && sums && loss_out && (n == 0 || (cls && boxes && y)), "null pointers");
  const int nb = (int)rgnn_detection_loss_blocks(n);
  const LossParams p{cls, ldc, n_classes, boxes, ldb, box_width, y, ldy, class_weight, n, bg_index, delta};
  hipLaunchKernelGGL(k_loss_partial, dim3(nb), dim3(LOSS_THREADS), 0, (hipStream_t)stream, p, partial_tmp);
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, partial_tmp, nb, cls_loss_weight,
                     bb_loss_weight, sums, loss_out);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_detection_loss_bwd(const float* cls, int64_t ldc, int32_t n_classes, const float* boxes, int64_t ldb,
                                       int32_t box_width, const float* y, int64_t ldy, const float* class_weight, int64_t n,
                                       int32_t bg_index, float delta, float cls_loss_weight, float bb_loss_weight,
                                       const double* sums, const float* grad_loss, float* d_cls, int64_t lddc, float* d_boxes,
                                       int64_t lddb, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_classes >= 1 && box_width >= 1, "bad sizes");
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(cls && boxes && y && sums && d_cls && d_boxes, "null pointers");
  const LossParams p{cls, ldc, n_classes, boxes, ldb, box_width, y, ldy, class_weight, n, bg_index, delta};
  hipLaunchKernelGGL(k_loss_bwd, dim3(rgnn_blocks(n, LOSS_THREADS)), dim3(LOSS_THREADS), 0, (hipStream_t)stream, p, sums,
                     cls_loss_weight, bb_loss_weight, grad_loss, d_cls, lddc, d_boxes, lddb);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
