// Post-processor front half on the device: what the reference does per graph in numpy loops after the
// forward pass -- PredictionExtractor.get_predicted_label / get_prediction_scores / get_clutter_scores and
// get_absolute_object_bounding_box_predictions (postprocessor/postprocessing.py:177-196,198-319), with the box algebra of
// preprocessor/bounding_box.py:21-66 (absolute rotated), :68-153 (E(n)-invariant), :156-199 (relative rotated), :275-312
// (relative aligned) and :566-589 (inverse of the sin-smoothed angle).
//
// One thread per node: label = FIRST index of the row maximum, score = that maximum, keep flag from the three removal
// rules, and the four corners of the absolute box.  The box arithmetic runs in float64 on the float32 predictions (numpy
// scalars of the reference's pinned numpy 1.x promote to float64 with Python floats); the threshold comparisons keep the
// reference's dtypes: background probability float32 >= float32(threshold), score float64 <= float64 threshold.
// HBM-bound: reads 4 (K + W + 2) bytes per node, writes 76.
#include "common.h"
#include <math.h>
#include <type_traits>

namespace {

constexpr double PI_D = 3.141592653589793;

struct DecodeParams {
  const float* prob; int64_t ldp; int K;
  const float* bb; int64_t ldb; int W;
  const float* pos;
  const int32_t* nn;
  int64_t n;
  int bg_index; float max_bg;
  const double* min_score; int n_min;
  int invariance;            // 0 none, 1 translation, 2 en
  int adapt_angle;
  int32_t* label; float* score; int32_t* keep; double* corners;
};

__device__ __forceinline__ double round5(double x) { return rint(x * 100000.0) / 100000.0; }   // np.round(x, 5)

__global__ __launch_bounds__(256) void k_decode(const DecodeParams p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const float* pr = p.prob + i * p.ldp;
  int best = 0;
  float m = pr[0];
  for (int k = 1; k < p.K; k++) {
    const float v = pr[k];
    if (v > m) { m = v; best = k; }                     // np.where(vec == max)[0][0]: the first maximum
  }
  bool remove = false;
  if (p.bg_index >= 0 && p.bg_index < p.K) remove = pr[p.bg_index] >= p.max_bg;     // postprocessing.py:222
  remove = remove || best == p.bg_index;                                            // :223
  if (best < p.n_min) remove = remove || ((double)m <= p.min_score[best]);          // :227-228
  p.label[i] = best;
  p.score[i] = m;
  p.keep[i] = remove ? 0 : 1;

  const float* b = p.bb + i * p.ldb;
  const double px = (double)p.pos[2 * i], py = (double)p.pos[2 * i + 1];
  double cx, cy, hl, hw, theta_deg = 0.0;
  bool rotate = false;
  if (p.W == 4) {                                        // relative aligned box: bounding_box.py:275-312
    cx = px + (double)b[0]; cy = py + (double)b[1];
    hl = (double)b[2] / 2; hw = (double)b[3] / 2;
  } else if (p.invariance != 2) {
    hl = (double)b[2] / 2; hw = (double)b[3] / 2;
    double th = (double)b[4];
    if (p.adapt_angle) {                                 // invert_bb_orientation_angle_adaption, bounding_box.py:566-589
      th = fmax(fmin(th, 1.0), -1.0);
      th = asin(th);
      if (th < 0) th = th + PI_D;
    }
    theta_deg = th * 180 / PI_D;
    rotate = true;
    if (p.invariance == 1) { cx = px + (double)b[0]; cy = py + (double)b[1]; }      // relative rotated :156-199
    else { cx = (double)b[0]; cy = (double)b[1]; }                                  // absolute rotated :21-66
  } else {                                               // E(n)-invariant representation: bounding_box.py:97-153
    const int32_t j = p.nn[i];
    const double vx = (double)p.pos[2 * (int64_t)j] - px, vy = (double)p.pos[2 * (int64_t)j + 1] - py;
    const double nrm = sqrt(vx * vx + vy * vy);
    const double th_nn = atan2(vy / nrm, vx / nrm) * 180 / PI_D;
    const double d = (double)b[0];
    const double th_pc_rel = (double)b[1] * 180 / PI_D, th_dir_rel = (double)b[4] * 180 / PI_D;
    double th_dir = round5(th_dir_rel + th_nn);
    while (th_dir < 0) th_dir = 360 + th_dir;
    while (th_dir >= 180) th_dir = th_dir - 180;
    double th_pc = th_pc_rel + th_nn;
    while (th_pc > 360) th_pc = th_pc - 360;
    const double xc = d * cos((th_pc * PI_D) / 180), yc = d * sin((th_pc * PI_D) / 180);
    hl = (double)b[2] / 2; hw = (double)b[3] / 2;
    theta_deg = th_dir;
    rotate = true;
    cx = px + xc; cy = py + yc;
  }
  double c = 1.0, s = 0.0;
  if (rotate) {
    const double rad = (theta_deg * PI_D) / 180;
    c = cos(rad); s = sin(rad);
  }
  const double ox[4] = {hl, hl, -hl, -hl}, oy[4] = {hw, -hw, -hw, hw};              // c1..c4 before rotation
  double* out = p.corners + i * 8;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const double rx = rotate ? c * ox[q] + (-s) * oy[q] : ox[q];
    const double ry = rotate ? s * ox[q] + c * oy[q] : oy[q];
    out[2 * q] = rx + cx;
    out[2 * q + 1] = ry + cy;
  }
}

}  // namespace

extern "C" int rgnn_decode_predictions(const float* class_prob, int64_t ldp, int32_t n_classes, const float* boxes,
                                       int64_t ldb, int32_t box_width, const float* pos, const int32_t* nn_index, int64_t n,
                                       int32_t bg_index, float max_score_for_background, const double* min_object_score,
                                       int32_t n_min_scores, int32_t invariance, int32_t adapt_orientation_angle,
                                       int32_t* label, float* score, int32_t* keep, double* corners, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_classes >= 1, "bad sizes");
  RGNN_CHECK_ARG(box_width == 4 || box_width == 5, "boxes are [x, y, dx, dy] or [x, y, l, w, theta]");
  RGNN_CHECK_ARG(invariance >= 0 && invariance <= 2, "invariance: 0 none, 1 translation, 2 en");
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(!(invariance == 2 && box_width == 5) || nn_index != nullptr, "the en representation needs nearest neighbours");
  RGNN_CHECK_ARG(n_min_scores == 0 || min_object_score != nullptr, "null thresholds");
  RGNN_CHECK_ARG(class_prob && boxes && pos && label && score && keep && corners, "null pointers");
  DecodeParams p{class_prob, ldp, n_classes, boxes, ldb, box_width, pos, nn_index, n, bg_index, max_score_for_background,
                 min_object_score, n_min_scores, invariance, adapt_orientation_angle, label, score, keep, corners};
  hipLaunchKernelGGL(k_decode, dim3(rgnn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, p);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Non-maximum suppression (BoxSuppressor.apply_nms, postprocessor/postprocessing.py:336-431): greedy, by descending score;
// a box is dropped when its IoU with an already kept box exceeds the threshold.
//   kind 0  aligned boxes [x_min, y_min, x_max, y_max] float32 -- torchvision.ops.nms: suppress when IoU >  threshold
//   kind 1  rotated boxes [x, y, l, w, theta(deg)]     float64 -- detectron2 nms_rotated (CPU): suppress when IoU >= threshold
// Two kernels: (1) the M x M "j is suppressed by i" bit matrix over the score-sorted boxes, 64 x 64 tiles, one row per lane,
// the 64 column boxes of a tile staged in LDS; (2) one work-group walks the sorted boxes 64 at a time: the in-tile greedy
// pass runs on the tile's diagonal words with wave broadcasts, then the rows of the boxes kept are OR-ed into the
// `removed` bitmap (LDS) by all threads.  Rotated IoU: the rectangle of box 1 clipped by the four half-planes of box 2
// (Sutherland-Hodgman, <= 8 vertices), shoelace area -- the same quantity detectron2 obtains via intersection points +
// convex hull.
// ------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float iou_aligned(const float* a, const float* b) {
  const float area_a = (a[2] - a[0]) * (a[3] - a[1]), area_b = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = fmaxf(0.f, fminf(a[2], b[2]) - fmaxf(a[0], b[0]));
  const float h = fmaxf(0.f, fminf(a[3], b[3]) - fmaxf(a[1], b[1]));
  const float inter = w * h;
  return inter / (area_a + area_b - inter);
}

// Corner convention of detectron2's box_iou_rotated (get_rotated_vertices in layers/csrc/box_iou_rotated/
// box_iou_rotated_utils.h, image coordinates with y pointing down): the long side points along (cos t, -sin t),
//   pts[0] = ctr + l/2 (cos, -sin) + w/2 (sin, cos).
// The reference hands theta_x = atan2(v_y, v_x) of its y-up frame straight to nms_rotated (postprocessing.py:356-370), so
// what it suppresses on are the IoUs of boxes mirrored about their own centre -- and a mirrored pair overlaps differently
// from the original pair unless the boxes are axis aligned.  Parity with the reference means reproducing that.
__device__ __forceinline__ void rect_vertices(double x, double y, double l, double w, double deg, double (&vx)[4], double (&vy)[4]) {
  const double t = deg * PI_D / 180;
  const double c = cos(t), s = -sin(t), a = l / 2, b = w / 2;
  vx[0] = x + a * c - b * s; vy[0] = y + a * s + b * c;      // (one orientation; the clipping below works with |area|)
  vx[1] = x - a * c - b * s; vy[1] = y - a * s + b * c;
  vx[2] = x - a * c + b * s; vy[2] = y - a * s - b * c;
  vx[3] = x + a * c + b * s; vy[3] = y + a * s - b * c;
}

__device__ double iou_rotated(const double* p, const double* q) {
  const double area_p = p[2] * p[3], area_q = q[2] * q[3];
  if (area_p < 1e-14 || area_q < 1e-14) return 0.0;
  const double sx = (p[0] + q[0]) / 2, sy = (p[1] + q[1]) / 2;     // work around the common centre (precision)
  double ax[4], ay[4], bx[4], by[4];
  rect_vertices(p[0] - sx, p[1] - sy, p[2], p[3], p[4], ax, ay);
  rect_vertices(q[0] - sx, q[1] - sy, q[2], q[3], q[4], bx, by);
  double px[8], py[8], qx[8], qy[8];
  int n = 4;
  for (int i = 0; i < 4; i++) { px[i] = ax[i]; py[i] = ay[i]; }
  for (int e = 0; e < 4 && n > 0; e++) {
    const double ex = bx[(e + 1) & 3] - bx[e], ey = by[(e + 1) & 3] - by[e];
    int m = 0;
    for (int i = 0; i < n; i++) {
      const int k = (i + 1 == n) ? 0 : i + 1;
      const double d0 = ex * (py[i] - by[e]) - ey * (px[i] - bx[e]);      // >= 0: inside (left of the edge)
      const double d1 = ex * (py[k] - by[e]) - ey * (px[k] - bx[e]);
      // (a convex polygon gains at most one vertex per clip edge; rounding in near-degenerate inputs could produce more
      // sign changes, so the writes are bounded by the array, not by the geometry)
      if (d0 >= 0 && m < 8) { qx[m] = px[i]; qy[m] = py[i]; m++; }
      if ((d0 >= 0) != (d1 >= 0) && m < 8) {
        const double t = d0 / (d0 - d1);
        qx[m] = px[i] + t * (px[k] - px[i]); qy[m] = py[i] + t * (py[k] - py[i]); m++;
      }
    }
    n = m < 8 ? m : 8;
    for (int i = 0; i < n; i++) { px[i] = qx[i]; py[i] = qy[i]; }
  }
  double inter = 0.0;
  for (int i = 0; i < n; i++) {
    const int k = (i + 1 == n) ? 0 : i + 1;
    inter += px[i] * py[k] - px[k] * py[i];
  }
  inter = fabs(inter) / 2;
  return inter / (area_p + area_q - inter);
}

template <int KIND>
__global__ __launch_bounds__(64) void k_nms_mask(const void* __restrict__ boxes_, const int64_t* __restrict__ order, int64_t m,
                                                double thr, unsigned long long* __restrict__ mask, int words) {
  using T = typename std::conditional<KIND == 0, float, double>::type;
  constexpr int BW = KIND == 0 ? 4 : 5;
  const T* boxes = (const T*)boxes_;
  const int bi = blockIdx.y, bj = blockIdx.x, lane = threadIdx.x;
  const int64_t i = (int64_t)bi * 64 + lane;
  if (bj < bi) {                                           // columns before the row: never suppressed by it
    if (i < m) mask[i * words + bj] = 0ull;
    return;
  }
  __shared__ T col[64][BW];
  const int64_t jc = (int64_t)bj * 64 + lane;
  if (jc < m) {
    const T* src = boxes + order[jc] * BW;
    for (int c = 0; c < BW; c++) col[lane][c] = src[c];
  }
  __syncthreads();
  if (i >= m) return;
  T mine[BW];
  const T* src = boxes + order[i] * BW;
  for (int c = 0; c < BW; c++) mine[c] = src[c];
  unsigned long long bits = 0ull;
  const int ncol = (int)((m - (int64_t)bj * 64 < 64) ? m - (int64_t)bj * 64 : 64);
  for (int c = 0; c < ncol; c++) {
    const int64_t j = (int64_t)bj * 64 + c;
    if (j <= i) continue;
    bool hit;
    if constexpr (KIND == 0) hit = iou_aligned(mine, col[c]) > (float)thr;
    else hit = iou_rotated(mine, col[c]) >= thr;
    if (hit) bits |= 1ull << c;
  }
  mask[i * words + bj] = bits;
}

constexpr int NMS_MAX_WORDS = 2048;     // 131 072 boxes per call

__global__ __launch_bounds__(256) void k_nms_reduce(const unsigned long long* __restrict__ mask, const int64_t* __restrict__ order,
                                                   int64_t m, int words, int64_t* __restrict__ keep, int64_t* __restrict__ count) {
  __shared__ unsigned long long removed[NMS_MAX_WORDS];
  __shared__ unsigned long long kept_bits;
  const int t = threadIdx.x, lane = t & 63;
  for (int w = t; w < words; w += 256) removed[w] = 0ull;
  __syncthreads();
  int64_t n_keep = 0;
  for (int b = 0; b < words; b++) {
    if (t < 64) {                                          // in-tile greedy pass on the diagonal words (wave 0)
      const int64_t i = (int64_t)b * 64 + lane;
      const unsigned long long diag = (i < m) ? mask[i * words + b] : 0ull;
      unsigned long long cur = removed[b], kb = 0ull;
      const int nrow = (int)((m - (int64_t)b * 64 < 64) ? m - (int64_t)b * 64 : 64);
      for (int l = 0; l < nrow; l++) {
        const unsigned long long row = __shfl(diag, l, 64);
        if (!((cur >> l) & 1ull)) { kb |= 1ull << l; cur |= row; }
      }
      if ((kb >> lane) & 1ull) keep[n_keep + __popcll(kb & ((1ull << lane) - 1ull))] = order[i];
      if (lane == 0) kept_bits = kb;
    }
    __syncthreads();
    const unsigned long long kb = kept_bits;
    n_keep += __popcll(kb);
    for (int w = b + 1 + t; w < words; w += 256) {         // rows of the boxes kept suppress the later tiles
      unsigned long long acc = removed[w], rest = kb;
      while (rest) {
        const int l = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        acc |= mask[((int64_t)b * 64 + l) * words + w];
      }
      removed[w] = acc;
    }
    __syncthreads();
  }
  if (t == 0) *count = n_keep;
}

}  // namespace

namespace {
// BoundingBox.get_to_two_point_representation / get_absolute_rotated_box_representations (preprocessor/bounding_box.py:447-
// 540) for all boxes at once: [x_min, y_min, x_max, y_max] and [x_centre, y_centre, l, w, theta in [0, 180]] per box.
// l / w: the two shorter of the distances corner 1 -> corners 2, 3, 4 (the longest is the diagonal); theta: direction of
// the side of length l; a box whose l matches none of the three (NaN corners) becomes the reference's default [0,0,1,1,0].
__global__ __launch_bounds__(256) void k_box_repr(const double* __restrict__ corners, int64_t m, double* __restrict__ two_point,
                                                 double* __restrict__ rotated) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double x[4], y[4];
  for (int q = 0; q < 4; q++) { x[q] = corners[i * 8 + 2 * q]; y[q] = corners[i * 8 + 2 * q + 1]; }
  if (two_point) {
    two_point[i * 4 + 0] = fmin(fmin(x[0], x[1]), fmin(x[2], x[3]));
    two_point[i * 4 + 1] = fmin(fmin(y[0], y[1]), fmin(y[2], y[3]));
    two_point[i * 4 + 2] = fmax(fmax(x[0], x[1]), fmax(x[2], x[3]));
    two_point[i * 4 + 3] = fmax(fmax(y[0], y[1]), fmax(y[2], y[3]));
  }
  if (rotated) {
    double d[3];
    for (int q = 0; q < 3; q++) {
      const double dx = x[0] - x[q + 1], dy = y[0] - y[q + 1];
      d[q] = sqrt(dx * dx + dy * dy);
    }
    int iw = 0;                                            // min(d), first occurrence; then min of the other two
    if (d[1] < d[iw]) iw = 1;
    if (d[2] < d[iw]) iw = 2;
    int il = -1;
    for (int q = 0; q < 3; q++)
      if (q != iw && (il < 0 || d[q] < d[il])) il = q;
    const double w = d[iw], l = d[il];
    int side = -1;                                         // first of d1, d2, d3 equal to l
    for (int q = 2; q >= 0; q--)
      if (d[q] == l) side = q;
    double* o = rotated + i * 5;
    if (side < 0) { o[0] = 0; o[1] = 0; o[2] = 1; o[3] = 1; o[4] = 0; return; }
    const double vx = x[0] - x[side + 1], vy = y[0] - y[side + 1];
    const double nrm = sqrt(vx * vx + vy * vy);
    double theta = atan2(vy / nrm, vx / nrm) * 180 / PI_D;
    if (theta < 0) theta = 180 + theta;
    o[0] = (x[0] + x[1] + x[2] + x[3]) / 4;
    o[1] = (y[0] + y[1] + y[2] + y[3]) / 4;
    o[2] = l; o[3] = w; o[4] = theta;
  }
}
}  // namespace

extern "C" int rgnn_box_representations(const double* corners, int64_t m, double* two_point, double* rotated, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0, "bad sizes");
  if (m == 0) return RGNN_OK;
  RGNN_CHECK_ARG(corners && (two_point || rotated), "null pointers");
  hipLaunchKernelGGL(k_box_repr, dim3(rgnn_blocks(m, 256)), dim3(256), 0, (hipStream_t)stream, corners, m, two_point, rotated);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// ------------------------------------------------------------------------------------------------
// Score order for NMS: box ids by descending score, ties by ascending id (a stable descending sort; torchvision / detectron2 sort
// the scores first, postprocessing.py:336-435).  Bitonic network over (key, id) pairs: chunks of SORT_CHUNK pairs are sorted in
// LDS by one launch; for more boxes every merge stage runs its long-distance steps as one launch each and its short-distance
// steps in LDS again.  NaN scores order as the largest value (torch.sort's convention).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SORT_CHUNK = 4096;
__device__ __forceinline__ unsigned long long score_key(double v) {
  if (v != v) return ~0ull;                                    // NaN: first
  if (v == 0.0) v = 0.0;                                       // -0.0 and +0.0 tie (a comparison sort cannot tell them apart)
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);         // order-preserving: larger score <-> larger key
}
// a goes before b: larger key first, equal keys by ascending id
__device__ __forceinline__ bool sort_before(unsigned long long ka, unsigned ia, unsigned long long kb, unsigned ib) {
  return ka > kb || (ka == kb && ia < ib);
}
// MODE 0: load scores, sort the chunk completely (stages k <= SORT_CHUNK); MODE 1: steps j < SORT_CHUNK of stage k on loaded pairs
template <int MODE>
__global__ __launch_bounds__(1024) void k_sort_chunk(const void* __restrict__ scores, int is_f64, int64_t m, int64_t padded,
                                                    unsigned long long* __restrict__ keys, unsigned* __restrict__ ids, int64_t k_stage) {
  __shared__ unsigned long long sk[SORT_CHUNK];
  __shared__ unsigned si[SORT_CHUNK];
  const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
  for (int t = threadIdx.x; t < SORT_CHUNK; t += blockDim.x) {
    const int64_t i = base + t;
    if (MODE == 0) {
      if (i < m) {
        const double v = is_f64 ? ((const double*)scores)[i] : (double)((const float*)scores)[i];
        sk[t] = score_key(v); si[t] = (unsigned)i;
      } else { sk[t] = 0ull; si[t] = 0xffffffffu; }            // padding: after every real pair
    } else { sk[t] = keys[i]; si[t] = ids[i]; }
  }
  __syncthreads();
  const int64_t k_lo = MODE == 0 ? 2 : k_stage, k_hi = MODE == 0 ? SORT_CHUNK : k_stage;
  for (int64_t k = k_lo; k <= k_hi; k <<= 1) {
    for (int j = (int)((k >> 1) < SORT_CHUNK ? (k >> 1) : (SORT_CHUNK >> 1)); j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < SORT_CHUNK; t += blockDim.x) {
        const int p = t ^ j;
        if (p > t) {
          const bool up = (((base + t) & k) == 0);             // this pair belongs to an ascending (in sort_before order) run
          const unsigned long long ka = sk[t], kb = sk[p];
          const unsigned ia = si[t], ib = si[p];
          if (sort_before(kb, ib, ka, ia) == up) { sk[t] = kb; si[t] = ib; sk[p] = ka; si[p] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < SORT_CHUNK; t += blockDim.x) { keys[base + t] = sk[t]; ids[base + t] = si[t]; }
}
__global__ __launch_bounds__(256) void k_sort_step(unsigned long long* __restrict__ keys, unsigned* __restrict__ ids, int64_t padded,
                                                  int64_t j, int64_t k) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= padded) return;
  const int64_t p = t ^ j;
  if (p <= t) return;
  const bool up = ((t & k) == 0);
  const unsigned long long ka = keys[t], kb = keys[p];
  const unsigned ia = ids[t], ib = ids[p];
  if (sort_before(kb, ib, ka, ia) == up) { keys[t] = kb; ids[t] = ib; keys[p] = ka; ids[p] = ia; }
}
__global__ __launch_bounds__(256) void k_sort_emit(const unsigned* __restrict__ ids, int64_t m, int64_t* __restrict__ order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) order[t] = (int64_t)ids[t];
}
}  // namespace

static int64_t sort_padded(int64_t m) {
  int64_t p = SORT_CHUNK;
  while (p < m) p <<= 1;
  return p;
}
extern "C" int64_t rgnn_sort_scores_tmp_bytes(int64_t m) { return sort_padded(m) * 12 + 256; }

extern "C" int rgnn_sort_scores(const void* scores, int32_t is_f64, int64_t m, int64_t* order, void* tmp, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0 && m < ((int64_t)1 << 31), "bad size");
  if (m == 0) return RGNN_OK;
  RGNN_CHECK_ARG(scores && order && tmp, "null pointers");
  const int64_t padded = sort_padded(m);
  unsigned long long* keys = (unsigned long long*)tmp;
  unsigned* ids = (unsigned*)((char*)tmp + padded * 8);
  hipStream_t s = (hipStream_t)stream;
  const unsigned chunks = (unsigned)(padded / SORT_CHUNK);
  hipLaunchKernelGGL(k_sort_chunk<0>, dim3(chunks), dim3(1024), 0, s, scores, (int)is_f64, m, padded, keys, ids, (int64_t)0);
  for (int64_t k = 2 * SORT_CHUNK; k <= padded; k <<= 1) {
    for (int64_t j = k >> 1; j >= SORT_CHUNK; j >>= 1)
      hipLaunchKernelGGL(k_sort_step, dim3(rgnn_blocks(padded, 256)), dim3(256), 0, s, keys, ids, padded, j, k);
    hipLaunchKernelGGL(k_sort_chunk<1>, dim3(chunks), dim3(1024), 0, s, scores, (int)is_f64, m, padded, keys, ids, k);
  }
  hipLaunchKernelGGL(k_sort_emit, dim3(rgnn_blocks(m, 256)), dim3(256), 0, s, (const unsigned*)ids, m, order);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int64_t rgnn_nms_mask_words(int64_t m) { return m * ((m + 63) / 64); }

extern "C" int rgnn_nms(const void* boxes, int32_t kind, const int64_t* order, int64_t m, double iou_threshold,
                        uint64_t* mask_tmp, int64_t* keep, int64_t* count, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(kind == 0 || kind == 1, "kind: 0 aligned float32 [m, 4], 1 rotated float64 [m, 5]");
  RGNN_CHECK_ARG(m >= 0 && count != nullptr, "bad sizes");
  if (m == 0) {
    hipMemsetAsync(count, 0, 8, (hipStream_t)stream);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(boxes && order && mask_tmp && keep, "null pointers");
  const int words = (int)((m + 63) / 64);
  if (words > NMS_MAX_WORDS) {
    rgnn_set_error("rgnn_nms: at most %d boxes per call (got %lld)", NMS_MAX_WORDS * 64, (long long)m);
    return RGNN_ERR_UNSUPPORTED;
  }
  if (kind == 0)
    hipLaunchKernelGGL(k_nms_mask<0>, dim3(words, words), dim3(64), 0, (hipStream_t)stream, boxes, order, m, iou_threshold,
                       (unsigned long long*)mask_tmp, words);
  else
    hipLaunchKernelGGL(k_nms_mask<1>, dim3(words, words), dim3(64), 0, (hipStream_t)stream, boxes, order, m, iou_threshold,
                       (unsigned long long*)mask_tmp, words);
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)mask_tmp, order, m,
                     words, keep, count);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
