// Edge / node feature extraction and the per-frame time index (gfx950).
//
// Replaces the per-edge Python loop of graph_constructor/graph.py:139-223 (+ features.py:6-122), the
// column concatenation of graph.py:225-275 and the unique-timestamp loop of
// preprocessor/radarscenes/dataset_creation.py:214-223.  HBM-bound: one thread per edge (16 B of edge_index
// + two random 32 B point records in, 4*De B out), one thread per node.  Arithmetic is float64 like the
// reference's numpy, the store casts to float32 when the caller asks for the create_graph_data dtype
// (dataset_creation.py:804-806).
#include "common.h"
#include <math.h>

namespace {

struct Codes {
  int32_t n;
  int32_t c[RGNN_MAX_FEATURE_CODES];
};

__device__ __forceinline__ void unit_or_zero(double vx, double vy, double& ux, double& uy) {
  // features.py:24-40,62-65: an all-zero vector stays zero, otherwise v / ||v||_2
  if (vx == 0.0 && vy == 0.0) {
    ux = 0.0; uy = 0.0;
  } else {
    const double nrm = sqrt(vx * vx + vy * vy);
    ux = vx / nrm; uy = vy / nrm;
  }
}

__device__ __forceinline__ double clamped_angle_deg(double dot, int32_t* status) {
  // features.py:49-58: |dot| in (1, 1+1e-3) is clamped, beyond that the reference raises
  if (fabs(dot) > 1.0) {
    if ((fabs(dot) - 1.0) < 1e-3) dot = (dot > 0) ? 1.0 : -1.0;
    else atomicOr(status, RGNN_STATUS_DOT_PRODUCT);
  }
  return acos(dot) * 180 / M_PI;
}

__device__ __forceinline__ double angle_deg(double dot) { return acos(dot) * 180 / M_PI; }

template <typename OutT>
__global__ __launch_bounds__(256) void k_edge_features(const double* __restrict__ X, const double* __restrict__ V,
                                                      const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                      Codes codes, int width, int undirected, OutT* __restrict__ out,
                                                      int32_t* __restrict__ status,
                                                      const int32_t* __restrict__ reversed_of = nullptr, int64_t n_rows = 0) {
  // reversed_of == NULL: row e of `out` = features of edge e = (i, j).  Otherwise (rgnn_edge_features_reversed): row s = features of
  // the REVERSE (j, i) of edge reversed_of[s] -- the same code on swapped end points, i.e. bit for bit what the twin edge's own row
  // holds.
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (reversed_of ? n_rows : n_edges)) return;
  const int64_t e = reversed_of ? (int64_t)reversed_of[r] : r;
  const int64_t i = edge_index[reversed_of ? n_edges + e : e], j = edge_index[reversed_of ? e : n_edges + e];
  const double2 xi = ((const double2*)X)[i], xj = ((const double2*)X)[j];
  const double2 vi = ((const double2*)V)[i], vj = ((const double2*)V)[j];
  OutT* o = out + r * width;
  int w = 0;
  for (int c = 0; c < codes.n; c++) {
    switch (codes.c[c]) {
      case RGNN_EF_POINT_PAIR: {
        double v1x, v1y, v2x, v2y;
        unit_or_zero(vi.x, vi.y, v1x, v1y);
        unit_or_zero(vj.x, vj.y, v2x, v2y);
        const double dx = xi.x - xj.x, dy = xi.y - xj.y;
        const double d = sqrt(dx * dx + dy * dy);                               // features.py:43
        const double th_v = clamped_angle_deg(v1x * v2x + v1y * v2y, status);  // features.py:46-58
        double a, b;
        if (!undirected) {
          double ux, uy;
          unit_or_zero(xj.x - xi.x, xj.y - xi.y, ux, uy);                       // (p2 - p1) / ||.||
          a = clamped_angle_deg(v1x * ux + v1y * uy, status);                   // features.py:67-79
          b = clamped_angle_deg(v2x * ux + v2y * uy, status);                   // features.py:81-93
        } else {
          double d1x, d1y, d2x, d2y;
          unit_or_zero(xi.x - xj.x, xi.y - xj.y, d1x, d1y);
          unit_or_zero(xj.x - xi.x, xj.y - xi.y, d2x, d2y);
          const double a11 = angle_deg(v1x * d1x + v1y * d1y), a12 = angle_deg(v2x * d1x + v2y * d1y);
          const double a21 = angle_deg(v1x * d2x + v1y * d2y), a22 = angle_deg(v2x * d2x + v2y * d2y);
          const double t1 = (a21 < a11) ? a21 : a11;                            // python min(a, b)
          const double t2 = (a22 < a12) ? a22 : a12;
          a = (t2 < t1) ? t2 : t1;                                              // features.py:119
          b = (t2 > t1) ? t2 : t1;                                              // features.py:120
        }
        o[w++] = (OutT)d; o[w++] = (OutT)th_v; o[w++] = (OutT)a; o[w++] = (OutT)b;
      } break;
      case RGNN_EF_SPATIAL_DISTANCE: {
        const double dx = xi.x - xj.x, dy = xi.y - xj.y;
        o[w++] = (OutT)sqrt(dx * dx + dy * dy);
      } break;
      case RGNN_EF_VELOCITY_DISTANCE: {
        const double dx = vi.x - vj.x, dy = vi.y - vj.y;
        o[w++] = (OutT)sqrt(dx * dx + dy * dy);
      } break;
      case RGNN_EF_RELATIVE_POSITION: {
        double dx = xi.x - xj.x, dy = xi.y - xj.y;                              // graph.py:199-200
        if (undirected) { dx = fabs(dx); dy = fabs(dy); }
        o[w++] = (OutT)dx; o[w++] = (OutT)dy;
      } break;
      case RGNN_EF_RELATIVE_VELOCITY: {
        double dx = vi.x - vj.x, dy = vi.y - vj.y;
        if (undirected) { dx = fabs(dx); dy = fabs(dy); }
        o[w++] = (OutT)dx; o[w++] = (OutT)dy;
      } break;
    }
  }
}

// The shipped feature list (configuration_radarscenes.yml:22: relative_position only) on its own: two 16-byte gathers and one
// 8-byte store per edge, nothing of the general kernel's velocity loads and angle code in the way (27 -> 10 us on the C2 batch).
template <typename OutT>
__global__ __launch_bounds__(256) void k_edge_relative_position(const double* __restrict__ X,
                                                               const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                               int undirected, OutT* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t i = edge_index[e], j = edge_index[n_edges + e];
  const double2 xi = ((const double2*)X)[i], xj = ((const double2*)X)[j];
  double dx = xi.x - xj.x, dy = xi.y - xj.y;                                      // graph.py:199-200
  if (undirected) { dx = fabs(dx); dy = fabs(dy); }
  out[e * 2] = (OutT)dx;
  out[e * 2 + 1] = (OutT)dy;
}

// feature row of node i (graph.py:225-275); tidx_i: the node's time index (used when the list asks for it)
template <typename OutT>
__device__ __forceinline__ void node_feature_row(const double* __restrict__ X, const double* __restrict__ V,
                                                 const double* __restrict__ rcs, double tidx_i,
                                                 const int32_t* __restrict__ degree, int64_t i, const Codes& codes, int width,
                                                 OutT* __restrict__ out) {
  OutT* o = out + i * width;
  int w = 0;
  for (int c = 0; c < codes.n; c++) {
    switch (codes.c[c]) {
      case RGNN_NF_RCS: o[w++] = (OutT)rcs[i]; break;
      case RGNN_NF_TIME_INDEX: o[w++] = (OutT)tidx_i; break;
      case RGNN_NF_DEGREE: o[w++] = (OutT)degree[i]; break;
      case RGNN_NF_VELOCITY_LENGTH: {
        const double2 v = ((const double2*)V)[i];
        o[w++] = (OutT)sqrt(v.x * v.x + v.y * v.y);
      } break;
      case RGNN_NF_VELOCITY_VECTOR: {
        const double2 v = ((const double2*)V)[i];
        o[w++] = (OutT)v.x; o[w++] = (OutT)v.y;
      } break;
      case RGNN_NF_SPATIAL_COORDINATES: {
        const double2 x = ((const double2*)X)[i];
        o[w++] = (OutT)x.x; o[w++] = (OutT)x.y;
      } break;
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void k_node_features(const double* __restrict__ X, const double* __restrict__ V,
                                                      const double* __restrict__ rcs, const double* __restrict__ tidx,
                                                      const int32_t* __restrict__ degree, int64_t n, Codes codes,
                                                      int width, OutT* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  node_feature_row<OutT>(X, V, rcs, tidx ? tidx[i] : 0.0, degree, i, codes, width, out);
}

// ------------------------------------------------------------------------------------------------
// time index: one block per frame; distinct timestamps are collected in an LDS hash set, sorted
// (bitonic, in LDS) and every point binary-searches its rank.
// ------------------------------------------------------------------------------------------------
constexpr int TI_CAP = 4096;       // hash slots (power of two)
constexpr int TI_MAX_UNIQUE = 3072;
constexpr unsigned long long TI_EMPTY = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ unsigned long long ts_key(double t) {
  if (t == 0.0) t = 0.0;  // -0.0 and +0.0 are one value for np.unique
  return (unsigned long long)__double_as_longlong(t);
}

// FEAT: instead of the time index alone (out), write the node feature rows of the frame -- the index goes straight into its
// column (one launch for k_time_index + k_node_features, and the [N] float64 index array is never written nor read back).
template <bool FEAT, typename OutT>
__global__ __launch_bounds__(1024) void k_time_index(const double* __restrict__ ts, const int64_t* __restrict__ frame_ptr,
                                                   double* __restrict__ out, int32_t* __restrict__ status,
                                                   const double* __restrict__ X, const double* __restrict__ V,
                                                   const double* __restrict__ rcs, const int32_t* __restrict__ degree,
                                                   Codes codes, int width, OutT* __restrict__ feat,
                                                   int32_t* __restrict__ frame_nonempty = nullptr) {
  __shared__ unsigned long long table[TI_CAP];
  __shared__ double vals[TI_CAP];
  __shared__ int n_unique;
  __shared__ int overflow;
  const int f = blockIdx.x;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  for (int s = threadIdx.x; s < TI_CAP; s += blockDim.x) { table[s] = TI_EMPTY; vals[s] = INFINITY; }
  if (threadIdx.x == 0) { n_unique = 0; overflow = 0; }
  __syncthreads();
  if (FEAT && frame_nonempty != nullptr) {
    // side output for rgnn_split_by_degree_frames: how many nodes of this frame have a non-zero degree
    __shared__ int nz_total;
    if (threadIdx.x == 0) nz_total = 0;
    __syncthreads();
    int nz = 0;
    for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) nz += degree[i] > 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o, 64);
    if ((threadIdx.x & 63) == 0 && nz) atomicAdd(&nz_total, nz);
    __syncthreads();
    if (threadIdx.x == 0) frame_nonempty[f] = nz_total;
  }
  // (four points per thread and round, their loads issued together: one block walks the whole frame, and a 100 000-point
  //  frame at one load latency per 1 024 points was 186 us)
  for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += 4 * (int64_t)blockDim.x) {
   double tv[4];
#pragma unroll
   for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * (int64_t)blockDim.x; tv[u] = ts[i < end ? i : end - 1]; }
#pragma unroll
   for (int u = 0; u < 4; u++) {
    if (i0 + u * (int64_t)blockDim.x >= end) break;
    const unsigned long long key = ts_key(tv[u]);
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 52) & (TI_CAP - 1);
    int probes = 0;
    for (;;) {
      // (a plain look first: a frame holds few distinct timestamps, so after the first few points every probe finds its key
      //  already there -- without this all 100 000 points of a one-frame cloud queue up on the same LDS atomics: 188 us)
      if (((volatile unsigned long long*)table)[h] == key) break;
      const unsigned long long old = atomicCAS(&table[h], TI_EMPTY, key);
      if (old == TI_EMPTY) {
        if (atomicAdd(&n_unique, 1) >= TI_MAX_UNIQUE) overflow = 1;
        break;
      }
      if (old == key) break;
      h = (h + 1) & (TI_CAP - 1);
      if (++probes >= TI_CAP) { overflow = 1; break; }
    }
    if (overflow) break;
   }
   if (overflow) break;
  }
  __syncthreads();
  if (overflow) {
    if (threadIdx.x == 0) atomicOr(status, RGNN_STATUS_TIME_INDEX_OVERFLOW);
    return;
  }
  // compact: every occupied slot contributes its value (order irrelevant, sorted next)
  __syncthreads();
  if (threadIdx.x == 0) n_unique = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < TI_CAP; s += blockDim.x) {
    const unsigned long long k = table[s];
    if (k != TI_EMPTY) vals[atomicAdd(&n_unique, 1)] = __longlong_as_double((long long)k);
  }
  __syncthreads();
  const int U = n_unique;
  int P = 1;
  while (P < U) P <<= 1;
  // bitonic sort of vals[0..P) ascending (+inf padding beyond U)
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int s = threadIdx.x; s < P; s += blockDim.x) {
        const int partner = s ^ j;
        if (partner > s) {
          const double a = vals[s], b = vals[partner];
          const bool up = ((s & k) == 0);
          if ((a > b) == up) { vals[s] = b; vals[partner] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int64_t i0 = beg + threadIdx.x; i0 < end; i0 += 4 * (int64_t)blockDim.x) {
    double tv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t i = i0 + u * (int64_t)blockDim.x; tv[u] = ts[i < end ? i : end - 1]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = i0 + u * (int64_t)blockDim.x;
      if (i >= end) break;
      const double t = tv[u];
      int lo = 0, hi = U;  // first position with vals[pos] >= t
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (vals[mid] < t) lo = mid + 1; else hi = mid;
      }
      if (FEAT) node_feature_row<OutT>(X, V, rcs, (double)lo, degree, i, codes, width, feat);
      else out[i] = (double)lo;
    }
  }
}

// ---- time index of LARGE frames (r05): the frame's points spread over the chip --------------------------------------------------
// One block per frame walks a 100 000-point cloud alone (160 us of the C5 step).  Here: (1) every point puts its timestamp into the
// frame's hash set in GLOBAL memory -- after the first few points a probe finds its key with a plain load (an L2 hit), so the
// atomics are a handful per distinct value --, (2) one block per frame compacts and sorts the <= 3 072 distinct values, (3) every
// point looks its rank up (binary search in 24 KB that stay in L2).  Same keys (np.unique folds -0.0 into 0.0), same ranks.
struct TiBigWs { unsigned long long* table; double* vals; int32_t* n_unique; };

__device__ __forceinline__ int ti_find_frame(const int64_t* __restrict__ frame_ptr, int n_frames, int64_t i) {
  int lo = 0, hi = n_frames;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (frame_ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_ti_insert(const double* __restrict__ ts, const int64_t* __restrict__ frame_ptr, int n_frames, int64_t n,
                                                  unsigned long long* __restrict__ table, int32_t* __restrict__ n_unique,
                                                  int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = ti_find_frame(frame_ptr, n_frames, i);
  unsigned long long* tab = table + (int64_t)f * TI_CAP;
  const unsigned long long key = ts_key(ts[i]);
  unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 52) & (TI_CAP - 1);
  for (int probes = 0; probes < TI_CAP; probes++) {
    if (__hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key) return;
    const unsigned long long old = atomicCAS(tab + h, TI_EMPTY, key);
    if (old == TI_EMPTY) {
      if (atomicAdd(n_unique + f, 1) >= TI_MAX_UNIQUE) atomicOr(status, RGNN_STATUS_TIME_INDEX_OVERFLOW);
      return;
    }
    if (old == key) return;
    h = (h + 1) & (TI_CAP - 1);
  }
  atomicOr(status, RGNN_STATUS_TIME_INDEX_OVERFLOW);
}

__global__ __launch_bounds__(1024) void k_ti_sort(const unsigned long long* __restrict__ table, double* __restrict__ vals_out,
                                                 int32_t* __restrict__ n_unique) {
  __shared__ double vals[TI_CAP];
  __shared__ int cnt;
  const int f = blockIdx.x;
  const unsigned long long* tab = table + (int64_t)f * TI_CAP;
  for (int s = threadIdx.x; s < TI_CAP; s += blockDim.x) vals[s] = INFINITY;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < TI_CAP; s += blockDim.x) {
    const unsigned long long k = tab[s];
    if (k != TI_EMPTY) vals[atomicAdd(&cnt, 1)] = __longlong_as_double((long long)k);
  }
  __syncthreads();
  const int U = cnt;
  int P = 1;
  while (P < U) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int s = threadIdx.x; s < P; s += blockDim.x) {
        const int partner = s ^ j;
        if (partner > s) {
          const double a = vals[s], b = vals[partner];
          const bool up = ((s & k) == 0);
          if ((a > b) == up) { vals[s] = b; vals[partner] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int s = threadIdx.x; s < U; s += blockDim.x) vals_out[(int64_t)f * TI_CAP + s] = vals[s];
  if (threadIdx.x == 0) n_unique[f] = U;
}

__global__ __launch_bounds__(256) void k_ti_rank(const double* __restrict__ ts, const int64_t* __restrict__ frame_ptr, int n_frames, int64_t n,
                                                const double* __restrict__ vals, const int32_t* __restrict__ n_unique,
                                                double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = ti_find_frame(frame_ptr, n_frames, i);
  const double* v = vals + (int64_t)f * TI_CAP;
  double t = ts[i];
  if (t == 0.0) t = 0.0;
  int lo = 0, hi = min(n_unique[f], TI_MAX_UNIQUE);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < t) lo = mid + 1; else hi = mid;
  }
  out[i] = (double)lo;
}

int edge_width(const int32_t* codes, int n) {
  int w = 0;
  for (int i = 0; i < n; i++) {
    switch (codes[i]) {
      case RGNN_EF_POINT_PAIR: w += 4; break;
      case RGNN_EF_SPATIAL_DISTANCE:
      case RGNN_EF_VELOCITY_DISTANCE: w += 1; break;
      case RGNN_EF_RELATIVE_POSITION:
      case RGNN_EF_RELATIVE_VELOCITY: w += 2; break;
      default: return -1;
    }
  }
  return w;
}

int node_width(const int32_t* codes, int n) {
  int w = 0;
  for (int i = 0; i < n; i++) {
    switch (codes[i]) {
      case RGNN_NF_RCS:
      case RGNN_NF_TIME_INDEX:
      case RGNN_NF_DEGREE:
      case RGNN_NF_VELOCITY_LENGTH: w += 1; break;
      case RGNN_NF_VELOCITY_VECTOR:
      case RGNN_NF_SPATIAL_COORDINATES: w += 2; break;
      default: return -1;
    }
  }
  return w;
}

}  // namespace

extern "C" int rgnn_edge_features(const double* X, const double* V, const int64_t* edge_index, int64_t n_edges,
                                  const int32_t* codes, int32_t n_codes, int32_t undirected, void* out,
                                  int32_t out_is_f64, int32_t* status, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_codes >= 0 && n_codes <= RGNN_MAX_FEATURE_CODES && (n_codes == 0 || codes), "bad feature code list");
  const int width = edge_width(codes, n_codes);
  if (width < 0) {
    rgnn_set_error("Invalid feature specified");  // graph.py:219-220
    return RGNN_ERR_INVALID_ARGUMENT;
  }
  if (n_edges == 0 || width == 0) return RGNN_OK;
  RGNN_CHECK_ARG(X && V && edge_index && out && status, "null pointers");
  Codes c;
  c.n = n_codes;
  for (int i = 0; i < n_codes; i++) c.c[i] = codes[i];
  hipStream_t s = (hipStream_t)stream;
  if (n_codes == 1 && codes[0] == RGNN_EF_RELATIVE_POSITION) {
    if (out_is_f64)
      hipLaunchKernelGGL(k_edge_relative_position<double>, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, X, edge_index,
                         n_edges, undirected, (double*)out);
    else
      hipLaunchKernelGGL(k_edge_relative_position<float>, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, X, edge_index,
                         n_edges, undirected, (float*)out);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  if (out_is_f64)
    hipLaunchKernelGGL(k_edge_features<double>, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, X, V, edge_index,
                       n_edges, c, width, undirected, (double*)out, status);
  else
    hipLaunchKernelGGL(k_edge_features<float>, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, X, V, edge_index,
                       n_edges, c, width, undirected, (float*)out, status);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// Edge features of the REVERSED edges at the rows of a list: out[s] = features(E[1][e], E[0][e]), e = reversed_of[s].  For a symmetric
// edge list whose CSR by target was built without the twin search (rgnn_csr_by_target_symmetric_own: own_edge[slot] = the out-edge
// (t -> i) at the slot of the in-edge (i -> t)) this IS the attribute list in target order -- the same arithmetic on the same end
// points as the twin's own row, so bit-identical to gathering the twin's row -- without looking the twin up (a binary search per edge:
// 179 us and 1.1 GB of reads on the 100 000-point cloud).
extern "C" int rgnn_edge_features_reversed(const double* X, const double* V, const int64_t* edge_index, int64_t n_edges,
                                           const int32_t* reversed_of, int64_t n_rows, const int32_t* codes, int32_t n_codes,
                                           int32_t undirected, void* out, int32_t out_is_f64, int32_t* status, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_codes >= 0 && n_codes <= RGNN_MAX_FEATURE_CODES && (n_codes == 0 || codes), "bad feature code list");
  const int width = edge_width(codes, n_codes);
  if (width < 0) {
    rgnn_set_error("Invalid feature specified");  // graph.py:219-220
    return RGNN_ERR_INVALID_ARGUMENT;
  }
  if (n_rows == 0 || width == 0) return RGNN_OK;
  RGNN_CHECK_ARG(X && V && edge_index && out && status && reversed_of && n_edges > 0, "null pointers");
  Codes c;
  c.n = n_codes;
  for (int i = 0; i < n_codes; i++) c.c[i] = codes[i];
  hipStream_t s = (hipStream_t)stream;
  if (out_is_f64)
    hipLaunchKernelGGL(k_edge_features<double>, dim3(rgnn_blocks(n_rows, 256)), dim3(256), 0, s, X, V, edge_index, n_edges, c, width,
                       undirected, (double*)out, status, reversed_of, n_rows);
  else
    hipLaunchKernelGGL(k_edge_features<float>, dim3(rgnn_blocks(n_rows, 256)), dim3(256), 0, s, X, V, edge_index, n_edges, c, width,
                       undirected, (float*)out, status, reversed_of, n_rows);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_node_features(const double* X, const double* V, const double* rcs, const double* time_index,
                                  const int32_t* degree, int64_t n, const int32_t* codes, int32_t n_codes, void* out,
                                  int32_t out_is_f64, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_codes >= 0 && n_codes <= RGNN_MAX_FEATURE_CODES && (n_codes == 0 || codes), "bad feature code list");
  const int width = node_width(codes, n_codes);
  if (width < 0) {
    rgnn_set_error("Invalid node feature specified");
    return RGNN_ERR_INVALID_ARGUMENT;
  }
  if (n == 0 || width == 0) return RGNN_OK;
  for (int i = 0; i < n_codes; i++) {
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_RCS || rcs, "rcs requested but NULL");
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_TIME_INDEX || time_index, "time_index requested but NULL");
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_DEGREE || degree, "degree requested but NULL");
    RGNN_CHECK_ARG((codes[i] != RGNN_NF_VELOCITY_LENGTH && codes[i] != RGNN_NF_VELOCITY_VECTOR) || V, "V is NULL");
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_SPATIAL_COORDINATES || X, "X is NULL");
  }
  RGNN_CHECK_ARG(out, "null out");
  Codes c;
  c.n = n_codes;
  for (int i = 0; i < n_codes; i++) c.c[i] = codes[i];
  hipStream_t s = (hipStream_t)stream;
  if (out_is_f64)
    hipLaunchKernelGGL(k_node_features<double>, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, X, V, rcs, time_index, degree,
                       n, c, width, (double*)out);
  else
    hipLaunchKernelGGL(k_node_features<float>, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, X, V, rcs, time_index, degree,
                       n, c, width, (float*)out);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

namespace {
// Nodes with / without edges as two lists in ascending node order (what the row-subset dense launches of a conv layer walk),
// from the degrees of a SYMMETRIC graph (radius graphs: in-degree = out-degree = row length of the search) and the per-frame
// counts the node-feature kernel left behind: one block per frame adds up the counts of the frames before it and compacts its
// own nodes -- one launch instead of flags + two scan kernels + compaction.
__global__ __launch_bounds__(1024) void k_split_frames(const int32_t* __restrict__ degree, const int64_t* __restrict__ frame_ptr,
                                                      int n_frames, const int32_t* __restrict__ frame_nonempty,
                                                      int32_t* __restrict__ list_e, int64_t* __restrict__ count_e,
                                                      int32_t* __restrict__ slot, int32_t* __restrict__ list_ne,
                                                      int64_t* __restrict__ count_ne) {
  __shared__ int wsum[16];
  __shared__ int before_s;
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  int part = 0;
  for (int g = t; g < f; g += 1024) part += frame_nonempty[g];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) wsum[w] = part;
  __syncthreads();
  if (t == 0) { int b = 0; for (int i = 0; i < 16; i++) b += wsum[i]; before_s = b; }
  __syncthreads();
  int64_t ne_run = before_s;                            // non-empty nodes before the current strip
  for (int64_t base = beg; base < end; base += 1024) {
    const int64_t i = base + t;
    const int nz = (i < end && degree[i] > 0) ? 1 : 0;
    int inc = nz;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int v = wsum[k]; if (k < w) add += v; tot += v; }
    if (i < end) {
      const int64_t ne_pos = ne_run + add + inc - nz;    // non-empty nodes with a smaller id
      if (nz) { list_ne[ne_pos] = (int32_t)i; if (slot) slot[i] = -1; }
      else { list_e[i - ne_pos] = (int32_t)i; if (slot) slot[i] = (int32_t)(i - ne_pos); }
    }
    ne_run += tot;
  }
  if (f == n_frames - 1 && t == 0) { *count_ne = ne_run; *count_e = end - ne_run; }
}
// An ascending node list, cut at the segment (frame) borders and padded with -1 so that every segment starts a tile of `pad` rows
// of its own (rgnn_linear_args.a1_panel_segment).  One block per segment: where the segment's entries lie in the list (binary
// searches -- also for every earlier segment, whose padded lengths give this one's offset), copy, pad, name the tiles.
__device__ __forceinline__ int64_t lower_bound_i32(const int32_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
struct PadList { const int32_t* list; const int64_t* count; int32_t* out; int64_t* out_total; int32_t* tile_segment; int32_t* stat_start; };
__global__ __launch_bounds__(256) void k_pad_list_segments(PadList la, PadList lb, const int64_t* __restrict__ seg_ptr, int n_seg,
                                                          int pad, int stat_rows) {
  __shared__ long long red[256];
  const int f = blockIdx.x, t = threadIdx.x;
  const PadList L = blockIdx.y ? lb : la;             // (two lists in one launch: the targets with and without edges of a batch)
  const int32_t* __restrict__ list = L.list;
  const int64_t* __restrict__ count = L.count;
  int32_t* __restrict__ out = L.out;
  int64_t* __restrict__ out_total = L.out_total;
  int32_t* __restrict__ tile_segment = L.tile_segment;
  int32_t* __restrict__ stat_start = L.stat_start;
  const int64_t n = *count;
  long long before = 0;                                 // padded length of the segments in front of this one
  for (int g = t; g < f; g += 256) {
    const int64_t c = lower_bound_i32(list, n, seg_ptr[g + 1]) - lower_bound_i32(list, n, seg_ptr[g]);
    before += (c + pad - 1) / pad * pad;
  }
  red[t] = before;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  const int64_t off = red[0];
  const int64_t lo = lower_bound_i32(list, n, seg_ptr[f]), hi = lower_bound_i32(list, n, seg_ptr[f + 1]);
  const int64_t len = hi - lo, plen = (len + pad - 1) / pad * pad;
  for (int64_t i = t; i < plen; i += 256) out[off + i] = (i < len) ? list[lo + i] : -1;
  for (int64_t i = t; i < plen / pad; i += 256) tile_segment[off / pad + i] = f;
  if (t == 0) {
    stat_start[f] = (int32_t)(off / stat_rows);
    if (f == n_seg - 1) { stat_start[n_seg] = (int32_t)((off + plen) / stat_rows); *out_total = off + plen; }
  }
}
}  // namespace

extern "C" int rgnn_pad_list_by_segment(const int32_t* list, const int64_t* count, const int64_t* seg_ptr, int64_t n_seg,
                                        int32_t* out_list, int64_t* out_count, int32_t* tile_segment, int32_t* stat_panel_start,
                                        rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_seg >= 1 && n_seg < ((int64_t)1 << 24), "bad segment count");
  RGNN_CHECK_ARG(list && count && seg_ptr && out_list && out_count && tile_segment && stat_panel_start, "null pointers");
  const PadList la{list, count, out_list, out_count, tile_segment, stat_panel_start};
  hipLaunchKernelGGL(k_pad_list_segments, dim3((unsigned)n_seg, 1), dim3(256), 0, (hipStream_t)stream, la, la, seg_ptr, (int)n_seg, 256,
                     RGNN_STAT_PANEL_ROWS);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_pad_list_pair_by_segment(const int32_t* list_a, const int64_t* count_a, const int32_t* list_b, const int64_t* count_b,
                                             const int64_t* seg_ptr, int64_t n_seg, int32_t* out_list_a, int64_t* out_count_a,
                                             int32_t* tile_segment_a, int32_t* stat_panel_start_a, int32_t* out_list_b,
                                             int64_t* out_count_b, int32_t* tile_segment_b, int32_t* stat_panel_start_b,
                                             rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_seg >= 1 && n_seg < ((int64_t)1 << 24), "bad segment count");
  RGNN_CHECK_ARG(list_a && count_a && list_b && count_b && seg_ptr && out_list_a && out_count_a && tile_segment_a && stat_panel_start_a &&
                 out_list_b && out_count_b && tile_segment_b && stat_panel_start_b, "null pointers");
  const PadList la{list_a, count_a, out_list_a, out_count_a, tile_segment_a, stat_panel_start_a};
  const PadList lb{list_b, count_b, out_list_b, out_count_b, tile_segment_b, stat_panel_start_b};
  hipLaunchKernelGGL(k_pad_list_segments, dim3((unsigned)n_seg, 2), dim3(256), 0, (hipStream_t)stream, la, lb, seg_ptr, (int)n_seg, 256,
                     RGNN_STAT_PANEL_ROWS);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_split_by_degree_frames(const int32_t* degree, const int64_t* frame_ptr, int64_t n_frames,
                                           const int32_t* frame_nonempty, int32_t* list_empty, int64_t* count_empty,
                                           int32_t* slot_of_node, int32_t* list_nonempty, int64_t* count_nonempty,
                                           rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_frames >= 1 && degree && frame_ptr && frame_nonempty && list_empty && count_empty && list_nonempty && count_nonempty,
                 "null pointers");
  hipLaunchKernelGGL(k_split_frames, dim3((unsigned)n_frames), dim3(1024), 0, (hipStream_t)stream, degree, frame_ptr, (int)n_frames,
                     frame_nonempty, list_empty, count_empty, slot_of_node, list_nonempty, count_nonempty);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// What the host reads back of a radius graph before it sizes the edge arrays, in ONE launch and one 8-byte copy: out[0] = rowptr[n]
// (the edge count), out[1] = the number of edges in rows longer than `threshold` (a symmetric graph's in-degrees are its row lengths:
// the share of the edges in targets too large for a stream of the window kernel picks the form of the max aggregation).  One block,
// 16-byte loads, four in flight per thread.  (The torch form was five launches: compare, multiply, reduce, two copies into a stack.)
namespace {
__global__ __launch_bounds__(1024) void k_radius_counts(const int32_t* __restrict__ deg, int64_t n, const int32_t* __restrict__ rowptr,
                                                       int threshold, int32_t* __restrict__ out) {
  __shared__ long long red[16];
  long long s = 0;
  // 16-byte loads, four of them in flight per thread and trip (one block: a dependent 4-byte load per trip was 188 latencies = 57 us)
  const int64_t n4 = ((((uintptr_t)deg) & 15) == 0) ? n / 4 : 0;
  const int4* d4 = (const int4*)deg;
  for (int64_t i = threadIdx.x; i < n4; i += 4 * 1024) {
    int4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = (i + u * 1024 < n4) ? d4[i + u * 1024] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++)
      s += (long long)(v[u].x > threshold ? v[u].x : 0) + (v[u].y > threshold ? v[u].y : 0) + (v[u].z > threshold ? v[u].z : 0) +
           (v[u].w > threshold ? v[u].w : 0);
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 1024) {
    const int d = deg[i];
    s += d > threshold ? d : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < 16; w++) t += red[w];
    out[0] = rowptr[n];
    out[1] = (int32_t)(t > 0x7fffffffLL ? 0x7fffffffLL : t);
  }
}
}  // namespace

extern "C" int rgnn_radius_counts(const int32_t* deg, int64_t n, const int32_t* rowptr, int32_t threshold, int32_t* out2,
                                  rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && rowptr && out2 && (n == 0 || deg), "null pointers");
  hipLaunchKernelGGL(k_radius_counts, dim3(1), dim3(1024), 0, (hipStream_t)stream, deg, n, rowptr, (int)threshold, out2);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_time_index(const double* timestamp, const int64_t* frame_ptr, int64_t n_frames, double* time_index,
                               int32_t* status, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_frames >= 0, "negative n_frames");
  if (n_frames == 0) return RGNN_OK;
  RGNN_CHECK_ARG(timestamp && frame_ptr && time_index && status, "null pointers");
  hipLaunchKernelGGL((k_time_index<false, float>), dim3((unsigned)n_frames), dim3(1024), 0, (hipStream_t)stream, timestamp, frame_ptr,
                     time_index, status, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr,
                     (const int32_t*)nullptr, Codes{}, 0, (float*)nullptr);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int64_t rgnn_time_index_ws_bytes(int64_t n_frames) { return n_frames * (int64_t)(TI_CAP * 16 + 64); }

extern "C" int rgnn_time_index_ws(const double* timestamp, const int64_t* frame_ptr, int64_t n_frames, int64_t n, double* time_index,
                                  int32_t* status, void* ws, int64_t ws_bytes, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_frames >= 0 && n >= 0, "negative sizes");
  if (n_frames == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(timestamp && frame_ptr && time_index && status && ws, "null pointers");
  RGNN_CHECK_ARG(ws_bytes >= rgnn_time_index_ws_bytes(n_frames) && (((uintptr_t)ws) & 15) == 0, "workspace too small or misaligned");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* table = (unsigned long long*)ws;
  double* vals = (double*)(table + n_frames * TI_CAP);
  int32_t* n_unique = (int32_t*)(vals + n_frames * TI_CAP);
  hipMemsetAsync(table, 0xff, (size_t)n_frames * TI_CAP * 8, s);
  hipMemsetAsync(n_unique, 0, (size_t)n_frames * 4, s);
  hipLaunchKernelGGL(k_ti_insert, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, timestamp, frame_ptr, (int)n_frames, n, table, n_unique, status);
  hipLaunchKernelGGL(k_ti_sort, dim3((unsigned)n_frames), dim3(1024), 0, s, (const unsigned long long*)table, vals, n_unique);
  hipLaunchKernelGGL(k_ti_rank, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, timestamp, frame_ptr, (int)n_frames, n, (const double*)vals,
                     (const int32_t*)n_unique, time_index);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_node_features_time_index(const double* X, const double* V, const double* rcs, const double* timestamp,
                                             const int64_t* frame_ptr, int64_t n_frames, const int32_t* degree, int64_t n,
                                             const int32_t* codes, int32_t n_codes, void* out, int32_t out_is_f64,
                                             int32_t* status, int32_t* frame_nonempty, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(frame_nonempty == nullptr || degree != nullptr, "frame_nonempty needs the degrees");
  RGNN_CHECK_ARG(n_codes >= 0 && n_codes <= RGNN_MAX_FEATURE_CODES && (n_codes == 0 || codes), "bad feature code list");
  const int width = node_width(codes, n_codes);
  if (width < 0) {
    rgnn_set_error("Invalid node feature specified");
    return RGNN_ERR_INVALID_ARGUMENT;
  }
  if (n == 0 || width == 0 || n_frames == 0) return RGNN_OK;
  RGNN_CHECK_ARG(timestamp && frame_ptr && out && status, "null pointers");
  for (int i = 0; i < n_codes; i++) {
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_RCS || rcs, "rcs requested but NULL");
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_DEGREE || degree, "degree requested but NULL");
    RGNN_CHECK_ARG((codes[i] != RGNN_NF_VELOCITY_LENGTH && codes[i] != RGNN_NF_VELOCITY_VECTOR) || V, "velocity requested but NULL");
    RGNN_CHECK_ARG(codes[i] != RGNN_NF_SPATIAL_COORDINATES || X, "coordinates requested but NULL");
  }
  Codes c;
  c.n = n_codes;
  for (int i = 0; i < n_codes; i++) c.c[i] = codes[i];
  if (out_is_f64)
    hipLaunchKernelGGL((k_time_index<true, double>), dim3((unsigned)n_frames), dim3(1024), 0, (hipStream_t)stream, timestamp, frame_ptr,
                       (double*)nullptr, status, X, V, rcs, degree, c, width, (double*)out, frame_nonempty);
  else
    hipLaunchKernelGGL((k_time_index<true, float>), dim3((unsigned)n_frames), dim3(1024), 0, (hipStream_t)stream, timestamp, frame_ptr,
                       (double*)nullptr, status, X, V, rcs, degree, c, width, (float*)out, frame_nonempty);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
