// Backward kernels of the hot path (the reference trains through these layers with
// loss.backward(), gnn/trainer.py:176-231).  What autograd would derive for the reference's op-by-op forward:
//
//   * ReLU fused into a dense-layer epilogue:      dx = (y > 0) ? dy : 0
//   * train-mode BatchNorm1d (+ fused ReLU):       dx = gamma rstd (g - mean(g) - xhat mean(g xhat)),  g = relu'(y) dy
//                                                  dgamma = sum g xhat,  dbeta = sum g
//   * fused gather / mat-vec / segmented reduce:   M[t] = aggr_{e -> t} (Q[s_e] + W_e a_e)
//       max : the gradient of channel c of target t goes to the FIRST edge that attains the maximum (torch-scatter's
//             arg_out convention), i.e. dQ[s_e*, c] += dM[t, c],  da[e*, :] += dM[t, c] W_e[c, :],  dW_e[c, :] += dM[t, c] a_e*
//       mean / add : every edge of the segment receives dM[t] (/ deg)
//
// The dense-layer gradients themselves (dX = dY W, dW = dY^T X) are plain GEMMs: dX runs on rgnn_linear_fwd with the
// transposed weight, dW on the BLAS behind torch.mm (radargnn_amd/gnn/autograd.py).
#include "common.h"
#include <math.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                 float* __restrict__ dx, int64_t count) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < count) {
    const float4 g = *(const float4*)(dy + i);
    const float4 v = *(const float4*)(y + i);
    *(float4*)(dx + i) = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
  } else {
    for (int64_t j = i; j < count; j++) dx[j] = y[j] > 0.f ? dy[j] : 0.f;
  }
}

// per 128-row panel: column sums of g and of g * h, g = (y > 0 or no mask) ? dy : 0  -- same partial layout as the
// forward column statistics ([panels, 2, n]), so rgnn_column_stats consumers can reduce both
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                     int64_t ldy, const float* __restrict__ h, int64_t ldh, int64_t m,
                                                     int n, float* __restrict__ partial) {
  __shared__ float red[2][4][64];
  const int panel = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int g4 = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < n) {
    const int64_t r0 = (int64_t)panel * 128 + g4 * 32;
    for (int i = 0; i < 32; i++) {
      const int64_t r = r0 + i;
      if (r < m) {
        float g = dy[r * lddy + c];
        if (y && !(y[r * ldy + c] > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * h[r * ldh + c];
      }
    }
  }
  red[0][g4][threadIdx.x & 63] = s1;
  red[1][g4][threadIdx.x & 63] = s2;
  __syncthreads();
  if (g4 == 0 && c < n) {
    const int l = threadIdx.x;
    partial[((int64_t)panel * 2 + 0) * n + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    partial[((int64_t)panel * 2 + 1) * n + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
  }
}

// dx = A[c] g + B[c] h + C[c]   (coef = [A | B | C], each n floats)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                     int64_t ldy, const float* __restrict__ h, int64_t ldh,
                                                     const float* __restrict__ coef, int64_t m, int n,
                                                     float* __restrict__ dx, int64_t lddx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * n) return;
  const int64_t r = idx / n;
  const int c = (int)(idx - r * n);
  float g = dy[r * lddy + c];
  if (y && !(y[r * ldy + c] > 0.f)) g = 0.f;
  dx[r * lddx + c] = coef[c] * g + coef[n + c] * h[r * ldh + c] + coef[2 * n + c];
}

// ---------------------------------------------------------------------------------------------- message passing
// sum over the 64 lanes of DEP per-lane values in DEP/2 + DEP/4 + ... + 1 + (6 - log2 DEP) shuffles: the first
// log2(DEP) steps halve the vector while exchanging with the partner lane, after which lane l holds the 2^k-lane
// partial of component idx(l); the remaining steps are a plain butterfly.  Returns the total of component
// `comp` (set per lane); every component appears in 64 / DEP lanes.
template <int DEP>
__device__ __forceinline__ float wave_reduce_vec(float (&v)[DEP], int lane, int& comp) {
  comp = 0;
  int width = DEP;
#pragma unroll
  for (int bit = 0; (1 << bit) < DEP; bit++) {
    const int half = width >> 1;
    const bool up = (lane >> bit) & 1;
#pragma unroll
    for (int i = 0; i < DEP / 2; i++) {
      if (i < half) {
        const float send = up ? v[i] : v[i + half];
        const float keep = up ? v[i + half] : v[i];
        v[i] = keep + __shfl_xor(send, 1 << bit, 64);
      }
    }
    if (up) comp += half;
    width = half;
  }
  float s = v[0];
#pragma unroll
  for (int off = DEP; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
  return s;
}

// One wave per CSR segment (persistent, strided), lanes across channel groups of 4 (NCH groups per lane).
// MODE 0 max, 1 mean, 2 add.
// row[cb .. cb+3] as a float4; VEC: one 16-byte load, else element-wise with the channel bound (d % 4 != 0 or unaligned rows)
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ row, int cb, int d) {
  if (VEC) return *(const float4*)(row + cb);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cb + 0 < d) v.x = row[cb + 0];
  if (cb + 1 < d) v.y = row[cb + 1];
  if (cb + 2 < d) v.z = row[cb + 2];
  if (cb + 3 < d) v.w = row[cb + 3];
  return v;
}

template <int NCH, int DEP, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_mpnn_bwd(const float* __restrict__ dM, int64_t lddm, const float* __restrict__ Q,
                                                 int64_t ldq, const float* __restrict__ We, int64_t ldwe,
                                                 const float* __restrict__ ea, int de, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ src, const int32_t* __restrict__ node_order,
                                                 int64_t n, int d, float* __restrict__ dQ, int64_t lddq,
                                                 float* __restrict__ dea, float* __restrict__ dWe) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int groups = (d + 3) >> 2;
  int cb[NCH];
  bool act[NCH];
  float w[NCH][4][DEP], dw[NCH][4][DEP];
#pragma unroll
  for (int q = 0; q < NCH; q++) {
    const int cg = lane + 64 * q;
    act[q] = cg < groups;
    cb[q] = act[q] ? cg * 4 : 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < DEP; j++) {
        w[q][k][j] = (act[q] && j < de && cb[q] + k < d) ? We[(int64_t)(cb[q] + k) * ldwe + j] : 0.f;
        dw[q][k][j] = 0.f;
      }
  }
  for (int64_t p = wave; p < n; p += n_waves) {
    const int r0 = rowptr[p], r1 = rowptr[p + 1];
    if (r1 == r0) continue;
    const int64_t t = node_order ? (int64_t)node_order[p] : p;
    float4 g[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++)
      g[q] = act[q] ? load4<VEC>(dM + t * lddm, cb[q], d) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0) {
      float4 best[NCH];
      int arg[NCH][4];
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        best[q] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        arg[q][0] = arg[q][1] = arg[q][2] = arg[q][3] = -1;
      }
      for (int e = r0; e < r1; e++) {
        const int64_t s = src[e];
        float z[DEP];
#pragma unroll
        for (int j = 0; j < DEP; j++) z[j] = (j < de) ? ea[(int64_t)e * de + j] : 0.f;
#pragma unroll
        for (int q = 0; q < NCH; q++) {
          if (!act[q]) continue;
          const float4 qv = load4<VEC>(Q + s * ldq, cb[q], d);
          float v[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < DEP; j++) a += w[q][k][j] * z[j];
            v[k] += a;
          }
          if (v[0] > best[q].x) { best[q].x = v[0]; arg[q][0] = e; }
          if (v[1] > best[q].y) { best[q].y = v[1]; arg[q][1] = e; }
          if (v[2] > best[q].z) { best[q].z = v[2]; arg[q][2] = e; }
          if (v[3] > best[q].w) { best[q].w = v[3]; arg[q][3] = e; }
        }
      }
      for (int e = r0; e < r1; e++) {
        const int64_t s = src[e];
        float z[DEP], dz[DEP];
#pragma unroll
        for (int j = 0; j < DEP; j++) { z[j] = (j < de) ? ea[(int64_t)e * de + j] : 0.f; dz[j] = 0.f; }
#pragma unroll
        for (int q = 0; q < NCH; q++) {
          const float gv[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (act[q] && arg[q][k] == e && cb[q] + k < d) {
              const float gg = gv[k];
              unsafeAtomicAdd(dQ + s * lddq + cb[q] + k, gg);
#pragma unroll
              for (int j = 0; j < DEP; j++) { dz[j] += gg * w[q][k][j]; dw[q][k][j] += gg * z[j]; }
            }
          }
        }
        int comp;
        const float tot = wave_reduce_vec<DEP>(dz, lane, comp);
        if (lane < DEP && comp < de) dea[(int64_t)e * de + comp] = tot;
      }
    } else {
      const float sc = (MODE == 1) ? 1.f / (float)(r1 - r0) : 1.f;
      float dz[DEP], zs[DEP];
#pragma unroll
      for (int j = 0; j < DEP; j++) { dz[j] = 0.f; zs[j] = 0.f; }
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        g[q].x *= sc; g[q].y *= sc; g[q].z *= sc; g[q].w *= sc;
        const float gv[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int j = 0; j < DEP; j++) dz[j] += gv[k] * w[q][k][j];
      }
      int comp;
      const float tot = wave_reduce_vec<DEP>(dz, lane, comp);
      for (int e = r0; e < r1; e++) {
        const int64_t s = src[e];
#pragma unroll
        for (int j = 0; j < DEP; j++) zs[j] += (j < de) ? ea[(int64_t)e * de + j] : 0.f;
#pragma unroll
        for (int q = 0; q < NCH; q++) {
          if (!act[q]) continue;
          if (cb[q] + 0 < d) unsafeAtomicAdd(dQ + s * lddq + cb[q] + 0, g[q].x);
          if (cb[q] + 1 < d) unsafeAtomicAdd(dQ + s * lddq + cb[q] + 1, g[q].y);
          if (cb[q] + 2 < d) unsafeAtomicAdd(dQ + s * lddq + cb[q] + 2, g[q].z);
          if (cb[q] + 3 < d) unsafeAtomicAdd(dQ + s * lddq + cb[q] + 3, g[q].w);
        }
        if (lane < DEP && comp < de) dea[(int64_t)e * de + comp] = tot;
      }
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const float gv[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int j = 0; j < DEP; j++) dw[q][k][j] += gv[k] * zs[j];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NCH; q++) {
    if (!act[q]) continue;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < DEP; j++)
        if (j < de && cb[q] + k < d && dw[q][k][j] != 0.f) unsafeAtomicAdd(dWe + (int64_t)(cb[q] + k) * de + j, dw[q][k][j]);
  }
}

template <int NCH, int DEP, bool VEC>
void launch_bwd(int mode, dim3 g, dim3 b, hipStream_t s, const float* dM, int64_t lddm, const float* Q, int64_t ldq,
                const float* We, int64_t ldwe, const float* ea, int de, const int32_t* rowptr, const int32_t* src,
                const int32_t* node_order, int64_t n, int d, float* dQ, int64_t lddq, float* dea, float* dWe) {
  if (mode == RGNN_AGGR_MAX)
    hipLaunchKernelGGL((k_mpnn_bwd<NCH, DEP, 0, VEC>), g, b, 0, s, dM, lddm, Q, ldq, We, ldwe, ea, de, rowptr, src, node_order, n, d, dQ, lddq, dea, dWe);
  else if (mode == RGNN_AGGR_MEAN)
    hipLaunchKernelGGL((k_mpnn_bwd<NCH, DEP, 1, VEC>), g, b, 0, s, dM, lddm, Q, ldq, We, ldwe, ea, de, rowptr, src, node_order, n, d, dQ, lddq, dea, dWe);
  else
    hipLaunchKernelGGL((k_mpnn_bwd<NCH, DEP, 2, VEC>), g, b, 0, s, dM, lddm, Q, ldq, We, ldwe, ea, de, rowptr, src, node_order, n, d, dQ, lddq, dea, dWe);
}

}  // namespace

extern "C" int rgnn_relu_bwd(const float* dy, const float* y, float* dx, int64_t count, rgnn_stream_t stream) {
  if (count == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dy && y && dx, "null pointers");
  RGNN_CHECK_ARG((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15) == 0, "pointers must be 16-byte aligned");
  hipLaunchKernelGGL(k_relu_bwd, dim3(rgnn_blocks((count + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, count);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_bn_bwd_stats(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                                 int64_t m, int32_t n, float* partial, rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dy && h && partial, "null pointers");
  const unsigned panels = (unsigned)((m + 127) / 128);
  hipLaunchKernelGGL(k_bn_bwd_stats, dim3(panels, (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y,
                     ldy, h, ldh, m, n, partial);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_bn_bwd_apply(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                                 const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dy && h && coef && dx, "null pointers");
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(rgnn_blocks(m * n, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, h, ldh,
                     coef, m, n, dx, lddx);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_mpnn_aggregate_bwd(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We,
                                       int64_t ldwe, const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t,
                                       const int32_t* src_sorted, const int32_t* node_order, int64_t n, int32_t d,
                                       int32_t aggr, float* dQ, int64_t lddq, float* d_edge_attr, float* dWe,
                                       rgnn_stream_t stream) {
  if (n == 0 || d == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dM && Q && We && rowptr_t && src_sorted && dQ && dWe, "null pointers");
  RGNN_CHECK_ARG(de >= 0 && de <= 16, "edge attribute width must be <= 16");
  RGNN_CHECK_ARG(de == 0 || (edge_attr_sorted && d_edge_attr), "edge attribute pointers");
  RGNN_CHECK_ARG(aggr >= 0 && aggr <= 2, "unknown aggregation");
  RGNN_CHECK_ARG(d <= 1024, "message width must be <= 1024");
  const bool vec = d % 4 == 0 && lddm % 4 == 0 && ldq % 4 == 0 && (((uintptr_t)dM | (uintptr_t)Q) & 15) == 0;
  const int nch = ((d + 3) / 4 + 63) / 64;
  const dim3 b(256), g((unsigned)(n < 8192 ? (n + 3) / 4 : 2048));
  hipStream_t s = (hipStream_t)stream;
#define RGNN_BWD(NCH)                                                                                                   \
  do {                                                                                                                  \
    if (vec) RGNN_BWD2(NCH, true); else RGNN_BWD2(NCH, false);                                                          \
  } while (0)
#define RGNN_BWD2(NCH, V)                                                                                               \
  do {                                                                                                                  \
    if (de <= 4) launch_bwd<NCH, 4, V>(aggr, g, b, s, dM, lddm, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, dQ, lddq, d_edge_attr, dWe); \
    else if (de <= 8) launch_bwd<NCH, 8, V>(aggr, g, b, s, dM, lddm, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, dQ, lddq, d_edge_attr, dWe); \
    else launch_bwd<NCH, 16, V>(aggr, g, b, s, dM, lddm, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, dQ, lddq, d_edge_attr, dWe); \
  } while (0)
  switch (nch) {
    case 1: RGNN_BWD(1); break;
    case 2: RGNN_BWD(2); break;
    case 3: RGNN_BWD(3); break;
    default: RGNN_BWD(4); break;
  }
#undef RGNN_BWD
#undef RGNN_BWD2
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
