// Backward kernels of the hot path (the reference trains through these layers with
// loss.backward(), gnn/trainer.py:176-231).  What autograd would derive for the reference's op-by-op forward:
//
//   * ReLU fused into a dense-layer epilogue:      dx = (y > 0) ? dy : 0
//   * train-mode BatchNorm1d (+ fused ReLU):       dx = gamma rstd (g - mean(g) - xhat mean(g xhat)),  g = relu'(y) dy
//                                                  dgamma = sum g xhat,  dbeta = sum g
//   * fused gather / mat-vec / segmented reduce:   M[t] = aggr_{e -> t} (Q[s_e] + W_e a_e)
//       max : the gradient of channel c of target t goes to the FIRST edge that attains the maximum (torch-scatter's
//             arg_out convention): dQ[s_e*, c] += dM[t, c],  da[e*, :] += dM[t, c] W_e[c, :],  dW_e[c, :] += dM[t, c] a_e*
//       mean / add : every edge of the segment receives dM[t] (/ deg)
//     Three kernels and no atomics (deterministic): k_mpnn_bwd_arg finds the winning edge per (t, c) over the CSR by
//     target, k_mpnn_bwd_edge routes the gradient to the edge attributes and to per-wave partials of dW_e, and
//     k_mpnn_bwd_src computes dQ as a GATHER over the CSR by source (a scatter with 89 M float atomics measured 4x slower).
//
// The dense-layer gradients themselves (dX = dY W, dW = dY^T X) are plain GEMMs: dX runs on rgnn_linear_fwd with the
// transposed weight, dW on rgnn_wgrad (wgrad.hip) -- no BLAS anywhere in the training path (radargnn_amd/gnn/autograd.py).
#include "common.h"
#include <stdlib.h>
#include <math.h>

#ifndef RGNN_BWD_NT_STORE
#define RGNN_BWD_NT_STORE 1
#endif

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
// `table` (optional, instead of y): the BatchNorm's apply table {mean_hi, g, t} -- the ReLU mask is recomputed from h exactly as the
// forward pass computed y (k_scale_shift_act: fmaf(h - mean_hi, g, t)), so the [m, n] matrix y is not read again
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                     int64_t ldy, const float* __restrict__ table, const float* __restrict__ h,
                                                     int64_t ldh, int64_t m, int n, float* __restrict__ partial) {
  __shared__ float red[2][4][64];
  const int panel = blockIdx.x;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int g4 = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (c < n) {
    const int64_t r0 = (int64_t)panel * 128 + g4 * 32;
    float mu = 0.f, sc = 0.f, sh = 0.f;
    if (table) { mu = table[c]; sc = table[n + c]; sh = table[2 * n + c]; }
    for (int i = 0; i < 32; i++) {
      const int64_t r = r0 + i;
      if (r < m) {
        float g = dy[r * lddy + c];
        const float hv = h[r * ldh + c];
        if (table) { if (!(fmaf(hv - mu, sc, sh) > 0.f)) g = 0.f; }
        else if (y && !(y[r * ldy + c] > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * hv;
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

// dx = A[c] g + B[c] h + C[c]   (coef = [A | B | C], each n floats); VEC: four columns per thread (16-byte loads / stores).
// dx_absmax (optional): max |dx| for the f16x2 form of the dgrad launches that read dx -- one atomic per wave at the END of a
// grid-stride loop (an atomic per wave and element group, 1.4 M of them on 256 words, cost 450 us on a [192 000 x 464] matrix).
template <bool VEC>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                     int64_t ldy, const float* __restrict__ table, const float* __restrict__ h,
                                                     int64_t ldh, const float* __restrict__ coef, int64_t m, int n,
                                                     float* __restrict__ dx, int64_t lddx, float* __restrict__ dx_absmax) {
  const int gc = VEC ? (n >> 2) : n;                   // column groups per row
  const int64_t total = m * gc;
  float amax = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r; int c;
    if (total < ((int64_t)1 << 32)) { const unsigned q = (unsigned)idx / (unsigned)gc; r = q; c = (int)((unsigned)idx - q * (unsigned)gc); }
    else { r = idx / gc; c = (int)(idx - r * gc); }
    if (VEC) {
      c <<= 2;
      float4 g = *(const float4*)(dy + r * lddy + c);
      const float4 hh = *(const float4*)(h + r * ldh + c);
      if (table) {                                     // (the mask from h and the apply table: the bits of the forward pass's y)
        const float4 mu = *(const float4*)(table + c), sc = *(const float4*)(table + n + c), sh = *(const float4*)(table + 2 * n + c);
        if (!(fmaf(hh.x - mu.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
        if (!(fmaf(hh.y - mu.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
        if (!(fmaf(hh.z - mu.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
        if (!(fmaf(hh.w - mu.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
      } else if (y) {
        const float4 yy = *(const float4*)(y + r * ldy + c);
        if (!(yy.x > 0.f)) g.x = 0.f;
        if (!(yy.y > 0.f)) g.y = 0.f;
        if (!(yy.z > 0.f)) g.z = 0.f;
        if (!(yy.w > 0.f)) g.w = 0.f;
      }
      const float4 A = *(const float4*)(coef + c), B = *(const float4*)(coef + n + c), Cc = *(const float4*)(coef + 2 * n + c);
      float4 v;                                        // (the scalar form's order of operations: A g + B h, then + C)
      v.x = A.x * g.x + B.x * hh.x + Cc.x; v.y = A.y * g.y + B.y * hh.y + Cc.y;
      v.z = A.z * g.z + B.z * hh.z + Cc.z; v.w = A.w * g.w + B.w * hh.w + Cc.w;
      *(float4*)(dx + r * lddx + c) = v;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    } else {
      float g = dy[r * lddy + c];
      const float hv = h[r * ldh + c];
      if (table) { if (!(fmaf(hv - table[c], table[n + c], table[2 * n + c]) > 0.f)) g = 0.f; }
      else if (y && !(y[r * ldy + c] > 0.f)) g = 0.f;
      const float v = coef[c] * g + coef[n + c] * hv + coef[2 * n + c];
      dx[r * lddx + c] = v;
      amax = fmaxf(amax, fabsf(v));
    }
  }
  if (dx_absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0)
      atomicMax((unsigned int*)dx_absmax + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(amax));
  }
}

// ---------------------------------------------------------------------------------------------- message passing
// sum over the 64 lanes of DEP per-lane values in DEP/2 + DEP/4 + ... + 1 + (6 - log2 DEP) shuffles: the first
// log2(DEP) steps halve the vector while exchanging with the partner lane, after which lane l holds the 2^k-lane
// partial of component idx(l); the remaining steps are a plain butterfly.  Returns the total of component
// `comp` (set per lane); every component appears in 64 / DEP lanes.
template <int DEP, int HALF, int BIT>
__device__ __forceinline__ void wave_reduce_step(float (&v)[DEP], int lane, int& comp) {
  if constexpr (HALF >= 1) {
    const bool up = (lane >> BIT) & 1;
#pragma unroll
    for (int i = 0; i < HALF; i++) {                            // (all indices are compile-time constants)
      const float send = up ? v[i] : v[i + HALF];
      const float keep = up ? v[i + HALF] : v[i];
      v[i] = keep + __shfl_xor(send, 1 << BIT, 64);
    }
    if (up) comp += HALF;
    wave_reduce_step<DEP, HALF / 2, BIT + 1>(v, lane, comp);
  }
}

template <int DEP>
__device__ __forceinline__ float wave_reduce_vec(float (&v)[DEP], int lane, int& comp) {
  comp = 0;
  wave_reduce_step<DEP, DEP / 2, 0>(v, lane, comp);
  float s = v[0];
#pragma unroll
  for (int off = DEP; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
  return s;
}

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

// Edge half, two kernels.  CS waves per CSR-by-target segment (persistent, strided); wave h of a group owns the channel
// groups (of 4) lane + 64 h, so the W_e slice stays at 4 x DEP registers per lane for any message width <= 1024.
//   k_mpnn_bwd_arg  (max only) repeats the forward's gather to find, per channel, the FIRST edge that attains the
//                   maximum and records its position in arg[t, :];
//   k_mpnn_bwd_edge routes dM[t, c] to that edge (mean / add: to every edge, / deg): d_edge_attr (atomic when CS > 1:
//                   every wave adds its channels' share) and the per-slot partial of dW_e.
// The node half (k_mpnn_bwd_src) turns arg into dQ without atomics.
template <int CS, int DEP, bool VEC, bool LOC = false>
__global__ __launch_bounds__(256) void k_mpnn_bwd_arg(const float* __restrict__ Q, int64_t ldq, const float* __restrict__ We,
                                                     int64_t ldwe, const float* __restrict__ ea, int de,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ node_order, int64_t n, int d,
                                                     int32_t* __restrict__ arg_out) {
  // LOC: arg_out is a uint16 [n, d] array of indices INSIDE the segment (rgnn_mpnn_max_bwd); else int32 edge positions
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_slots = (int64_t)gridDim.x * (blockDim.x >> 6) / CS;
  const int cg = lane + 64 * (int)(wave % CS);
  if (cg >= ((d + 3) >> 2)) return;
  const int cb = cg * 4;
  float w[4][DEP];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < DEP; j++) w[k][j] = (j < de && cb + k < d) ? We[(int64_t)(cb + k) * ldwe + j] : 0.f;
  for (int64_t p = wave / CS; p < n; p += n_slots) {
    const int r0 = rowptr[p], r1 = rowptr[p + 1];
    if (r1 == r0) continue;
    const int64_t t = node_order ? (int64_t)node_order[p] : p;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int arg[4] = {-1, -1, -1, -1};
    // gathers two edges ahead of the compare (the row address depends on a loaded index)
    float4 q0 = load4<VEC>(Q + (int64_t)src[r0] * ldq, cb, d);
    float4 q1 = (r0 + 1 < r1) ? load4<VEC>(Q + (int64_t)src[r0 + 1] * ldq, cb, d) : q0;
    for (int e = r0; e < r1; e++) {
      const float4 qv = q0;
      q0 = q1;
      if (e + 2 < r1) q1 = load4<VEC>(Q + (int64_t)src[e + 2] * ldq, cb, d);
      float v[4] = {qv.x, qv.y, qv.z, qv.w};
      if (de > 0) {
        float z[DEP];
#pragma unroll
        for (int j = 0; j < DEP; j++) z[j] = (j < de) ? ea[(int64_t)e * de + j] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < DEP; j++) a += w[k][j] * z[j];
          v[k] += a;
        }
      }
      if (v[0] > best.x) { best.x = v[0]; arg[0] = e; }
      if (v[1] > best.y) { best.y = v[1]; arg[1] = e; }
      if (v[2] > best.z) { best.z = v[2]; arg[2] = e; }
      if (v[3] > best.w) { best.w = v[3]; arg[3] = e; }
    }
    if (LOC) {
      uint2 pk;
      pk.x = (unsigned)((arg[0] - r0) & 0xffff) | ((unsigned)(arg[1] - r0) << 16);
      pk.y = (unsigned)((arg[2] - r0) & 0xffff) | ((unsigned)(arg[3] - r0) << 16);
      *(uint2*)((uint16_t*)arg_out + t * (int64_t)d + cb) = pk;
      continue;
    }
    int32_t* ap = arg_out + t * (int64_t)d + cb;
    if (VEC) {
      *(int4*)ap = make_int4(arg[0], arg[1], arg[2], arg[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (cb + k < d) ap[k] = arg[k];
    }
  }
}

template <int CS, int DEP, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_mpnn_bwd_edge(const float* __restrict__ dM, int64_t lddm, const float* __restrict__ We,
                                                      int64_t ldwe, const float* __restrict__ ea, int de,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ node_order,
                                                      int64_t n, int d, const int32_t* __restrict__ arg_in,
                                                      int64_t n_edges, float* __restrict__ dea, float* __restrict__ dWe) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_slots = (int64_t)gridDim.x * (blockDim.x >> 6) / CS;   // == rgnn_mpnn_bwd_slots(n)
  const int cg = lane + 64 * (int)(wave % CS);
  const bool act = cg < ((d + 3) >> 2);
  const int cb = act ? cg * 4 : 0;
  // CS > 1: wave h writes ITS channels' share of d_edge_attr to slice h of a [CS, E, de] buffer (plain stores; summed
  // by k_reduce_slots) -- an atomic here would put a memory-side round trip into every iteration of the edge loop
  float* dea_w = dea + (wave % CS) * n_edges * de;
  // Results of up to BUF edges wait in LDS and are flushed together: a global store inside the edge loop would make
  // every iteration wait for its acknowledgement (loads and stores share vmcnt on gfx9 and cannot be told apart).
  constexpr int BUF = 32;
  __shared__ float buf_v[4][BUF][DEP];
  __shared__ int buf_e[4][BUF];
  const int wib = threadIdx.x >> 6;
  int n_buf = 0;
  auto flush = [&]() {
    for (int i = lane; i < n_buf * DEP; i += 64) {
      const int r = i / DEP, c = i % DEP;
      if (c < de) dea_w[(int64_t)buf_e[wib][r] * de + c] = buf_v[wib][r][c];
    }
    n_buf = 0;
  };
  float w[4][DEP], dw[4][DEP];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < DEP; j++) {
      w[k][j] = (act && j < de && cb + k < d) ? We[(int64_t)(cb + k) * ldwe + j] : 0.f;
      dw[k][j] = 0.f;
    }
  // The per-segment work is a chain of dependent loads (rowptr -> node id -> dM / arg rows -> attributes), so the loop is
  // software-pipelined across segments: segment descriptors are fetched two segments ahead, rows one segment ahead.
  struct Meta { int r0, r1; int64_t t; };
  struct Rows { float4 g; int arg[4]; float z0[DEP]; };
  auto load_meta = [&](int64_t p) {
    Meta m = {0, 0, 0};
    if (p < n) {
      m.r0 = rowptr[p]; m.r1 = rowptr[p + 1];
      m.t = node_order ? (int64_t)node_order[p] : p;
    }
    return m;
  };
  auto load_rows = [&](const Meta& m) {
    Rows r;
    r.g = make_float4(0.f, 0.f, 0.f, 0.f);
    r.arg[0] = r.arg[1] = r.arg[2] = r.arg[3] = -1;
#pragma unroll
    for (int j = 0; j < DEP; j++) r.z0[j] = 0.f;
    if (m.r1 > m.r0) {
      if (act) {
        r.g = load4<VEC>(dM + m.t * lddm, cb, d);
        if (MODE == 0) {
          const int32_t* ap = arg_in + m.t * (int64_t)d + cb;
          if (VEC) { const int4 a = *(const int4*)ap; r.arg[0] = a.x; r.arg[1] = a.y; r.arg[2] = a.z; r.arg[3] = a.w; }
          else {
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (cb + k < d) r.arg[k] = ap[k];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < DEP; j++)
        if (j < de) r.z0[j] = ea[(int64_t)m.r0 * de + j];
    }
    return r;
  };
  const int64_t p_first = wave / CS;
  Meta m0 = load_meta(p_first), m1 = load_meta(p_first + n_slots);
  Rows rw0 = load_rows(m0);
  for (int64_t p = p_first; p < n; p += n_slots) {
    const Meta m2 = load_meta(p + 2 * n_slots);
    const Rows rw1 = load_rows(m1);
    const int r0 = m0.r0, r1 = m0.r1;
    if (r1 > r0) {
    float4 g = rw0.g;
    if (MODE == 0) {
      const int* arg = rw0.arg;
      const float gv[4] = {g.x, g.y, g.z, g.w};
      float zn[DEP];                                           // attributes of the next edge, in flight during the reduce
#pragma unroll
      for (int j = 0; j < DEP; j++) zn[j] = rw0.z0[j];
      for (int e = r0; e < r1; e++) {
        float z[DEP], dz[DEP];
#pragma unroll
        for (int j = 0; j < DEP; j++) { z[j] = zn[j]; dz[j] = 0.f; }
        if (e + 1 < r1) {
#pragma unroll
          for (int j = 0; j < DEP; j++) zn[j] = (j < de) ? ea[(int64_t)(e + 1) * de + j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float gg = (arg[k] == e) ? gv[k] : 0.f;        // branch-free: one select, then plain FMAs
#pragma unroll
          for (int j = 0; j < DEP; j++) { dz[j] += gg * w[k][j]; dw[k][j] += gg * z[j]; }
        }
        int comp;
        const float tot = wave_reduce_vec<DEP>(dz, lane, comp);
        if (lane < DEP) buf_v[wib][n_buf][comp] = tot;
        if (lane == 0) buf_e[wib][n_buf] = e;
        if (++n_buf == BUF) flush();                           // (same wave wrote and reads: LDS operations stay in order)
      }
    } else {
      const float sc = (MODE == 1) ? 1.f / (float)(r1 - r0) : 1.f;
      g.x *= sc; g.y *= sc; g.z *= sc; g.w *= sc;
      const float gv[4] = {g.x, g.y, g.z, g.w};
      float dz[DEP], zs[DEP];
#pragma unroll
      for (int j = 0; j < DEP; j++) { dz[j] = 0.f; zs[j] = 0.f; }
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < DEP; j++) dz[j] += gv[k] * w[k][j];
      int comp;
      const float tot = wave_reduce_vec<DEP>(dz, lane, comp);
      for (int e = r0; e < r1; e++) {
#pragma unroll
        for (int j = 0; j < DEP; j++) zs[j] += (j < de) ? ea[(int64_t)e * de + j] : 0.f;
        if (lane < DEP) buf_v[wib][n_buf][comp] = tot;
        if (lane == 0) buf_e[wib][n_buf] = e;
        if (++n_buf == BUF) flush();
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < DEP; j++) dw[k][j] += gv[k] * zs[j];
    }
    }
    m0 = m1; m1 = m2; rw0 = rw1;
  }
  flush();
  // per-slot partial of dW_e (plain stores; k_reduce_slots sums the slots: 30 M same-address atomics otherwise)
  if (act && wave / CS < n_slots) {
    float* o = dWe + (wave / CS) * (int64_t)d * de;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < DEP; j++)
        if (j < de && cb + k < d) o[(int64_t)(cb + k) * de + j] = dw[k][j];
  }
}

// Edge half of the max aggregation for message widths <= 512 and <= 8 edge attributes (the shipped models), two kernels
// that each keep their reduction INSIDE a lane (the general kernel above asks, edge after edge, which channels it won: 64
// FMAs and a 10-shuffle wave reduction per edge, 850 us per layer at C2).  The gradient of channel c of target t goes to
// ONE edge, arg[t, c]:
//   k_mpnn_bwd_dwe_max  lanes = channels (one wave per target segment, 8 channels per lane): dW_e[c, :] += dM[t, c] a_arg is
//                       a per-lane FMA on an attribute row the wave staged in LDS; one partial of dW_e per wave.
//   k_mpnn_bwd_dea_max  lanes = edges (64 consecutive rows of the target-sorted list, i.e. a handful of segments): an edge
//                       scans its target's arg row (16-byte gathers that hit L1: the row is shared by the segment) and adds
//                       dM[t, c] W_e[c, :] for the channels it won, W_e in LDS; d a_e is stored by the lane that owns the
//                       edge -- no reduction across lanes, no atomics, channels summed in ascending order.
template <int DEP>
__global__ __launch_bounds__(256) void k_mpnn_bwd_dwe_max(const float* __restrict__ dM, int64_t lddm, const float* __restrict__ ea,
                                                         int de, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ node_order, int64_t n, int d,
                                                         const uint16_t* __restrict__ arg_in, float* __restrict__ dWe) {
  constexpr int CAP = 128;
  __shared__ __attribute__((aligned(16))) float s_ea[4][CAP * DEP];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wib;
  const int64_t n_slots = (int64_t)gridDim.x * 4;
  int cb[2];
  bool ok[2];
  float dw[2][4][DEP];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    cb[t] = (lane + 64 * t) * 4;
    ok[t] = cb[t] < d;                                  // (d % 4 == 0: the dispatcher takes these kernels for vectorisable rows only)
    if (!ok[t]) cb[t] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int j = 0; j < DEP; j++) dw[t][k][j] = 0.f;
  }
  float* const my_ea = s_ea[wib];
  // One target per trip, the NEXT target's segment bounds, gradient row and arg row requested while this one is processed (behind
  // this trip's attribute loads: returns are in order, so the wait for the attributes does not wait for the prefetch).  A trip used
  // to be three dependent round trips to memory -- rowptr, then the rows, then the attributes (192 us per layer -> MEASUREMENTS.md section 8).
  int r0 = 0, r1 = 0;
  int64_t tn = 0;
  float4 gv[2];
  uint2 av[2];
  auto fetch = [&](int64_t p, int& q0, int& q1, int64_t& tq, float4 (&gq)[2], uint2 (&aq)[2]) {
    q0 = rowptr[p]; q1 = rowptr[p + 1];
    tq = node_order ? (int64_t)node_order[p] : p;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      gq[t] = *(const float4*)(dM + tq * lddm + cb[t]);
      aq[t] = *(const uint2*)(arg_in + tq * (int64_t)d + cb[t]);          // four indices inside the segment
    }
  };
  if (wave < n) fetch(wave, r0, r1, tn, gv, av);
  for (int64_t p = wave; p < n; p += n_slots) {
    const int64_t pn = p + n_slots;
    int r0n = 0, r1n = 0;
    int64_t tnn = 0;
    float4 gvn[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    uint2 avn[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
    bool fetched = false;
    if (r1 > r0) {
      float g[2][4];
      int a[2][4];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        g[t][0] = gv[t].x; g[t][1] = gv[t].y; g[t][2] = gv[t].z; g[t][3] = gv[t].w;
        a[t][0] = r0 + (int)(av[t].x & 0xffff); a[t][1] = r0 + (int)(av[t].x >> 16);
        a[t][2] = r0 + (int)(av[t].y & 0xffff); a[t][3] = r0 + (int)(av[t].y >> 16);
      }
      for (int c0 = r0; c0 < r1; c0 += CAP) {
        const int cnt = min(CAP, r1 - c0);
        if (cnt * DEP <= 64) {                            // (up to 8 edges: one element per lane)
          const int e = lane / DEP, j = lane % DEP;
          const float v = (lane < cnt * DEP && j < de) ? ea[(int64_t)(c0 + e) * de + j] : 0.f;
          if (!fetched && pn < n) { fetch(pn, r0n, r1n, tnn, gvn, avn); fetched = true; }
          if (lane < cnt * DEP) my_ea[lane] = v;
        } else {
          for (int i = lane; i < cnt * DEP; i += 64) {    // the pass's attribute rows -> LDS (zero-padded to DEP)
            const int e = i / DEP, j = i % DEP;
            my_ea[i] = (j < de) ? ea[(int64_t)(c0 + e) * de + j] : 0.f;
          }
        }
        // (the wave's own LDS writes and reads execute in issue order: a compiler barrier is all the hand-over needs -- a
        //  wavefront-scope fence also waits for every outstanding GLOBAL load, i.e. for the prefetch)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int j = a[t][k] - c0;
            const bool hit = ok[t] && j >= 0 && j < cnt;
            const float gg = hit ? g[t][k] : 0.f;
            const int jj = hit ? j : 0;
            const float4 z0 = *(const float4*)(my_ea + jj * DEP);
            dw[t][k][0] += gg * z0.x; dw[t][k][1] += gg * z0.y; dw[t][k][2] += gg * z0.z; dw[t][k][3] += gg * z0.w;
            if (DEP > 4) {
              const float4 z1 = *(const float4*)(my_ea + jj * DEP + 4);
              dw[t][k][4] += gg * z1.x; dw[t][k][5] += gg * z1.y; dw[t][k][6] += gg * z1.z; dw[t][k][7] += gg * z1.w;
            }
          }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!fetched && pn < n) fetch(pn, r0n, r1n, tnn, gvn, avn);
    r0 = r0n; r1 = r1n; tn = tnn;
#pragma unroll
    for (int t = 0; t < 2; t++) { gv[t] = gvn[t]; av[t] = avn[t]; }
  }
  // one partial per WORK-GROUP: waves 1 .. 3 hand their sums to wave 0 through LDS, a quarter of the registers at a time, in a
  // fixed order (deterministic).  Four times the waves of r02's one-partial-per-wave grid for the same partial buffer: the kernel
  // waits on memory, and 2 waves per SIMD hid little of it.
  float* const xch = &s_ea[0][0];                      // 4 x CAP x DEP floats >= 3 waves x 64 lanes x 16 floats
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (wib > 0) {
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
          for (int j = 0; j < DEP; j++) xch[((wib - 1) * 16 + k * DEP + j) * 64 + lane] = dw[t][2 * half + k][j];
      }
      __syncthreads();
      if (wib == 0) {
#pragma unroll
        for (int w = 0; w < 3; w++)
#pragma unroll
          for (int k = 0; k < 2; k++)
#pragma unroll
            for (int j = 0; j < DEP; j++) dw[t][2 * half + k][j] += xch[(w * 16 + k * DEP + j) * 64 + lane];
      }
      __syncthreads();
    }
  if (wib == 0) {
    float* o = dWe + blockIdx.x * (int64_t)d * de;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < DEP; j++)
          if (ok[t] && j < de) o[(int64_t)(cb[t] + k) * de + j] = dw[t][k][j];
  }
}

template <int DEP>
__global__ __launch_bounds__(256) void k_mpnn_bwd_dea_max(const float* __restrict__ dM, int64_t lddm, const float* __restrict__ We,
                                                         int64_t ldwe, int de, const int32_t* __restrict__ tgt_sorted,
                                                         const int32_t* __restrict__ eloc_sorted, int64_t n_edges, int d,
                                                         const uint16_t* __restrict__ arg_in, float* __restrict__ dea) {
  __shared__ __attribute__((aligned(16))) float sW[512 * DEP];   // W_e rows, zero-padded to DEP
  for (int i = threadIdx.x; i < d * DEP; i += 256) {
    const int c = i / DEP, j = i % DEP;
    sW[i] = (j < de) ? We[(int64_t)c * ldwe + j] : 0.f;
  }
  __syncthreads();
  const int d8 = d >> 3;                                // 16-byte pieces of an arg row: 8 channels each (d % 8 == 0)
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * 256) {
    const int64_t t = tgt_sorted[e];
    const uint4* __restrict__ ar = (const uint4*)(arg_in + t * (int64_t)d);
    const float4* __restrict__ gr = (const float4*)(dM + t * lddm);
    const unsigned me = (unsigned)eloc_sorted[e];
    float acc[DEP];
#pragma unroll
    for (int j = 0; j < DEP; j++) acc[j] = 0.f;
    // A DENSE masked mat-vec: m[c] = (arg[t, c] == me) ? dM[t, c] : 0, d a_e = sum_c m[c] W_e[c, :].  Both rows are read with
    // 16-byte loads (the segment's edges sit in neighbouring lanes and share them: L1 hits), every channel is multiplied --
    // 8x more FMAs than the channels the edge actually won, but no branch, no scattered 4-byte gradient loads (the first
    // version, which only visited the hits, spent its time in exactly those: 419 us against this one's -- see MEASUREMENTS.md 7), and
    // the W_e rows are LDS broadcasts (all lanes read the same address).
    for (int i8 = 0; i8 < d8; i8++) {
      const uint4 a = ar[i8];
      const float4 g0 = gr[2 * i8], g1 = gr[2 * i8 + 1];
      float m[8];
      m[0] = ((a.x & 0xffffu) == me) ? g0.x : 0.f; m[1] = ((a.x >> 16) == me) ? g0.y : 0.f;
      m[2] = ((a.y & 0xffffu) == me) ? g0.z : 0.f; m[3] = ((a.y >> 16) == me) ? g0.w : 0.f;
      m[4] = ((a.z & 0xffffu) == me) ? g1.x : 0.f; m[5] = ((a.z >> 16) == me) ? g1.y : 0.f;
      m[6] = ((a.w & 0xffffu) == me) ? g1.z : 0.f; m[7] = ((a.w >> 16) == me) ? g1.w : 0.f;
      const float* wrow = sW + i8 * 8 * DEP;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float4 w0 = *(const float4*)(wrow + c * DEP);
        acc[0] = fmaf(m[c], w0.x, acc[0]); acc[1] = fmaf(m[c], w0.y, acc[1]);
        acc[2] = fmaf(m[c], w0.z, acc[2]); acc[3] = fmaf(m[c], w0.w, acc[3]);
        if (DEP > 4) {
          const float4 w1 = *(const float4*)(wrow + c * DEP + 4);
          acc[4] = fmaf(m[c], w1.x, acc[4]); acc[5] = fmaf(m[c], w1.y, acc[5]);
          acc[6] = fmaf(m[c], w1.z, acc[6]); acc[7] = fmaf(m[c], w1.w, acc[7]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DEP; j++)
      if (j < de) dea[e * de + j] = acc[j];
  }
}

// Node half for max aggregation with uint16 in-segment winners: dQ[s, :] = sum over the out-edges j of s of dM[t_j, c] where
// arg[t_j, c] names that edge (tloc[j] = its index inside the segment of its target) -- a gather over the CSR by source,
// plain stores, no atomics.  The arg row of an out-edge is read first (a quarter of the bytes of the gradient row), and the
// gradient is only loaded by the lanes that found a hit in their four channels.
template <int NCH>
__global__ __launch_bounds__(256) void k_mpnn_bwd_src_max16(const float* __restrict__ dM, int64_t lddm, const uint16_t* __restrict__ arg,
                                                           const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ tnode,
                                                           const int32_t* __restrict__ tloc, const int32_t* __restrict__ node_order,
                                                           int64_t n, int d, float* __restrict__ dQ, int64_t lddq,
                                                           float* __restrict__ dq_absmax) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int groups = d >> 2;
  float amax = 0.f;
  for (int64_t p = wave; p < n; p += n_waves) {
    const int r0 = rowptr_s[p], r1 = rowptr_s[p + 1];
    const int64_t s_ = node_order ? (int64_t)node_order[p] : p;
    float4 acc[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the (target, index) pairs of up to 64 out-edges at a time: lane i holds those of edge r0 + i
    for (int jb = r0; jb < r1; jb += 64) {
      const int my_t = (jb + lane < r1) ? tnode[jb + lane] : 0;
      const int my_l = (jb + lane < r1) ? tloc[jb + lane] : 0;
      const int nb = min(64, r1 - jb);
      // four out-edges at a time: their arg rows are requested together, then the gradient rows of the lanes that found a hit --
      // a source of the radius graph has ~4 out-edges, and one edge per trip left every load waiting for the one before it
      // (362 -> see MEASUREMENTS.md section 8).  The sums run in edge order as before: same bits.
      for (int i = 0; i < nb; i += 4) {
        int64_t t[4];
        unsigned pos[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int e = min(i + u, nb - 1);            // (a trip's unused slots re-read the last edge with an index nothing matches)
          t[u] = __builtin_amdgcn_readlane(my_t, e);
          pos[u] = (i + u < nb) ? (unsigned)__builtin_amdgcn_readlane(my_l, e) : 0x10000u;
        }
        uint2 a[4][NCH];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < NCH; q++) {
            const int cg = lane + 64 * q;
            a[u][q] = (cg < groups) ? *(const uint2*)(arg + t[u] * (int64_t)d + cg * 4) : make_uint2(0xffffffffu, 0xffffffffu);
          }
        float4 g[4][NCH];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < NCH; q++) {
            const int cg = lane + 64 * q;
            const bool hit = (a[u][q].x & 0xffffu) == pos[u] || (a[u][q].x >> 16) == pos[u] || (a[u][q].y & 0xffffu) == pos[u] ||
                             (a[u][q].y >> 16) == pos[u];
            g[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hit && cg < groups) g[u][q] = *(const float4*)(dM + t[u] * lddm + cg * 4);
          }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < NCH; q++) {
            const bool h0 = (a[u][q].x & 0xffffu) == pos[u], h1 = (a[u][q].x >> 16) == pos[u], h2 = (a[u][q].y & 0xffffu) == pos[u],
                       h3 = (a[u][q].y >> 16) == pos[u];
            if (h0 | h1 | h2 | h3) {
              acc[q].x += h0 ? g[u][q].x : 0.f; acc[q].y += h1 ? g[u][q].y : 0.f;
              acc[q].z += h2 ? g[u][q].z : 0.f; acc[q].w += h3 ? g[u][q].w : 0.f;
            }
          }
      }
    }
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int cg = lane + 64 * q;
      if (cg >= groups) continue;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[q].x), fabsf(acc[q].y)), fmaxf(fabsf(acc[q].z), fabsf(acc[q].w))));
#if RGNN_BWD_NT_STORE
      {                                              // streaming store (as in the forward edge kernel: dQ is not read again here and
        float* o_ = dQ + s_ * lddq + cg * 4;          //  should leave L2 to the arg / gradient rows that are)
        __builtin_nontemporal_store(acc[q].x, o_); __builtin_nontemporal_store(acc[q].y, o_ + 1);
        __builtin_nontemporal_store(acc[q].z, o_ + 2); __builtin_nontemporal_store(acc[q].w, o_ + 3);
      }
#else
      *(float4*)(dQ + s_ * lddq + cg * 4) = acc[q];
#endif
    }
  }
  if (dq_absmax) {                                     // one atomic per wave, spread over the slots of the bound (rgnn.h)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) atomicMax((unsigned int*)dq_absmax + (wave & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(amax));
  }
}

// out[i] = sum_s part[s, i]: one block = 64 columns x 16 slot groups (coalesced 256-B reads), LDS combine
__global__ __launch_bounds__(1024) void k_reduce_slots(const float* __restrict__ part, int64_t slots, int64_t width,
                                                      float* __restrict__ out) {
  __shared__ float red[16][64];
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lc;
  float a0 = 0.f, a1 = 0.f;
  if (i < width) {
    int64_t s = g;
    for (; s + 16 < slots; s += 32) { a0 += part[s * width + i]; a1 += part[(s + 16) * width + i]; }
    if (s < slots) a0 += part[s * width + i];
  }
  red[g][lc] = a0 + a1;
  __syncthreads();
  if (g == 0 && i < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][lc];
    out[i] = t;
  }
}

// Node half: dQ[s, :] = sum over the edges leaving s of the gradient their target routed to them -- a gather over the
// CSR keyed on the SOURCE (tnode[j] = target of out-edge j, tpos[j] = its position in the target-sorted edge list,
// which is what arg[] holds), plain stores, no atomics; nodes without out-edges get 0.
template <int NCH, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_mpnn_bwd_src(const float* __restrict__ dM, int64_t lddm, const int32_t* __restrict__ arg,
                                                     const float* __restrict__ target_scale, const int32_t* __restrict__ rowptr_s,
                                                     const int32_t* __restrict__ tnode, const int32_t* __restrict__ tpos,
                                                     const int32_t* __restrict__ node_order, int64_t n, int d,
                                                     float* __restrict__ dQ, int64_t lddq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int groups = (d + 3) >> 2;
  for (int64_t p = wave; p < n; p += n_waves) {
    const int r0 = rowptr_s[p], r1 = rowptr_s[p + 1];
    const int64_t s = node_order ? (int64_t)node_order[p] : p;
    float4 acc[NCH];
#pragma unroll
    for (int q = 0; q < NCH; q++) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = r0; j < r1; j++) {
      const int64_t t = tnode[j];
      const int pos = tpos[j];
      const float sc = (MODE == 1) ? target_scale[t] : 1.f;
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const int cg = lane + 64 * q;
        if (cg >= groups) continue;
        const int cb = cg * 4;
        const float4 g = load4<VEC>(dM + t * lddm, cb, d);
        if (MODE == 0) {
          int a0, a1, a2, a3;
          const int32_t* ap = arg + t * (int64_t)d + cb;
          if (VEC) { const int4 a = *(const int4*)ap; a0 = a.x; a1 = a.y; a2 = a.z; a3 = a.w; }
          else { a0 = ap[0]; a1 = (cb + 1 < d) ? ap[1] : -1; a2 = (cb + 2 < d) ? ap[2] : -1; a3 = (cb + 3 < d) ? ap[3] : -1; }
          acc[q].x += (a0 == pos) ? g.x : 0.f; acc[q].y += (a1 == pos) ? g.y : 0.f;
          acc[q].z += (a2 == pos) ? g.z : 0.f; acc[q].w += (a3 == pos) ? g.w : 0.f;
        } else {
          acc[q].x += g.x * sc; acc[q].y += g.y * sc; acc[q].z += g.z * sc; acc[q].w += g.w * sc;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NCH; q++) {
      const int cg = lane + 64 * q;
      if (cg >= groups) continue;
      const int cb = cg * 4;
      float* o = dQ + s * lddq + cb;
      if (VEC) *(float4*)o = acc[q];
      else {
        if (cb + 0 < d) o[0] = acc[q].x;
        if (cb + 1 < d) o[1] = acc[q].y;
        if (cb + 2 < d) o[2] = acc[q].z;
        if (cb + 3 < d) o[3] = acc[q].w;
      }
    }
  }
}

// Backward of rgnn_segment_reduce (general message path, pre_layers > 1): rows [E, d] in CSR-by-target order were
// reduced per segment.  max: dM[t, c] goes to the first row of the segment attaining the maximum; mean / add: to every row.
__global__ __launch_bounds__(256) void k_segment_reduce_bwd(const float* __restrict__ dM, int64_t lddm,
                                                           const float* __restrict__ rows, int64_t ldr,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ order,
                                                           int64_t n, int d, int aggr, float* __restrict__ drows, int64_t lddr) {
  const int lane = threadIdx.x & 63;
  const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= n) return;
  const int64_t node = order ? (int64_t)order[pos] : pos;
  const int beg = rowptr[pos], end = rowptr[pos + 1];
  if (end == beg) return;
  const float sc = (aggr == RGNN_AGGR_MEAN) ? 1.f / (float)(end - beg) : 1.f;
  for (int c = lane; c < d; c += 64) {
    const float g = dM[node * lddm + c];
    if (aggr == RGNN_AGGR_MAX) {
      float best = -INFINITY;
      int arg = beg;
      for (int e = beg; e < end; e++) {
        const float v = rows[(int64_t)e * ldr + c];
        if (v > best) { best = v; arg = e; }
      }
      for (int e = beg; e < end; e++) drows[(int64_t)e * lddr + c] = (e == arg) ? g : 0.f;
    } else {
      for (int e = beg; e < end; e++) drows[(int64_t)e * lddr + c] = g * sc;
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dW[n, k] = sum_m G[m, n] * [A1 | A2][m, k]  (G = gradient of the layer output, A = the layer input): a GEMM whose
// reduction runs over the ROWS of both operands.  v_mfma_f32_32x32x2_f32 wants, per lane, one value of row (lane & 31)
// and reduction slot (lane >> 5) of each operand -- with the reduction along m that is G[m + (lane >> 5)][n + (lane &
// 31)]: 32 consecutive floats of a row, so the tiles go into LDS exactly as they lie in memory (no transpose) and the
// fragments are conflict-free ds_read_b32.  Work decomposition: 128 x 128 output tiles x `slabs` row slabs; every
// work-group reduces its slab into registers and stores the partial tile; k_reduce_slots sums the slabs (no atomics,
// deterministic).  Slabs are dealt to XCDs (block b -> XCD b % 8) with the tiles of one slab on the same XCD, so a slab
// of G / A leaves HBM once.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct WgParams {
  const float* G; int64_t ldg; const float* A1; int64_t lda1; int k1; const float* A2; int64_t lda2; int k2;
  int64_t m; int n; float* part; int slabs; int64_t rows_per_slab; int nt, kt;
};

__global__ __launch_bounds__(256) void k_wgrad(const WgParams p) {
  constexpr int BM = 32, BT = 128;
  __shared__ __attribute__((aligned(16))) float Gs[2][BM][BT];
  __shared__ __attribute__((aligned(16))) float As[2][BM][BT];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = p.nt * p.kt;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile = idx % tiles;
  const int slab = (idx / tiles) * 8 + xcd;
  if (slab >= p.slabs) return;
  const int n0 = (tile / p.kt) * BT, k0 = (tile % p.kt) * BT;
  const int K = p.k1 + p.k2;
  const int64_t m_beg = (int64_t)slab * p.rows_per_slab;
  const int64_t m_end = (m_beg + p.rows_per_slab < p.m) ? m_beg + p.rows_per_slab : p.m;

  // this thread's 4 + 4 float4 of a 32-row step: rows (t >> 5) + 8 s, 16-byte column t & 31
  const int c4 = (t & 31) * 4, r_base = t >> 5;
  const int gn = n0 + c4, gk = k0 + c4;
  const bool g_ok = gn < p.n, a_ok = gk < K;
  const float* a_col = nullptr;
  int64_t a_ld = 0;
  if (a_ok) {
    if (gk < p.k1) { a_col = p.A1 + gk; a_ld = p.lda1; }
    else { a_col = p.A2 + (gk - p.k1); a_ld = p.lda2; }
  }
  const float* g_col = p.G + (g_ok ? gn : 0);
  float4 rg[4], ra[4];
  auto load_step = [&](int64_t m0) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int64_t m = m0 + r_base + 8 * s;
      const bool row_ok = m < m_end;
      const int64_t mc = row_ok ? m : m_beg;
      float4 v = *(const float4*)(g_col + mc * p.ldg);
      if (!(row_ok && g_ok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      rg[s] = v;
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_ok) {
        u = *(const float4*)(a_col + mc * a_ld);
        if (!row_ok) u = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ra[s] = u;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      *(float4*)&Gs[buf][r_base + 8 * s][c4] = rg[s];
      *(float4*)&As[buf][r_base + 8 * s][c4] = ra[s];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  if (m_beg < m_end) {
    load_step(m_beg);
    int cur = 0;
    const int frag_row = lane >> 5, frag_col = lane & 31;
    for (int64_t m0 = m_beg; m0 < m_end; m0 += BM) {
      store_step(cur);
      __syncthreads();
      if (m0 + BM < m_end) load_step(m0 + BM);
      const float* gp = &Gs[cur][frag_row][wn * 64 + frag_col];
      const float* ap = &As[cur][frag_row][wk * 64 + frag_col];
#pragma unroll
      for (int s = 0; s < BM / 2; s++) {
        const float g0 = gp[(2 * s) * BT], g1 = gp[(2 * s) * BT + 32];
        const float a0 = ap[(2 * s) * BT], a1 = ap[(2 * s) * BT + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, a1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, a0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, a1, acc[1][1], 0, 0, 0);
      }
      cur ^= 1;   // the other buffer was last read before this step's barrier: safe to overwrite next iteration
    }
  }
  // partial tile: part[slab][n][k]; D layout: row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (k) = lane & 31
  float* out = p.part + (int64_t)slab * p.n * K;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int kk = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int nn = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (nn < p.n && kk < K) out[(int64_t)nn * K + kk] = acc[i][j][r];
      }
    }
}

struct BwdArgs {
  const float* dM; int64_t lddm; const float* Q; int64_t ldq; const float* We; int64_t ldwe; const float* ea; int de;
  const int32_t* rowptr; const int32_t* src; const int32_t* node_order; int64_t n; int d; int32_t* arg; float* dea; float* dWe;  /* dWe: per-slot partials */
  const float* target_scale; const int32_t* rowptr_s; const int32_t* tnode; const int32_t* tpos; float* dQ; int64_t lddq;
  int64_t n_edges;
};

template <int CS, int DEP, bool VEC>
void launch_bwd(int mode, dim3 g, dim3 b, hipStream_t s, const BwdArgs& a) {
#define RGNN_EDGE(M) hipLaunchKernelGGL((k_mpnn_bwd_edge<CS, DEP, M, VEC>), g, b, 0, s, a.dM, a.lddm, a.We, a.ldwe, a.ea, a.de, a.rowptr, a.node_order, a.n, a.d, a.arg, a.n_edges, a.dea, a.dWe)
  if (mode == RGNN_AGGR_MAX) {
    // the arg pass holds fewer registers: give it 4x the waves (it is a chain of dependent gathers)
    hipLaunchKernelGGL((k_mpnn_bwd_arg<CS, DEP, VEC>), dim3(g.x * 4), b, 0, s, a.Q, a.ldq, a.We, a.ldwe, a.ea, a.de, a.rowptr,
                       a.src, a.node_order, a.n, a.d, a.arg);
    if (a.de > 0) RGNN_EDGE(0);
  } else if (a.de > 0) {
    if (mode == RGNN_AGGR_MEAN) RGNN_EDGE(1); else RGNN_EDGE(2);
  }
#undef RGNN_EDGE
}

template <int NCH, bool VEC>
void launch_src(int mode, dim3 g, dim3 b, hipStream_t s, const BwdArgs& a) {
#define RGNN_SRC(M) hipLaunchKernelGGL((k_mpnn_bwd_src<NCH, M, VEC>), g, b, 0, s, a.dM, a.lddm, a.arg, a.target_scale, a.rowptr_s, a.tnode, a.tpos, a.node_order, a.n, a.d, a.dQ, a.lddq)
  if (mode == RGNN_AGGR_MAX) RGNN_SRC(0);
  else if (mode == RGNN_AGGR_MEAN) RGNN_SRC(1);
  else RGNN_SRC(2);
#undef RGNN_SRC
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
  return rgnn_bn_bwd_stats_table(dy, lddy, y, ldy, nullptr, h, ldh, m, n, partial, stream);
}

extern "C" int rgnn_bn_bwd_stats_table(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* table, const float* h,
                                       int64_t ldh, int64_t m, int32_t n, float* partial, rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dy && h && partial, "null pointers");
  const unsigned panels = (unsigned)((m + 127) / 128);
  hipLaunchKernelGGL(k_bn_bwd_stats, dim3(panels, (unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y,
                     ldy, table, h, ldh, m, n, partial);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_bn_bwd_apply(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                                 const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, rgnn_stream_t stream) {
  return rgnn_bn_bwd_apply_table(dy, lddy, y, ldy, nullptr, h, ldh, coef, m, n, dx, lddx, nullptr, stream);
}

extern "C" int rgnn_bn_bwd_apply_absmax(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* h, int64_t ldh,
                                        const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx, float* dx_absmax,
                                        rgnn_stream_t stream) {
  return rgnn_bn_bwd_apply_table(dy, lddy, y, ldy, nullptr, h, ldh, coef, m, n, dx, lddx, dx_absmax, stream);
}

extern "C" int rgnn_bn_bwd_apply_table(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* table, const float* h,
                                       int64_t ldh, const float* coef, int64_t m, int32_t n, float* dx, int64_t lddx,
                                       float* dx_absmax, rgnn_stream_t stream) {
  if (m == 0 || n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dy && h && coef && dx, "null pointers");
  const bool vec = n % 4 == 0 && lddy % 4 == 0 && ldh % 4 == 0 && lddx % 4 == 0 && (y == nullptr || ldy % 4 == 0) &&
                   (((uintptr_t)dy | (uintptr_t)h | (uintptr_t)dx | (uintptr_t)coef | (uintptr_t)(y ? y : dy) |
                     (uintptr_t)(table ? table : dy)) & 15) == 0;
  int64_t blocks = rgnn_blocks(vec ? m * (n / 4) : m * n, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;            // grid-stride: 16 work-groups per CU
  if (vec)
    hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, table, h,
                       ldh, coef, m, n, dx, lddx, dx_absmax);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, table, h,
                       ldh, coef, m, n, dx, lddx, dx_absmax);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int32_t rgnn_mpnn_bwd_split(int32_t d) {
  const int nch = ((d + 3) / 4 + 63) / 64;
  return nch == 3 ? 4 : nch;
}

extern "C" int64_t rgnn_mpnn_bwd_slots(int64_t n) {
  int64_t s = n < 2048 ? n : 2048;
  s = (s + 3) / 4 * 4;                                  // whole 4-wave blocks for every waves-per-segment factor
  return s > 0 ? s : 4;
}

extern "C" int rgnn_mpnn_aggregate_bwd(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We,
                                       int64_t ldwe, const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t,
                                       const int32_t* src_sorted, const int32_t* node_order, int64_t n, int32_t d,
                                       int32_t aggr, const int32_t* rowptr_s, const int32_t* tnode, const int32_t* tpos,
                                       const float* target_scale, int64_t n_edges, int32_t* arg_tmp, float* dwe_partial,
                                       float* dea_partial, float* dQ, int64_t lddq, float* d_edge_attr, float* dWe,
                                       rgnn_stream_t stream) {
  if (n == 0 || d == 0) return RGNN_OK;
  if (n_edges == 0) {                                   // a graph without edges: no source received anything, no edge exists
    RGNN_CHECK_ARG(dQ != nullptr, "null pointers");
    hipStream_t s0 = (hipStream_t)stream;
    hipMemset2DAsync(dQ, (size_t)lddq * sizeof(float), 0, (size_t)d * sizeof(float), (size_t)n, s0);
    if (de > 0 && dWe) hipMemsetAsync(dWe, 0, (size_t)d * de * sizeof(float), s0);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(dM && Q && rowptr_t && src_sorted && dQ && rowptr_s && tnode && tpos, "null pointers");
  RGNN_CHECK_ARG(de >= 0 && de <= 16, "edge attribute width must be <= 16");
  RGNN_CHECK_ARG(de == 0 || (edge_attr_sorted && d_edge_attr && We && dWe && dwe_partial), "edge attribute pointers");
  RGNN_CHECK_ARG(aggr >= 0 && aggr <= 2, "unknown aggregation");
  RGNN_CHECK_ARG(aggr != RGNN_AGGR_MAX || arg_tmp, "max aggregation needs the arg workspace");
  RGNN_CHECK_ARG(aggr != RGNN_AGGR_MEAN || target_scale, "mean aggregation needs 1 / in-degree per node");
  RGNN_CHECK_ARG(d <= 1024, "message width must be <= 1024");
  const bool vec = d % 4 == 0 && lddm % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0 &&
                   (((uintptr_t)dM | (uintptr_t)Q | (uintptr_t)dQ | (uintptr_t)arg_tmp) & 15) == 0;
  const int nch = ((d + 3) / 4 + 63) / 64;             // channel groups per lane (node half) / waves per segment (edge half)
  const int64_t slots = rgnn_mpnn_bwd_slots(n);
  const int cs = rgnn_mpnn_bwd_split(d);
  RGNN_CHECK_ARG(de == 0 || cs == 1 || dea_partial, "d_edge_attr partial workspace");
  float* dea_target = (cs == 1) ? d_edge_attr : dea_partial;
  hipStream_t s = (hipStream_t)stream;
  const BwdArgs a = {dM, lddm, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, arg_tmp,
                     dea_target, dwe_partial, target_scale, rowptr_s, tnode, tpos, dQ, lddq, n_edges};
  const dim3 b(256), g((unsigned)(n < 8192 ? (n + 3) / 4 : 2048));
#define RGNN_BWD2(NCH, V)                                                   \
  do {                                                                      \
    constexpr int CS = NCH == 3 ? 4 : NCH;                                  \
    const dim3 ge((unsigned)((slots * CS + 3) / 4));                        \
    if (de <= 4) launch_bwd<CS, 4, V>(aggr, ge, b, s, a);                   \
    else if (de <= 8) launch_bwd<CS, 8, V>(aggr, ge, b, s, a);              \
    else launch_bwd<CS, 16, V>(aggr, ge, b, s, a);                          \
    launch_src<NCH, V>(aggr, g, b, s, a);                                   \
  } while (0)
#define RGNN_BWD(NCH) do { if (vec) RGNN_BWD2(NCH, true); else RGNN_BWD2(NCH, false); } while (0)
  switch (nch) {
    case 1: RGNN_BWD(1); break;
    case 2: RGNN_BWD(2); break;
    case 3: RGNN_BWD(3); break;
    default: RGNN_BWD(4); break;
  }
#undef RGNN_BWD
#undef RGNN_BWD2
  if (de > 0) {
    hipLaunchKernelGGL(k_reduce_slots, dim3(rgnn_blocks((int64_t)d * de, 64)), dim3(1024), 0, s, dwe_partial, slots, (int64_t)d * de, dWe);
    if (cs > 1 && n_edges > 0)
      hipLaunchKernelGGL(k_reduce_slots, dim3(rgnn_blocks(n_edges * de, 64)), dim3(1024), 0, s, dea_partial, (int64_t)cs, n_edges * de, d_edge_attr);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int32_t rgnn_mpnn_max_bwd_supported(int32_t d, int32_t de) { return (d % 8 == 0 && d <= 512 && de > 0 && de <= 8) ? 1 : 0; }

extern "C" int rgnn_mpnn_max_bwd(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                 const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                 const int32_t* tgt_sorted, const int32_t* eloc_sorted, const int32_t* node_order, int64_t n,
                                 int32_t d, const int32_t* rowptr_s, const int32_t* tnode, const int32_t* tloc, int64_t n_edges,
                                 uint16_t* arg, int32_t arg_is_valid, float* dwe_partial, float* dQ, int64_t lddq,
                                 float* d_edge_attr, float* dWe, rgnn_stream_t stream) {
  return rgnn_mpnn_max_bwd_absmax(dM, lddm, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, tgt_sorted, eloc_sorted,
                                  node_order, n, d, rowptr_s, tnode, tloc, n_edges, arg, arg_is_valid, dwe_partial, dQ, lddq,
                                  d_edge_attr, dWe, nullptr, stream);
}

extern "C" int rgnn_mpnn_max_bwd_absmax(const float* dM, int64_t lddm, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                        const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                        const int32_t* tgt_sorted, const int32_t* eloc_sorted, const int32_t* node_order, int64_t n,
                                        int32_t d, const int32_t* rowptr_s, const int32_t* tnode, const int32_t* tloc, int64_t n_edges,
                                        uint16_t* arg, int32_t arg_is_valid, float* dwe_partial, float* dQ, int64_t lddq,
                                        float* d_edge_attr, float* dWe, float* dq_absmax, rgnn_stream_t stream) {
  if (n == 0 || d == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rgnn_mpnn_max_bwd_supported(d, de), "needs d % 8 == 0, d <= 512, 1 <= de <= 8 (else rgnn_mpnn_aggregate_bwd)");
  RGNN_CHECK_ARG(dM && Q && We && edge_attr_sorted && rowptr_t && src_sorted && tgt_sorted && eloc_sorted && rowptr_s && tnode &&
                     tloc && arg && dwe_partial && dQ && d_edge_attr && dWe, "null pointers");
  RGNN_CHECK_ARG(lddm % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0 &&
                     (((uintptr_t)dM | (uintptr_t)Q | (uintptr_t)dQ | (uintptr_t)arg) & 15) == 0, "rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int nch = ((d + 3) / 4 + 63) / 64;             // 1 or 2
  const int64_t slots = rgnn_mpnn_bwd_slots(n);
  const dim3 b(256);
  if (!arg_is_valid) {                                  // the forward pass did not record the winners: repeat its gather
    const dim3 ga((unsigned)((slots * nch + 3) / 4 * 4));
    if (nch == 1) hipLaunchKernelGGL((k_mpnn_bwd_arg<1, 8, true, true>), ga, b, 0, s, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, (int32_t*)arg);
    else hipLaunchKernelGGL((k_mpnn_bwd_arg<2, 8, true, true>), ga, b, 0, s, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, n, d, (int32_t*)arg);
  }
  hipLaunchKernelGGL((k_mpnn_bwd_dwe_max<8>), dim3((unsigned)slots), b, 0, s, dM, lddm, edge_attr_sorted, de, rowptr_t, node_order, n, d,
                     arg, dwe_partial);                 // (one partial per work-group: `slots` work-groups)
  if (n_edges > 0) {
    int64_t blocks = (n_edges + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL((k_mpnn_bwd_dea_max<8>), dim3((unsigned)blocks), b, 0, s, dM, lddm, We, ldwe, de, tgt_sorted, eloc_sorted, n_edges, d,
                       arg, d_edge_attr);
  }
  const dim3 g((unsigned)(n < 8192 ? (n + 3) / 4 : 2048));
  if (nch == 1) hipLaunchKernelGGL((k_mpnn_bwd_src_max16<1>), g, b, 0, s, dM, lddm, arg, rowptr_s, tnode, tloc, node_order, n, d, dQ, lddq, dq_absmax);
  else hipLaunchKernelGGL((k_mpnn_bwd_src_max16<2>), g, b, 0, s, dM, lddm, arg, rowptr_s, tnode, tloc, node_order, n, d, dQ, lddq, dq_absmax);
  hipLaunchKernelGGL(k_reduce_slots, dim3(rgnn_blocks((int64_t)d * de, 64)), dim3(1024), 0, s, dwe_partial, slots, (int64_t)d * de, dWe);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int32_t rgnn_linear_wgrad_slabs(int64_t m, int32_t n, int32_t k) {
  const int64_t tiles = (int64_t)((n + 127) / 128) * ((k + 127) / 128);
  int64_t slabs = (2048 + tiles - 1) / tiles;                  // ~2048 work-groups
  const int64_t max_slabs = (m + 255) / 256;                   // at least 256 rows per slab
  if (slabs > max_slabs) slabs = max_slabs;
  slabs = (slabs + 7) / 8 * 8;
  return (int32_t)(slabs < 8 ? 8 : slabs);
}

extern "C" int rgnn_linear_wgrad(const float* G, int64_t ldg, const float* A1, int64_t lda1, int32_t k1, const float* A2,
                                 int64_t lda2, int32_t k2, int64_t m, int32_t n, float* partial, float* dW,
                                 rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0 && n >= 0 && k1 >= 0 && k2 >= 0, "negative sizes");
  const int K = k1 + k2;
  if (n == 0 || K == 0) return RGNN_OK;
  RGNN_CHECK_ARG(G && dW && partial && (k1 == 0 || A1) && (k2 == 0 || A2), "null pointers");
  RGNN_CHECK_ARG(n % 4 == 0 && k1 % 4 == 0 && k2 % 4 == 0 && ldg % 4 == 0 && (k1 == 0 || lda1 % 4 == 0) &&
                     (k2 == 0 || lda2 % 4 == 0) &&
                     (((uintptr_t)G | (uintptr_t)(k1 ? A1 : G) | (uintptr_t)(k2 ? A2 : G)) & 15) == 0,
                 "widths / row strides must be multiples of 4 floats and the pointers 16-byte aligned");
  WgParams p;
  p.G = G; p.ldg = ldg; p.A1 = k1 ? A1 : A2; p.lda1 = k1 ? lda1 : lda2; p.k1 = k1; p.A2 = A2; p.lda2 = lda2; p.k2 = k2;
  p.m = m; p.n = n; p.part = partial;
  p.slabs = rgnn_linear_wgrad_slabs(m, n, K);
  p.rows_per_slab = ((m + p.slabs - 1) / p.slabs + 31) / 32 * 32;
  if (p.rows_per_slab < 32) p.rows_per_slab = 32;
  p.nt = (n + 127) / 128; p.kt = (K + 127) / 128;
  hipStream_t s = (hipStream_t)stream;
  const int64_t blocks = (int64_t)p.slabs * p.nt * p.kt;       // slabs is a multiple of 8: blockIdx -> (xcd, slab / 8, tile)
  hipLaunchKernelGGL(k_wgrad, dim3((unsigned)blocks), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_reduce_slots, dim3(rgnn_blocks((int64_t)n * K, 64)), dim3(1024), 0, s, partial, (int64_t)p.slabs,
                     (int64_t)n * K, dW);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_segment_reduce_bwd(const float* dM, int64_t lddm, const float* rows, int64_t ldr, const int32_t* rowptr_t,
                                       const int32_t* node_order, int64_t n, int32_t d, int32_t aggr, float* d_rows,
                                       int64_t lddr, rgnn_stream_t stream) {
  if (n == 0 || d == 0) return RGNN_OK;
  RGNN_CHECK_ARG(dM && rows && rowptr_t && d_rows, "null pointers");
  RGNN_CHECK_ARG(aggr >= 0 && aggr <= 2, "unknown aggregation");
  hipLaunchKernelGGL(k_segment_reduce_bwd, dim3(rgnn_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, dM, lddm, rows, ldr,
                     rowptr_t, node_order, n, d, aggr, d_rows, lddr);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
