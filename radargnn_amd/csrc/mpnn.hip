// Message passing: fused gather + per-edge linear + segmented reduction over a CSR keyed on the target node.
//
// Replaces MessagePassing.propagate of torch_geometric as used by gnn/mpnn_layers.py:88,94-101 (MPNNConv) and
// :173,179-184 (RadarPointGNNConv): index_select of both endpoint rows, torch.cat, a [E,D]x[D,D] addmm and a
// torch-scatter reduce.  With a single-Linear message function the node terms are hoisted out of the edge loop
// (see rgnn.h); what is left per edge is one D-wide row gather of Q (L2 resident: frames are independent and
// the XCD-aware chunking keeps a frame on one XCD), a de x D mat-vec held in registers, and the reduction.
//
// Work decomposition: one wave per contiguous chunk of target nodes, lanes across channels (4 consecutive
// channels per lane and pass -> 1 KiB coalesced row reads); the W_e block of the lane's channels lives in VGPRs
// for the whole chunk, edge attributes and CSR indices are wave-uniform (scalar loads).
#include "common.h"

namespace {

constexpr int MP_THREADS = 256;
constexpr int MP_WAVES = MP_THREADS / 64;

struct MpParams {
  const float* P; int64_t ldp; const float* p_bias;
  const float* Q; int64_t ldq;
  const float* We; int64_t ldwe;
  const float* ea; int de;
  const int32_t* rowptr; const int32_t* src;
  int64_t n; int d; int aggr; int relu;
  float* out; int64_t ldo;
  int64_t chunk;  // nodes per wave
};

// MODE 0: reduce into out[n, d];  MODE 1: store the per-edge hidden row (general pre_layers > 1 path)
template <int VEC, int NCH, int DEP, int MODE>
__global__ __launch_bounds__(MP_THREADS) void k_mpnn(const MpParams p) {
  const int lane = threadIdx.x & 63;
  // XCD-aware: consecutive chunks of nodes (hence frames) go to the same XCD (workgroup b runs on XCD b % 8)
  const int nb = gridDim.x;
  const int b = blockIdx.x;
  const int per_xcd = nb >> 3;  // grid is a multiple of 8
  const int vb = (b & 7) * per_xcd + (b >> 3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t gw = (int64_t)vb * MP_WAVES + wave;
  const int64_t node_beg = gw * p.chunk;
  int64_t node_end = node_beg + p.chunk;
  if (node_end > p.n) node_end = p.n;
  if (node_beg >= p.n) return;

  constexpr int PASS = 64 * VEC * NCH;  // channels per pass
  for (int c0 = 0; c0 < p.d; c0 += PASS) {
    // this lane's channels in this pass and its slice of W_e
    int ch[NCH];
    bool ok[NCH];
    float we[NCH][VEC][DEP];
#pragma unroll
    for (int t = 0; t < NCH; t++) {
      ch[t] = c0 + (lane + 64 * t) * VEC;
      ok[t] = ch[t] < p.d;  // d % VEC == 0 is guaranteed by the dispatcher
#pragma unroll
      for (int v = 0; v < VEC; v++)
#pragma unroll
        for (int k = 0; k < DEP; k++)
          we[t][v][k] = (ok[t] && k < p.de) ? p.We[(int64_t)(ch[t] + v) * p.ldwe + k] : 0.f;
    }
    for (int64_t node = node_beg; node < node_end; node++) {
      const int beg = p.rowptr[node], end = p.rowptr[node + 1];
      float acc[NCH][VEC];
      float pn[NCH][VEC];
#pragma unroll
      for (int t = 0; t < NCH; t++)
#pragma unroll
        for (int v = 0; v < VEC; v++) {
          acc[t][v] = (p.aggr == RGNN_AGGR_MAX) ? -INFINITY : 0.f;
          pn[t][v] = 0.f;
        }
      if (MODE == 1 || end > beg) {
#pragma unroll
        for (int t = 0; t < NCH; t++)
          if (ok[t]) {
#pragma unroll
            for (int v = 0; v < VEC; v++) {
              float x = p.p_bias ? p.p_bias[ch[t] + v] : 0.f;
              if (p.P) x += p.P[node * p.ldp + ch[t] + v];
              pn[t][v] = x;
            }
          }
      }
      for (int e = beg; e < end; e++) {
        const int s = p.src[e];
        float a[DEP];
#pragma unroll
        for (int k = 0; k < DEP; k++) a[k] = (k < p.de) ? p.ea[(int64_t)e * p.de + k] : 0.f;
#pragma unroll
        for (int t = 0; t < NCH; t++) {
          if (!ok[t]) continue;
          float q[VEC];
          if (VEC == 4) {
            const float4 q4 = *(const float4*)(p.Q + (int64_t)s * p.ldq + ch[t]);
            q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
          } else {
#pragma unroll
            for (int v = 0; v < VEC; v++) q[v] = p.Q[(int64_t)s * p.ldq + ch[t] + v];
          }
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            float r = q[v];
#pragma unroll
            for (int k = 0; k < DEP; k++) r = __builtin_fmaf(we[t][v][k], a[k], r);
            q[v] = r;
          }
          if (MODE == 0) {
#pragma unroll
            for (int v = 0; v < VEC; v++)
              acc[t][v] = (p.aggr == RGNN_AGGR_MAX) ? fmaxf(acc[t][v], q[v]) : acc[t][v] + q[v];
          } else {
            float* h = p.out + (int64_t)e * p.ldo + ch[t];
#pragma unroll
            for (int v = 0; v < VEC; v++) {
              float x = pn[t][v] + q[v];
              if (p.relu) x = fmaxf(x, 0.f);
              h[v] = x;
            }
          }
        }
      }
      if (MODE == 0) {
        const int cnt = end - beg;
#pragma unroll
        for (int t = 0; t < NCH; t++) {
          if (!ok[t]) continue;
          float o[VEC];
#pragma unroll
          for (int v = 0; v < VEC; v++) {
            float x = 0.f;  // empty segment -> exactly 0 (torch-scatter)
            if (cnt > 0) {
              if (p.aggr == RGNN_AGGR_MAX) x = pn[t][v] + acc[t][v];
              else if (p.aggr == RGNN_AGGR_MEAN) x = pn[t][v] + acc[t][v] / (float)cnt;
              else x = (float)cnt * pn[t][v] + acc[t][v];
            }
            o[v] = x;
          }
          float* dst = p.out + node * p.ldo + ch[t];
          if (VEC == 4) *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
          else
#pragma unroll
            for (int v = 0; v < VEC; v++) dst[v] = o[v];
        }
      }
    }
  }
}

template <int MODE>
int dispatch(MpParams& p, hipStream_t s) {
  // one wave per chunk; enough waves to fill 256 CUs several times over, few enough to amortise the W_e load
  const int64_t target_waves = 256 * 16;
  int64_t chunk = (p.n + target_waves - 1) / target_waves;
  if (chunk < 8) chunk = 8;
  p.chunk = chunk;
  const int64_t waves = (p.n + chunk - 1) / chunk;
  int64_t blocks = (waves + MP_WAVES - 1) / MP_WAVES;
  blocks = (blocks + 7) / 8 * 8;
  const bool vec4 = (p.d % 4 == 0) && (p.ldq % 4 == 0) && (p.ldo % 4 == 0) && (((uintptr_t)p.Q & 15) == 0) &&
                    (((uintptr_t)p.out & 15) == 0);
  const dim3 grid((unsigned)blocks), block(MP_THREADS);
#define RGNN_MP_LAUNCH(VEC, NCH, DEP) hipLaunchKernelGGL((k_mpnn<VEC, NCH, DEP, MODE>), grid, block, 0, s, p)
  if (vec4) {
    if (p.de <= 4) { if (p.d <= 256) RGNN_MP_LAUNCH(4, 1, 4); else RGNN_MP_LAUNCH(4, 2, 4); }
    else if (p.de <= 16) { if (p.d <= 256) RGNN_MP_LAUNCH(4, 1, 16); else RGNN_MP_LAUNCH(4, 2, 16); }
    else RGNN_MP_LAUNCH(4, 1, 32);
  } else {
    if (p.de <= 4) RGNN_MP_LAUNCH(1, 2, 4);
    else if (p.de <= 16) RGNN_MP_LAUNCH(1, 2, 16);
    else RGNN_MP_LAUNCH(1, 2, 32);
  }
#undef RGNN_MP_LAUNCH
  return 0;
}

// rows [E, d] in CSR order -> out [n, d]
__global__ __launch_bounds__(256) void k_segment_reduce(const float* __restrict__ rows, int64_t ldr,
                                                       const int32_t* __restrict__ rowptr, int64_t n, int d, int aggr,
                                                       float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n) return;
  const int beg = rowptr[node], end = rowptr[node + 1];
  for (int c = lane; c < d; c += 64) {
    float acc = (aggr == RGNN_AGGR_MAX) ? -INFINITY : 0.f;
    for (int e = beg; e < end; e++) {
      const float v = rows[(int64_t)e * ldr + c];
      acc = (aggr == RGNN_AGGR_MAX) ? fmaxf(acc, v) : acc + v;
    }
    float x = 0.f;
    if (end > beg) x = (aggr == RGNN_AGGR_MEAN) ? acc / (float)(end - beg) : acc;
    out[node * ldo + c] = x;
  }
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ in, int64_t ldi,
                                                    const int32_t* __restrict__ perm, int64_t n_rows, int width,
                                                    float* __restrict__ out, int64_t ldo) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * width) return;
  const int64_t r = idx / width;
  const int c = (int)(idx - r * width);
  out[r * ldo + c] = in[(int64_t)perm[r] * ldi + c];
}

int check_common(const float* Q, const float* We, const float* ea, int de, const int32_t* rowptr, const int32_t* src,
                 int64_t n, int d, int aggr) {
  RGNN_CHECK_ARG(n >= 0 && d >= 1, "bad sizes");
  RGNN_CHECK_ARG(aggr >= 0 && aggr <= 2, "aggr must be 0 (max), 1 (mean) or 2 (add)");
  RGNN_CHECK_ARG(Q && rowptr && src, "null pointers");
  RGNN_CHECK_ARG(de == 0 || (We && ea), "edge attributes given without weights");
  if (de > 32) {
    rgnn_set_error("fused message kernel supports at most 32 edge attributes (got %d)", de);
    return RGNN_ERR_UNSUPPORTED;
  }
  return RGNN_OK;
}

}  // namespace

extern "C" int rgnn_mpnn_aggregate(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                   const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                   const int32_t* rowptr_t, const int32_t* src_sorted, int64_t n, int32_t d,
                                   int32_t aggr, float* out, int64_t ldo, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  int rc = check_common(Q, We, edge_attr_sorted, de, rowptr_t, src_sorted, n, d, aggr);
  if (rc) return rc;
  RGNN_CHECK_ARG(out, "null out");
  MpParams p;
  p.P = P; p.ldp = ldp; p.p_bias = p_bias; p.Q = Q; p.ldq = ldq; p.We = We; p.ldwe = ldwe; p.ea = edge_attr_sorted;
  p.de = de; p.rowptr = rowptr_t; p.src = src_sorted; p.n = n; p.d = d; p.aggr = aggr; p.relu = 0; p.out = out;
  p.ldo = ldo;
  dispatch<0>(p, (hipStream_t)stream);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_mpnn_edge_hidden(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                     const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                     const int32_t* rowptr_t, const int32_t* src_sorted, int64_t n, int32_t d,
                                     int32_t relu, float* hidden, int64_t ldh, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  int rc = check_common(Q, We, edge_attr_sorted, de, rowptr_t, src_sorted, n, d, 0);
  if (rc) return rc;
  RGNN_CHECK_ARG(hidden, "null hidden");
  MpParams p;
  p.P = P; p.ldp = ldp; p.p_bias = p_bias; p.Q = Q; p.ldq = ldq; p.We = We; p.ldwe = ldwe; p.ea = edge_attr_sorted;
  p.de = de; p.rowptr = rowptr_t; p.src = src_sorted; p.n = n; p.d = d; p.aggr = 0; p.relu = relu; p.out = hidden;
  p.ldo = ldh;
  dispatch<1>(p, (hipStream_t)stream);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_segment_reduce(const float* rows, int64_t ldr, const int32_t* rowptr_t, int64_t n, int32_t d,
                                   int32_t aggr, float* out, int64_t ldo, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rowptr_t && out && d >= 1 && aggr >= 0 && aggr <= 2, "bad arguments");
  hipLaunchKernelGGL(k_segment_reduce, dim3(rgnn_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, rows, ldr, rowptr_t, n,
                     d, aggr, out, ldo);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_gather_rows_f32(const float* in, int64_t ldi, const int32_t* perm, int64_t n_rows, int32_t width,
                                    float* out, int64_t ldo, rgnn_stream_t stream) {
  if (n_rows == 0 || width == 0) return RGNN_OK;
  RGNN_CHECK_ARG(in && perm && out, "null pointers");
  hipLaunchKernelGGL(k_gather_rows, dim3(rgnn_blocks(n_rows * width, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi,
                     perm, n_rows, width, out, ldo);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
