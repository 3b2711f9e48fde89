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
#include <stdlib.h>

#ifndef RGNN_MPNN_NT_STORE
#define RGNN_MPNN_NT_STORE 1   // the aggregated rows leave with streaming stores (they are not read again here; kept out of L2 they leave it to the rows of Q: -27 % HBM reads, -5 % time)
#endif
#ifndef RGNN_MPNN_SCALAR_EA
#define RGNN_MPNN_SCALAR_EA 0  // k_mpnn_max: an edge's attribute row through the scalar cache (s_load_dwordx8, one edge ahead) instead of a 64-edge
                               // block in VGPRs + 8 v_readlane per edge: 54 -> 46 VALU per edge, bit-identical, and SLOWER -- 203 -> 212 us on the C2
                               // graph, 539 -> 631 us at k = 20 (tools/mpnn_bench.py, same call): the kernel is not bound by its VALU issue slots
#endif
#ifndef RGNN_MPNN_NT_LOAD
#define RGNN_MPNN_NT_LOAD 0    // streaming loads for the edge stream (sources, attributes): -4 % in tools/mpnn_bench.py, nothing inside the step
#endif
#ifndef RGNN_MPNN_ABL
#define RGNN_MPNN_ABL 0     // experiments only (wrong results): 1 cache-resident gathers, 2 one of the eight FMA terms, 4 no stores
#endif

namespace {

constexpr int RGNN_MPNN_QUEUE_INTS = 8 * 8 * 16;  // ticket counters: up to 8 channel blocks x 8 XCDs, 64 B apart
constexpr int MP_THREADS = 256;
constexpr int MP_WAVES = MP_THREADS / 64;

// (the builtin's result must be received in a GCC-style vector, see linear_common.h)
typedef unsigned int mp_u32x4 __attribute__((__vector_size__(16)));
typedef float mp_f32x4 __attribute__((ext_vector_type(4)));
typedef float mp_f32x2 __attribute__((ext_vector_type(2)));
// fmaxf semantics (a NaN operand loses) in ONE v_max_f32: the accumulator and the fused multiply-add results are already
// canonical, the extra `v_max_f32 x, x, x` hipcc puts in front of every fmaxf to quiet signalling NaNs buys nothing here
__device__ __forceinline__ float mp_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float4 mp_buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const mp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  const mp_f32x4 f = __builtin_bit_cast(mp_f32x4, v);
  return make_float4(f.x, f.y, f.z, f.w);
}

struct MpParams {
  const float* P; int64_t ldp; const float* p_bias;
  const float* Q; int64_t ldq;
  const float* We; int64_t ldwe;
  const float* ea; int de;
  const int32_t* rowptr; const int32_t* src; const int32_t* order;
  int64_t n; int d; int aggr; int relu;
  float* out; int64_t ldo;
  int64_t chunk;  // nodes per wave (generic kernel)
  const int32_t* chunk_start; int n_chunks;  // work-balanced chunks (fast kernel); 1024 ticket ints follow the table
  int skip_empty;                            // leave the rows of targets without incoming edges unwritten
  uint16_t* arg_out;                         // max kernel: record the winning edge per (target, channel); NULL = not wanted
  float* out_absmax;                         // optional device word: atomic max of |out| over the rows written (f16x2 dense form)
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
    for (int64_t pos = node_beg; pos < node_end; pos++) {
      const int64_t node = p.order ? (int64_t)p.order[pos] : pos;
      const int beg = p.rowptr[pos], end = p.rowptr[pos + 1];  // CSR is laid out in visiting order
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


// ------------------------------------------------------------------------------------------------
// Fast path (d % 4 == 0).  The edge loop of the simple kernel above is latency bound: every edge costs a dependent
// chain  index load -> 1.8 KB row gather -> FMAs, and with the W_e slice of 2 x 4 channels in registers only two
// waves fit on a SIMD.  This version
//   * splits the channels over workgroups (blockIdx.y: 256 channels = one float4 per lane) -> 64 weight VGPRs,
//     ~100 VGPRs in total, 4-5 waves per SIMD;
//   * takes the CSR indices and edge attributes of a node with ONE coalesced load each (lane j holds src[beg+j],
//     lane l holds a[l / DEP][l % DEP]) and broadcasts them with v_readlane -> no dependent scalar loads;
//   * software-pipelines the row gathers two edges ahead (4 rows in flight per wave);
//   * visits the targets in `order` (grid-cell order): neighbouring targets share sources -> the gathers hit L2.
// ------------------------------------------------------------------------------------------------
// (at least 3 waves per SIMD: the D = 464 instance wants 175 VGPRs, seven more than three resident waves leave it)
#ifndef RGNN_MPNN_WAVES
#define RGNN_MPNN_WAVES __attribute__((amdgpu_waves_per_eu(3)))
#endif
template <int NCH, int DEP, int MODE>
__global__ __launch_bounds__(MP_THREADS) RGNN_MPNN_WAVES void k_mpnn_fast(const float* __restrict__ P, int64_t ldp,
                                                         const float* __restrict__ p_bias,
                                                         const float* __restrict__ Q, int64_t ldq,
                                                         const float* __restrict__ We, int64_t ldwe,
                                                         const float* __restrict__ ea, int de,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ chunk_start, int n_chunks,
                                                         int32_t* __restrict__ queue, int64_t n, int d, int aggr,
                                                         int relu, float* __restrict__ out, int64_t ldo) {
  constexpr int EPL = 64 / DEP;  // edges whose attributes fit one 64-lane load
  const int lane = threadIdx.x & 63;
  // Persistent waves pull chunks from a per-XCD ticket counter: XCD x (= workgroup id % 8, observed placement) owns the
  // contiguous chunk range [x, x+1) * n_chunks / 8 -- neighbouring targets stay on one L2 -- and inside it the waves
  // take the next chunk as they finish the previous one, so there is no tail of idle SIMDs (static assignment ran at
  // 1.7 of 3 possible waves per SIMD on average).
  const int xcd = blockIdx.x & 7;
  const int c_lo = (int)((int64_t)n_chunks * xcd / 8), c_hi = (int)((int64_t)n_chunks * (xcd + 1) / 8);
  int32_t* ticket = queue + (blockIdx.y * 8 + xcd) * 16;  // one counter per (channel block, XCD), 64 B apart

  // this lane's channels: NCH groups of 4, group t at channel blockIdx.y*256*NCH + (lane + 64 t) * 4
  int ch[NCH];
  bool ok[NCH];
  float4 we[NCH][DEP];
#pragma unroll
  for (int t = 0; t < NCH; t++) {
    const int c = (blockIdx.y * NCH + t) * 256 + lane * 4;
    ok[t] = c < d;
    ch[t] = ok[t] ? c : 0;
#pragma unroll
    for (int k = 0; k < DEP; k++) {
      const bool kk = ok[t] && k < de;
      we[t][k].x = kk ? We[(int64_t)(ch[t] + 0) * ldwe + k] : 0.f;
      we[t][k].y = kk ? We[(int64_t)(ch[t] + 1) * ldwe + k] : 0.f;
      we[t][k].z = kk ? We[(int64_t)(ch[t] + 2) * ldwe + k] : 0.f;
      we[t][k].w = kk ? We[(int64_t)(ch[t] + 3) * ldwe + k] : 0.f;
    }
  }
  const int ea_edge = lane / DEP, ea_k = lane % DEP;  // which attribute this lane fetches in an attribute block

  for (;;) {
    int cidx = 0;
    if (lane == 0) cidx = c_lo + atomicAdd(ticket, 1);
    cidx = __builtin_amdgcn_readfirstlane(cidx);
    if (cidx >= c_hi) {
      // this wave is done with the queue; the last of the queue's waves (all work-groups of this XCD and channel block)
      // puts both counters back to zero for the next launch
      if (lane == 0) {
        const int waves = (int)(gridDim.x >> 3) * MP_WAVES;
        if (atomicAdd(ticket + 1, 1) == waves - 1) { ticket[0] = 0; ticket[1] = 0; }
      }
      break;
    }
    const int pos_beg = __builtin_amdgcn_readfirstlane(chunk_start[cidx]);
    const int cn = __builtin_amdgcn_readfirstlane(chunk_start[cidx + 1]) - pos_beg;  // targets of this chunk (<= 63)
    if (cn <= 0) continue;
    // the chunk's CSR slice: lane i holds rowptr[pos_beg + i] (i <= cn) and the node visited at position pos_beg + i
    const int my_rp = rowptr[pos_beg + min(lane, cn)];
    const int my_node = order ? order[pos_beg + min(lane, cn - 1)] : (pos_beg + min(lane, cn - 1));
    const int e_lo = __builtin_amdgcn_readlane(my_rp, 0);
    const int e_hi = __builtin_amdgcn_readlane(my_rp, cn);

    // node cursor (all wave-uniform)
    int ni = 0, node = 0, node_end = 0, cnt = 0;
    float4 pn[NCH], acc[NCH];
    const float init = (aggr == RGNN_AGGR_MAX) ? -INFINITY : 0.f;
    auto open_node = [&](int i) {
      node = __builtin_amdgcn_readlane(my_node, i);
      node_end = __builtin_amdgcn_readlane(my_rp, i + 1);
      cnt = node_end - __builtin_amdgcn_readlane(my_rp, i);
  #pragma unroll
      for (int t = 0; t < NCH; t++) {
        acc[t] = make_float4(init, init, init, init);
        pn[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cnt > 0) {
          if (p_bias) pn[t] = *(const float4*)(p_bias + ch[t]);
          if (P) {
            const float4 x = *(const float4*)(P + (int64_t)node * ldp + ch[t]);
            pn[t].x += x.x; pn[t].y += x.y; pn[t].z += x.z; pn[t].w += x.w;
          }
        }
      }
    };
    auto close_node = [&]() {
      if (MODE != 0) return;
      const float fc = (float)cnt;
  #pragma unroll
      for (int t = 0; t < NCH; t++) {
        if (!ok[t]) continue;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);  // empty segment -> exactly 0 (torch-scatter)
        if (cnt > 0) {
          if (aggr == RGNN_AGGR_MAX)
            o = make_float4(pn[t].x + acc[t].x, pn[t].y + acc[t].y, pn[t].z + acc[t].z, pn[t].w + acc[t].w);
          else if (aggr == RGNN_AGGR_MEAN)
            o = make_float4(pn[t].x + acc[t].x / fc, pn[t].y + acc[t].y / fc, pn[t].z + acc[t].z / fc,
                            pn[t].w + acc[t].w / fc);
          else
            o = make_float4(fc * pn[t].x + acc[t].x, fc * pn[t].y + acc[t].y, fc * pn[t].z + acc[t].z,
                            fc * pn[t].w + acc[t].w);
        }
        *(float4*)(out + (int64_t)node * ldo + ch[t]) = o;
      }
    };
    open_node(0);

    // flat edge stream [e_lo, e_hi): 64 indices per coalesced load, row gathers two edges ahead, across node borders
    int src_cur = (e_lo + lane < e_hi) ? src[e_lo + lane] : 0;
    float4 qa[NCH], qb[NCH];
    float my_ea = 0.f;
    for (int eb = e_lo; eb < e_hi; eb += 64) {
      const int src_nxt = (eb + 64 + lane < e_hi) ? src[eb + 64 + lane] : 0;
      auto row_of = [&](int j, float4* q) {  // Q row of edge eb + j (clamped to the stream; surplus gathers are discarded)
        const int jj = min(j, e_hi - 1 - eb);
        const int s = (jj < 64) ? __builtin_amdgcn_readlane(src_cur, jj) : __builtin_amdgcn_readlane(src_nxt, jj - 64);
  #pragma unroll
        for (int t = 0; t < NCH; t++) q[t] = *(const float4*)(Q + (int64_t)s * ldq + ch[t]);
      };
      auto attr_block = [&](int blk) {  // attributes of edges eb + [blk*EPL, blk*EPL+EPL), one value per lane
        const int e = eb + blk * EPL + ea_edge;
        return (e < e_hi && ea_k < de) ? ea[(int64_t)e * de + ea_k] : 0.f;
      };
      if (eb == e_lo) {
        row_of(0, qa); row_of(1, qb);
        my_ea = attr_block(0);
      }
      const int nbk = min(64, e_hi - eb);
      for (int j = 0; j < nbk; j += 2) {
        float4 qc[NCH], qd[NCH];
        row_of(j + 2, qc); row_of(j + 3, qd);
        float ea_next = my_ea;
        if (((j + 2) % EPL) == 0) ea_next = attr_block((j + 2) / EPL);
  #pragma unroll
        for (int u = 0; u < 2; u++) {
          const int e = eb + j + u;
          if (e >= e_hi) break;
          while (e >= node_end) {  // crossed into the next target (possibly over empty ones)
            close_node();
            ni++;
            open_node(ni);
          }
          float4 q[NCH];
  #pragma unroll
          for (int t = 0; t < NCH; t++) q[t] = (u == 0) ? qa[t] : qb[t];
          const int lane0 = ((j + u) % EPL) * DEP;
  #pragma unroll
          for (int k = 0; k < DEP; k++) {
            const float ak = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_ea), lane0 + k));
  #pragma unroll
            for (int t = 0; t < NCH; t++) {
              q[t].x = __builtin_fmaf(we[t][k].x, ak, q[t].x); q[t].y = __builtin_fmaf(we[t][k].y, ak, q[t].y);
              q[t].z = __builtin_fmaf(we[t][k].z, ak, q[t].z); q[t].w = __builtin_fmaf(we[t][k].w, ak, q[t].w);
            }
          }
  #pragma unroll
          for (int t = 0; t < NCH; t++) {
            if (MODE == 0) {
              if (aggr == RGNN_AGGR_MAX) {
                acc[t].x = fmaxf(acc[t].x, q[t].x); acc[t].y = fmaxf(acc[t].y, q[t].y);
                acc[t].z = fmaxf(acc[t].z, q[t].z); acc[t].w = fmaxf(acc[t].w, q[t].w);
              } else {
                acc[t].x += q[t].x; acc[t].y += q[t].y; acc[t].z += q[t].z; acc[t].w += q[t].w;
              }
            } else if (ok[t]) {
              float4 h = make_float4(pn[t].x + q[t].x, pn[t].y + q[t].y, pn[t].z + q[t].z, pn[t].w + q[t].w);
              if (relu) { h.x = fmaxf(h.x, 0.f); h.y = fmaxf(h.y, 0.f); h.z = fmaxf(h.z, 0.f); h.w = fmaxf(h.w, 0.f); }
              *(float4*)(out + (int64_t)e * ldo + ch[t]) = h;
            }
          }
        }
  #pragma unroll
        for (int t = 0; t < NCH; t++) { qa[t] = qc[t]; qb[t] = qd[t]; }
        my_ea = ea_next;
      }
      src_cur = src_nxt;
    }
    // the target that was open when the stream ended, then any trailing targets without edges
    for (;;) {
      close_node();
      if (++ni >= cn) break;
      open_node(ni);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The shipped case -- max aggregation, target term folded away (no per-target P rows), reduce into out[n, d] -- as its own
// kernel.  Same work decomposition and arithmetic as k_mpnn_fast (bit-identical results); what changes is what the wave
// waits for.  gfx950 has ONE in-order counter for vector memory: a wait for the row gathered for edge j also waits for
// every load issued before it.  In k_mpnn_fast
//   * opening a target loaded its bias / P row and hipcc put `s_waitcnt vmcnt(0)` in front of the first use: every target
//     boundary (six edges on average) drained the gathers prefetched for the next edges;
//   * the attributes of eight edges at a time came in through one small load whose miss latency every younger gather then
//     inherited, and the register copies that rotated the prefetched rows (qc -> qa) forced `vmcnt(0)` at the end of every
//     iteration.
// Here the bias sits in registers for the whole launch, a 64-edge block's indices AND attributes arrive with three loads
// issued one block ahead (lane j holds edge j's attributes; v_readlane broadcasts them), and the edge loop is unrolled so
// that the two row-register sets swap roles instead of being copied.  With every gather served by the cache the old kernel
// ran at 232 us of its 263 us (tools/mpnn_bench.py): latency structure, not bandwidth, was the bound.
template <int NCH, int DEP, bool ARG = false, bool AMAX = false>
__global__ __launch_bounds__(MP_THREADS) RGNN_MPNN_WAVES void k_mpnn_max(const float* __restrict__ p_bias,
                                                        const float* __restrict__ Q, int64_t ldq,
                                                        const float* __restrict__ We, int64_t ldwe,
                                                        const float* __restrict__ ea, int de,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ src,
                                                        const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ chunk_start, int n_chunks,
                                                        int32_t* __restrict__ queue, int64_t n, int d,
                                                        float* __restrict__ out, int64_t ldo, int q_bytes,
                                                        int skip_empty, uint16_t* __restrict__ arg_out = nullptr,
                                                        float* __restrict__ out_absmax = nullptr) {
  // ARG (training): also records, per target and channel, the FIRST edge that attains the maximum (torch-scatter's arg_out
  // convention) as its index INSIDE the target's segment (arg_out uint16 [n, d]: a quarter of the bytes of an edge position,
  // and the backward kernels are bound by reading these rows) -- the backward pass then routes the gradient to exactly the
  // edge this kernel's arithmetic chose instead of repeating the gather (rgnn_mpnn_max_bwd).
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7;
  const int c_lo = (int)((int64_t)n_chunks * xcd / 8), c_hi = (int)((int64_t)n_chunks * (xcd + 1) / 8);
  int32_t* ticket = queue + (blockIdx.y * 8 + xcd) * 16;
  // rows of Q come in through a buffer descriptor: the row offset is wave-uniform (an SGPR), the lane's channel offset a
  // constant VGPR -- no per-gather 64-bit address arithmetic, and no address temporaries for hipcc to alias with row
  // registers whose loads are still in flight (that aliasing cost a vmcnt(0) per iteration)
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Q, (short)0, q_bytes, 0x00020000);
  const int ldq4 = (int)ldq * 4;

  int ch[NCH];
  bool ok[NCH];
  float4 we[NCH][DEP], bias[NCH];
#pragma unroll
  for (int t = 0; t < NCH; t++) {
    const int c = (blockIdx.y * NCH + t) * 256 + lane * 4;
    ok[t] = c < d;
    ch[t] = ok[t] ? c : 0;
#pragma unroll
    for (int k = 0; k < DEP; k++) {
      const bool kk = ok[t] && k < de;
      we[t][k].x = kk ? We[(int64_t)(ch[t] + 0) * ldwe + k] : 0.f;
      we[t][k].y = kk ? We[(int64_t)(ch[t] + 1) * ldwe + k] : 0.f;
      we[t][k].z = kk ? We[(int64_t)(ch[t] + 2) * ldwe + k] : 0.f;
      we[t][k].w = kk ? We[(int64_t)(ch[t] + 3) * ldwe + k] : 0.f;
    }
    bias[t] = (p_bias && ok[t]) ? *(const float4*)(p_bias + ch[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float amax = 0.f;                                    // AMAX: |out| seen by this lane (the bound of the update GEMM's A2 operand)

  for (;;) {
    int cidx = 0;
    if (lane == 0) cidx = c_lo + atomicAdd(ticket, 1);
    cidx = __builtin_amdgcn_readfirstlane(cidx);
    if (cidx >= c_hi) {
      if (lane == 0) {
        const int waves = (int)(gridDim.x >> 3) * MP_WAVES;
        if (atomicAdd(ticket + 1, 1) == waves - 1) { ticket[0] = 0; ticket[1] = 0; }
      }
      break;
    }
    const int pos_beg = __builtin_amdgcn_readfirstlane(chunk_start[cidx]);
    const int cn = __builtin_amdgcn_readfirstlane(chunk_start[cidx + 1]) - pos_beg;
    if (cn <= 0) continue;
    const int my_rp = rowptr[pos_beg + min(lane, cn)];
    const int my_node = order ? order[pos_beg + min(lane, cn - 1)] : (pos_beg + min(lane, cn - 1));
    const int e_lo = __builtin_amdgcn_readlane(my_rp, 0);
    const int e_hi = __builtin_amdgcn_readlane(my_rp, cn);

    int ni = 0, node = 0, node_end = 0, cnt = 0;     // target cursor (wave-uniform)
    float4 acc[NCH];
    int4 win[NCH];
    auto open_node = [&](int i) {
      node = __builtin_amdgcn_readlane(my_node, i);
      node_end = __builtin_amdgcn_readlane(my_rp, i + 1);
      cnt = node_end - __builtin_amdgcn_readlane(my_rp, i);
#pragma unroll
      for (int t = 0; t < NCH; t++) {
        acc[t] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (ARG) win[t] = make_int4(-1, -1, -1, -1);
      }
    };
    auto close_node = [&]() {
      if (cnt == 0 && skip_empty) return;             // (the caller never reads the rows of targets without edges)
#pragma unroll
      for (int t = 0; t < NCH; t++) {
        if (!ok[t]) continue;
        if (ARG && cnt > 0) {
          uint2 pk;
          pk.x = (unsigned)(win[t].x & 0xffff) | ((unsigned)win[t].y << 16);
          pk.y = (unsigned)(win[t].z & 0xffff) | ((unsigned)win[t].w << 16);
          *(uint2*)(arg_out + (int64_t)node * d + ch[t]) = pk;
        }
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);   // empty segment -> exactly 0 (torch-scatter)
        if (cnt > 0) o = make_float4(bias[t].x + acc[t].x, bias[t].y + acc[t].y, bias[t].z + acc[t].z, bias[t].w + acc[t].w);
        if (AMAX) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        if (!(RGNN_MPNN_ABL & 4) || node == 0) {
#if RGNN_MPNN_NT_STORE
          // streaming store: the aggregated rows are not read again by this kernel and should not push rows of Q out of L2
          __builtin_nontemporal_store(o.x, out + (int64_t)node * ldo + ch[t] + 0);
          __builtin_nontemporal_store(o.y, out + (int64_t)node * ldo + ch[t] + 1);
          __builtin_nontemporal_store(o.z, out + (int64_t)node * ldo + ch[t] + 2);
          __builtin_nontemporal_store(o.w, out + (int64_t)node * ldo + ch[t] + 3);
#else
          *(float4*)(out + (int64_t)node * ldo + ch[t]) = o;
#endif
        }
      }
    };
    open_node(0);

    // block of 64 edges: lane j holds the source and the DEP attributes of edge eb + j
#if RGNN_MPNN_NT_LOAD
    auto load_src = [&](int eb) { return (eb + lane < e_hi) ? __builtin_nontemporal_load(src + eb + lane) : 0; };
#else
    auto load_src = [&](int eb) { return (eb + lane < e_hi) ? src[eb + lane] : 0; };
#endif
    auto load_ea = [&](int eb, float (&a)[DEP]) {
      const int e = eb + lane;
      if (DEP == 8 && de == 8) {                       // (rows of 32 bytes: two 16-byte loads)
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
#if RGNN_MPNN_NT_LOAD
        if (e < e_hi) {
          const float* q_ = ea + (int64_t)e * 8;
          lo = make_float4(__builtin_nontemporal_load(q_), __builtin_nontemporal_load(q_ + 1), __builtin_nontemporal_load(q_ + 2), __builtin_nontemporal_load(q_ + 3));
          hi = make_float4(__builtin_nontemporal_load(q_ + 4), __builtin_nontemporal_load(q_ + 5), __builtin_nontemporal_load(q_ + 6), __builtin_nontemporal_load(q_ + 7));
        }
#else
        if (e < e_hi) { lo = *(const float4*)(ea + (int64_t)e * 8); hi = *(const float4*)(ea + (int64_t)e * 8 + 4); }
#endif
        a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w;
        if (DEP == 8) { a[4 % DEP] = hi.x; a[5 % DEP] = hi.y; a[6 % DEP] = hi.z; a[7 % DEP] = hi.w; }
      } else {
#pragma unroll
        for (int k = 0; k < DEP; k++) a[k] = (e < e_hi && k < de) ? ea[(int64_t)e * de + k] : 0.f;
      }
    };
    int src_cur = load_src(e_lo);
    float ea_cur[DEP];
    if (!RGNN_MPNN_SCALAR_EA) load_ea(e_lo, ea_cur);
    // RGNN_MPNN_SCALAR_EA (r03 experiment, off): an edge belongs to ONE wave, so the address of its attribute row is wave-uniform
    // -- the row can come in through the scalar cache, one edge ahead, and the per-edge FMAs take it straight from SGPRs
    float a_nx[DEP];
    auto sload_ea = [&](int e, float (&a)[DEP]) {
      const int ee = __builtin_amdgcn_readfirstlane(min(e, e_hi - 1));
      const float* r = ea + (int64_t)ee * de;
      if (de == DEP) {
#pragma unroll
        for (int k = 0; k < DEP; k++) a[k] = r[k];
      } else {
#pragma unroll
        for (int k = 0; k < DEP; k++) a[k] = (k < de) ? r[k] : 0.f;
      }
    };
    if (RGNN_MPNN_SCALAR_EA) sload_ea(e_lo, a_nx);
    float4 qa[NCH], qb[NCH], qc[NCH], qd[NCH];
    // A block holds 64 edges in its registers but only 60 are consumed before the next block takes over: the gathers run up
    // to five edges ahead (j + 5 <= 61), so they never need the NEXT block's indices.  (Reading a lane of a register whose
    // load is still in flight makes hipcc wait for it with vmcnt(0) -- on every iteration, draining all prefetched rows.)
    constexpr int BLK = 60;
    for (int eb = e_lo; eb < e_hi; eb += BLK) {
      const int src_nxt = load_src(eb + BLK);         // (issued before this block's gathers: the first gather wait absorbs them)
      float ea_nxt[DEP];
      if (!RGNN_MPNN_SCALAR_EA) load_ea(eb + BLK, ea_nxt);
      auto row_of = [&](int j, float4* q) {           // Q row of edge eb + j (clamped to the stream; surplus gathers are discarded)
        const int jj = min(j, e_hi - 1 - eb);
        int s_ = __builtin_amdgcn_readlane(src_cur, jj);
        if (RGNN_MPNN_ABL & 1) s_ &= 63;             // experiment: every gather hits the same 64 rows
        const int soff = s_ * ldq4;
#pragma unroll
        for (int t = 0; t < NCH; t++) q[t] = mp_buf_load16(rq, ch[t] * 4, soff);
      };
      auto edge = [&](int j, const float4* qin) {     // edge eb + j, its row in qin
        const int e = eb + j;
        while (e >= node_end) {                        // crossed into the next target (possibly over empty ones)
          close_node();
          ni++;
          open_node(ni);
        }
        // q = row + W_e a, channel pairs as <2 x float>: v_pk_fma_f32 does two lanes' worth of FMA per issue slot (hipcc packs
        // the float4 form of k_mpnn_fast by itself but not this loop: 64 v_fmac_f32 per edge instead of 32 v_pk_fma_f32).
        // Same per-element operation order (k ascending, one fused multiply-add each): same bits.
        mp_f32x2 q[NCH][2];
#pragma unroll
        for (int t = 0; t < NCH; t++) {
          q[t][0] = mp_f32x2{qin[t].x, qin[t].y};
          q[t][1] = mp_f32x2{qin[t].z, qin[t].w};
        }
        float aks[DEP];                               // (all broadcasts first: a v_readlane result needs wait states before a VALU
#pragma unroll                                        //  instruction may read it, and one s_nop per attribute is an issue slot each)
        for (int k = 0; k < DEP; k++) {
          if (RGNN_MPNN_SCALAR_EA) aks[k] = a_nx[k];
          else aks[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ea_cur[k]), j));
        }
        if (RGNN_MPNN_SCALAR_EA) sload_ea(e + 1, a_nx);   // (the next edge of the stream, whichever block or target it is in)
#pragma unroll
        for (int k = 0; k < ((RGNN_MPNN_ABL & 2) ? 1 : DEP); k++) {
          const mp_f32x2 a2 = mp_f32x2{aks[k], aks[k]};
#pragma unroll
          for (int t = 0; t < NCH; t++) {
            q[t][0] = __builtin_elementwise_fma(mp_f32x2{we[t][k].x, we[t][k].y}, a2, q[t][0]);
            q[t][1] = __builtin_elementwise_fma(mp_f32x2{we[t][k].z, we[t][k].w}, a2, q[t][1]);
          }
        }
#pragma unroll
        for (int t = 0; t < NCH; t++) {
          if (ARG) {                                  // strictly greater: the first edge that attains the maximum keeps it
            const int el = e - (node_end - cnt);      // index inside the segment (wave-uniform)
            if (q[t][0].x > acc[t].x) { acc[t].x = q[t][0].x; win[t].x = el; }
            if (q[t][0].y > acc[t].y) { acc[t].y = q[t][0].y; win[t].y = el; }
            if (q[t][1].x > acc[t].z) { acc[t].z = q[t][1].x; win[t].z = el; }
            if (q[t][1].y > acc[t].w) { acc[t].w = q[t][1].y; win[t].w = el; }
          } else {
            acc[t].x = mp_max(acc[t].x, q[t][0].x); acc[t].y = mp_max(acc[t].y, q[t][0].y);
            acc[t].z = mp_max(acc[t].z, q[t][1].x); acc[t].w = mp_max(acc[t].w, q[t][1].y);
          }
        }
      };
      if (eb == e_lo) { row_of(0, qa); row_of(1, qb); }
      const int nbk = min(BLK, e_hi - eb);
      for (int j = 0; j < nbk; j += 4) {
        row_of(j + 2, qc); row_of(j + 3, qd);          // two edges ahead, across target and block borders
        edge(j, qa);
        if (j + 1 < nbk) edge(j + 1, qb);
        if (j + 2 >= nbk) break;
        row_of(j + 4, qa); row_of(j + 5, qb);          // (the register sets swap roles: no copies, no wait forced by a copy)
        edge(j + 2, qc);
        if (j + 3 < nbk) edge(j + 3, qd);
      }
      src_cur = src_nxt;
      if (!RGNN_MPNN_SCALAR_EA) {
#pragma unroll
        for (int k = 0; k < DEP; k++) ea_cur[k] = ea_nxt[k];
      }
    }
    for (;;) {                                         // the target that was open when the stream ended, then trailing empty ones
      close_node();
      if (++ni >= cn) break;
      open_node(ni);
    }
  }
  if (AMAX) {                                          // one atomic per wave, spread over the slots of the bound (rgnn.h)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0)
      atomicMax((unsigned int*)out_absmax + (((blockIdx.y * gridDim.x + blockIdx.x) * MP_WAVES + (threadIdx.x >> 6)) & (RGNN_BOUND_SLOTS - 1)),
                __float_as_uint(amax));
  }
}

// |out| of a whole [n, d] matrix into a device word (the kernels that do not track it themselves)
__global__ __launch_bounds__(256) void k_absmax_matrix(const float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                      float* __restrict__ word) {
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * d; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / d;
    m = fmaxf(m, fabsf(x[r * ldx + (idx - r * d)]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)word + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(m));
}

// Work-balanced chunking of the visiting sequence: chunk c covers positions [chunk_start[c], chunk_start[c+1]) with
// about `work` units of (edges + 2 * targets) each -> equal wave run times although in-degrees are very uneven in
// grid-cell order (35 points of one cluster next to each other, then isolated clutter).  <= work/2 < 64 targets.
__global__ __launch_bounds__(256) void k_partition(const int32_t* __restrict__ rowptr, int64_t n, int work, int alpha, int n_chunks,
                                                  int32_t* __restrict__ chunk_start) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < RGNN_MPNN_QUEUE_INTS) chunk_start[n_chunks + 1 + c] = 0;   // ticket counters behind the table start at zero
  if (c > n_chunks) return;
  if (c == n_chunks) { chunk_start[c] = (int32_t)n; return; }
  const int64_t target = (int64_t)c * work;
  int64_t lo = 0, hi = n;  // first position p with rowptr[p] + 2p >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)rowptr[mid] + (int64_t)alpha * mid < target) lo = mid + 1; else hi = mid;
  }
  chunk_start[c] = (int32_t)lo;
}

template <int MODE>
int dispatch(MpParams& p, hipStream_t s) {
  const bool vec4 = (p.d % 4 == 0) && (p.ldq % 4 == 0) && (p.ldo % 4 == 0) && (((uintptr_t)p.Q & 15) == 0) &&
                    (((uintptr_t)p.out & 15) == 0) && (p.P == nullptr || ((p.ldp % 4 == 0) && (((uintptr_t)p.P & 15) == 0))) &&
                    (p.p_bias == nullptr || (((uintptr_t)p.p_bias & 15) == 0));
  if (vec4 && p.chunk_start != nullptr) {
    const int nch = (p.de <= 8 && p.d > 256 && RGNN_ENV("RGNN_MPNN_NCH1") == nullptr) ? 2 : 1;  // 64 weight registers either way
    const unsigned ny = (unsigned)((p.d + 256 * nch - 1) / (256 * nch));
    int64_t blocks = (p.n_chunks + MP_WAVES - 1) / MP_WAVES;
    static const int per_cu = RGNN_ENV("RGNN_MPNN_WG_PER_CU") ? atoi(RGNN_ENV("RGNN_MPNN_WG_PER_CU")) : 3;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;  // persistent: 3 workgroups of 4 waves per CU
    blocks = (blocks + 7) / 8 * 8;
    const dim3 grid((unsigned)blocks, ny), block(MP_THREADS);
    // ticket counters live behind the chunk table: zeroed by rgnn_mpnn_partition, and every launch leaves them zero again
    // (the last wave of a queue to run dry resets it) -- a memset per launch was three fill kernels (unaligned head /
    // body / tail), 15 us per layer
    int32_t* queue = const_cast<int32_t*>(p.chunk_start) + p.n_chunks + 1;
    // max aggregation without per-target rows: k_mpnn_max
    const int64_t q_bytes = ((p.n - 1) * p.ldq + p.d) * 4;   // (sources are node ids < n)
    if (MODE == 0 && p.aggr == RGNN_AGGR_MAX && p.P == nullptr && p.de <= 8 && q_bytes < ((int64_t)1 << 31) &&
        RGNN_ENV("RGNN_MPNN_NOSPEC") == nullptr) {
#define RGNN_MPX(NCH, DEP)                                                                                          \
  do {                                                                                                              \
    if (p.arg_out && p.out_absmax)                                                                                  \
      hipLaunchKernelGGL((k_mpnn_max<NCH, DEP, true, true>), grid, block, 0, s, p.p_bias, p.Q, p.ldq, p.We, p.ldwe, p.ea, p.de, p.rowptr, \
                         p.src, p.order, p.chunk_start, p.n_chunks, queue, p.n, p.d, p.out, p.ldo, (int)q_bytes, p.skip_empty, p.arg_out, p.out_absmax); \
    else if (p.arg_out)                                                                                             \
      hipLaunchKernelGGL((k_mpnn_max<NCH, DEP, true>), grid, block, 0, s, p.p_bias, p.Q, p.ldq, p.We, p.ldwe, p.ea, p.de, p.rowptr, \
                         p.src, p.order, p.chunk_start, p.n_chunks, queue, p.n, p.d, p.out, p.ldo, (int)q_bytes, p.skip_empty, p.arg_out); \
    else if (p.out_absmax)                                                                                          \
      hipLaunchKernelGGL((k_mpnn_max<NCH, DEP, false, true>), grid, block, 0, s, p.p_bias, p.Q, p.ldq, p.We, p.ldwe, p.ea, p.de, p.rowptr, \
                         p.src, p.order, p.chunk_start, p.n_chunks, queue, p.n, p.d, p.out, p.ldo, (int)q_bytes, p.skip_empty, nullptr, p.out_absmax); \
    else                                                                                                            \
      hipLaunchKernelGGL((k_mpnn_max<NCH, DEP, false>), grid, block, 0, s, p.p_bias, p.Q, p.ldq, p.We, p.ldwe, p.ea, p.de, p.rowptr, \
                         p.src, p.order, p.chunk_start, p.n_chunks, queue, p.n, p.d, p.out, p.ldo, (int)q_bytes, p.skip_empty, nullptr); \
  } while (0)
      if (nch == 2) { if (p.de <= 4) RGNN_MPX(2, 4); else RGNN_MPX(2, 8); }
      else if (p.de <= 4) RGNN_MPX(1, 4);
      else RGNN_MPX(1, 8);
#undef RGNN_MPX
      return p.out_absmax ? 2 : 1;                      // (the kernel that can record the winners; 2: it tracked |out| as well)
    }
#define RGNN_MPF(NCH, DEP)                                                                                          \
  hipLaunchKernelGGL((k_mpnn_fast<NCH, DEP, MODE>), grid, block, 0, s, p.P, p.ldp, p.p_bias, p.Q, p.ldq, p.We, p.ldwe,   \
                     p.ea, p.de, p.rowptr, p.src, p.order, p.chunk_start, p.n_chunks, queue, p.n, p.d, p.aggr, p.relu,   \
                     p.out, p.ldo)
    if (nch == 2) { if (p.de <= 4) RGNN_MPF(2, 4); else RGNN_MPF(2, 8); }
    else if (p.de <= 4) RGNN_MPF(1, 4);
    else if (p.de <= 8) RGNN_MPF(1, 8);
    else if (p.de <= 16) RGNN_MPF(1, 16);
    else RGNN_MPF(1, 32);
#undef RGNN_MPF
    return 0;
  }
  // generic path: one wave per chunk, W_e slice in registers, scalar channels
  const int64_t target_waves = 256 * 16;
  int64_t chunk = (p.n + target_waves - 1) / target_waves;
  if (chunk < 8) chunk = 8;
  p.chunk = chunk;
  const int64_t waves = (p.n + chunk - 1) / chunk;
  int64_t blocks = (waves + MP_WAVES - 1) / MP_WAVES;
  blocks = (blocks + 7) / 8 * 8;
  const dim3 grid((unsigned)blocks), block(MP_THREADS);
#define RGNN_MP_LAUNCH(VEC, NCH, DEP) hipLaunchKernelGGL((k_mpnn<VEC, NCH, DEP, MODE>), grid, block, 0, s, p)
  if (p.de <= 4) RGNN_MP_LAUNCH(1, 2, 4);
  else if (p.de <= 16) RGNN_MP_LAUNCH(1, 2, 16);
  else RGNN_MP_LAUNCH(1, 2, 32);
#undef RGNN_MP_LAUNCH
  return 0;
}

// rows [E, d] in CSR order -> out [n, d]
__global__ __launch_bounds__(256) void k_segment_reduce(const float* __restrict__ rows, int64_t ldr,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ order, int64_t n, int d, int aggr,
                                                       float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= n) return;
  const int64_t node = order ? (int64_t)order[pos] : pos;
  const int beg = rowptr[pos], end = rowptr[pos + 1];
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

extern "C" int rgnn_mpnn_aggregate_flags(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                         const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                         const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                                         const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t aggr,
                                         float* out, int64_t ldo, int32_t flags, rgnn_stream_t stream) {
  return rgnn_mpnn_aggregate_absmax(P, ldp, p_bias, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order,
                                    chunk_start, n_chunks, n, d, aggr, out, ldo, flags, nullptr, stream);
}

extern "C" int rgnn_mpnn_aggregate_absmax(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                          const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                          const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                                          const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t aggr,
                                          float* out, int64_t ldo, int32_t flags, float* out_absmax, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  int rc = check_common(Q, We, edge_attr_sorted, de, rowptr_t, src_sorted, n, d, aggr);
  if (rc) return rc;
  RGNN_CHECK_ARG(out, "null out");
  RGNN_CHECK_ARG((flags & ~RGNN_MPNN_SKIP_EMPTY_ROWS) == 0, "unknown flags");
  MpParams p;
  p.P = P; p.ldp = ldp; p.p_bias = p_bias; p.Q = Q; p.ldq = ldq; p.We = We; p.ldwe = ldwe; p.ea = edge_attr_sorted;
  p.de = de; p.rowptr = rowptr_t; p.src = src_sorted; p.order = node_order; p.n = n; p.d = d; p.aggr = aggr; p.relu = 0;
  p.chunk_start = chunk_start; p.n_chunks = n_chunks;
  p.skip_empty = (flags & RGNN_MPNN_SKIP_EMPTY_ROWS) ? 1 : 0;   // (honoured by the max kernel; the others write the zeros)
  p.out = out;
  p.ldo = ldo;
  p.arg_out = nullptr;
  p.out_absmax = out_absmax;
  rgnn_prof_begin((hipStream_t)stream);
  const int which = dispatch<0>(p, (hipStream_t)stream);
  rgnn_prof_end((hipStream_t)stream);
  if (out_absmax != nullptr && which != 2) {
    // another kernel took the launch (mean / add, per-target rows, wide edge attributes): one pass over the rows it wrote --
    // those kernels write every row, the zeros of empty targets included
    hipLaunchKernelGGL(k_absmax_matrix, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float*)out, ldo, n, d, out_absmax);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_mpnn_aggregate_max_arg(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                           const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t,
                                           const int32_t* src_sorted, const int32_t* node_order, const int32_t* chunk_start,
                                           int32_t n_chunks, int64_t n, int32_t d, float* out, int64_t ldo, uint16_t* arg_out,
                                           int32_t flags, int32_t* arg_written, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(arg_written != nullptr, "null arg_written");
  const int rc = rgnn_mpnn_aggregate_max_arg_absmax(p_bias, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order,
                                                    chunk_start, n_chunks, n, d, out, ldo, arg_out, flags, arg_written, nullptr, stream);
  *arg_written &= 1;
  return rc;
}

extern "C" int rgnn_mpnn_aggregate_max_arg_absmax(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                                  const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t,
                                                  const int32_t* src_sorted, const int32_t* node_order, const int32_t* chunk_start,
                                                  int32_t n_chunks, int64_t n, int32_t d, float* out, int64_t ldo, uint16_t* arg_out,
                                                  int32_t flags, int32_t* arg_written, float* out_absmax, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(arg_written != nullptr, "null arg_written");
  *arg_written = 0;
  if (n == 0) return RGNN_OK;
  int rc = check_common(Q, We, edge_attr_sorted, de, rowptr_t, src_sorted, n, d, RGNN_AGGR_MAX);
  if (rc) return rc;
  RGNN_CHECK_ARG(out && arg_out, "null out / arg_out");
  RGNN_CHECK_ARG((flags & ~RGNN_MPNN_SKIP_EMPTY_ROWS) == 0, "unknown flags");
  MpParams p;
  p.P = nullptr; p.ldp = 0; p.p_bias = p_bias; p.Q = Q; p.ldq = ldq; p.We = We; p.ldwe = ldwe; p.ea = edge_attr_sorted;
  p.de = de; p.rowptr = rowptr_t; p.src = src_sorted; p.order = node_order; p.n = n; p.d = d; p.aggr = RGNN_AGGR_MAX; p.relu = 0;
  p.chunk_start = chunk_start; p.n_chunks = n_chunks;
  p.skip_empty = (flags & RGNN_MPNN_SKIP_EMPTY_ROWS) ? 1 : 0;
  p.out = out; p.ldo = ldo;
  p.arg_out = (d % 8 == 0 && ((uintptr_t)arg_out & 15) == 0) ? arg_out : nullptr;
  p.out_absmax = out_absmax;
  rgnn_prof_begin((hipStream_t)stream);
  const int which = dispatch<0>(p, (hipStream_t)stream);
  rgnn_prof_end((hipStream_t)stream);
  *arg_written = ((which >= 1 && p.arg_out != nullptr) ? 1 : 0) | ((which == 2) ? 2 : 0);
  if (out_absmax != nullptr && which != 2) {                    // (another kernel took the launch: it wrote every row -- one pass over them)
    hipLaunchKernelGGL(k_absmax_matrix, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float*)out, ldo, n, d, out_absmax);
    *arg_written |= 2;
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_mpnn_aggregate(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                   const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                   const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                                   const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t aggr,
                                   float* out, int64_t ldo, rgnn_stream_t stream) {
  return rgnn_mpnn_aggregate_flags(P, ldp, p_bias, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order,
                                   chunk_start, n_chunks, n, d, aggr, out, ldo, 0, stream);
}

extern "C" int rgnn_mpnn_edge_hidden(const float* P, int64_t ldp, const float* p_bias, const float* Q, int64_t ldq,
                                     const float* We, int64_t ldwe, const float* edge_attr_sorted, int32_t de,
                                     const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order,
                                     const int32_t* chunk_start, int32_t n_chunks, int64_t n, int32_t d, int32_t relu,
                                     float* hidden, int64_t ldh, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  int rc = check_common(Q, We, edge_attr_sorted, de, rowptr_t, src_sorted, n, d, 0);
  if (rc) return rc;
  RGNN_CHECK_ARG(hidden, "null hidden");
  MpParams p;
  p.P = P; p.ldp = ldp; p.p_bias = p_bias; p.Q = Q; p.ldq = ldq; p.We = We; p.ldwe = ldwe; p.ea = edge_attr_sorted;
  p.de = de; p.rowptr = rowptr_t; p.src = src_sorted; p.order = node_order; p.n = n; p.d = d; p.aggr = 0; p.relu = relu;
  p.chunk_start = chunk_start; p.n_chunks = n_chunks; p.skip_empty = 0; p.arg_out = nullptr; p.out_absmax = nullptr;
  p.out = hidden;
  p.ldo = ldh;
  dispatch<1>(p, (hipStream_t)stream);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_segment_reduce(const float* rows, int64_t ldr, const int32_t* rowptr_t, const int32_t* node_order,
                                   int64_t n, int32_t d, int32_t aggr, float* out, int64_t ldo, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rowptr_t && out && d >= 1 && aggr >= 0 && aggr <= 2, "bad arguments");
  hipLaunchKernelGGL(k_segment_reduce, dim3(rgnn_blocks(n, 4)), dim3(256), 0, (hipStream_t)stream, rows, ldr, rowptr_t,
                     node_order, n, d, aggr, out, ldo);
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

namespace {
__global__ __launch_bounds__(256) void k_empty_flags(const int32_t* __restrict__ rowptr, int64_t n, int32_t* __restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) flags[p] = (rowptr[p + 1] == rowptr[p]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_empty_compact(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ order, int64_t n,
                                                      const int32_t* __restrict__ pos, int32_t* __restrict__ list,
                                                      int64_t* __restrict__ count, int32_t* __restrict__ slot,
                                                      int32_t* __restrict__ list_ne, int64_t* __restrict__ count_ne) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) {
    *count = (int64_t)pos[n];
    if (count_ne) *count_ne = n - (int64_t)pos[n];
  }
  if (p >= n) return;
  const int32_t node = order ? order[p] : (int32_t)p;
  const bool empty = rowptr[p + 1] == rowptr[p];
  if (empty) list[pos[p]] = node;
  else if (list_ne) list_ne[p - pos[p]] = node;          // pos[p] empties precede p: its rank among the others is p - pos[p]
  if (slot) slot[node] = empty ? pos[p] : -1;
}
// The same split with the lists in ascending NODE order (thread = node id; its CSR segment is rank[node]): a dense layer that
// runs on such a list walks the activation matrix front to back instead of jumping around inside every frame (the visiting
// order is a cell order) -- the row-subset launches of the C2 step measure 7 % faster on it.
__global__ __launch_bounds__(256) void k_empty_flags_node(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rank,
                                                         int64_t n, int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t sgm = rank ? rank[i] : i;
  flags[i] = (rowptr[sgm + 1] == rowptr[sgm]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_empty_compact_node(const int32_t* __restrict__ flags, int64_t n,
                                                           const int32_t* __restrict__ pos, int32_t* __restrict__ list,
                                                           int64_t* __restrict__ count, int32_t* __restrict__ slot,
                                                           int32_t* __restrict__ list_ne, int64_t* __restrict__ count_ne) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    *count = (int64_t)pos[n];
    if (count_ne) *count_ne = n - (int64_t)pos[n];
  }
  if (i >= n) return;
  const bool empty = flags[i] != 0;
  if (empty) list[pos[i]] = (int32_t)i;
  else if (list_ne) list_ne[i - pos[i]] = (int32_t)i;
  if (slot) slot[i] = empty ? pos[i] : -1;
}
}  // namespace

extern "C" int rgnn_split_targets(const int32_t* rowptr_t, const int32_t* node_order, int64_t n, int32_t* flags_tmp,
                                  int32_t* pos_tmp, void* scan_tmp, int32_t* list, int64_t* count, int32_t* slot_of_node,
                                  int32_t* list_nonempty, int64_t* count_nonempty, rgnn_stream_t stream);

extern "C" int rgnn_split_targets_by_node(const int32_t* rowptr_t, const int32_t* rank, int64_t n, int32_t* flags_tmp,
                                          int32_t* pos_tmp, void* scan_tmp, int32_t* list, int64_t* count,
                                          int32_t* slot_of_node, int32_t* list_nonempty, int64_t* count_nonempty,
                                          rgnn_stream_t stream) {
  RGNN_CHECK_ARG(count != nullptr, "null count");
  RGNN_CHECK_ARG((list_nonempty == nullptr) == (count_nonempty == nullptr), "list_nonempty and count_nonempty go together");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipMemsetAsync(count, 0, 8, s);
    if (count_nonempty) hipMemsetAsync(count_nonempty, 0, 8, s);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(rowptr_t && flags_tmp && pos_tmp && scan_tmp && list, "null pointers");
  hipLaunchKernelGGL(k_empty_flags_node, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, rowptr_t, rank, n, flags_tmp);
  int rc = rgnn_exclusive_scan_i32(flags_tmp, pos_tmp, n, scan_tmp, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_empty_compact_node, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, flags_tmp, n, pos_tmp, list, count,
                     slot_of_node, list_nonempty, count_nonempty);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_empty_targets(const int32_t* rowptr_t, const int32_t* node_order, int64_t n, int32_t* flags_tmp,
                                  int32_t* pos_tmp, void* scan_tmp, int32_t* list, int64_t* count, int32_t* slot_of_node,
                                  rgnn_stream_t stream) {
  return rgnn_split_targets(rowptr_t, node_order, n, flags_tmp, pos_tmp, scan_tmp, list, count, slot_of_node, nullptr, nullptr,
                            stream);
}

extern "C" int rgnn_split_targets(const int32_t* rowptr_t, const int32_t* node_order, int64_t n, int32_t* flags_tmp,
                                  int32_t* pos_tmp, void* scan_tmp, int32_t* list, int64_t* count, int32_t* slot_of_node,
                                  int32_t* list_nonempty, int64_t* count_nonempty, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(count != nullptr, "null count");
  RGNN_CHECK_ARG((list_nonempty == nullptr) == (count_nonempty == nullptr), "list_nonempty and count_nonempty go together");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipMemsetAsync(count, 0, 8, s);
    if (count_nonempty) hipMemsetAsync(count_nonempty, 0, 8, s);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(rowptr_t && flags_tmp && pos_tmp && scan_tmp && list, "null pointers");
  hipLaunchKernelGGL(k_empty_flags, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, rowptr_t, n, flags_tmp);
  int rc = rgnn_exclusive_scan_i32(flags_tmp, pos_tmp, n, scan_tmp, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_empty_compact, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, rowptr_t, node_order, n, pos_tmp, list,
                     count, slot_of_node, list_nonempty, count_nonempty);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// Work units (edges + alpha per target) per chunk = per wave visit.  80 on full batches (three waves per SIMD: 80 measured
// 0.8 % better than 120, 60 - 100 within noise); a small graph (one frame: 36 k units) is cut finer, down to 16, so that
// its chunks still cover the 3 072 wave slots of the chip instead of 451 waves doing 80 edges one after the other
// (C1: 27 -> see MEASUREMENTS.md).  RGNN_MPNN_WORK overrides for experiments.
static int mpnn_work(int64_t n, int64_t n_edges) {
  static int forced = -1;
  if (forced < 0) {
    const char* e = RGNN_ENV("RGNN_MPNN_WORK");
    forced = e ? atoi(e) : 0;
    if (forced && forced < 8) forced = 8;
  }
  if (forced) return forced;
  const int64_t w = (n_edges + 2 * n) / 3072;
  return (int)(w < 16 ? 16 : (w > 80 ? 80 : w));
}
static int mpnn_alpha(int64_t n, int64_t n_edges) {  // weight of a target in the work estimate; chunk holds <= work / alpha <= 63 targets
  const int w = mpnn_work(n, n_edges);
  int a = (w + 62) / 63;
  return a < 2 ? 2 : a;
}
extern "C" int32_t rgnn_mpnn_work_units(int64_t n, int64_t n_edges) { return mpnn_work(n, n_edges); }
extern "C" int32_t rgnn_mpnn_target_weight(int64_t n, int64_t n_edges) { return mpnn_alpha(n, n_edges); }
extern "C" int32_t rgnn_mpnn_num_chunks(int64_t n, int64_t n_edges) {
  const int w = mpnn_work(n, n_edges);
  return (int32_t)((n_edges + (int64_t)mpnn_alpha(n, n_edges) * n + w - 1) / w + 1);
}

extern "C" int rgnn_mpnn_partition(const int32_t* rowptr_t, int64_t n, int64_t n_edges, int32_t* chunk_start,
                                   rgnn_stream_t stream) {
  RGNN_CHECK_ARG(rowptr_t && chunk_start && n >= 0, "bad arguments");
  const int nc = rgnn_mpnn_num_chunks(n, n_edges);
  hipLaunchKernelGGL(k_partition, dim3(rgnn_blocks(nc + 1 > RGNN_MPNN_QUEUE_INTS ? nc + 1 : RGNN_MPNN_QUEUE_INTS, 256)), dim3(256), 0,
                     (hipStream_t)stream, rowptr_t, n,
                     mpnn_work(n, n_edges), mpnn_alpha(n, n_edges), nc,
                     chunk_start);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
