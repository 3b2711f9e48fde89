// fp32 dense layer on the gfx950 matrix cores:  out = act([A1|A2] * W^T + bias) (+ residual), optional
// per-row-panel column statistics for the BatchNorm that follows.
//
// Replaces ATen addmm behind torch_geometric's dense Linear (gnn/gnn_models.py:137-178,
// gnn/mpnn_layers.py:64-74,89-90).  The reference tolerance (1e-5 relative) rules out bf16/fp8, so the
// kernel uses v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the fp32 vector rate (157 TFLOP/s peak).
//
// Tiling (one 256-thread workgroup = 4 waves of 64):
//   block tile  BM=128 x BN in {128, 64, 32},  BK = 32
//   wave tile   64x64 (2x2 MFMA tiles), 64x32 (2x1) or 32x32 (1x1)
//   LDS         A tile [128][36] + W tile [BN][36] floats, rows padded 32 -> 36 floats so that both the
//               ds_write_b128 of the staging pass and the ds_read_b128 of the fragment reads are bank-conflict
//               free (row stride 9 x 16 B, odd -> the 16 lanes of a b128 group hit 16 distinct 16-B slots)
//   fragments   lane l reads 4 consecutive k of row (l & 31) at k-offset 8*s + 4*(l >> 5): one ds_read_b128 feeds
//               4 MFMAs (MFMA j pairs k = 8s+j from lanes 0-31 with k = 8s+4+j from lanes 32-63; A and W use the
//               same pairing so the products line up)
//   staging     global -> registers (next tile, in flight during the MFMAs) -> LDS
//   grid        1-D, XCD-aware: workgroup b runs on XCD b % 8 (observed placement, speed only), so all column
//               tiles of one 128-row panel are issued back to back on ONE XCD and the A panel is fetched from
//               HBM once into that XCD's L2.
#include "linear_common.h"

#ifndef RGNN_NBUF
#define RGNN_NBUF 2
#endif
#ifndef RGNN_MINW
#define RGNN_MINW 2
#endif
#ifndef RGNN_STAGGER
#define RGNN_STAGGER 0
#endif
#ifndef RGNN_WAVES8
#define RGNN_WAVES8 0
#endif

namespace {


// Tile loads are branch-free: out-of-range rows / k are clamped to a valid address and zeroed with a select, so the
// eight loads of a k-step are straight-line code the scheduler can hoist over the MFMAs.
// `row` = matrix row already resolved through row_index (IDX) and clamped; `gm` only decides validity.
template <bool VEC, bool IDX>
__device__ __forceinline__ float4 load_a(const LinParams& p, int64_t M, int64_t gm, int64_t row, int gk) {
  const int K = p.k1 + p.k2;
  if (VEC) {
    const bool ok = (gm < M) && (gk < K);
    const int64_t r = row;
    const int k = (gk < K) ? gk : (K - 4);
    const float* ptr = (k < p.k1) ? (p.A1 + r * p.lda1 + k) : (p.A2 + r * p.lda2 + (k - p.k1));
    float4 v = *(const float4*)ptr;
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  } else {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gm >= M) return v;
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = gk + j;
      t[j] = (k < K) ? ((k < p.k1) ? p.A1[row * p.lda1 + k] : p.A2[row * p.lda2 + (k - p.k1)]) : 0.f;
    }
    return make_float4(t[0], t[1], t[2], t[3]);
  }
}

template <bool VEC>
__device__ __forceinline__ float4 load_w(const LinParams& p, int gn, int gk) {
  const int K = p.k1 + p.k2;
  if (VEC) {
    const bool ok = (gn < p.n) && (gk < K);
    const int n = (gn < p.n) ? gn : (p.n - 1);
    const int k = (gk < K) ? gk : (K - 4);
    const float* row = (n < p.w_split) ? (p.W1 + (int64_t)n * p.ldw) : (p.W2 + (int64_t)(n - p.w_split) * p.ldw);
    float4 v = *(const float4*)(row + k);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  } else {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gn >= p.n) return v;
    const float* row = (gn < p.w_split) ? (p.W1 + (int64_t)gn * p.ldw) : (p.W2 + (int64_t)(gn - p.w_split) * p.ldw);
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = (gk + j < K) ? row[gk + j] : 0.f;
    return make_float4(t[0], t[1], t[2], t[3]);
  }
}



// BUFL: operand tiles come in through buffer descriptors (buffer_load_dwordx4 v, voffset, srsrc, soffset): the
// per-thread byte offset is computed once per tile, the k-step advances in an SGPR, rows / k beyond the matrix are
// given an out-of-range offset and the hardware returns 0 -- no per-load address arithmetic, selects or exec masks.
// That matters more than usual here: v_mfma_f32_32x32x2_f32 runs at the fp32 VALU rate and every VALU instruction a
// wave issues between MFMAs costs matrix throughput (tools/attic/mfma_probe.hip: 154 TFLOP/s pure, 141 with 2 VALU ops
// per MFMA, 118 with 6 -- independent of the number of waves per SIMD).
template <int BN, int WGM, int WGN, int TM, int TN, int NBUF, bool VEC, bool IDX, bool BUFL>
__global__ __launch_bounds__(WGM * WGN * 64) void k_linear(const LinParams p) {
  constexpr int THREADS = WGM * WGN * 64;
  static_assert(WGM * TM * 32 == BM && WGN * TN * 32 == BN, "tile config");
  constexpr int NA = BM * (BK / 4) / THREADS;  // float4 per thread for the A tile (4)
  constexpr int NB = BN * (BK / 4) / THREADS;  // for the W tile (4 / 2 / 1)
  static_assert(NB >= 1, "W tile too small");
  constexpr int BUF = (BM + BN) * LDK;         // floats per LDS buffer
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware work list: XCD x (= workgroup id % 8, observed placement: speed only) owns the row panels
  // x, x+8, ...; its items are (panel, column tile) pairs, panel-major, dealt round-robin to its workgroups, so
  // the column tiles of a panel run concurrently on ONE XCD and the A panel is fetched from HBM once.
  const int64_t M = (IDX && p.m_dev) ? *p.m_dev : p.m;
  const int mt = (int)((M + BM - 1) / BM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, g8 = gridDim.x >> 3;
  const int my_panels = (mt > xcd) ? (mt - xcd + 7) / 8 : 0;
  const int n_items = my_panels * p.nt;
  int item = slot;
  if (item >= n_items) return;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int K = p.k1 + p.k2;
  const int nk = (K + BK - 1) / BK;

  float4 ra[NA], rb[NB];
  int64_t arow[NA];  // matrix rows of this thread's A loads, resolved once per tile (row_index gathers are a dependent hop)
  // BUFL: byte offsets of this thread's rows inside A1 / A2 / W (OOB when out of range), current and next tile
  int va1[NA], va2[NA], vw[NB], nva1[NA], nva2[NA], nvw[NB];
  __amdgpu_buffer_rsrc_t ra1_d, ra2_d, rw_d;
  const int col_b = (t & 7) * 16;  // this thread's 16-B column inside a 128-B k-step row
  if (BUFL) {
    ra1_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, (short)0, p.ext_a1, 0x00020000);
    ra2_d = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A1), (short)0, p.A2 ? p.ext_a2 : 0, 0x00020000);
    rw_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, (short)0, p.ext_w, 0x00020000);
  }
  // BUFL: byte offsets of this thread's rows of tile (m0, n0) inside A1 / A2 / W; computed once per tile (for the NEXT
  // tile while the current one is being multiplied, so the k-loop carries no per-iteration copies of them)
  auto tile_offsets = [&](int64_t m0, int n0, int (&o1)[NA], int (&o2)[NA], int (&ow)[NB]) {
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int64_t gm = m0 + ((t + THREADS * s) >> 3);
      int64_t row = -1;
      if (gm < M) row = IDX ? (int64_t)p.row_index[gm] : gm;
      o1[s] = (row >= 0) ? (int)(row * p.lda1 * 4) + col_b : OOB;
      o2[s] = (row >= 0) ? (int)(row * p.lda2 * 4) + col_b : OOB;
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
      const int gn = n0 + ((t + THREADS * s) >> 3);
      ow[s] = (gn < p.n) ? (int)((int64_t)gn * p.ldw * 4) + col_b : OOB;
    }
  };
  auto load_bufl = [&](int kt, const int (&o1)[NA], const int (&o2)[NA], const int (&ow)[NB]) {
    // the k-step offset is wave-uniform; readfirstlane tells the compiler so (otherwise it wraps every buffer
    // load in a waterfall loop over the "divergent" soffset)
    const int k0 = __builtin_amdgcn_readfirstlane(kt * BK);
    if (k0 + BK <= K) {
      if (k0 < p.k1) {
#pragma unroll
        for (int s = 0; s < NA; s++) ra[s] = buf_load16(ra1_d, o1[s], k0 * 4);
      } else {
        const int ko = __builtin_amdgcn_readfirstlane((k0 - p.k1) * 4);
#pragma unroll
        for (int s = 0; s < NA; s++) ra[s] = buf_load16(ra2_d, o2[s], ko);
      }
#pragma unroll
      for (int s = 0; s < NB; s++) rb[s] = buf_load16(rw_d, ow[s], k0 * 4);
    } else {  // last, partial k-step: the 16-B columns beyond K read as zero
      const bool tail = (k0 + (col_b >> 2)) >= K;
      if (k0 < p.k1) {
#pragma unroll
        for (int s = 0; s < NA; s++) ra[s] = buf_load16(ra1_d, tail ? OOB : o1[s], k0 * 4);
      } else {
        const int ko = __builtin_amdgcn_readfirstlane((k0 - p.k1) * 4);
#pragma unroll
        for (int s = 0; s < NA; s++) ra[s] = buf_load16(ra2_d, tail ? OOB : o2[s], ko);
      }
#pragma unroll
      for (int s = 0; s < NB; s++) rb[s] = buf_load16(rw_d, tail ? OOB : ow[s], k0 * 4);
    }
  };
  auto load_tiles = [&](int64_t m0, int n0, int kt) {  // pointer path (operands that do not fit a buffer descriptor)
    if (kt == 0) {
#pragma unroll
      for (int s = 0; s < NA; s++) {
        const int64_t gm = m0 + ((t + THREADS * s) >> 3);
        const int64_t r = (gm < M) ? gm : (M - 1);
        arow[s] = IDX ? (int64_t)p.row_index[r] : r;
      }
    }
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int qq = t + THREADS * s;
      ra[s] = load_a<VEC, IDX>(p, M, m0 + (qq >> 3), arow[s], kt * BK + (qq & 7) * 4);
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
      const int qq = t + THREADS * s;
      rb[s] = load_w<VEC>(p, n0 + (qq >> 3), kt * BK + (qq & 7) * 4);
    }
  };
  // kt: the k-step the registers hold.  Optional scale / shift (+ ReLU) of the A1 columns (the BatchNorm in front of the
  // layer, see rgnn_linear_args.a1_scale_shift): every thread owns ONE 16-byte column group of the step, so it is one pair of
  // 16-byte loads of the table and NA x 8 VALU operations per step.
  auto store_tiles = [&](float* buf, int kt) {
    if (BUFL && p.a1_aff != nullptr) {
      const int k = kt * BK + (t & 7) * 4;
      if (k < p.k1) {
        // table rows (rgnn.h, RGNN_AFFINE_ROWS): mean, g, t -- y = (x - mean) g + t
        const float4 mu = *(const float4*)(p.a1_aff + k), sc = *(const float4*)(p.a1_aff + p.k1 + k), sh = *(const float4*)(p.a1_aff + 2 * p.k1 + k);
        const float lo = p.a1_relu ? 0.f : -INFINITY;
#pragma unroll
        for (int s = 0; s < NA; s++) {
          ra[s].x = fmaxf(fmaf(ra[s].x - mu.x, sc.x, sh.x), lo); ra[s].y = fmaxf(fmaf(ra[s].y - mu.y, sc.y, sh.y), lo);
          ra[s].z = fmaxf(fmaf(ra[s].z - mu.z, sc.z, sh.z), lo); ra[s].w = fmaxf(fmaf(ra[s].w - mu.w, sc.w, sh.w), lo);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int qq = t + THREADS * s;
      *(float4*)&buf[(qq >> 3) * LDK + (qq & 7) * 4] = ra[s];
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
      const int qq = t + THREADS * s;
      *(float4*)&buf[BM * LDK + (qq >> 3) * LDK + (qq & 7) * 4] = rb[s];
    }
  };
  auto decode = [&](int it, int64_t& m0, int& n0, int& panel) {
    panel = xcd + 8 * (it / p.nt);
    m0 = (int64_t)panel * BM;
    n0 = (it % p.nt) * BN;
  };

  int64_t m0, nm0 = 0;
  int n0, panel, nn0 = 0, npanel = 0;
  decode(item, m0, n0, panel);
  int cur = 0;
  if (BUFL) tile_offsets(m0, n0, va1, va2, vw);
  if (BUFL) load_bufl(0, va1, va2, vw);
  else load_tiles(m0, n0, 0);
  const int frag_k = (lane >> 5) * 4;
  const int a_off = (wm * TM * 32 + (lane & 31)) * LDK + frag_k;
  const int b_off = BM * LDK + (wn * TN * 32 + (lane & 31)) * LDK + frag_k;

  for (;;) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int next_item = item + g8;
    const bool has_next = next_item < n_items;
    if (has_next) {
      decode(next_item, nm0, nn0, npanel);
      if (BUFL) tile_offsets(nm0, nn0, nva1, nva2, nvw);
    }

    for (int kt = 0; kt < nk; kt++) {
      float* buf = smem + cur * BUF;
      store_tiles(buf, kt);
      __syncthreads();
      // (last k-step: first k-step of the next tile, in flight during the epilogue)
      if (BUFL) {
        if (kt + 1 < nk) load_bufl(kt + 1, va1, va2, vw);
        else if (has_next) load_bufl(0, nva1, nva2, nvw);
      } else {
        if (kt + 1 < nk) load_tiles(m0, n0, kt + 1);
        else if (has_next) load_tiles(nm0, nn0, 0);
      }
      const float* a_base = buf + a_off;
      const float* b_base = buf + b_off;
      // fragments are double-buffered in registers: the ds_read_b128 of group s+1 are issued before the 4*TM*TN
      // MFMAs of group s, so the LDS latency hides behind the matrix pipe
      float4 a4[2][TM], b4[2][TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a4[0][i] = *(const float4*)(a_base + i * 32 * LDK);
#pragma unroll
      for (int j = 0; j < TN; j++) b4[0][j] = *(const float4*)(b_base + j * 32 * LDK);
#pragma unroll
      for (int s = 0; s < BK / 8; s++) {
        const int c = s & 1, nx = c ^ 1;
        if (s + 1 < BK / 8) {
#pragma unroll
          for (int i = 0; i < TM; i++) a4[nx][i] = *(const float4*)(a_base + i * 32 * LDK + (s + 1) * 8);
#pragma unroll
          for (int j = 0; j < TN; j++) b4[nx][j] = *(const float4*)(b_base + j * 32 * LDK + (s + 1) * 8);
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[c][i].x, b4[c][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[c][i].y, b4[c][j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[c][i].z, b4[c][j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[c][i].w, b4[c][j].w, acc[i][j], 0, 0, 0);
      }
      if (NBUF == 2) cur ^= 1;      // the other buffer was last read before the barrier above: safe to overwrite
      else __syncthreads();         // single buffer: everyone must be done reading before the next store
    }

    // ---- epilogue: bias, activation, residual, store, column statistics
    // (the 32x160 wave tile stays on the staged epilogue: the direct form would push it past 256 VGPRs, 1 wave / SIMD)
    constexpr bool DIRECT_OK = !IDX && (TM * TN != 5);
    float* stage = smem + cur * BUF;  // the buffer the NEXT store will overwrite: nobody reads it any more
    if (DIRECT_OK && p.direct_epilogue) {
      direct_epilogue<BN, WGM, WGN, TM, TN>(p, acc, m0, n0, panel, M, stage);
    } else
    if (p.fast_epilogue) {
      // The accumulators go through LDS (one 32x32 MFMA tile per wave at a time) so that every lane ends up with 4
      // consecutive columns of a row: 16-B stores (8 lanes cover a 128-B row segment), one bias / residual vector per
      // lane, ~4 VALU operations per element instead of ~20 for the per-element form below.
      constexpr int LDE = 36;                // staging row stride (floats): conflict-free b32 writes / b128 reads
      float* my = stage + wave * 32 * LDE;   // this wave's 32 x 32 staging slice (WAVES * 1152 floats <= BUF)
      const int c4 = lane & 7, r0 = lane >> 3;
      float4 s1[TN], s2[TN], pv[TN];           // column statistics about a pivot (the lane's first stored value: linear_common.h)
      float cn[TN];
      const bool do_stats = p.col_stats != nullptr;
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int gn = n0 + (wn * TN + j) * 32 + c4 * 4;
        const bool ncol = gn < p.n;          // n % 4 == 0 on this path
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ncol) {
          const float* bp = (gn < p.w_split) ? p.bias1 : p.bias2;
          if (bp) bias = *(const float4*)(bp + ((gn < p.w_split) ? gn : gn - p.w_split));
        }
        s1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        s2[j] = s1[j]; pv[j] = s1[j]; cn[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
          // C/D layout of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
          for (int r = 0; r < 16; r++)
            my[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDE + (lane & 31)] = acc[i][j][r];
          // the same wave wrote and now reads its own slice: LDS operations of one wave complete in order
          const int64_t gm0 = m0 + (wm * TM + i) * 32 + r0;
#pragma unroll
          for (int pass = 0; pass < 4; pass++) {
            float4 v = *(const float4*)(my + (r0 + pass * 8) * LDE + c4 * 4);
            const int64_t gmr = gm0 + pass * 8;
            const bool okr = ncol && (gmr < M);
            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
            if (p.relu_out) {
              v.x = fmaxf(v.x, gn + 0 >= p.relu_lo ? 0.f : -INFINITY); v.y = fmaxf(v.y, gn + 1 >= p.relu_lo ? 0.f : -INFINITY);
              v.z = fmaxf(v.z, gn + 2 >= p.relu_lo ? 0.f : -INFINITY); v.w = fmaxf(v.w, gn + 3 >= p.relu_lo ? 0.f : -INFINITY);
            }
            if (okr) {
              const int64_t row = (IDX && !p.gather_only) ? (int64_t)p.row_index[gmr] : gmr;
              if (p.residual) {
                const int64_t rrow = p.res_index ? (int64_t)p.res_index[row] : row;
                if (rrow >= 0) {
                  const float4 rr = *(const float4*)(p.residual + rrow * p.ldr + gn);
                  v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
              }
              float* optr = p.out + row * p.ldo + gn;
              if (IDX && p.accumulate) {
                const float4 o = *(const float4*)optr;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
              }
              *(float4*)optr = v;
              if (do_stats) {  // wave-uniform: launches without BatchNorm statistics skip these VALU ops per store
                if (cn[j] == 0.f) pv[j] = v;
                const float4 d = make_float4(v.x - pv[j].x, v.y - pv[j].y, v.z - pv[j].z, v.w - pv[j].w);
                cn[j] += 1.f;
                s1[j].x += d.x; s1[j].y += d.y; s1[j].z += d.z; s1[j].w += d.w;
                s2[j].x += d.x * d.x; s2[j].y += d.y * d.y; s2[j].z += d.z * d.z; s2[j].w += d.w * d.w;
              }
            }
          }
        }
      }
      if (p.col_stats) {
        float* stat_lds = stage;             // [WGM][BN][RGNN_STAT_ROWS], reuses the staging slices after a barrier
        ColStat cs[TN][4];
#pragma unroll
        for (int j = 0; j < TN; j++) {
          cs[j][0] = stat_make(cn[j], pv[j].x, s1[j].x, s2[j].x); cs[j][1] = stat_make(cn[j], pv[j].y, s1[j].y, s2[j].y);
          cs[j][2] = stat_make(cn[j], pv[j].z, s1[j].z, s2[j].z); cs[j][3] = stat_make(cn[j], pv[j].w, s1[j].w, s2[j].w);
#pragma unroll
          for (int off = 8; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < 4; q++) cs[j][q] = stat_merge_xor(cs[j][q], off);
        }
        __syncthreads();
        if (lane < 8) {
#pragma unroll
          for (int j = 0; j < TN; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) stat_store(stat_lds + (wm * BN + (wn * TN + j) * 32 + lane * 4 + q) * RGNN_STAT_ROWS, 1, cs[j][q]);
        }
        __syncthreads();
        for (int c = t; c < BN; c += THREADS) {
          const int gc = n0 + c;
          if (gc < p.n) {
            ColStat a = stat_make(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < WGM; w++) a = stat_merge(a, stat_load(stat_lds + (w * BN + c) * RGNN_STAT_ROWS, 1));
            stat_store(p.col_stats + ((int64_t)panel * RGNN_STAT_ROWS) * p.n + gc, p.n, a);
          }
        }
      }
      __syncthreads();  // the staging slices become the next tile's first operand buffer
    } else {
    float* stat_lds = stage;
    const int col_l = lane & 31, row_h = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int gn = n0 + (wn * TN + j) * 32 + col_l;
      const bool ncol = gn < p.n;
      float bias = 0.f;
      if (ncol) {
        const float* bp = (gn < p.w_split) ? p.bias1 : p.bias2;
        const int bi = (gn < p.w_split) ? gn : gn - p.w_split;
        if (bp) bias = bp[bi];
      }
      float s1 = 0.f, s2 = 0.f, pv = 0.f, cn = 0.f;        // column statistics about a pivot (linear_common.h)
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int64_t gmr = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          if (ncol && gmr < M) {
            const int64_t gm = (IDX && !p.gather_only) ? (int64_t)p.row_index[gmr] : gmr;
            float v = acc[i][j][r] + bias;
            if (p.relu_out && gn >= p.relu_lo) v = fmaxf(v, 0.f);
            if (p.residual) {
              const int64_t rrow = p.res_index ? (int64_t)p.res_index[gm] : gm;
              if (rrow >= 0) v += p.residual[rrow * p.ldr + gn];
            }
            if (IDX && p.accumulate) {
              const float o = p.out[gm * p.ldo + gn];
              v += o;
            }
            p.out[gm * p.ldo + gn] = v;
            if (cn == 0.f) pv = v;
            const float d = v - pv;
            cn += 1.f; s1 += d; s2 += d * d;
          }
        }
      }
      if (p.col_stats) {
        const ColStat both = stat_merge_xor(stat_make(cn, pv, s1, s2), 32);
        if (lane < 32) stat_store(stat_lds + (wm * BN + (wn * TN + j) * 32 + col_l) * RGNN_STAT_ROWS, 1, both);
      }
    }
    if (p.col_stats) {
      __syncthreads();
      for (int c = t; c < BN; c += THREADS) {
        const int gn = n0 + c;
        if (gn < p.n) {
          ColStat a = stat_make(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w = 0; w < WGM; w++) a = stat_merge(a, stat_load(stat_lds + (w * BN + c) * RGNN_STAT_ROWS, 1));
          stat_store(p.col_stats + ((int64_t)panel * RGNN_STAT_ROWS) * p.n + gn, p.n, a);
        }
      }
      __syncthreads();  // stat_lds is the next tile's first staging buffer
    }
    }
    if (!has_next) break;
    item = next_item;
    m0 = nm0; n0 = nn0; panel = npanel;
    if (BUFL) {
#pragma unroll
      for (int s = 0; s < NA; s++) { va1[s] = nva1[s]; va2[s] = nva2[s]; }
#pragma unroll
      for (int s = 0; s < NB; s++) vw[s] = nvw[s];
    }
  }
}


// ------------------------------------------------------------------------------------------------ bf16x3 path
// fp32 GEMM on the bf16 matrix pipe.  Every fp32 operand is split into three bf16 terms a = h + m + l (each the
// round-to-nearest bf16 of what the previous ones leave, |a - h - m - l| <= 2^-27 |a|) and the product is taken as the
// six terms h h', h m', m h', h l', l h', m m' accumulated in fp32 by v_mfma_f32_32x32x16_bf16: measured against
// float64 the result is as accurate as the fp32 MFMA path (3e-7 vs 7e-7 norm-wise at K = 688; three terms alone give
// 5e-6 and are not used).  Per k = 32 step a 64x64 wave tile costs 48 of these MFMAs = 1536 matrix-pipe cycles instead
// of 4096 for v_mfma_f32_32x32x2_f32, and -- unlike that instruction -- they do not share the issue port with the fp32
// VALU (tools/attic/bf16x6_probe.hip: 343 fp32-equivalent TFLOP/s pure, 264 with the split's 88 VALU instructions and 24
// ds_read_b128 per step).  The weight arrives pre-split (rgnn_linear_split_weights, three planes [3][n][kp], kp = K
// rounded up to 32 with zeros); the activations are split in registers on their way from HBM to LDS.
// LDS image: per plane, rows of 32 bf16 = four 16-byte chunks, unpadded; chunk c of row r sits at position
// c ^ ((r >> 1) & 3) (fragment reads: 8 lanes = 8 rows, one chunk each; weight-chunk and 8-byte activation writes: whole
// rows).  256 x 128 tiles, 8 waves of 64 x 64, two LDS buffers of 72 KB, one work-group per CU.
// Ablation at K = 4096 (tools/attic/gemm_bench.hip, fp32-equivalent TFLOP/s): 170 as built, 240 with the global loads dropped
// by the range check, 184 without the barrier, 193 without the split, 312 without all three (= the probe's ceiling):
// what limits this kernel is operand delivery from L2 / HBM into the CU (56 KB per k-step and CU), not the matrix pipe.

// BKX: k per step (32, or 16 for the 256 x 256 tile, whose two LDS buffers then still fit); NSETS: register sets the
// loads alternate between (2 = requested three steps ahead of the MFMAs, 1 = two steps ahead, 28 registers less).
template <int BMT, int BN, int WGM, int WGN, int TM, int TN, int BKX, int NSETS, bool IDX>
__global__ __launch_bounds__(WGM * WGN * 64) void k_linear_x3(const LinParams p) {
  constexpr int THREADS = WGM * WGN * 64;
  static_assert(WGM * TM * 32 == BMT && WGN * TN * 32 == BN, "tile config");
  static_assert(BKX == 16 || BKX == 32, "k-step");
  constexpr int RS = BKX * 2;                     // LDS row stride in bytes (BKX bf16, chunks XOR-swizzled with the row)
  constexpr int CPR = BKX / 8;                    // 16-byte chunks per LDS row
  constexpr int F4R = BKX / 4;                    // float4 per fp32 activation row of a k-step
  constexpr int HN = BKX / 16;                    // k16 halves (one MFMA k-extent each) per step
  constexpr int A_PLANE = BMT * RS, W_PLANE = BN * RS;
  constexpr int BUFB = 3 * (A_PLANE + W_PLANE);   // bytes per LDS buffer
  constexpr int NAQ = BMT * F4R;                  // float4 of the fp32 activation tile
  constexpr int NA = (NAQ + THREADS - 1) / THREADS;
  constexpr int NWQ = BN * 3 * CPR;               // 16-byte chunks of the weight tile (3 planes x BN rows x CPR)
  constexpr int NW = (NWQ + THREADS - 1) / THREADS;
  static_assert(NAQ % THREADS == 0, "activation tile must divide evenly");
  constexpr bool W_EXACT = NWQ % THREADS == 0;    // every thread owns exactly NW weight chunks: no guard, no branch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = (char*)smem;
  float* const stat_lds = (float*)(lds + 2 * BUFB);   // stat_lds_floats(WGM, BN) floats, used by the epilogue only

  // IDX: the layer runs on a row subset (tile row r = matrix row row_index[r]; the subset size lives on the device)
  const int64_t M = (IDX && p.m_dev) ? *p.m_dev : p.m;
  const int mt = (int)((M + BMT - 1) / BMT);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, g8 = gridDim.x >> 3;
  const int my_panels = (mt > xcd) ? (mt - xcd + 7) / 8 : 0;
  const int n_items = my_panels * p.nt;
  if (slot >= n_items) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int K = p.k1 + p.k2;
  const int nk = (K + BKX - 1) / BKX;
  // chunk c of LDS row r sits at position swz(r, c): conflict-free fragment reads (8 lanes = 8 rows) and row-wise writes
  // (ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): the 16 rows of a group must
  // land on 16 distinct 16-byte slots of the 256-byte bank row.  Rows 64 B apart (CPR 4): four rows share a residue
  // mod 4 and differ in bits 2-3; rows 32 B apart (CPR 2): two rows share a residue mod 8 and differ in bit 3.  The
  // r01 image used bits 1-2 / bit 2 and paid a 2-way conflict on every fragment read: SQ_LDS_BANK_CONFLICT = 1.5 x MFMAs.)
  auto swz = [](int r, int c) { return CPR == 4 ? (c ^ ((r >> 2) & 3)) : (c ^ ((r >> 3) & 1)); };

  const __amdgpu_buffer_rsrc_t ra1_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, (short)0, p.ext_a1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra2_d =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A1), (short)0, p.A2 ? p.ext_a2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw_d = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, (short)0, p.ext_wp, 0x00020000);
  const int col_b = (t % F4R) * 16;               // this thread's 16-byte column of a fp32 k-step row
  // Two streams walk the same sequence of (tile, k-step) pairs: the MFMAs and, THREE steps ahead of them, the global
  // loads.  A k-step is only ~1500 matrix-pipe cycles per wave here -- less than an HBM round trip -- so the operands of
  // a step are requested two steps before they are written to LDS (register sets X / Y alternate) and written one step
  // before they are multiplied (two LDS buffers, ONE barrier per step; the split / ds_write of step g + 1 sits in the
  // same basic block as the MFMAs of step g, so the scheduler interleaves them).
  struct Regs { float4 a[NA]; float4 w[NW]; };
  Regs rx, ry;                                  // (ry is unused with NSETS == 1)
  int va1[NA], va2[NA], vw[NW];
  int ld_item = slot, ld_kt = 0;
  auto tile_offsets = [&](int it) {
    const int64_t m0 = (int64_t)(xcd + 8 * (it / p.nt)) * BMT;
    const int n0 = (it % p.nt) * BN;
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int64_t gm = m0 + ((t + THREADS * s) / F4R);
      int64_t row = -1;
      if (gm < M) row = IDX ? (int64_t)p.row_index[gm] : gm;
      va1[s] = (row >= 0) ? (int)(row * p.lda1 * 4) + col_b : OOB;
      va2[s] = (row >= 0) ? (int)(row * p.lda2 * 4) + col_b : OOB;
    }
#pragma unroll
    for (int s = 0; s < NW; s++) {
      const int q = t + THREADS * s;              // chunk -> (plane, row, 16-byte column)
      const int plane = q / (BN * CPR), row = (q / CPR) % BN, c = q % CPR;
      const int gn = n0 + row;
      // planes: [k / 16][plane][n][16 bf16]; chunk c of a k-step = k16 block c >> 1, 16-byte half c & 1
      vw[s] = ((W_EXACT || q < NWQ) && gn < p.n) ? (int)((((int64_t)(c >> 1) * 3 + plane) * p.n + gn) * 32) + (c & 1) * 16 : OOB;
    }
  };
  // request the operands of the load stream's current step -- branch-free (descriptor / offsets picked with uniform
  // selects, an exhausted stream reads with out-of-range offsets, i.e. touches no memory), so that it can share a basic
  // block with the MFMAs; `advance_loads` moves the stream on.
  auto load_next = [&](Regs& r) {
    const bool live = ld_item < n_items;
    const int k0 = __builtin_amdgcn_readfirstlane(ld_kt * BKX);
    const bool use1 = k0 < p.k1;
    // (partial last k-step: columns >= K; bit-wise operators: `||` / `&&` on uniform values become scalar branches)
    const bool dead = (!live) | ((k0 + BKX > K) & ((k0 + (col_b >> 2)) >= K));
    const __amdgpu_buffer_rsrc_t ra_d = use1 ? ra1_d : ra2_d;
    const int soff = __builtin_amdgcn_readfirstlane(use1 ? k0 * 4 : (k0 - p.k1) * 4);
    // (an offset with the top bit set lies beyond every extent; OR-ing it in keeps this straight-line code -- a select
    // against the constant makes hipcc emit two predicated loads and split the block)
    const int a_kill = dead ? OOB : 0, w_kill = live ? 0 : OOB;
#pragma unroll
    for (int s = 0; s < NA; s++) r.a[s] = buf_load16(ra_d, (use1 ? va1[s] : va2[s]) | a_kill, soff);
    const int wsoff = __builtin_amdgcn_readfirstlane(ld_kt * HN * 3 * p.n * 32);
#pragma unroll
    for (int s = 0; s < NW; s++) r.w[s] = buf_load16(rw_d, vw[s] | w_kill, wsoff);   // (zero-padded to kp)
  };
  auto advance_loads = [&]() {
    if (++ld_kt == nk) {
      ld_kt = 0;
      ld_item += g8;
      if (ld_item < n_items) tile_offsets(ld_item);
    }
  };
  auto store_regs = [&](const Regs& r, char* buf) {
    char* As = buf;
    char* Ws = buf + 3 * A_PLANE;
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int qq = t + THREADS * s;
      bf16x4_t h, m, l;
      split3(r.a[s], h, m, l);
      const int row = qq / F4R, c8 = qq % F4R;       // 8-byte half (c8 & 1) of chunk c8 >> 1
      char* d = As + row * RS + swz(row, c8 >> 1) * 16 + (c8 & 1) * 8;
      *(bf16x4_t*)(d) = h;
      *(bf16x4_t*)(d + A_PLANE) = m;
      *(bf16x4_t*)(d + 2 * A_PLANE) = l;
    }
#pragma unroll
    for (int s = 0; s < NW; s++) {
      const int q = t + THREADS * s;
      if (W_EXACT || q < NWQ) {
        const int plane = q / (BN * CPR), row = (q / CPR) % BN, c = q % CPR;
        *(float4*)(Ws + plane * W_PLANE + row * RS + swz(row, c) * 16) = r.w[s];
      }
    }
  };

  const int a_row = (wm * TM * 32 + (lane & 31)) * RS;
  const int b_row = 3 * A_PLANE + (wn * TN * 32 + (lane & 31)) * RS;
  // k-half h: this lane's chunk 2 h + (lane >> 5), at its swizzled position (sub-tiles start at multiples of 32 rows)
  const int frag_off[2] = {swz(lane & 31, lane >> 5) * 16, swz(lane & 31, (2 + (lane >> 5)) % CPR) * 16};

  int c_item = slot, c_kt = 0;                  // compute stream
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  };
  zero_acc();
  tile_offsets(slot);
  load_next(rx); advance_loads();               // step 0
  if (NSETS == 2) { load_next(ry); advance_loads(); }   // step 1
  store_regs(rx, lds);                          // step 0 -> buffer 0
  load_next(rx); advance_loads();               // step 2 (NSETS == 1: step 1)
  // step g: multiply from buffer g & 1 while the operands of step g + 1 (register set `r`) go into the other buffer and
  // `r` is re-requested for step g + 3.  Returns false after the last tile.
  auto step = [&](Regs& r, const char* cur, char* nxt) -> bool {
    __syncthreads();    // buffer `cur` is complete; nobody reads `nxt` (last multiplied in step g - 1) any more
    store_regs(r, nxt);
    load_next(r);
#pragma unroll
    for (int h = 0; h < HN; h++) {
      bf16x8_t a[TM][3], b[TN][3];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[i][pl] = *(const bf16x8_t*)(cur + a_row + pl * A_PLANE + i * 32 * RS + frag_off[h]);
#pragma unroll
        for (int j = 0; j < TN; j++) b[j][pl] = *(const bf16x8_t*)(cur + b_row + pl * W_PLANE + j * 32 * RS + frag_off[h]);
      }
      // smallest terms first: l h', h l', m m', m h', h m', h h'
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
    advance_loads();
    if (++c_kt == nk) {
      const int panel = xcd + 8 * (c_item / p.nt);
      direct_epilogue<BN, WGM, WGN, TM, TN, BMT, IDX>(p, acc, (int64_t)panel * BMT, (c_item % p.nt) * BN, panel, M, stat_lds, (int*)(stat_lds + stat_lds_floats(WGM, BN)));
      c_kt = 0;
      c_item += g8;
      if (c_item >= n_items) return false;
      zero_acc();
    }
    return true;
  };
  for (;;) {
    if (!step(NSETS == 2 ? ry : rx, lds, lds + BUFB)) break;   // even step: multiply buffer 0, fill buffer 1
    if (!step(rx, lds + BUFB, lds)) break;                     // odd step
  }
}

// the weight of a dense layer as three bf16 planes, laid out for the kernels above: [kp / 16][3][n][16] -- the 32 bytes
// (16 k) of one row, plane and k16 block are contiguous, and so are the n rows of a plane, so that a weight tile of one
// k-step is a few runs of BN * 32 bytes (whole cache lines; a plain [3][n][kp] image makes every fetch use a fraction of
// each 128-byte line it touches).  kp = K rounded up to 32, zero padded.
__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ W1, const float* __restrict__ W2, int64_t ldw,
                                                      int w_split, int n, int k, int kp, __bf16* __restrict__ planes) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * kp) return;
  const int row = (int)(idx / kp), col = (int)(idx - (int64_t)row * kp);
  float v = 0.f;
  if (col < k) v = (row < w_split) ? W1[(int64_t)row * ldw + col] : W2[(int64_t)(row - w_split) * ldw + col];
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  const int kb = col >> 4, kk = col & 15;
  __bf16* o = planes + (((int64_t)kb * 3) * n + row) * 16 + kk;
  o[0] = h;
  o[(int64_t)n * 16] = m;
  o[(int64_t)2 * n * 16] = l;
}

// f16x2 form (linear_dma.hip, FMT 1): the weight as TWO f16 planes of W 2^sw, [kp / 16][2][n][16], followed by a 16-byte footer
// {absmax |W|, 2^sw, 2^-sw, 0} and 64 scratch words; sw is the largest exponent with absmax 2^sw < 2^15.  Two launches, no memset and
// no atomics (a training step rebuilds the planes of every weight: r04 measured 53 us per weight for memset + 1024 work-groups of
// atomic max + split): up to 64 work-groups leave one partial maximum each in the scratch words, every wave of the split launch
// reduces the 64 words with six shuffles.
constexpr int W_ABSMAX_PARTS = 64;
__global__ __launch_bounds__(256) void k_weights_absmax(const float* __restrict__ W1, const float* __restrict__ W2, int64_t ldw,
                                                       int w_split, int n, int k, float* __restrict__ parts) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n * k; idx += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / k), col = (int)(idx - (int64_t)row * k);
    const float v = (row < w_split) ? W1[(int64_t)row * ldw + col] : W2[(int64_t)(row - w_split) * ldw + col];
    m = fmaxf(m, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (blockIdx.x == 0 && threadIdx.x >= gridDim.x && threadIdx.x < W_ABSMAX_PARTS) parts[threadIdx.x] = 0.f;   // (slots nobody owns)
}
__global__ __launch_bounds__(256) void k_split_weights_f16(const float* __restrict__ W1, const float* __restrict__ W2, int64_t ldw,
                                                          int w_split, int n, int k, int kp, _Float16* __restrict__ planes,
                                                          float* __restrict__ foot) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float amax = foot[4 + (threadIdx.x & 63)];                            // the partial maxima of k_weights_absmax
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const int be = (int)((__float_as_uint(amax) >> 23) & 255u);           // absmax < 2^(be - 126)
  int se = 268 - be;
  se = se > 253 ? 253 : se;
  const float scale = __uint_as_float((unsigned)se << 23);
  if (idx == 0) { foot[0] = amax; foot[1] = scale; foot[2] = __uint_as_float((unsigned)(254 - se) << 23); foot[3] = 0.f; }
  if (idx >= (int64_t)n * kp) return;
  const int row = (int)(idx / kp), col = (int)(idx - (int64_t)row * kp);
  float v = 0.f;
  if (col < k) v = (row < w_split) ? W1[(int64_t)row * ldw + col] : W2[(int64_t)(row - w_split) * ldw + col];
  v *= scale;
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  const int kb = col >> 4, kk = col & 15;
  _Float16* o = planes + (((int64_t)kb * 2) * n + row) * 16 + kk;
  o[0] = h;
  o[(int64_t)n * 16] = l;
}

template <bool IDX, int BMT, int BN, int WGM, int WGN, int TM, int TN, int BKX = 32, int NSETS = 2>
void launch_x3(LinParams p, hipStream_t s) {
  const size_t lds = (size_t)(2 * 3 * (BMT + BN) * BKX * 2 + stat_lds_floats(WGM, BN) * 4 + (IDX ? BMT * 4 : 0));
  p.mt = (int)((p.m + BMT - 1) / BMT);
  const int64_t tiles = (int64_t)p.mt * p.nt;
  int64_t grid = 256;                            // one 8-wave work-group per CU (two LDS buffers of 72 KB at 256 x 128)
  if (grid > tiles) grid = tiles;
  grid = (grid + 7) / 8 * 8;
  static RgnnOncePerDevice attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_linear_x3<BMT, BN, WGM, WGN, TM, TN, BKX, NSETS, IDX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((k_linear_x3<BMT, BN, WGM, WGN, TM, TN, BKX, NSETS, IDX>), dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, s, p);
}

constexpr int NBUF_DEFAULT = RGNN_NBUF;

template <int BN, int WGM, int WGN, int TM, int TN, int NBUF = NBUF_DEFAULT, bool BUFL_ONLY = false>
void launch(const LinParams& p, bool vec, bool bufl, hipStream_t s) {
  const size_t lds = (size_t)NBUF * (BM + BN) * LDK * sizeof(float);
  // persistent grid: enough workgroups to fill 256 CUs at the occupancy the LDS / register budget admits
  int per_cu = (int)(160 * 1024 / lds) < RGNN_MINW ? (int)(160 * 1024 / lds) : RGNN_MINW;
  const int64_t tiles = (int64_t)p.mt * p.nt;
  int64_t grid = 256 * per_cu;
  if (grid > tiles) grid = tiles;
  grid = (grid + 7) / 8 * 8;
  static RgnnOncePerDevice attr_once;  // once per template instance and device (not a stream operation; safe during graph capture)
  if (lds > 64 * 1024 && attr_once.first()) {
#define RGNN_SET_ATTR(V, I, B)                                                                   \
  hipFuncSetAttribute((const void*)k_linear<BN, WGM, WGN, TM, TN, NBUF, V, I, B>,               \
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    RGNN_SET_ATTR(true, false, true);
    if constexpr (!BUFL_ONLY) {
      RGNN_SET_ATTR(true, false, false); RGNN_SET_ATTR(false, false, false); RGNN_SET_ATTR(true, true, false);
      RGNN_SET_ATTR(false, true, false); RGNN_SET_ATTR(true, true, true);
    }
#undef RGNN_SET_ATTR
  }
  const dim3 g((unsigned)grid), b(WGM * WGN * 64);
  const bool idx = p.row_index != nullptr;
#define RGNN_GO(V, I, B) hipLaunchKernelGGL((k_linear<BN, WGM, WGN, TM, TN, NBUF, V, I, B>), g, b, lds, s, p)
  if constexpr (BUFL_ONLY) {
    RGNN_GO(true, false, true);
  } else {
    if (bufl) { if (idx) RGNN_GO(true, true, true); else RGNN_GO(true, false, true); }
    else if (vec) { if (idx) RGNN_GO(true, true, false); else RGNN_GO(true, false, false); }
    else { if (idx) RGNN_GO(false, true, false); else RGNN_GO(false, false, false); }
  }
#undef RGNN_GO
}

// Dense layer with a tiny reduction dimension (K <= 8: the first Linear of the node / edge embeddings, 5 -> 32 and 2 -> 4
// on N / E rows).  An MFMA tile would spend its time on the scalar operand loads of rows that are not 16-byte aligned;
// here a thread owns four neighbouring outputs of a row: K input words (broadcast within the row's lanes), the 4 x K
// weight slice from LDS, K FMAs per output in k order, one 16-byte store -- the layer is bound by writing its output.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_linear_tiny(const float* __restrict__ A, int64_t lda, int k, const float* __restrict__ W,
                                                    int64_t ldw, const float* __restrict__ bias, int64_t m, int n, int relu,
                                                    float* __restrict__ out, int64_t ldo) {
  __shared__ float w_s[64 * 8 + 64];
  for (int i = threadIdx.x; i < n * k; i += 256) w_s[i] = W[(int64_t)(i / k) * ldw + (i % k)];
  for (int i = threadIdx.x; i < n; i += 256) w_s[64 * 8 + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int groups = (n + 3) / 4;
  // (grid-stride: a few thousand blocks walk all rows, so the weight preamble above is paid once per block, not per 256
  // outputs)
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < m * groups; idx += (int64_t)gridDim.x * blockDim.x) {
  const int64_t row = idx / groups;
  const int c0 = (int)(idx - row * groups) * 4;
  float a[8];
#pragma unroll
  for (int q = 0; q < 8; q++) a[q] = (q < k) ? A[row * lda + q] : 0.f;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = c0 + j;
    float acc = 0.f;
    if (c < n) {
#pragma unroll
      for (int q = 0; q < 8; q++)                      // (static register indices; a[q] = 0 beyond k)
        if (q < k) acc = fmaf(a[q], w_s[c * k + q], acc);
      acc += w_s[64 * 8 + c];
      if (relu) acc = fmaxf(acc, 0.f);
    }
    v[j] = acc;
  }
  if (VEC4) {
    *(float4*)(out + row * ldo + c0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (c0 + j < n) out[row * ldo + c0 + j] = v[j];
  }
  }
}

// Two tiny Linear layers back to back on gathered rows (r03): out[r] = act2(W2 act1(W1 a[row_index[r]] + b1) + b2) with at
// most 8 inputs, 8 hidden and 16 output features -- the hidden layers of DetNetBasic's edge embedding (2 -> 4 -> 8 on E rows,
// gnn_models.py:48-52,137-178) on the edge attributes in target order.  One thread per row: the gather (rgnn_gather_rows_f32),
// both k_linear_tiny launches and the two intermediate [E, .] arrays in between become one pass that reads 8 + 4 bytes and
// writes 32 per edge.  Same arithmetic per output as k_linear_tiny (K FMAs in k order, then the bias, then the clamp): same bits.
__global__ __launch_bounds__(256) void k_tiny_mlp2(const float* __restrict__ A, int64_t lda, int k0,
                                                  const int32_t* __restrict__ row_index, int64_t m,
                                                  const float* __restrict__ W1, int64_t ldw1, const float* __restrict__ b1, int n1,
                                                  int relu1, const float* __restrict__ W2, int64_t ldw2,
                                                  const float* __restrict__ b2, int n2, int relu2, float* __restrict__ out,
                                                  int64_t ldo, int vec4) {
  __shared__ float w1_s[8 * 8 + 8], w2_s[16 * 8 + 16];
  for (int i = threadIdx.x; i < n1 * k0; i += 256) w1_s[i] = W1[(int64_t)(i / k0) * ldw1 + (i % k0)];
  for (int i = threadIdx.x; i < n1; i += 256) w1_s[64 + i] = b1 ? b1[i] : 0.f;
  for (int i = threadIdx.x; i < n2 * n1; i += 256) w2_s[i] = W2[(int64_t)(i / n1) * ldw2 + (i % n1)];
  for (int i = threadIdx.x; i < n2; i += 256) w2_s[128 + i] = b2 ? b2[i] : 0.f;
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = row_index ? (int64_t)row_index[r] : r;
    float a[8], h[8], o[16];
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = (q < k0) ? A[src * lda + q] : 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      float acc = 0.f;
      if (c < n1) {
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (q < k0) acc = fmaf(a[q], w1_s[c * k0 + q], acc);
        acc += w1_s[64 + c];
        if (relu1) acc = fmaxf(acc, 0.f);
      }
      h[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      float acc = 0.f;
      if (c < n2) {
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (q < n1) acc = fmaf(h[q], w2_s[c * n1 + q], acc);
        acc += w2_s[128 + c];
        if (relu2) acc = fmaxf(acc, 0.f);
      }
      o[c] = acc;
    }
    if (vec4) {
#pragma unroll
      for (int c = 0; c < 16; c += 4)
        if (c < n2) *(float4*)(out + r * ldo + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < 16; c++)
        if (c < n2) out[r * ldo + c] = o[c];
    }
  }
}

inline bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

}  // namespace

extern "C" int rgnn_tiny_mlp2(const float* A, int64_t lda, int32_t k0, const int32_t* row_index, int64_t m, const float* W1,
                              int64_t ldw1, const float* b1, int32_t n1, int32_t relu1, const float* W2, int64_t ldw2,
                              const float* b2, int32_t n2, int32_t relu2, float* out, int64_t ldo, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0 && k0 >= 1 && k0 <= 8 && n1 >= 1 && n1 <= 8 && n2 >= 1 && n2 <= 16, "widths: <= 8 in, <= 8 hidden, <= 16 out");
  if (m == 0) return RGNN_OK;
  RGNN_CHECK_ARG(A && W1 && W2 && out, "null pointers");
  const int vec4 = (n2 % 4 == 0) && (ldo % 4 == 0) && aligned16(out);
  const int64_t blocks = rgnn_blocks(m, 256) < 8192 ? rgnn_blocks(m, 256) : 8192;
  hipLaunchKernelGGL(k_tiny_mlp2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, lda, k0, row_index, m, W1, ldw1, b1, n1,
                     relu1, W2, ldw2, b2, n2, relu2, out, ldo, vec4);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

int rgnn_linear_dma_launch(const void* lin_params, int subset, hipStream_t s);   // linear_dma.hip
int rgnn_linear_dma_lds_bytes(int n, int64_t m, int f16_form);

// Does this call take the LDS-DMA bf16x3 kernel (k_linear_dma)?  One predicate for the dispatcher below and for
// rgnn_linear_fwd_fuses_a1_affine (only that kernel applies a scale / shift to its A1 fragments).
static bool takes_dma_kernel(const rgnn_linear_args* a) {
  if (a == nullptr || a->m <= 0 || a->n <= 0 || a->k1 <= 0 || !a->W_planes || !a->A1 || !a->W1 || !a->out) return false;
  if (a->k2 > 0 && !a->A2) return false;
  if (a->w_split < a->n) return false;
  const bool tiny = a->k2 == 0 && a->k1 <= 8 && a->n <= 64 && a->residual == nullptr && a->row_index == nullptr &&
                    a->col_stats == nullptr && a->m >= 4096 && RGNN_ENV("RGNN_LINEAR_NO_TINY") == nullptr;
  if (tiny) return false;
  const bool vec = (a->k1 % 4 == 0) && (a->k2 % 4 == 0) && (a->ldw % 4 == 0) && aligned16(a->W1) &&
                   (a->W2 == nullptr || aligned16(a->W2)) && (a->lda1 % 4 == 0 && aligned16(a->A1)) &&
                   (a->k2 == 0 || (a->lda2 % 4 == 0 && aligned16(a->A2)));
  const int64_t e1 = ((a->m - 1) * a->lda1 + a->k1) * 4;
  const int64_t e2 = a->k2 ? ((a->m - 1) * a->lda2 + a->k2) * 4 : 0;
  const int64_t ew = ((int64_t)(a->n - 1) * a->ldw + a->k1 + a->k2) * 4;
  const int64_t eo = ((a->m - 1) * a->ldo + a->n) * 4;
  const int64_t lim = ((int64_t)1 << 31) - 64;
  const bool bufl = vec && e1 < lim && e2 < lim && ew < lim && (a->k2 == 0 || a->k1 % BK == 0) &&
                    RGNN_ENV("RGNN_LINEAR_NO_BUFL") == nullptr;
  const bool direct = a->row_index == nullptr && a->residual == nullptr && eo < lim && RGNN_ENV("RGNN_LINEAR_NO_DIRECT") == nullptr;
  const bool x3_subset = a->row_index != nullptr && !a->accumulate && !a->gather_only && a->residual == nullptr && eo < lim &&
                         a->relu_from_col <= 0;
  if (!(bufl && (direct || x3_subset) && a->w_planes_kp >= a->k1 + a->k2 && a->w_planes_kp % BK == 0 &&
        (int64_t)3 * a->n * a->w_planes_kp * 2 < lim && RGNN_ENV("RGNN_LINEAR_FP32") == nullptr))
    return false;
  const int dma_min_n = RGNN_ENV("RGNN_DMA_MIN_N") ? atoi(RGNN_ENV("RGNN_DMA_MIN_N")) : 32;
  return a->n > dma_min_n && (a->k1 + a->k2) % 16 == 0 && a->k1 % 16 == 0 && RGNN_ENV("RGNN_X3_NODMA") == nullptr;
}

// ... or the fp32-MFMA kernel with buffer-descriptor operands (k_linear<..., BUFL = true>: the narrow layers, no weight planes)?
static bool takes_fp32_bufl_kernel(const rgnn_linear_args* a) {
  if (a == nullptr || a->m <= 0 || a->n <= 0 || a->k1 <= 0 || a->W_planes || !a->A1 || !a->W1 || !a->out) return false;
  if (a->k2 > 0 && !a->A2) return false;
  if (a->w_split < a->n) return false;
  const bool tiny = a->k2 == 0 && a->k1 <= 8 && a->n <= 64 && a->residual == nullptr && a->row_index == nullptr &&
                    a->col_stats == nullptr && a->m >= 4096 && a->relu_from_col <= 0 && RGNN_ENV("RGNN_LINEAR_NO_TINY") == nullptr;
  if (tiny) return false;
  const bool vec = (a->k1 % 4 == 0) && (a->k2 % 4 == 0) && (a->ldw % 4 == 0) && aligned16(a->W1) &&
                   (a->lda1 % 4 == 0 && aligned16(a->A1)) && (a->k2 == 0 || (a->lda2 % 4 == 0 && aligned16(a->A2)));
  const int64_t e1 = ((a->m - 1) * a->lda1 + a->k1) * 4;
  const int64_t e2 = a->k2 ? ((a->m - 1) * a->lda2 + a->k2) * 4 : 0;
  const int64_t ew = ((int64_t)(a->n - 1) * a->ldw + a->k1 + a->k2) * 4;
  const int64_t lim = ((int64_t)1 << 31) - 64;
  return vec && e1 < lim && e2 < lim && ew < lim && (a->k2 == 0 || a->k1 % BK == 0) && RGNN_ENV("RGNN_LINEAR_NO_BUFL") == nullptr;
}

// ... and in its f16x2 form (two f16 terms per operand, three products): f16 planes + bounds of both activation blocks given
static bool takes_f16_form(const rgnn_linear_args* a) {
  return a->W_planes_f16 != nullptr && a->a1_bound != nullptr && (a->k2 == 0 || a->a2_bound != nullptr) &&
         RGNN_ENV("RGNN_LINEAR_NO_F16") == nullptr;
}
extern "C" int32_t rgnn_linear_fwd_path(const rgnn_linear_args* a) {
  if (!takes_dma_kernel(a)) return RGNN_LINEAR_PATH_OTHER;
  return takes_f16_form(a) ? RGNN_LINEAR_PATH_DMA_F16X2 : RGNN_LINEAR_PATH_DMA_BF16X3;
}

extern "C" int32_t rgnn_linear_fwd_fuses_a1_affine(const rgnn_linear_args* a) {
  if (RGNN_ENV("RGNN_DMA_NO_AFFINE") != nullptr) return 0;
  if (a->a1_panel_segment != nullptr)           // per-segment tables: the LDS-DMA kernel on a row list, two tables resident
    return (a->row_index != nullptr && takes_dma_kernel(a) && a->k1 <= 512 &&
            rgnn_linear_dma_lds_bytes(a->n, a->m, takes_f16_form(a)) + 2 * RGNN_AFFINE_ROWS * 4 * (int64_t)a->k1 <= 160 * 1024) ? 1 : 0;
  if (takes_fp32_bufl_kernel(a)) return 1;
  if (!takes_dma_kernel(a)) return 0;
  return rgnn_linear_dma_lds_bytes(a->n, a->m, takes_f16_form(a)) + RGNN_AFFINE_ROWS * 4 * (int64_t)a->k1 <= 160 * 1024 ? 1 : 0;
}

extern "C" int64_t rgnn_linear_stat_panels(int64_t m) { return (m + BM - 1) / BM; }
// 256 work-group slots of 256 KiB (accumulators of a 256 x 256 tile) + one flag word each
extern "C" int64_t rgnn_linear_splitk_ws_bytes(void) { return (int64_t)256 * 8 * 16 * 512 * 4 + 4096; }

extern "C" int rgnn_linear_fwd(const rgnn_linear_args* a, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(a != nullptr, "null args");
  RGNN_CHECK_ARG(a->m >= 0 && a->n >= 0 && a->k1 >= 0 && a->k2 >= 0, "negative sizes");
  if (a->m == 0 || a->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(a->k1 + a->k2 > 0, "empty reduction dimension");
  RGNN_CHECK_ARG(a->out && a->W1 && (a->k1 == 0 || a->A1) && (a->k2 == 0 || a->A2), "null pointers");
  RGNN_CHECK_ARG(a->w_split >= a->n || a->W2, "w_split < n needs W2");
  RGNN_CHECK_ARG(a->m < ((int64_t)1 << 31) * BM, "m too large");
  LinParams p;
  p.sk_ws = nullptr; p.sk_flags = nullptr; p.no_split_k = RGNN_ENV("RGNN_DMA_NOPSK") != nullptr;
  { static const int stg = RGNN_ENV("RGNN_DMA_STAGGER") ? atoi(RGNN_ENV("RGNN_DMA_STAGGER")) : 100; p.stagger = stg; }   // per cent of the default start stagger (0: off)
  p.a1_aff = a->a1_scale_shift; p.a1_relu = a->a1_relu; p.a1_aff_panel = a->a1_panel_segment;
  RGNN_CHECK_ARG(a->a1_panel_segment == nullptr || a->a1_scale_shift != nullptr, "a1_panel_segment needs a1_scale_shift");
  p.relu_lo = a->relu_from_col > 0 ? a->relu_from_col : 0;
  p.fmt = 0; p.a1_bound = a->a1_bound; p.a2_bound = a->a2_bound; p.out_absmax = a->out_absmax;
  if (a->out_absmax != nullptr && !takes_dma_kernel(a)) {
    rgnn_set_error("rgnn_linear_fwd: out_absmax needs the LDS-DMA kernel (rgnn_linear_fwd_path != 0)");
    return RGNN_ERR_UNSUPPORTED;
  }
  if (a->a1_scale_shift != nullptr && !rgnn_linear_fwd_fuses_a1_affine(a)) {
    rgnn_set_error("rgnn_linear_fwd: a1_scale_shift needs the LDS-DMA kernel (rgnn_linear_fwd_fuses_a1_affine): apply "
                   "rgnn_scale_shift_act to A1 instead");
    return RGNN_ERR_UNSUPPORTED;
  }
  p.A1 = a->A1; p.A2 = a->A2; p.lda1 = a->lda1; p.lda2 = a->lda2; p.k1 = a->k1; p.k2 = a->k2;
  p.W1 = a->W1; p.W2 = a->W2; p.ldw = a->ldw; p.w_split = a->w_split >= a->n ? a->n : a->w_split;
  p.bias1 = a->bias1; p.bias2 = a->bias2;
  p.residual = a->residual; p.ldr = a->ldr;
  p.out = a->out; p.ldo = a->ldo; p.m = a->m; p.n = a->n; p.relu_out = a->relu_out; p.col_stats = a->col_stats;
  p.row_index = a->row_index; p.m_dev = a->m_dev; p.accumulate = a->accumulate;
  p.gather_only = a->gather_only; p.res_index = a->residual_index;
  RGNN_CHECK_ARG(a->row_index != nullptr || (a->m_dev == nullptr && a->accumulate == 0), "m_dev / accumulate need row_index");
  RGNN_CHECK_ARG(!(a->accumulate && a->col_stats), "col_stats of an accumulating launch are not defined (statistics are sums about a pivot of the stored values)");
  p.mt = (int)((a->m + BM - 1) / BM);
  const bool vec = (a->k1 % 4 == 0) && (a->k2 % 4 == 0) && (a->ldw % 4 == 0) && aligned16(a->W1) &&
                   (a->W2 == nullptr || aligned16(a->W2)) &&
                   (a->k1 == 0 || (a->lda1 % 4 == 0 && aligned16(a->A1))) &&
                   (a->k2 == 0 || (a->lda2 % 4 == 0 && aligned16(a->A2)));
  // buffer-descriptor path: single weight block, extents below 2 GiB, [A1|A2] split on a k-step boundary
  const int64_t rows_a = a->m;  // row_index launches pass the full row count of the matrices as m
  const int64_t e1 = a->k1 ? ((rows_a - 1) * a->lda1 + a->k1) * 4 : 0;
  const int64_t e2 = a->k2 ? ((rows_a - 1) * a->lda2 + a->k2) * 4 : 0;
  const int64_t ew = ((int64_t)(a->n - 1) * a->ldw + a->k1 + a->k2) * 4;
  const int64_t lim = ((int64_t)1 << 31) - 64;
  const bool bufl = vec && a->k1 > 0 && (a->w_split >= a->n) && e1 < lim && e2 < lim && ew < lim &&
                    (a->k2 == 0 || a->k1 % BK == 0) && RGNN_ENV("RGNN_LINEAR_NO_BUFL") == nullptr;
  p.ext_a1 = (int)e1; p.ext_a2 = (int)e2; p.ext_w = (int)ew;
  p.fast_epilogue = (a->n % 4 == 0) && (a->ldo % 4 == 0) && aligned16(a->out) &&
                    (a->residual == nullptr || (a->ldr % 4 == 0 && aligned16(a->residual))) &&
                    (a->bias1 == nullptr || aligned16(a->bias1)) && (a->bias2 == nullptr || aligned16(a->bias2)) &&
                    (a->w_split >= a->n || a->w_split % 4 == 0);
  const int64_t eo = ((a->m - 1) * a->ldo + a->n) * 4;
  p.direct_epilogue = a->row_index == nullptr && a->residual == nullptr && eo < lim && RGNN_ENV("RGNN_LINEAR_NO_DIRECT") == nullptr;
  p.ext_out = (int)(eo < lim ? eo : 0);
  hipStream_t s = (hipStream_t)stream;
  if (a->k2 == 0 && a->k1 <= 8 && a->n <= 64 && a->w_split >= a->n && a->residual == nullptr && a->row_index == nullptr &&
      a->col_stats == nullptr && a->m >= 4096 && a->relu_from_col <= 0 && RGNN_ENV("RGNN_LINEAR_NO_TINY") == nullptr) {
    const bool v4 = (a->n % 4 == 0) && (a->ldo % 4 == 0) && aligned16(a->out);
    const int64_t threads = a->m * ((a->n + 3) / 4);
    const int64_t tiny_blocks = rgnn_blocks(threads, 256) < 4096 ? rgnn_blocks(threads, 256) : 4096;
    rgnn_prof_begin(s);
    if (v4)
      hipLaunchKernelGGL(k_linear_tiny<true>, dim3((unsigned)tiny_blocks), dim3(256), 0, s, (const float*)a->A1, a->lda1,
                         a->k1, (const float*)a->W1, a->ldw, (const float*)a->bias1, a->m, a->n, a->relu_out, (float*)a->out, a->ldo);
    else
      hipLaunchKernelGGL(k_linear_tiny<false>, dim3((unsigned)tiny_blocks), dim3(256), 0, s, (const float*)a->A1, a->lda1,
                         a->k1, (const float*)a->W1, a->ldw, (const float*)a->bias1, a->m, a->n, a->relu_out, (float*)a->out, a->ldo);
    rgnn_prof_end(s);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  // bf16x3 path (see k_linear_x3): pre-split weight planes supplied, buffer-descriptor operands, direct epilogue
  p.Wp = a->W_planes; p.kp = a->w_planes_kp;
  p.ext_wp = 0;
  const bool x3_subset = a->row_index != nullptr && !a->accumulate && !a->gather_only && a->residual == nullptr && eo < lim &&
                         a->relu_from_col <= 0;   // (the row-subset epilogue clamps every column or none)
  if (a->W_planes && bufl && (p.direct_epilogue || x3_subset) && a->w_planes_kp >= a->k1 + a->k2 && a->w_planes_kp % BK == 0 &&
      (int64_t)3 * a->n * a->w_planes_kp * 2 < lim && RGNN_ENV("RGNN_LINEAR_FP32") == nullptr) {
    p.ext_wp = (int)((int64_t)3 * a->n * a->w_planes_kp * 2);
    rgnn_prof_begin(s);
    // LDS-DMA staged kernel (linear_dma.hip): wide layers whose reduction splits into whole k-steps of 16
    // (32 < n <= 64, e.g. the last conv layer's update: few MFMAs per k-step, the kernel then runs at the rate its
    // activation stream arrives -- still ahead of the fp32-MFMA kernel the narrow layers used to take)
    static const int dma_min_n = RGNN_ENV("RGNN_DMA_MIN_N") ? atoi(RGNN_ENV("RGNN_DMA_MIN_N")) : 32;
    if (a->n > dma_min_n && (a->k1 + a->k2) % 16 == 0 && a->k1 % 16 == 0 && RGNN_ENV("RGNN_X3_NODMA") == nullptr) {
      if (a->splitk_ws && a->splitk_ws_bytes >= rgnn_linear_splitk_ws_bytes() && RGNN_ENV("RGNN_DMA_NOSK") == nullptr) {
        p.sk_ws = a->splitk_ws;
        p.sk_flags = (int*)((char*)a->splitk_ws + (int64_t)256 * 8 * 16 * 512 * 4);
      }
      if (takes_f16_form(a)) {                          // two f16 planes (+ footer) instead of three bf16 planes
        p.fmt = 1; p.Wp = a->W_planes_f16;
        p.ext_wp = (int)((int64_t)2 * a->n * a->w_planes_kp * 2);
      }
      const int rc_dma = rgnn_linear_dma_launch(&p, x3_subset ? 1 : 0, s);
      rgnn_prof_end(s);
      if (rc_dma != RGNN_OK) return rc_dma;
      RGNN_CHECK_LAUNCH();
      return RGNN_OK;
    }
    if (a->a1_scale_shift != nullptr) {               // (takes_dma_kernel and this dispatcher must agree)
      rgnn_set_error("rgnn_linear_fwd: internal: a1_scale_shift accepted but the LDS-DMA kernel was not selected");
      return RGNN_ERR_UNSUPPORTED;
    }
    // 256 x 256 tiles (k-step 16) move 28 % fewer operand bytes per flop than 256 x 128 (k-step 32) and measure 5 - 15 %
    // faster, unless they pad more columns (N = 272: 512 against 384)
    const int pad_w = (a->n + 255) / 256 * 256, pad_n = (a->n + 127) / 128 * 128;
    const bool wide = a->n > 128 && pad_w * 100 <= pad_n * 107 && RGNN_ENV("RGNN_X3_NARROW") == nullptr;
#define RGNN_X3(IDX)                                                                                    \
  do {                                                                                                  \
    if (wide) { p.nt = (a->n + 255) / 256; launch_x3<IDX, 256, 256, 2, 4, 4, 2, 16, 1>(p, s); }        \
    else if (a->n > 64) { p.nt = (a->n + 127) / 128; launch_x3<IDX, 256, 128, 4, 2, 2, 2>(p, s); }     \
    else if (a->n > 32) { p.nt = 1; launch_x3<IDX, 256, 64, 4, 2, 2, 1>(p, s); }                       \
    else { p.nt = 1; launch_x3<IDX, 256, 32, 8, 1, 1, 1>(p, s); }                                      \
  } while (0)
    if (x3_subset) RGNN_X3(true); else RGNN_X3(false);
#undef RGNN_X3
    rgnn_prof_end(s);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  if (a->a1_scale_shift != nullptr && !bufl) {        // (the fp32 kernel applies it on its buffer-descriptor path only)
    rgnn_set_error("rgnn_linear_fwd: internal: a1_scale_shift accepted but no kernel that applies it was selected");
    return RGNN_ERR_UNSUPPORTED;
  }
  rgnn_prof_begin(s);
  // column tiling: 32*TN-wide tiles (4 waves stacked in M, each 32 x 32*TN) when that wastes fewer padded columns
  // than 128-wide tiles -- N = 224 -> one 224 tile (0 % instead of 12.5 % padding), 464 -> 3 x 160, 928 -> 6 x 160
  int wide_tn = 0;
  // (measured, M = 192k: N = 464 / K = 224 +5 %, N = 224 / K = 688 +4 %; not worth it for short reductions or N > 512)
  if (bufl && a->row_index == nullptr && a->n > 128 && a->n <= 512 && a->k1 + a->k2 >= 192) {
    const int base = ((a->n + 127) / 128) * 128;
    int best = base;
    for (int tn = 3; tn <= 7; tn++) {
      const int w = 32 * tn, padded = ((a->n + w - 1) / w) * w;
      if (padded < best || (padded == best && wide_tn)) { best = padded; wide_tn = tn; }
    }
    if (best * 100 > base * 95) wide_tn = 0;  // needs >= 5 % fewer padded columns
  }
  if (wide_tn) {
    p.nt = (a->n + 32 * wide_tn - 1) / (32 * wide_tn);
    switch (wide_tn) {
      case 3: launch<96, 4, 1, 1, 3, 1, true>(p, vec, bufl, s); break;
      case 4: launch<128, 4, 1, 1, 4, 1, true>(p, vec, bufl, s); break;
      case 5: launch<160, 4, 1, 1, 5, 1, true>(p, vec, bufl, s); break;
      case 6: launch<192, 4, 1, 1, 6, 1, true>(p, vec, bufl, s); break;
      default: launch<224, 4, 1, 1, 7, 1, true>(p, vec, bufl, s); break;
    }
  } else if (a->n > 64) {
    p.nt = (a->n + 127) / 128;
#if RGNN_WAVES8
    launch<128, 4, 2, 1, 2>(p, vec, bufl, s);   // 8 waves of 32x64
#else
    launch<128, 2, 2, 2, 2>(p, vec, bufl, s);   // 4 waves of 64x64
#endif
  } else if (a->n > 32) {
    p.nt = 1;
    launch<64, 2, 2, 2, 1>(p, vec, bufl, s);
  } else {
    p.nt = 1;
    launch<32, 4, 1, 1, 1>(p, vec, bufl, s);
  }
  rgnn_prof_end(s);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int32_t rgnn_linear_planes_kp(int32_t k) { return (k + BK - 1) / BK * BK; }

extern "C" int rgnn_linear_split_weights(const float* W1, const float* W2, int64_t ldw, int32_t w_split, int32_t n, int32_t k,
                                         void* planes, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && k >= 0, "negative sizes");
  if (n == 0 || k == 0) return RGNN_OK;
  RGNN_CHECK_ARG(W1 && planes && (w_split >= n || W2), "null pointers");
  const int kp = rgnn_linear_planes_kp(k);
  hipLaunchKernelGGL(k_split_weights, dim3(rgnn_blocks((int64_t)n * kp, 256)), dim3(256), 0, (hipStream_t)stream, W1, W2, ldw,
                     w_split >= n ? n : w_split, n, k, kp, (__bf16*)planes);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int64_t rgnn_linear_planes_f16_bytes(int32_t n, int32_t k) {
  return (int64_t)2 * n * rgnn_linear_planes_kp(k) * 2 + 16 + W_ABSMAX_PARTS * 4;      // planes, footer, scratch of the absmax launch
}

extern "C" int rgnn_linear_split_weights_f16(const float* W1, const float* W2, int64_t ldw, int32_t w_split, int32_t n, int32_t k,
                                             void* planes, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && k >= 0, "negative sizes");
  if (n == 0 || k == 0) return RGNN_OK;
  RGNN_CHECK_ARG(W1 && planes && (w_split >= n || W2), "null pointers");
  const int kp = rgnn_linear_planes_kp(k);
  float* foot = (float*)((char*)planes + (int64_t)2 * n * kp * 2);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = rgnn_blocks((int64_t)n * k, 1024);
  hipLaunchKernelGGL(k_weights_absmax, dim3((unsigned)(nb < W_ABSMAX_PARTS ? nb : W_ABSMAX_PARTS)), dim3(256), 0, s, W1, W2, ldw,
                     w_split >= n ? n : w_split, n, k, foot + 4);
  hipLaunchKernelGGL(k_split_weights_f16, dim3(rgnn_blocks((int64_t)n * kp, 256)), dim3(256), 0, s, W1, W2, ldw,
                     w_split >= n ? n : w_split, n, k, kp, (_Float16*)planes, foot);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
