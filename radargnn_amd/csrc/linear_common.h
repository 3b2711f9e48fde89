// Pieces shared by the dense-layer kernels (linear.hip: fp32 MFMA and register-staged bf16x3; linear_dma.hip: the
// LDS-DMA staged bf16x3 kernel): launch parameters, buffer-descriptor loads, the fp32 -> 3 x bf16 split and the
// epilogue that stores straight from the MFMA accumulator layout.
#pragma once
#include "common.h"
#include "stats.h"
#include <type_traits>
#include <stdlib.h>

#ifndef RGNN_EPI_AUX
#define RGNN_EPI_AUX 0     // cache policy of the epilogue's stores (2 = nt, streaming)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// A "bound" (f16x2 dense form) is RGNN_BOUND_SLOTS float words: producers raise slot (work-group id & 255) with one atomic max
// each, consumers take the maximum of all slots when they start.  (One shared word was measured first: the ~2 000 same-address
// atomics at the end of a launch serialise in L2 and cost 30 - 60 us per launch.)
constexpr int BOUND_SLOTS = RGNN_BOUND_SLOTS;
__device__ __forceinline__ float bound_read(const float* __restrict__ b) {     // every lane of the wave returns max over the slots
  const int lane = threadIdx.x & 63;
  float v = fmaxf(fmaxf(b[lane], b[lane + 64]), fmaxf(b[lane + 128], b[lane + 192]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void bound_raise(float* __restrict__ b, int slot, float v) {   // (non-negative floats order like their bits)
  atomicMax((unsigned int*)b + (slot & (BOUND_SLOTS - 1)), __float_as_uint(v));
}

constexpr int BM = RGNN_STAT_PANEL_ROWS;   // (also the height of a column-statistics panel)
constexpr int BK = 32;
constexpr int LDK = 36;

struct LinParams {
  const float* A1; const float* A2; int64_t lda1, lda2; int k1, k2;
  const float* W1; const float* W2; int64_t ldw; int w_split;
  const float* bias1; const float* bias2;
  const float* residual; int64_t ldr;
  float* out; int64_t ldo;
  int64_t m; int n;
  int relu_out;
  float* col_stats;
  int mt, nt;  // tiles
  int ext_a1, ext_a2, ext_w;  // byte extents of A1 / A2 / W for the buffer descriptors (BUFL path)
  const int32_t* row_index;  // tile row r works on matrix row row_index[r] (A, residual and out); NULL = identity
  const int64_t* m_dev;      // row count read from device memory (data-dependent subsets); NULL = use m
  int gather_only;           // IDX: only the A rows are gathered; output rows are the tile rows (compact result)
  const int32_t* res_index;  // per output row: row of `residual` to add, or -1 (NULL: residual row = output row)
  int accumulate;            // out += result (column statistics then hold the CHANGE of sum / sum of squares)
  int fast_epilogue;  // n, ldo, ldr multiples of 4 and 16-B aligned pointers: vectorised epilogue through LDS
  int direct_epilogue;  // no residual / row_index, out extent < 2 GiB: buffer stores straight from the MFMA layout
  int ext_out;
  const void* Wp;       // bf16x3 path: the weight as three bf16 planes [3][n][kp] (rgnn_linear_split_weights)
  int kp, ext_wp;
  void* sk_ws; int* sk_flags;   // stream-K hand-over workspace of the LDS-DMA kernel (NULL: static tile schedule)
  int no_split_k;               // never cut an item's k-loop over parallel work-groups (few-row launches; RGNN_DMA_NOPSK)
  int stagger;                  // LDS-DMA kernel, static schedule: start stagger of the work-groups in per cent of the default (linear_dma.hip)
  const float* a1_aff; int a1_relu;   // A1 := act(A1 * a1_aff[0][k] + a1_aff[1][k]) on its way into the kernel (NULL: none)
  int relu_lo;                        // relu_out applies to the columns >= relu_lo only (two heads in one launch)
  // f16x2 form of the LDS-DMA kernel (fmt 1): operands as two f16 terms after an exact power-of-two pre-scale, three MFMA
  // products per fp32 product.  Wp then holds the two f16 weight planes + their 16-byte footer (absmax, scale, 1 / scale);
  // a1_bound / a2_bound: device words holding an upper bound of |A1'| (after the optional affine) / |A2|.
  int fmt;
  const float* a1_bound; const float* a2_bound;
  float* out_absmax;                  // optional device word: atomic max of |out| (what the next layer's a*_bound reads)
  const int* a1_aff_panel;            // optional: a1_aff is [S][2][k1] and 256-row tile p of the row list takes table a1_aff_panel[p]
};

// IDX: row-subset form (row_index / m_dev / accumulate); kept out of the common instantiation, whose register
// budget is tight (215 VGPRs, SGPRs already spilling).
// NB: the builtin's result must be received in a GCC-style vector; an ext_vector_type(4) receiver silently turns
// the load into a 4-byte load splat over the four lanes (hipcc 7.2).
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;  // byte offset beyond any buffer extent (< 2^31): the buffer load returns 0

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  const f32x4v f = __builtin_bit_cast(f32x4v, v);
  return make_float4(f.x, f.y, f.z, f.w);
}

// Epilogue straight from the accumulator layout (column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): a
// lane owns ONE output column per 32-wide sub-tile, so bias is one register, the BatchNorm column sums are per-lane
// running sums (one cross-lane add at the end), and the store is a buffer_store_dword whose per-lane offset (column,
// + 4 rows for the upper half-wave) is fixed per sub-tile while the row advances in an SGPR; columns beyond n get an
// out-of-range offset and are dropped by the hardware (the SGPR offset takes no part in the range check, so the rows
// beyond M of the last panel are masked per element).  ~4 VALU operations and one store per element, no LDS round
// trip and -- without statistics -- no barrier.  `stage`: LDS nobody reads any more (stat_lds_floats(WGM, BN) floats are used).
//
// ROWS (row-subset launches: tile row r = matrix row row_index[r], M rows in the subset): the byte offsets of the panel's
// BMT output rows come from a small LDS table instead of the running SGPR (entries beyond M hold the out-of-range
// offset); the column statistics are those of the compact tile rows -- callers sum all panels, so the numbering is free.
// AMAX: the lane's running maximum of |v| over everything it stores is returned (amax_in carried on); callers that do not
// track it pass nothing and ignore the result.  (By value on purpose: through a pointer hipcc kept the running maximum in
// scratch memory and the epilogue slowed down by 30 - 60 us per launch.)
template <int BN, int WGM, int WGN, int TM, int TN, int BMT = 128, bool ROWS = false, bool AMAX = false>
__device__ __forceinline__ float direct_epilogue(const LinParams& p, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int panel,
                                                 int64_t M, float* stage, int* row_tab = nullptr,
                                                 const float* bias_regs = nullptr, const float acc_scale = 1.f,
                                                 const float amax_in = 0.f) {
  float amax = amax_in;
  constexpr int THREADS = WGM * WGN * 64;
  constexpr int H = BMT / RGNN_STAT_PANEL_ROWS, WH = WGM / H;   // the column statistics are kept per 128-row panel: H panels per tile
  static_assert(WGM * TM * 32 == BMT && H * WH == WGM, "tile config");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, (short)0, p.ext_out, 0x00020000);
  const int ldo4 = (int)p.ldo * 4;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int wm_u = wv / WGN, wn_u = wv % WGN;
  const bool do_stats = p.col_stats != nullptr;
  const bool full = m0 + BMT <= M;         // (ROWS: the table masks the rows beyond M of the last panel)
  float* stat_lds = stage;                 // [WGM][BN][STAT_LDS_ROWS] + [WGM] counts (stats.h)
  if constexpr (ROWS) {
#ifdef RGNN_EPI_ABL_NO_INDEX     // (experiment: no index loads -- wrong rows)
    for (int r = t; r < BMT; r += THREADS) row_tab[r] = (m0 + r < M) ? (int)(m0 + r) * ldo4 : OOB;
#else
    for (int r = t; r < BMT; r += THREADS) {           // (an entry of -1 is an absent row: padding of a segmented list)
      int ri = (m0 + r < M) ? p.row_index[m0 + r] : -1;
#ifdef RGNN_EPI_ABL_SMALLOUT     // (experiment: every store lands in the first 4 096 output rows -- cache-resident, no HBM write traffic; wrong results)
      if (ri >= 0) ri &= 4095;
#endif
      row_tab[r] = (ri >= 0) ? ri * ldo4 : OOB;
    }
#endif
    __syncthreads();
  }
  auto run = [&](auto relu_c, auto stats_c) {
    constexpr bool RELU = decltype(relu_c)::value;
    constexpr int STATS = decltype(stats_c)::value & 1;  // column statistics wanted
    constexpr int MASK = decltype(stats_c)::value >> 1;  // last row panel: rows >= M are neither stored nor counted
    float lane_rows = (float)(TM * 16);                    // rows this lane counts into the statistics (the same for every column)
    if constexpr (STATS != 0 && (ROWS || MASK != 0)) {
      lane_rows = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          bool okr;
          if constexpr (ROWS) okr = row_tab[(wm_u * TM + i) * 32 + rr] != OOB;
          else okr = (int64_t)((int)m0 + (wm_u * TM + i) * 32) + rr < M;
          lane_rows += okr ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int gn = n0 + (wn_u * TN + j) * 32 + (lane & 31);
      const bool ncol = gn < p.n;
      // (bias_regs: the caller holds this lane's TN bias values of the column tile in registers -- a load here is followed by
      //  `s_waitcnt vmcnt(0)`, which in the LDS-DMA kernel also waits for every operand piece prefetched for the next tile)
      float bias = 0.f;
      if (bias_regs != nullptr) {
        bias = bias_regs[j];
      } else if (ncol) {
        const float* bp = (gn < p.w_split) ? p.bias1 : p.bias2;
        if (bp) bias = bp[(gn < p.w_split) ? gn : gn - p.w_split];
      }
      const float rlo = (gn >= p.relu_lo) ? 0.f : -INFINITY;   // (columns below relu_lo keep their sign)
      const int vo = ncol ? ((ROWS ? 0 : (lane >> 5) * 4 * (int)p.ldo) + gn) * 4 : OOB;
      float s1 = 0.f, s2 = 0.f, piv = 0.f;                 // sums about the pivot = the lane's first value of this column
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int rowb = (int)m0 + (wm_u * TM + i) * 32;  // (m < 2^31 / ldo on this path)
        int so = ROWS ? 0 : __builtin_amdgcn_readfirstlane(rowb * ldo4);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rr = (r & 3) + 8 * (r >> 2);
          // (acc_scale: exact power of two that undoes the operand pre-scale of the f16x2 form; 1 elsewhere -- the product is
          //  exact, so this is the same value as before for the other kernels)
          float v = acc[i][j][r] * acc_scale + bias;
          if (RELU) v = fmaxf(v, rlo);
#if defined(__HIP_DEVICE_COMPILE__)
          // (opaque on purpose: hipcc re-associates a chain of fmaxf into a tree, keeps all 16 TN values of the tile alive for
          //  it and spills hundreds of registers; one dependent v_max per element costs nothing next to its store)
          if constexpr (AMAX) asm("v_max_f32 %0, %0, |%1|" : "+v"(amax) : "v"(v));
#endif
          if constexpr (ROWS) {
            const int rof = row_tab[(wm_u * TM + i) * 32 + rr + 4 * (lane >> 5)];
            const bool okr = rof != OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), ro, (rof | (vo & OOB)) + (vo & 0x7fffffff), 0, RGNN_EPI_AUX);
            if (STATS) { if (i == 0 && r == 0) piv = v; const float d = v - piv; s1 += okr ? d : 0.f; s2 += okr ? d * d : 0.f; }
          } else {
            const bool okr = !MASK || ((int64_t)rowb + rr + 4 * (lane >> 5) < M);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), ro, okr ? vo : OOB, so, RGNN_EPI_AUX);
            so += ((r & 3) == 3) ? 5 * ldo4 : ldo4;  // one running SGPR instead of 16 precomputed row offsets
            if (STATS) { if (i == 0 && r == 0) piv = v; const float d = v - piv; s1 += okr ? d : 0.f; s2 += okr ? d * d : 0.f; }
          }
        }
      }
      if (STATS) {
        const ColStat both = stat_merge_xor(stat_make(lane_rows, piv, s1, s2), 32);
        if (lane < 32) {
          float* d3 = stat_lds + (wm_u * BN + (wn_u * TN + j) * 32 + lane) * STAT_LDS_ROWS;
          d3[0] = both.piv; d3[1] = both.s1; d3[2] = both.s2;
          if (j == 0 && lane == 0 && wn_u == 0) stat_lds[WGM * BN * STAT_LDS_ROWS + wm_u] = both.n;   // (the same for every column)
        }
      }
    }
  };
  using T = std::true_type; using F = std::false_type;
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
  if (p.relu_out) {
    if (full) { if (do_stats) run(T{}, S1{}); else run(T{}, S0{}); }
    else { if (do_stats) run(T{}, S3{}); else run(T{}, S2{}); }
  } else {
    if (full) { if (do_stats) run(F{}, S1{}); else run(F{}, S0{}); }
    else { if (do_stats) run(F{}, S3{}); else run(F{}, S2{}); }
  }
  if (do_stats) {
    __syncthreads();
    for (int c = t; c < BN * H; c += THREADS) {
      const int hh = c / BN, cc = c - hh * BN;
      const int gc = n0 + cc;
      const int64_t sp = (int64_t)panel * H + hh;        // 128-row statistics panel
      if (gc < p.n && sp * RGNN_STAT_PANEL_ROWS < M) {
        ColStat a = stat_make(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < WH; w++) {
          const float* s3 = stat_lds + ((hh * WH + w) * BN + cc) * STAT_LDS_ROWS;
          a = stat_merge(a, stat_make(stat_lds[WGM * BN * STAT_LDS_ROWS + hh * WH + w], s3[0], s3[1], s3[2]));
        }
        stat_store(p.col_stats + (sp * RGNN_STAT_ROWS) * p.n + gc, p.n, a);
      }
    }
    __syncthreads();  // stat_lds is free again (the fp32 kernel: it is the next tile's first staging buffer)
  }
  return amax;
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

// fp32 -> two f16 terms, a = h + l up to 2^-23 |a| (h = rn(a): 11 significand bits, l = rn(a - h): the next 11; a - h is exact
// in fp32).  Callers scale a by a power of two first so that h cannot overflow and l stays a normal f16 number for every
// element within 2^-19 of the tensor's bound (linear_dma.hip).
__device__ __forceinline__ void split2(const float4 v, f16x4_t& h, f16x4_t& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const _Float16 hh = (_Float16)x[i];
    h[i] = hh;
    l[i] = (_Float16)(x[i] - (float)hh);
  }
}

__device__ __forceinline__ void split3(const float4 v, bf16x4_t& h, bf16x4_t& m, bf16x4_t& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const __bf16 hh = (__bf16)x[i];
    const float r1 = x[i] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    h[i] = hh; m[i] = mm; l[i] = (__bf16)r2;
  }
}

}  // namespace
