// Weight gradient of a dense layer on the bf16 matrix pipe:
//
//     dW[n, k] = sum over rows r of  G[r, n] * [A1 | A2 | 1][r, k]          (G = gradient of the layer output, A = its input)
//
// -- what autograd derives for the reference's torch_geometric Linear / torch.nn.Linear inside the MLPs and convs
// (gnn/gnn_models.py:137-178, gnn/mpnn_layers.py:64-74; trained by gnn/trainer.py:176-231).  The optional column of ones
// makes the bias gradient the last column of the same product.  Rows may be a device-side row list (row_index / m_dev:
// the targets with / without incoming edges that the forward pass updates with different weights).
//
// A GEMM whose reduction runs over the ROWS of both operands.  fp32 products as in the forward kernels (linear.hip): every
// operand value is split into three bf16 terms, six MFMA products per fp32 product (l h', h l', m m', m h', h m', h h'), fp32
// accumulate -- 2.7x the rate of v_mfma_f32_32x32x2_f32 at the same error class.  v_mfma_f32_32x32x16_bf16 wants, per lane,
// 8 consecutive REDUCTION slots of one output row: with the reduction along r that is a column of G (or A) -- so the tiles
// are transposed on their way into LDS: a lane loads 8 rows of ONE column (buffer_load_dword: the 64 lanes of a wave read
// 256 contiguous bytes of a row; the row offset travels in an SGPR), splits them and writes the three bf16x8 terms as
// ds_write_b128 into [plane][column][16 rows] -- the layout the forward kernel keeps its weight planes in, same XOR
// swizzle, conflict-free on both sides.  Every element is split once per work-group, not once per wave that uses it.
//
// Work decomposition: 128 (n) x 256 (k) output tiles x row slabs; 4 waves (2 x 2, each 64 x 128 = 2 x 4 MFMA tiles, 48
// MFMAs per 16-row step against 18 ds_read_b128); a work-group reduces its slab in registers and stores the partial tile,
// k_wg_reduce sums the slabs (no atomics, deterministic).  The tiles of one slab sit on the same XCD (block b -> XCD b % 8),
// so a slab of G / A leaves HBM once.  Two LDS stages (2 x 36 KiB): the global loads of step s + 2 are in flight while step
// s + 1 is split into the other stage and step s multiplies.
#include "linear_common.h"

#ifndef RGNN_WG_SCHED
#define RGNN_WG_SCHED 1
#endif

namespace {

constexpr int WG_BN = 128, WG_BK = 256, WG_THREADS = 256;
constexpr int WG_COLS = WG_BN + WG_BK;                 // columns of a step: [G tile | A tile]
constexpr int WG_PLANE = WG_COLS * 32;                 // bytes of one plane of a stage: 384 columns x 16 bf16
constexpr int WG_STAGE = 3 * WG_PLANE;                 // 36 KiB

struct Wg3Params {
  const float* G; int64_t ldg; int n;
  const float* A1; int64_t lda1; int k1;
  const float* A2; int64_t lda2; int k2;
  int ones;                        // a virtual column of ones behind [A1 | A2]
  int64_t m;                       // rows (of the row list, if any)
  const int32_t* row_index; const int64_t* m_dev;
  float* part; int slabs; int nt, kt;
  // f16x2 form (k_wgrad_x3<true>): bounds (RGNN_BOUND_SLOTS words each, rgnn.h) of |G| and of |A1|, |A2| -- both operands are
  // pre-scaled by exact powers of two derived from them, split into TWO f16 terms, three products (l h', h l', h h')
  const float* g_bound; const float* a1_bound; const float* a2_bound;
};

typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wg_bound_read(const float* __restrict__ b, int lane) {   // max over the slots of a bound, wave-uniform
  float m = fmaxf(fmaxf(b[lane], b[lane + 64]), fmaxf(b[lane + 128], b[lane + 192]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}
__device__ __forceinline__ int wg_scale_exp(float bound) {             // biased exponent se: bound * 2^(se - 127) < 2^15
  const int be = (int)((__float_as_uint(bound) >> 23) & 255u);
  const int se = 268 - be;
  return se > 253 ? 253 : se;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* base) {
  // num_records = 2^31 - 1: only the "killed" offset 0x80000000 is out of range; the row offset goes through the SGPR
  // operand, which takes no part in the range check (rows up to 4 GiB into the matrix)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, 0x7fffffff, 0x00020000);
}

template <bool F16>
__global__ __launch_bounds__(WG_THREADS) void k_wgrad_x3(const Wg3Params p) {
  extern __shared__ __attribute__((aligned(16))) char wg_lds[];
  constexpr int NPL = F16 ? 2 : 3;                  // planes of a stage
  constexpr int STAGE = NPL * WG_PLANE;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = p.nt * p.kt;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tile = idx % tiles;
  const int slab = (idx / tiles) * 8 + xcd;
  if (slab >= p.slabs) return;
  const int n0 = (tile / p.kt) * WG_BN, k0 = (tile % p.kt) * WG_BK;
  const int Kt = p.k1 + p.k2 + (p.ones ? 1 : 0);    // columns of dW
  const int64_t M = p.m_dev ? *p.m_dev : p.m;
  // slabs of whole 16-row steps
  const int64_t steps_total = (M + 15) / 16;
  const int64_t steps_per = (steps_total + p.slabs - 1) / p.slabs;
  const int64_t s_beg = (int64_t)slab * steps_per;
  const int64_t s_end = (s_beg + steps_per < steps_total) ? s_beg + steps_per : steps_total;

  // ---- this thread's items of a 16-row step.  G tile (128 columns x 2 row halves): column t % 128, half t / 128.  A tile
  // (256 columns x 2 halves): column t, both halves.  The k axis of dW is laid out VIRTUALLY as [A1 padded to 64 | A2 (| 1)
  // padded to 64], so the 64 columns of a wave come from ONE matrix: one descriptor per wave, chosen by scalar selects, one
  // load per element and no branch (the epilogue maps the virtual columns back).
  const int ones1 = (p.ones && p.k2 == 0) ? 1 : 0, ones2 = (p.ones && p.k2 > 0) ? 1 : 0;   // the ones follow the last matrix
  const int k1p = (p.k1 + ones1 + 63) & ~63;
  const int vk = k0 + t;                            // this thread's virtual k column
  const bool use1 = __builtin_amdgcn_readfirstlane(k0 + 64 * wave) < k1p;
  const __amdgpu_buffer_rsrc_t rg = wg_rsrc(p.G);
  const __amdgpu_buffer_rsrc_t ra = wg_rsrc(use1 ? (const void*)p.A1 : (p.A2 ? (const void*)p.A2 : (const void*)p.G));
  const unsigned ldg4 = (unsigned)(p.ldg * 4), lda4 = (unsigned)((use1 ? p.lda1 : p.lda2) * 4);
  // f16x2 form: G 2^sg and A 2^sa stay below 2^15; the column of ones holds 2^14 / 2^sa (2^14 after the pre-scale, whatever sa is),
  // and the epilogue multiplies by 2^-(sg + sa) -- by 2^-(sg + 14) in the column of ones
  float g_mul = 1.f, a_mul = 1.f, out_mul = 1.f, ones_mul = 1.f;
  unsigned one_val = 0x3f800000u;
  if (F16) {
    const int seg = wg_scale_exp(wg_bound_read(p.g_bound, lane));
    float ab = 0.f;
    if (p.k1 > 0) ab = wg_bound_read(p.a1_bound, lane);
    if (p.k2 > 0) ab = fmaxf(ab, wg_bound_read(p.a2_bound, lane));
    const int sea = wg_scale_exp(ab);
    g_mul = __uint_as_float((unsigned)seg << 23);
    a_mul = __uint_as_float((unsigned)sea << 23);
    const float g_inv = __uint_as_float((unsigned)(254 - seg) << 23), a_inv = __uint_as_float((unsigned)(254 - sea) << 23);
    out_mul = g_inv * a_inv;
    ones_mul = g_inv * 0x1p-14f;
    one_val = __float_as_uint(0x1p14f * a_inv);
  }
  int va = OOB;
  unsigned one_bits = 0;
  if (use1) {
    if (vk < p.k1) va = vk * 4;
    else if (vk == p.k1 && ones1) one_bits = one_val;
  } else {
    const int u = vk - k1p;
    if (u < p.k2) va = u * 4;
    else if (u == p.k2 && ones2) one_bits = one_val;
  }
  const int gc = t & (WG_BN - 1);
  const int g_half = __builtin_amdgcn_readfirstlane(t >> 7);
  const int vg = (n0 + gc < p.n) ? (n0 + gc) * 4 : OOB;
  const int ldsw_g = gc * 32 + ((g_half ^ ((gc >> 3) & 1)) * 16);
  const int ca = WG_BN + t;
  const int ldsw_a0 = ca * 32 + ((0 ^ ((ca >> 3) & 1)) * 16), ldsw_a1 = ca * 32 + ((1 ^ ((ca >> 3) & 1)) * 16);

  // row ids of a step: lane r < 16 of every wave holds the matrix row of step row r (-1: beyond M)
  auto step_rows = [&](int64_t s) -> int {
    const int64_t m = s * 16 + (lane & 15);
    int r = -1;
    if (m < M && s < s_end) r = p.row_index ? p.row_index[m] : (int)m;
    return r;
  };
  float rawg[8], rawa[16];
  auto load_a = [&](int rid, int h) {                // rows 8 h .. 8 h + 7 of this thread's A column
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int row = __builtin_amdgcn_readlane(rid, 8 * h + j);
      const int kill = row & OOB;                   // row = -1: the offset goes out of range, the load returns 0
      const unsigned x = __builtin_amdgcn_raw_buffer_load_b32(ra, va | kill, (int)((unsigned)row * lda4), 0);
      // the column of ones (at most one lane of the launch; its load is killed and returns 0): 1.0 on valid rows -- as bit
      // operations, so that hipcc keeps the load unconditional instead of branching around it per element
      rawa[8 * h + j] = __builtin_bit_cast(float, x | (one_bits & ~(unsigned)(row >> 31)));
    }
  };
  auto load_g = [&](int rid) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int row = __builtin_amdgcn_readlane(rid, g_half * 8 + j);
      const int kill = row & OOB;
      rawg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, vg | kill, (int)((unsigned)row * ldg4), 0));
    }
  };
  auto load_step = [&](int rid) { load_g(rid); load_a(rid, 0); load_a(rid, 1); };
  auto put = [&](char* st, int off, const float* v, float mul) {   // 8 rows of one column -> three bf16x8 (two f16x8) terms
    if constexpr (F16) {
      f16x4_t h0, l0, h1, l1;
      split2(make_float4(v[0] * mul, v[1] * mul, v[2] * mul, v[3] * mul), h0, l0);
      split2(make_float4(v[4] * mul, v[5] * mul, v[6] * mul, v[7] * mul), h1, l1);
      *(f16x8_t*)(st + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *(f16x8_t*)(st + WG_PLANE + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      bf16x4_t h0, m0, l0, h1, m1, l1;
      split3(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
      split3(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
      *(bf16x8_t*)(st + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *(bf16x8_t*)(st + WG_PLANE + off) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
      *(bf16x8_t*)(st + 2 * WG_PLANE + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto store_stage = [&](int stage) {               // split the loaded step and write it, transposed, into a stage
    char* st = wg_lds + stage * STAGE;
    put(st, ldsw_g, rawg, g_mul);
    put(st, ldsw_a0, rawa, a_mul);
    put(st, ldsw_a1, rawa + 8, a_mul);
  };
  auto mma = [&](const wg_u32x4 a, const wg_u32x4 b, const f32x16 c) -> f32x16 {
    if constexpr (F16)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment addresses inside a stage (plane 0): column c, chunk lane >> 5, swizzled like the writes
  int g_off[2], a_off[4];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int c = wn * 64 + i * 32 + (lane & 31);
    g_off[i] = c * 32 + (((lane >> 5) ^ ((c >> 3) & 1)) * 16);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = WG_BN + wk * 128 + j * 32 + (lane & 31);
    a_off[j] = c * 32 + (((lane >> 5) ^ ((c >> 3) & 1)) * 16);
  }

  // 32-column groups that lie entirely in the padding (beyond n; in the gap behind A1; behind the last column) multiply
  // nothing: their MFMAs are skipped (wave-uniform, loop-invariant)
  bool g_live[2], a_live[4];
#pragma unroll
  for (int i = 0; i < 2; i++) g_live[i] = n0 + wn * 64 + i * 32 < p.n;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c0 = k0 + wk * 128 + j * 32;
    a_live[j] = (c0 < p.k1 + ones1) || (c0 >= k1p && c0 - k1p < p.k2 + ones2);
  }

  if (s_beg < s_end) {
    int rid = step_rows(s_beg);
    load_step(rid);
    store_stage(0);
    rid = step_rows(s_beg + 1);
    load_step(rid);                                 // step s_beg + 1 (all rows killed if the slab has one step)
    int rid_next = step_rows(s_beg + 2);
    __syncthreads();
    int cur = 0;
    for (int64_t s = s_beg; s < s_end; s++) {
      // This step's MFMAs (four column groups) with the next step's split (registers -> the other stage, free since the
      // barrier that ended the previous iteration) and the global loads of the step after that dealt in between them, so
      // that the matrix pipe works while the wave's VALU / LDS / VMEM instructions issue.
      const char* st = wg_lds + cur * STAGE;
      char* nx = wg_lds + (cur ^ 1) * STAGE;
      wg_u32x4 gf[2][NPL];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) gf[i][pl] = *(const wg_u32x4*)(st + pl * WG_PLANE + g_off[i]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // (every register set is reloaded -- with the step after the next -- as soon as its split has consumed it: a full
        // iteration in flight)
        if (j == 0) { put(nx, ldsw_g, rawg, g_mul); load_g(rid_next); }
        if (j == 1) { put(nx, ldsw_a0, rawa, a_mul); load_a(rid_next, 0); }
        if (j == 2) { put(nx, ldsw_a1, rawa + 8, a_mul); load_a(rid_next, 1); }
        if (j == 3) rid_next = step_rows(s + 3);
        if (RGNN_WG_SCHED) __builtin_amdgcn_sched_barrier(0);
        if (!a_live[j]) continue;
        wg_u32x4 af[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) af[pl] = *(const wg_u32x4*)(st + pl * WG_PLANE + a_off[j]);
        // smallest terms first; the two accumulators alternate (no back-to-back MFMAs on one accumulator)
        if constexpr (F16) {
          if (g_live[0] && g_live[1]) {
            acc[0][j] = mma(gf[0][1], af[0], acc[0][j]);
            acc[1][j] = mma(gf[1][1], af[0], acc[1][j]);
            acc[0][j] = mma(gf[0][0], af[1], acc[0][j]);
            acc[1][j] = mma(gf[1][0], af[1], acc[1][j]);
            acc[0][j] = mma(gf[0][0], af[0], acc[0][j]);
            acc[1][j] = mma(gf[1][0], af[0], acc[1][j]);
          } else if (g_live[0]) {
            acc[0][j] = mma(gf[0][1], af[0], acc[0][j]);
            acc[0][j] = mma(gf[0][0], af[1], acc[0][j]);
            acc[0][j] = mma(gf[0][0], af[0], acc[0][j]);
          }
        } else {
          if (g_live[0] && g_live[1]) {
            acc[0][j] = mma(gf[0][NPL - 1], af[0], acc[0][j]);
            acc[1][j] = mma(gf[1][NPL - 1], af[0], acc[1][j]);
            acc[0][j] = mma(gf[0][0], af[NPL - 1], acc[0][j]);
            acc[1][j] = mma(gf[1][0], af[NPL - 1], acc[1][j]);
            acc[0][j] = mma(gf[0][1], af[1], acc[0][j]);
            acc[1][j] = mma(gf[1][1], af[1], acc[1][j]);
            acc[0][j] = mma(gf[0][1], af[0], acc[0][j]);
            acc[1][j] = mma(gf[1][1], af[0], acc[1][j]);
            acc[0][j] = mma(gf[0][0], af[1], acc[0][j]);
            acc[1][j] = mma(gf[1][0], af[1], acc[1][j]);
            acc[0][j] = mma(gf[0][0], af[0], acc[0][j]);
            acc[1][j] = mma(gf[1][0], af[0], acc[1][j]);
          } else if (g_live[0]) {
            acc[0][j] = mma(gf[0][NPL - 1], af[0], acc[0][j]);
            acc[0][j] = mma(gf[0][0], af[NPL - 1], acc[0][j]);
            acc[0][j] = mma(gf[0][1], af[1], acc[0][j]);
            acc[0][j] = mma(gf[0][1], af[0], acc[0][j]);
            acc[0][j] = mma(gf[0][0], af[1], acc[0][j]);
            acc[0][j] = mma(gf[0][0], af[0], acc[0][j]);
          }
        }
      }
      __syncthreads();                              // stage cur^1 is complete; nobody reads stage cur any more
      cur ^= 1;
    }
  }
  // partial tile: part[slab][n][Kt]; D layout: row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (k) = lane & 31
  float* out = p.part + (int64_t)slab * p.n * Kt;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int vkk = k0 + wk * 128 + j * 32 + (lane & 31);                 // virtual column -> column of dW (or none)
      int kk = -1;
      if (vkk < k1p) { if (vkk < p.k1 + ones1) kk = vkk; }
      else if (vkk - k1p < p.k2 + ones2) kk = p.k1 + (vkk - k1p);
      const float cm = (F16 && p.ones && kk == Kt - 1) ? ones_mul : out_mul;   // (exact powers of two; 1 in the bf16x3 form)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int nn = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (nn < p.n && kk >= 0) out[(int64_t)nn * Kt + kk] = F16 ? acc[i][j][r] * cm : acc[i][j][r];
      }
    }
}

// Tiny outputs (the first / last Linear of the embeddings and heads: [32 x 5], [16 x 8], [5 x 16] ... over 192 k node or
// 800 k edge rows): a 128 x 256 MFMA tile would multiply padding, and the launch is bound by streaming the rows once.
// Every thread keeps the whole NN x KK product in registers and walks rows (a lane reads ITS row: n + k contiguous floats,
// the wave a contiguous run of rows), then the wave sums the lanes with DPP/shuffle butterflies and the block its four
// waves through LDS; one partial per block, summed by k_wg_reduce.  Plain fp32 FMAs (exact products).
constexpr int WGN_BLOCKS = 256;

// sum over the 64 lanes of C per-lane values (C a power of two <= 64) in C/2 + C/4 + ... + 1 + (6 - log2 C) shuffles: the
// first log2(C) steps halve the vector while exchanging with the partner lane, after which lane l holds the partial of
// component comp(l); the remaining steps are a plain butterfly.  Lanes 0 .. C-1 hold the C distinct components.
template <int C, int HALF, int BIT>
__device__ __forceinline__ void wgn_reduce_step(float (&v)[C], int lane, int& comp) {
  if constexpr (HALF >= 1) {
    const bool up = (lane >> BIT) & 1;
#pragma unroll
    for (int i = 0; i < HALF; i++) {
      const float send = up ? v[i] : v[i + HALF];
      const float keep = up ? v[i + HALF] : v[i];
      v[i] = keep + __shfl_xor(send, 1 << BIT, 64);
    }
    if (up) comp += HALF;
    wgn_reduce_step<C, HALF / 2, BIT + 1>(v, lane, comp);
  }
}
template <int C>
__device__ __forceinline__ void wgn_reduce_store(const float* vals, int lane, float* dst) {   // dst[c] = sum over lanes of vals[c]
  float v[C];
#pragma unroll
  for (int i = 0; i < C; i++) v[i] = vals[i];
  int comp = 0;
  wgn_reduce_step<C, C / 2, 0>(v, lane, comp);
  float s_ = v[0];
#pragma unroll
  for (int off = C; off < 64; off <<= 1) s_ += __shfl_xor(s_, off, 64);
  if (lane < C) dst[comp] = s_;
}

template <int NN, int KK>
__global__ __launch_bounds__(256) void k_wgrad_narrow(const float* __restrict__ G, int64_t ldg, int n, const float* __restrict__ A,
                                                     int64_t lda, int k, int ones, int64_t m, float* __restrict__ part, int gvec,
                                                     int avec) {
  constexpr int NK = NN * KK;
  __shared__ float red[4][NK];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int kt = k + (ones ? 1 : 0);
  float acc[NK];
#pragma unroll
  for (int i = 0; i < NK; i++) acc[i] = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += (int64_t)gridDim.x * 256) {
    float gv[NN], av[KK];
    // a lane reads ITS row; 16-byte loads where the row geometry allows (the 64 lanes of a load then cover whole cache lines
    // between them over the row's 16-byte pieces), single floats otherwise
    if (gvec) {
#pragma unroll
      for (int i = 0; i < NN; i += 4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) t4 = *(const float4*)(G + r * ldg + i);
        gv[i] = t4.x; if (i + 1 < NN) gv[i + 1] = t4.y; if (i + 2 < NN) gv[i + 2] = t4.z; if (i + 3 < NN) gv[i + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NN; i++) gv[i] = (i < n) ? G[r * ldg + i] : 0.f;
    }
    if (avec) {
#pragma unroll
      for (int j = 0; j < (KK / 4) * 4; j += 4) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < k) t4 = *(const float4*)(A + r * lda + j);
        av[j] = t4.x; av[j + 1] = t4.y; av[j + 2] = t4.z; av[j + 3] = t4.w;
      }
#pragma unroll
      for (int j = (KK / 4) * 4; j < KK; j++) av[j] = 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < KK; j++) av[j] = (j < k) ? A[r * lda + j] : 0.f;
    }
    if (ones) {
#pragma unroll
      for (int j = 0; j < KK; j++) if (j == k) av[j] = 1.f;
    }
#pragma unroll
    for (int i = 0; i < NN; i++)
#pragma unroll
      for (int j = 0; j < KK; j++) acc[i * KK + j] = fmaf(gv[i], av[j], acc[i * KK + j]);
  }
  // wave totals -> LDS: the values go 64 (then 32, 16, 8 ...) at a time through the halving reduction
  constexpr int C64 = NK / 64;
#pragma unroll
  for (int c = 0; c < C64; c++) wgn_reduce_store<64>(acc + 64 * c, lane, red[wib] + 64 * c);
  constexpr int R0 = C64 * 64;
  if constexpr ((NK - R0) >= 32) wgn_reduce_store<32>(acc + R0, lane, red[wib] + R0);
  constexpr int R1 = R0 + (((NK - R0) >= 32) ? 32 : 0);
  if constexpr ((NK - R1) >= 16) wgn_reduce_store<16>(acc + R1, lane, red[wib] + R1);
  constexpr int R2 = R1 + (((NK - R1) >= 16) ? 16 : 0);
  if constexpr ((NK - R2) >= 8) wgn_reduce_store<8>(acc + R2, lane, red[wib] + R2);
  constexpr int R3 = R2 + (((NK - R2) >= 8) ? 8 : 0);
  static_assert(R3 == NK, "NN * KK must be a multiple of 8");
  __syncthreads();
  for (int o = threadIdx.x; o < n * kt; o += 256) {
    const int i = o / kt, j = o % kt;
    part[(int64_t)blockIdx.x * n * kt + o] = red[0][i * KK + j] + red[1][i * KK + j] + red[2][i * KK + j] + red[3][i * KK + j];
  }
}

// out[i] = sum_s part[s, i]: one block = 64 columns x 16 slab groups (coalesced 256-B reads), LDS combine
__global__ __launch_bounds__(1024) void k_wg_reduce(const float* __restrict__ part, int64_t slabs, int64_t width,
                                                   float* __restrict__ out) {
  __shared__ float red[16][64];
  const int lc = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lc;
  float a0 = 0.f, a1 = 0.f;
  if (i < width) {
    int64_t s = g;
    for (; s + 16 < slabs; s += 32) { a0 += part[s * width + i]; a1 += part[(s + 16) * width + i]; }
    if (s < slabs) a0 += part[s * width + i];
  }
  red[g][lc] = a0 + a1;
  __syncthreads();
  if (g == 0 && i < width) {
    float tsum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) tsum += red[k][lc];
    out[i] = tsum;
  }
}

}  // namespace

static int wg_virtual_k(int k1, int k2, int ones) {   // the kernel's virtual k axis: [A1 (| 1) padded to 64 | A2 | 1 padded to 64]
  const int o1 = (ones && k2 == 0) ? 1 : 0, o2 = (ones && k2 > 0) ? 1 : 0;
  return ((k1 + o1 + 63) & ~63) + ((k2 + o2 + 63) & ~63);
}

static int wg_narrow_class(int n, int k1, int k2, int ones) {   // 0: not a tiny output; else the register-tile instance
  if (k2 != 0 || n <= 0) return 0;
  const int kt = k1 + (ones ? 1 : 0);
  if (n <= 16 && kt <= 9) return 1;
  if (n <= 8 && kt <= 17) return 2;
  if (n <= 32 && kt <= 6) return 3;
  return 0;
}

extern "C" int32_t rgnn_wgrad_slabs(int64_t m, int32_t n, int32_t k1, int32_t k2, int32_t with_ones) {
  if (wg_narrow_class(n, k1, k2, with_ones)) return 512;       // (covers k_wgrad_narrow's 256 blocks and the MFMA kernel's slabs)
  const int64_t tiles = (int64_t)((n + WG_BN - 1) / WG_BN) * ((wg_virtual_k(k1, k2, with_ones ? 1 : 0) + WG_BK - 1) / WG_BK);
  if (tiles == 0) return 8;
  int64_t slabs = 512 / tiles / 8 * 8;                         // two work-groups per CU, all resident at once (one round)
  const int64_t max_slabs = ((m + 255) / 256 + 7) / 8 * 8;     // about 256 rows per slab at least
  if (slabs > max_slabs) slabs = max_slabs;
  return (int32_t)(slabs < 8 ? 8 : slabs);
}

extern "C" int rgnn_wgrad(const float* G, int64_t ldg, int32_t n, const float* A1, int64_t lda1, int32_t k1, const float* A2,
                          int64_t lda2, int32_t k2, int32_t with_ones, int64_t m, const int32_t* row_index, const int64_t* m_dev,
                          float* partial, float* dW, rgnn_stream_t stream) {
  return rgnn_wgrad_bounds(G, ldg, n, A1, lda1, k1, A2, lda2, k2, with_ones, m, row_index, m_dev, nullptr, nullptr, nullptr, partial, dW,
                           stream);
}

extern "C" int rgnn_wgrad_bounds(const float* G, int64_t ldg, int32_t n, const float* A1, int64_t lda1, int32_t k1, const float* A2,
                                 int64_t lda2, int32_t k2, int32_t with_ones, int64_t m, const int32_t* row_index, const int64_t* m_dev,
                                 const float* g_bound, const float* a1_bound, const float* a2_bound, float* partial, float* dW,
                                 rgnn_stream_t stream) {
  RGNN_CHECK_ARG(m >= 0 && n >= 0 && k1 >= 0 && k2 >= 0, "negative sizes");
  const int Kt = k1 + k2 + (with_ones ? 1 : 0);
  if (n == 0 || Kt == 0) return RGNN_OK;
  if (m == 0) {                                         // no rows: the sum over nothing (the edge MLPs of a graph without edges)
    RGNN_CHECK_ARG(dW != nullptr, "null pointers");
    hipMemsetAsync(dW, 0, (size_t)n * Kt * sizeof(float), (hipStream_t)stream);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(G && dW && partial && (k1 == 0 || A1) && (k2 == 0 || A2), "null pointers");
  RGNN_CHECK_ARG(row_index != nullptr || m_dev == nullptr, "m_dev needs row_index");
  RGNN_CHECK_ARG(ldg * 4 < ((int64_t)1 << 31) && lda1 * 4 < ((int64_t)1 << 31) && lda2 * 4 < ((int64_t)1 << 31), "row stride too large");
  hipStream_t s = (hipStream_t)stream;
  const int narrow = (row_index == nullptr && RGNN_ENV("RGNN_WGRAD_NO_NARROW") == nullptr) ? wg_narrow_class(n, k1, k2, with_ones) : 0;
  if (narrow) {
    int64_t blocks = (m + 1023) / 1024;
    if (blocks > WGN_BLOCKS) blocks = WGN_BLOCKS;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(256);
    const int gvec = (n % 4 == 0 && ldg % 4 == 0 && ((uintptr_t)G & 15) == 0) ? 1 : 0;
    const int avec = (k1 % 4 == 0 && k1 > 0 && lda1 % 4 == 0 && ((uintptr_t)A1 & 15) == 0) ? 1 : 0;
    const int on = with_ones ? 1 : 0;
    if (narrow == 1) hipLaunchKernelGGL((k_wgrad_narrow<16, 9>), g, b, 0, s, G, ldg, n, A1, lda1, k1, on, m, partial, gvec, avec);
    else if (narrow == 2) hipLaunchKernelGGL((k_wgrad_narrow<8, 17>), g, b, 0, s, G, ldg, n, A1, lda1, k1, on, m, partial, gvec, avec);
    else hipLaunchKernelGGL((k_wgrad_narrow<32, 6>), g, b, 0, s, G, ldg, n, A1, lda1, k1, on, m, partial, gvec, avec);
    hipLaunchKernelGGL(k_wg_reduce, dim3(rgnn_blocks((int64_t)n * Kt, 64)), dim3(1024), 0, s, partial, blocks, (int64_t)n * Kt, dW);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  Wg3Params p;
  p.G = G; p.ldg = ldg; p.n = n; p.A1 = k1 ? A1 : nullptr; p.lda1 = lda1; p.k1 = k1; p.A2 = k2 ? A2 : nullptr; p.lda2 = lda2; p.k2 = k2;
  p.ones = with_ones ? 1 : 0; p.m = m; p.row_index = row_index; p.m_dev = m_dev; p.part = partial;
  p.slabs = rgnn_wgrad_slabs(m, n, k1, k2, with_ones);
  const int kv = wg_virtual_k(k1, k2, p.ones);
  p.nt = (n + WG_BN - 1) / WG_BN; p.kt = (kv + WG_BK - 1) / WG_BK;
  p.g_bound = g_bound; p.a1_bound = a1_bound; p.a2_bound = a2_bound;
  // f16x2 form: every operand block carries a bound (three products instead of six; the caller's switch: ops.TRAIN_F16X2)
  const bool f16 = g_bound != nullptr && (k1 == 0 || a1_bound != nullptr) && (k2 == 0 || a2_bound != nullptr) &&
                   RGNN_ENV("RGNN_WGRAD_NO_F16X2") == nullptr;
  static RgnnOncePerDevice attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_wgrad_x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_STAGE);
    hipFuncSetAttribute((const void*)k_wgrad_x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * WG_PLANE);
  }
  const int64_t blocks = (int64_t)p.slabs * p.nt * p.kt;       // slabs is a multiple of 8: blockIdx -> (xcd, slab / 8, tile)
  if (f16) hipLaunchKernelGGL(k_wgrad_x3<true>, dim3((unsigned)blocks), dim3(WG_THREADS), 2 * 2 * WG_PLANE, s, p);
  else hipLaunchKernelGGL(k_wgrad_x3<false>, dim3((unsigned)blocks), dim3(WG_THREADS), 2 * WG_STAGE, s, p);
  hipLaunchKernelGGL(k_wg_reduce, dim3(rgnn_blocks((int64_t)n * Kt, 64)), dim3(1024), 0, s, partial, (int64_t)p.slabs,
                     (int64_t)n * Kt, dW);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
