// Max aggregation on the matrix pipe, window form:   M[t] = p_bias + max_{e : target(e) = t} ( Q[source(e)] + W_e z_e )
//
// Same contract as k_mpnn_max (mpnn.hip; replaces MessagePassing.propagate + torch-scatter max of gnn/mpnn_layers.py:88,94-101
// once the node terms are hoisted), different decomposition.  k_mpnn_max walks a target's edges one at a time with the lanes
// across channels; its 32 v_pk_fma_f32 per edge for W_e z_e keep the vector ALU 77 % busy at k = 20
// (profiles/r04base_c4_pmc_sq_k_mpnn_max.txt) and every edge fetches its own row of Q from L2.  Here
//   * the 32 rows of one v_mfma_f32_32x32x16_bf16 output tile are 32 CSR slots (edges in target order), its 32 columns are
//     32 channels: the gathered Q values are the accumulator's INITIAL value (C operand: lane l holds column l & 31 of rows
//     8 (i >> 2) + 4 (l >> 5) + (i & 3), i = 0..15), and W_e z_e arrives from the matrix pipe: z_e and W_e are each split
//     EXACTLY into three bf16 terms (top 8 significand bits, next 8, last 8), and six of the nine products -- everything down
//     to 2^-23 of |z||w| -- are accumulated in fp32 by three MFMAs whose K = 16 is two 8-wide products each:
//     [z1|z1]x[w1|w2],  [z2|z2]x[w1|w2],  [z3|z1]x[w1|w3];
//   * the accumulator layout gives each HALF of the wave sixteen rows: a half walks its OWN stream of slots, sixteen per tile,
//     serially through its sixteen registers with the running maximum in one register -- the segmented maximum needs no
//     cross-lane step, and a group of four slots without a segment end is two v_max3_f32;
//   * the distinct source rows of a WINDOW of targets are staged in LDS once per channel tile and shared by the window's edges
//     (see below).
// Per edge and channel the arithmetic differs from k_mpnn_max's fp32 FMA chain by ~2^-22 |z||w| (tests compare both with the
// float64 oracle); which kernel runs is decided per launch (rgnn_mpnn_aggregate_win refuses what it does not cover).
// (r04's tile-stream form of the same arithmetic -- one row piece per edge gathered straight into the accumulator layout -- and
//  its gather probes were measured out and are archived in tools/attic/mpnn_tiles_stream.hip.txt.)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int MT_THREADS = 256;
constexpr int MT_WAVES = MT_THREADS / 64;
constexpr int MT_QUEUE_INTS = 8 * 16;          // one ticket counter (+ exit counter) per XCD, 64 B apart

typedef float mt_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mt_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int mt_u32x4 __attribute__((ext_vector_type(4)));

// rows of the targets without incoming edges: exactly 0 (torch-scatter), for callers that read them
__global__ __launch_bounds__(256) void k_tiles_zero_empty(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ order,
                                                         int64_t n, int d, float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * MT_WAVES + (threadIdx.x >> 6);
  if (p >= n || rowptr[p + 1] != rowptr[p]) return;
  const int64_t node = order ? order[p] : p;
  for (int c = lane; c < d; c += 64) out[node * ldo + c] = 0.f;
}

// x = t1 + t2 + t3 exactly, each term the top 16 bits of an fp32 word (a bf16 value)
__device__ __forceinline__ void mt_split3(float x, unsigned& t1, unsigned& t2, unsigned& t3) {
  t1 = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(t1);
  t2 = __float_as_uint(r) & 0xffff0000u;
  const float r2 = r - __uint_as_float(t2);
  t3 = __float_as_uint(r2) & 0xffff0000u;
}
__device__ __forceinline__ unsigned mt_pack(unsigned lo, unsigned hi) { return (lo >> 16) | hi; }   // (both already masked to their top halves)
__device__ __forceinline__ void mt_split_row(const float (&x)[8], mt_u32x4& p1, mt_u32x4& p2, mt_u32x4& p3) {
  unsigned t1[8], t2[8], t3[8];
#pragma unroll
  for (int k = 0; k < 8; k++) mt_split3(x[k], t1[k], t2[k], t3[k]);
  p1 = mt_u32x4{mt_pack(t1[0], t1[1]), mt_pack(t1[2], t1[3]), mt_pack(t1[4], t1[5]), mt_pack(t1[6], t1[7])};
  p2 = mt_u32x4{mt_pack(t2[0], t2[1]), mt_pack(t2[2], t2[3]), mt_pack(t2[4], t2[5]), mt_pack(t2[6], t2[7])};
  p3 = mt_u32x4{mt_pack(t3[0], t3[1]), mt_pack(t3[2], t3[3]), mt_pack(t3[4], t3[5]), mt_pack(t3[6], t3[7])};
}

}  // namespace

// =====================================================================================================================
// Window form (r04; pipelined across windows in r05): the distinct source rows of a WINDOW of targets are staged in LDS once per
// channel tile and shared by the window's edges.
//
// A kernel that fetches one 128-byte piece of a Q row per edge and channel tile from L2 is bound by that path: the probes
// (profiles/r04_mpnn_tiles_probes.txt) put it at 27 TB/s when L2-resident and 11-13 TB/s at the 87-89 % hit rate of a real
// batch -- the per-edge kernel already sits there.  In grid-cell order ~20 consecutive targets of a k = 20
// graph name each source 2.5-3 times.  A window = up to 8 streams (one per half-wave of a 4-wave work-group) of <= 64
// slots, every target padded to a multiple of 4 slots (a repeated edge does not change a maximum) and packed whole into a
// stream, so that a segment can only end at the end of a group of four accumulator registers: the segmented maximum is two
// v_max3_f32 per group and one test per group.  Per window (plan, once per graph): the list of its distinct sources
// (<= 512), the LDS row of every slot, the edge of every slot (for z), end flags per group, the target of every ending group.
// Per channel tile the work-group DMAs the window's rows (128 B each, `buffer_load_dwordx4 ... lds`: eight rows per
// instruction) into LDS, and every wave initialises its accumulators -- register = slot, lane = channel -- with sixteen
// ds_read_b32: the layout change (a transposition of ~44 vector instructions per tile when done in registers) is done by the
// LDS addressing.  z is split once per window (its three bf16 terms stay in registers across the channel tiles).
namespace {

#ifndef RGNN_WIN_ONE_STORE
#define RGNN_WIN_ONE_STORE 1
#endif
constexpr int WN_SLOTS = 512;         // slots per window: 8 streams x 64
constexpr int WN_THREADS = 256;
typedef int wn_i32x4 __attribute__((ext_vector_type(4)));

struct WinParams {
  const float* p_bias;
  const float* Q; int ldq4; int q_bytes;
  const mt_u32x4* wplanes;             // [n_ct][4][32] bf16 terms of W_e (3) and the bias, per channel tile (k_win_wplanes)
  const float* ea; int de; int ea_vec;
  int n_win;                           // windows allocated; n_win_dev[0] = windows the plan really made
  const int32_t* n_win_dev;
  // the plan's per-window descriptors, addressed as byte offsets from `plan` (the kernel fetches them by LDS-DMA one window ahead)
  const int32_t* plan; int plan_bytes;
  int off_misc;                        // [n_win][16] ints: distinct sources nU (0: empty window), tiles per stream (8 bytes), end flags (32 bytes:
                                       //   per stream and tile, bit g = group g ends a segment)
  int off_eid;                         // [n_win][512] edge (row of ea) of every slot
  int off_lrow;                        // [n_win][512] bytes: local row (index into the window's distinct sources) of every slot
  int off_urow;                        // [n_win][WN_UMAX] source node id of local row u (entries up to the next multiple of 8 are valid ids)
  int off_tgt;                         // [n_win][8][4][4] target node of an ending group
  int32_t* queue;
  int d; int n_ct;
  float* out; int ldo4; int o_bytes;
  float* out_absmax;
  int abl;                             // experiments only (wrong results): 1 no row requests after the first tile, 2 no arithmetic, 4 no stores
};

__device__ __forceinline__ void wn_dma16(wn_i32x4 rsrc, int voff, int soff, unsigned lds_base) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %2 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
      : "memory");
}

__global__ __launch_bounds__(256) void k_win_wplanes(const float* __restrict__ We, int ldwe, int de, int d, int n_ct, const float* __restrict__ p_bias,
                                                     mt_u32x4* __restrict__ planes) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_ct * 32) return;
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = (c < d && k < de) ? We[(int64_t)c * ldwe + k] : 0.f;
  mt_u32x4 p1, p2, p3;
  mt_split_row(w, p1, p2, p3);
  // per channel tile: [term][32 channels] -- a lane reads ITS channel's 16 bytes next to its neighbours' (a [channel][term] image
  // put the lanes 64 bytes apart: four-way bank conflicts on every operand read, 12 % of the kernel's LDS cycles)
  mt_u32x4* t = planes + (c >> 5) * 128 + (c & 31);
  t[0] = p1; t[32] = p2; t[64] = p3;
  t[96] = mt_u32x4{__float_as_uint((p_bias != nullptr && c < d) ? p_bias[c] : 0.f), 0u, 0u, 0u};
}

// Segments padded to 2 slots instead of 4 (-DRGNN_WIN_PAD=2; r05, tests green): 13 % fewer slots on the r = 1 m graphs, and SLOWER
// everywhere -- C2 152 -> 202 us per launch, k = 20: 265 -> 366 -- eight possible segment ends per tile are eight copies of the
// straight-line end code per tile (the <true> instance spills), whatever the number of ends that occur.  4 it stays.
#ifndef RGNN_WIN_PAD
#define RGNN_WIN_PAD 4
#endif
constexpr int WN_PAD = RGNN_WIN_PAD;   // a target's slots are padded to a multiple of this (2 or 4): a GROUP of WN_PAD accumulator rows
constexpr int WN_GPT = 16 / WN_PAD;    // groups per 16-slot tile of a stream (a segment can end at every group)
constexpr int WN_GPS = 64 / WN_PAD;    // groups per stream
constexpr int WN_TMAX = 512 / WN_PAD;  // targets a window can hold
static_assert(WN_PAD == 2 || WN_PAD == 4, "segments are padded to 2 or 4 slots");
constexpr int WN_UMAX = 176;           // distinct source rows of a window (plan guarantee): 22 KB per staged channel tile
constexpr int WN_ROWBUF = WN_UMAX * 128;

__device__ __forceinline__ void wn_wait_leaving(int n) {        // s_waitcnt vmcnt(n) for a wave-uniform n in 0 .. 32
  switch (n) {
#define WN_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    WN_W(1) WN_W(2) WN_W(3) WN_W(4) WN_W(5) WN_W(6) WN_W(7) WN_W(8) WN_W(9) WN_W(10) WN_W(11) WN_W(12) WN_W(13) WN_W(14) WN_W(15) WN_W(16)
    WN_W(17) WN_W(18) WN_W(19) WN_W(20) WN_W(21) WN_W(22) WN_W(23) WN_W(24) WN_W(25) WN_W(26) WN_W(27) WN_W(28) WN_W(29) WN_W(30) WN_W(31) WN_W(32)
#undef WN_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

constexpr bool WN_WARM = false;        // warming requests one tile ahead: measured 521 -> 554 us at k = 20, D = 464: not the limiter
constexpr int WN_BBUF = 32 * 64;       // B operand + bias of one channel tile

constexpr int WN_DESC = 2 * WN_UMAX * 4 + 2048 + 512 + 8 * WN_GPS * 4 + 64;   // LDS bytes of the window descriptors: urow x 2, eid, lrow, tgt, misc
// (three work-groups per CU: 3 x WN_LDS rounded up to the 512-byte allocation granule must stay within 160 KiB -- 54 272 each)
static_assert((2 * WN_UMAX * 128 + 2 * 32 * 64 + WN_DESC + 16 + 511) / 512 * 512 * 3 <= 160 * 1024, "k_mpnn_win: LDS of three work-groups per CU");
constexpr int WN_LDS = 2 * WN_ROWBUF + 2 * WN_BBUF + WN_DESC + 16;

template <bool AMAX>
__global__ __launch_bounds__(WN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_mpnn_win(const WinParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Two row stages and two operand stages (a channel tile's bytes arrive while the one before it is multiplied), then the window
  // descriptors.  Everything a wave loads from the tile loop on comes through LDS-DMA requests of its own (invisible to hipcc's
  // wait bookkeeping) and is awaited with a COUNTED wait that leaves the streaming stores of the tile in between in flight: one
  // in-order counter per wave, and `vmcnt(0)` would make every channel tile wait for the HBM acknowledgement of the previous
  // one's stores.
  // r05: windows are pipelined ACROSS each other.  The descriptors of the NEXT window (its distinct sources, the edge / local row
  // of every slot, end flags, targets: 3.9 KB of the plan) are fetched by LDS-DMA during this window's first channel tile, and the
  // next window's first row stage is requested during this window's LAST channel tile -- r04 opened every window with a chain of
  // dependent global loads (ticket -> descriptors -> edge attributes; then the first rows), 21 % of the launch at k = 20
  // (gpurun_out/r05_win_abl.txt: 121 of 564 us with row requests and arithmetic switched off).
  char* bstage = smem + 2 * WN_ROWBUF;
  int* urowtab = (int*)(bstage + 2 * WN_BBUF);                  // [2][WN_UMAX] source node ids of the distinct rows (this window | next)
  int* eidtab = urowtab + 2 * WN_UMAX;                          // [512]
  const uint8_t* lrowtab = (const uint8_t*)(eidtab + 512);     // [512]
  int* tgttab = eidtab + 512 + 128;                             // [8 streams][WN_GPS]
  int* misctab = tgttab + 8 * WN_GPS;                           // [16]
  int* bcast = misctab + 16;                                    // [4]
  const unsigned rows_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned b_lds = rows_lds + 2 * WN_ROWBUF;
  const unsigned desc_lds = b_lds + 2 * WN_BBUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, col = lane & 31, col4 = col * 4;
  const int xcd = blockIdx.x & 7;
  const int n_win = min(p.n_win, __builtin_amdgcn_readfirstlane(p.n_win_dev[0]));
  const int i_lo = (int)((int64_t)n_win * xcd / 8), i_hi = (int)((int64_t)n_win * (xcd + 1) / 8);
  int32_t* ticket = p.queue + xcd * 16;
  wn_i32x4 rq, rw, rp;
  {
    const uint64_t a = (uint64_t)p.Q;
    rq.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    rq.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
    rq.z = __builtin_amdgcn_readfirstlane(p.q_bytes);
    rq.w = 0x00020000;
    const uint64_t b = (uint64_t)p.wplanes;
    rw.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    rw.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xffff));
    rw.z = __builtin_amdgcn_readfirstlane(p.n_ct * WN_BBUF);
    rw.w = 0x00020000;
    const uint64_t c = (uint64_t)p.plan;
    rp.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)c);
    rp.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((c >> 32) & 0xffff));
    rp.z = __builtin_amdgcn_readfirstlane(p.plan_bytes);
    rp.w = 0x00020000;
  }
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, (short)0, p.o_bytes, 0x00020000);
  const int rA = lane & 31, hA = (rA >> 2) & 1, iA = ((rA >> 3) << 2) | (rA & 3);
  const int my_stream = 2 * wave + half;
  float amax = 0.f;
  float ninf = -INFINITY;
  asm volatile("" : "+v"(ninf));                             // (kept in a register: v_cndmask takes no literal)

  // descriptors of window w -> LDS (urow into half `up` of its table): one or two requests per wave, lanes beyond a table masked
  auto desc_dma = [&](const int w, const int up) {
    if (wave == 0) {
      if (lane < WN_UMAX / 4) wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_urow + w * (WN_UMAX * 4)),
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(desc_lds + up * (WN_UMAX * 4))));
    } else if (wave == 1) {
      wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_eid + w * 2048), desc_lds + 8 * WN_UMAX);
      wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_eid + w * 2048 + 1024), desc_lds + 8 * WN_UMAX + 1024);
    } else if (wave == 2) {
      if (lane < 32) wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_lrow + w * 512), desc_lds + 8 * WN_UMAX + 2048);
    } else {
      if (lane < 2 * WN_GPS) wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_tgt + w * (32 * WN_GPS)), desc_lds + 8 * WN_UMAX + 2560);
      if (lane < 4) wn_dma16(rp, lane * 16, __builtin_amdgcn_readfirstlane(p.off_misc + w * 64), desc_lds + 8 * WN_UMAX + 2560 + 32 * WN_GPS);
    }
  };
  // this wave's share of the rows (wave 1: also the operands) of channel tile ct of a window whose distinct rows are urowtab[up]
  auto stage = [&](const int ct, const int st, const int up, const int nU8) {
    const unsigned base = rows_lds + st * WN_ROWBUF;
    const int* ut = urowtab + up * WN_UMAX;
    for (int k = wave; k < nU8; k += 4)
      wn_dma16(rq, (int)__umul24((unsigned)ut[8 * k + (lane >> 3)], (unsigned)p.ldq4) + (lane & 7) * 16, __builtin_amdgcn_readfirstlane(ct * 128),
               (unsigned)__builtin_amdgcn_readfirstlane((int)(base + k * 1024)));
    if (wave == 1) {
      const unsigned bb = b_lds + st * WN_BBUF;
      wn_dma16(rw, lane * 16, __builtin_amdgcn_readfirstlane(ct * WN_BBUF), (unsigned)__builtin_amdgcn_readfirstlane((int)bb));
      wn_dma16(rw, lane * 16 + 1024, __builtin_amdgcn_readfirstlane(ct * WN_BBUF), (unsigned)__builtin_amdgcn_readfirstlane((int)(bb + 1024)));
    }
  };

  // two tickets: this window and the next one (every later ticket is fetched a whole window ahead)
  if (tid == 0) { bcast[0] = i_lo + atomicAdd(ticket, 1); bcast[1] = i_lo + atomicAdd(ticket, 1); }
  __syncthreads();
  int win = bcast[0];
  int up = 0;                  // half of urowtab that holds this window's rows
  int gp = 0;                  // row / operand stage of this window's first channel tile
  bool staged = false;         // ... already requested (during the previous window's last tile)
  if (win < i_hi) desc_dma(win, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (;;) {
    if (win >= i_hi) {
      if (tid == 0) {
        const int wgs = (int)(gridDim.x >> 3);
        if (atomicAdd(ticket + 1, 1) == wgs - 1) { ticket[0] = 0; ticket[1] = 0; }
      }
      break;
    }
    // ---- this window's descriptors are in LDS; bcast[1] is the next window
    const int nxt = bcast[1];
    const int nU = misctab[0];
    const int nU8 = (nU + 7) >> 3;
    int nt = 0;
    unsigned endbits = 0, anyend = 0, endA = 0, endB = 0;
    mt_u32x4 a1[4], a2[4], a3[4];
    int aoff[4][16];
    int tv = 0;
    if (nU != 0) {
      // byte offsets (in `out`) of the target rows of this half's sixteen groups: lane j (and 32 + j) holds group j's -- where a
      // segment ends the offset comes out of the register with v_readlane, not out of LDS
      tv = tgttab[my_stream * WN_GPS + (lane & (WN_GPS - 1))] * p.ldo4;
      const uint8_t* m8 = (const uint8_t*)(misctab + 1);
      nt = max((int)m8[2 * wave], (int)m8[2 * wave + 1]);
      const uint8_t* e4 = m8 + 8 + my_stream * 4;
      endbits = (unsigned)e4[0] | ((unsigned)e4[1] << WN_GPT) | ((unsigned)e4[2] << (2 * WN_GPT)) | ((unsigned)e4[3] << (3 * WN_GPT));
      endA = (unsigned)__builtin_amdgcn_readlane((int)endbits, 0); endB = (unsigned)__builtin_amdgcn_readlane((int)endbits, 32);
      anyend = endA | endB;
      // z of this wave's (up to) four tiles, split into its bf16 terms once for all channel tiles
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int e = eidtab[(2 * wave + hA) * 64 + t * 16 + iA];
        float z[8];
        if (p.ea_vec) {
          const float4 mine = *(const float4*)(p.ea + (int64_t)e * 8 + half * 4);
          const float4 other = make_float4(__shfl_xor(mine.x, 32, 64), __shfl_xor(mine.y, 32, 64), __shfl_xor(mine.z, 32, 64), __shfl_xor(mine.w, 32, 64));
          const float4 lo = half ? other : mine, hi = half ? mine : other;
          z[0] = lo.x; z[1] = lo.y; z[2] = lo.z; z[3] = lo.w; z[4] = hi.x; z[5] = hi.y; z[6] = hi.z; z[7] = hi.w;
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) z[k] = (k < p.de) ? p.ea[(int64_t)e * p.de + k] : 0.f;
        }
        mt_u32x4 t3;
        mt_split_row(z, a1[t], a2[t], t3);
        a3[t] = half ? a1[t] : t3;
      }
      // LDS address (stage 0) of this lane's word of every one of this half's 64 slots -- local row * 128 + 4 * channel -- in 64
      // registers for the whole window: the tile loop spends NO vector instruction on addresses (r04 unpacked two 16-bit offsets
      // per register with one v_add per slot and tile: 40 % of the tile loop's vector instructions); the stage is an immediate.
      const mt_u32x4* lt = (const mt_u32x4*)(lrowtab + my_stream * 64);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const mt_u32x4 w4 = lt[t];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const unsigned w = w4[i >> 2];
          const unsigned r = (i & 3) == 3 ? (w >> 24) : ((w >> (8 * (i & 3))) & 0xffu);
          aoff[t][i] = (int)(rows_lds + (r << 7)) + col4;
          asm volatile("" : "+v"(aoff[t][i]));          // (opaque: hipcc otherwise keeps r << 7 and re-adds the lane's part at every use)
        }
      }
    }
    __syncthreads();                                            // everybody has read the descriptor tables and this pair of tickets
    if (tid == 0) bcast[1] = i_lo + atomicAdd(ticket, 1);       // the window after the next; read a whole window from now
    if (nU == 0) {                                              // (a window that went per target: only its successor's descriptors)
      if (nxt < i_hi) desc_dma(nxt, up ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      win = nxt; up ^= 1; staged = false;
      continue;
    }
    // stores this wave issues per channel tile: two (one per half, the exec mask of a half that does not end is empty) per group in
    // which either of its streams ends a segment
    const int nst = (RGNN_WIN_ONE_STORE ? 1 : 2) * __builtin_popcount(anyend & ((nt >= 4) ? (WN_GPT == 8 ? 0xffffffffu : 0xffffu) : ((1u << (WN_GPT * nt)) - 1u)));
    if (!staged) stage(0, gp, up, nU8);
    bool staged_next = false;

    auto tile_pass = [&](const int ct, auto stage_c) {
      constexpr int ST = decltype(stage_c)::value;              // row / operand stage of this channel tile: an instruction immediate
      // this tile's bytes (requested a whole tile ago) have landed; the stores issued since may still be under way
      if (ct == 0 || (p.abl & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else wn_wait_leaving(nst);
      __syncthreads();                                          // ... for everybody; and nobody reads the other stage any more
      if (ct == 0 && nxt < i_hi) desc_dma(nxt, up ^ 1);         // (complete behind the wait + barrier of the second tile)
      if (ct + 1 < p.n_ct) {
        if (!(p.abl & 1)) stage(ct + 1, ST ^ 1, up, nU8);
      } else if (ct > 0 && nxt < i_hi && !(p.abl & 1)) {
        // last tile: the first row stage of the NEXT window (its descriptors arrived during this window's first tile)
        const int nUn = misctab[0];
        if (nUn != 0) { stage(0, ST ^ 1, up ^ 1, (nUn + 7) >> 3); staged_next = true; }
      }
      const mt_u32x4* bl = (const mt_u32x4*)(bstage + ST * WN_BBUF);
      const mt_u32x4 bxc = bl[(half ? 32 : 0) + col], byc = bl[(half ? 64 : 0) + col];
      const float biasc = __uint_as_float(bl[96 + col].x);
      const int voffc = (ct * 32 + col < p.d) ? col4 : 0x7ffffff0;       // (out of range: the store is dropped)
      const float ninf_l = ninf;
      const __amdgpu_buffer_rsrc_t ro_l = ro;
      float rn = -INFINITY;
      // (what the segment ends derive from the end masks and the offset register is loop-invariant: without this hipcc hoists 64
      //  scalars out of the channel-tile loop and spills them)
      asm volatile("" : "+v"(tv), "+s"(endA), "+s"(endB), "+s"(anyend));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (t < nt && !(p.abl & 2)) {
          mt_f32x16 c;
#pragma unroll
          for (int i = 0; i < 16; i++) c[i] = *(const __attribute__((address_space(3))) float*)(uintptr_t)(unsigned)(aoff[t][i] + ST * WN_ROWBUF);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mt_bf16x8, a3[t]), __builtin_bit_cast(mt_bf16x8, byc), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mt_bf16x8, a2[t]), __builtin_bit_cast(mt_bf16x8, bxc), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mt_bf16x8, a1[t]), __builtin_bit_cast(mt_bf16x8, bxc), c, 0, 0, 0);
#pragma unroll
          for (int g = 0; g < WN_GPT; g++) {
            rn = __builtin_fmaxf(__builtin_fmaxf(rn, c[WN_PAD * g]), c[WN_PAD * g + 1]);
            if (WN_PAD == 4) rn = __builtin_fmaxf(__builtin_fmaxf(rn, c[WN_PAD * g + 2]), c[WN_PAD * g + 3]);
            if ((anyend >> (WN_GPT * t + g)) & 1u) {
              // A segment ends here in one half or in both.  Three vector instructions per end (bias, reset, maximum of |v|): the
              // store offset is a SCALAR per half (v_readlane out of `tv` + the channel tile's offset) handed to the store as its
              // soffset, one half-wave store per ending half under an exec mask set by hand; the reset is one v_cndmask under a
              // lane mask assembled from the two end bits.  (r04: table read + lgkmcnt(0) per end and sixteen lane masks in SGPR
              // pairs; a first r05 form with per-lane selects cost 14 vector instructions per end and lost 2.5 % on the r = 1 m
              // batches, where three of four groups end a segment.)
              const int k = WN_GPT * t + g;
              const float v = rn + biasc;
              // (straight-line on purpose: every taken branch refills the instruction buffer, and on the r = 1 m batches three of
              //  four groups come through here.  A half that does not end stores under an empty exec mask.)
              const unsigned long long mA = ((endA >> k) & 1u) ? 0xffffffffull : 0ull, mB = ((endB >> k) & 1u) ? 0xffffffff00000000ull : 0ull;
              const int soA = __builtin_amdgcn_readlane(tv, k) + ct * 128, soB = __builtin_amdgcn_readlane(tv, 32 + k) + ct * 128;
              if (!(p.abl & 4)) {
#if RGNN_WIN_ONE_STORE
                // ONE store per ending group: the two halves' scalar offsets selected per lane, lanes of a half that does not end masked
                const int off = (half ? soB : soA) + voffc;
                asm volatile("s_mov_b64 exec, %3\n\tbuffer_store_dword %0, %1, %2, 0 offen nt\n\ts_mov_b64 exec, -1"
                             : : "v"(v), "v"(off), "s"(ro_l), "s"(mA | mB) : "memory");
#else
                asm volatile("s_mov_b64 exec, %4\n\tbuffer_store_dword %0, %1, %2, %3 offen nt\n\t"
                             "s_mov_b64 exec, %6\n\tbuffer_store_dword %0, %1, %2, %5 offen nt\n\ts_mov_b64 exec, -1"
                             : : "v"(v), "v"(voffc), "s"(ro_l), "s"(soA), "s"(mA), "s"(soB), "s"(mB) : "memory");
#endif
              }
              if (AMAX) amax = fmaxf(amax, fabsf(v));           // (a half that does not end contributes a partial maximum: a bound all the same)
              const unsigned long long m64 = mA | mB;
              asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(rn) : "v"(ninf_l), "s"(m64));
            }
          }
        }
      }
    };
    // (lanes beyond d in the last channel tile hold whatever the padding columns of Q held: their stores are dropped by an out-of-range
    //  offset, and their part of max |v| is discarded below)
    float amax_full = 0.f;
    for (int ct = 0; ct < p.n_ct; ct++) {
      const bool partial = AMAX && (ct + 1) * 32 > p.d;
      if (partial) { amax_full = amax; amax = 0.f; }
      if ((gp + ct) & 1) tile_pass(ct, std::integral_constant<int, 1>());
      else tile_pass(ct, std::integral_constant<int, 0>());
      if (partial) amax = fmaxf(amax_full, (ct * 32 + col < p.d) ? amax : 0.f);
    }
    if (p.n_ct < 2) {                                           // (one channel tile: the next window's descriptors were requested in it)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    gp = (gp + p.n_ct) & 1;
    win = nxt; up ^= 1; staged = staged_next;
  }
  if (AMAX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0)
      atomicMax((unsigned int*)p.out_absmax + ((blockIdx.x * 4 + wave) & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(amax));
  }
}

// ---- the window plan, built on the device once per graph ---------------------------------------------------------------
constexpr int WN_BIG = 64;            // a target with more padded slots than a stream holds goes to the per-target kernel

constexpr int WN_SEG = 512;          // positions per greedy segment (one lane packs a segment's targets in order)
struct WinPlanLayout {
  int n_win, n_seg;
  int64_t off_assign, off_segcnt, off_segbase, off_wend, off_pdeg, off_wstart, off_left, off_leftcnt, off_queue, off_misc, off_tgt, off_eid, off_lrow, off_urow,
      off_wplanes, total_ints;
};
WinPlanLayout win_layout(int64_t n, int64_t n_edges) {
  WinPlanLayout L;
  L.n_seg = (int)((n + WN_SEG - 1) / WN_SEG);
  // a greedy window closes only when a target fits none of its 8 streams, i.e. every stream holds more than 64 - 64 slots ... at
  // least 8 x 33 = 264 slots unless it is the last of its segment
  L.n_win = (int)((n_edges + 3 * n) / 256 + L.n_seg + 2);
  int64_t o = 16;
  auto take = [&](int64_t ints) { const int64_t at = o; o = (o + ints + 15) / 16 * 16; return at; };
  L.off_assign = take(n + 1); L.off_segcnt = take(L.n_seg + 1); L.off_segbase = take(L.n_seg + 2);
  L.off_wend = take(L.n_win + 1);
  L.off_pdeg = take(n + 1); L.off_wstart = take(L.n_win + 1); L.off_left = take(n + 1); L.off_leftcnt = take(4);
  L.off_queue = take(MT_QUEUE_INTS); L.off_misc = take(16 * (int64_t)L.n_win);
  L.off_tgt = take(8 * WN_GPS * (int64_t)L.n_win); L.off_eid = take(512 * (int64_t)L.n_win); L.off_lrow = take(128 * (int64_t)L.n_win);
  L.off_urow = take(WN_UMAX * (int64_t)L.n_win); L.off_wplanes = take(16 * 64 * 32);       // up to 64 channel tiles (d <= 2048)
  L.total_ints = o;
  return L;
}

// padded slots of every target position (0: no edges, or too many for a stream -> the per-target list)
__global__ __launch_bounds__(256) void k_win_pdeg(const int32_t* __restrict__ rowptr, int64_t n, int32_t* __restrict__ pdeg,
                                                 int32_t* __restrict__ left, int32_t* __restrict__ leftcnt, int32_t* __restrict__ queue) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < MT_QUEUE_INTS) queue[p] = 0;
  if (p > n) return;
  if (p == n) { pdeg[p] = 0; return; }
  const int d = rowptr[p + 1] - rowptr[p];
  const int pd = (d + WN_PAD - 1) & ~(WN_PAD - 1);
  if (pd > WN_BIG) left[atomicAdd(leftcnt, 1)] = (int32_t)p;
  pdeg[p] = (pd > WN_BIG) ? 0 : pd;
}
// Greedy packing, in visiting order, one lane per segment of WN_SEG positions (a window never spans two segments): a target goes
// into the first of the open window's 8 streams that still has room for its padded slots; when none has, the window is closed
// and the target opens the next one.  assign[p] = (window inside the segment << 16) | (stream * 64 + first slot), -1 for targets
// without a place here (no edges; more than 64 padded slots: those are already on the per-target list).
__global__ __launch_bounds__(64) void k_win_greedy(const int32_t* __restrict__ pdeg, int64_t n, int32_t* __restrict__ assign,
                                                  int32_t* __restrict__ segcnt) {
  __shared__ int spd[WN_SEG];
  __shared__ int sas[WN_SEG];
  const int seg = blockIdx.x, lane = threadIdx.x;
  const int64_t p0 = (int64_t)seg * WN_SEG;
  const int cnt = (int)min((int64_t)WN_SEG, n - p0);
  for (int i = lane; i < cnt; i += 64) spd[i] = pdeg[p0 + i];
  __syncthreads();
  {
    // wave-uniform on purpose: sizes through readfirstlane, the eight fill levels and the decision in scalar registers (a lane-0
    // branch ran this loop on the vector unit at ~440 cycles per target)
    // lanes 0 .. 7 hold the fill levels of the 8 streams: "which streams still take this target" is one compare + ballot, the
    // first of them one s_ff1 -- no chain of branches (the scalar if / else form took ~360 cycles per target)
    int fill = 0;
    int win = 0, any = 0;
    for (int ib = 0; ib < cnt; ib += 64) {                      // 64 sizes per LDS read, handed out with v_readlane
      const int mine_pd = (ib + lane < cnt) ? spd[ib + lane] : 0;
      int mine_as = -1;
      const int nb = min(64, cnt - ib);
      for (int j = 0; j < nb; j++) {
        const int pd = __builtin_amdgcn_readlane(mine_pd, j);
        if (pd == 0) continue;
        const unsigned long long fits = __ballot(lane < 8 && fill + pd <= 64);
        int b = 0;
        if (fits == 0) { win++; fill = 0; }                     // full: next window, this target first
        else b = __builtin_ctzll(fits);
        const int off = __builtin_amdgcn_readlane(fill, b);
        if (lane == b) fill += pd;
        any = 1;
        if (lane == j) mine_as = (win << 16) | (b * 64 + off);
      }
      if (ib + lane < cnt) sas[ib + lane] = mine_as;
    }
    if (lane == 0) segcnt[seg] = any ? win + 1 : 0;
  }
  __syncthreads();
  for (int i = lane; i < cnt; i += 64) assign[p0 + i] = sas[i];
}
// first window of every segment (prefix sum of the segments' window counts: a few hundred values, one block), and the first
// position of every window: thread per position -- a target whose place is stream 0, slot 0 opens its window
__global__ __launch_bounds__(1024) void k_win_segbase(const int32_t* __restrict__ segcnt, int n_seg, int32_t* __restrict__ segbase) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n_seg + 1023) / 1024;
  int sum = 0;
  for (int i = t * per; i < min(n_seg, (t + 1) * per); i++) sum += segcnt[i];
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int i = t * per; i < min(n_seg, (t + 1) * per); i++) { segbase[i] = run; run += segcnt[i]; }
  if (t == 1023) segbase[n_seg] = part[1023];
}
__global__ __launch_bounds__(256) void k_win_starts(const int32_t* __restrict__ assign, const int32_t* __restrict__ segbase, int64_t n, int n_seg,
                                                   int n_win, int32_t* __restrict__ wstart, int32_t* __restrict__ wend) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int a = assign[p];
  if (a < 0) return;
  const int seg = (int)(p / WN_SEG);
  const int w = segbase[seg] + (a >> 16);
  if (w >= n_win) return;
  if ((a & 0xffff) == 0) wstart[w] = (int32_t)p;                // (stream 0, slot 0: the target that opened the window)
  atomicMax(&wend[w], (int32_t)p + 1);
}

// one wave per window: first fit of its targets (in visiting order) into 8 streams of 64 slots, the slot lists, the distinct sources
__global__ __launch_bounds__(256) void k_win_pack(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                 const int32_t* __restrict__ order, const int32_t* __restrict__ pdeg,
                                                 const int32_t* __restrict__ assign, const int32_t* __restrict__ segbase,
                                                 const int32_t* __restrict__ wstart, const int32_t* __restrict__ wend, int n_win,
                                                 int32_t* __restrict__ misc_out, int32_t* __restrict__ tgt_out,
                                                 int32_t* __restrict__ eid_out, uint8_t* __restrict__ lrow_out, int32_t* __restrict__ urow_out,
                                                 int32_t* __restrict__ left, int32_t* __restrict__ leftcnt) {
  __shared__ int s_src[4][512];
  __shared__ int s_key[4][1024];
  __shared__ int s_val[4][1024];
  __shared__ short s_assign[4][WN_TMAX];
  __shared__ int s_pos[4][WN_TMAX];
  __shared__ int s_pd[4][WN_TMAX];
  __shared__ int s_e0[4][WN_TMAX];
  __shared__ int s_misc[4][16];
  __shared__ int s_end[4][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = blockIdx.x * 4 + wv;
  if (w >= n_win) return;
  const int p1 = wend[w], p0 = p1 > 0 ? wstart[w] : 0;          // (wend = 0: no window with this number)
  int* ssrc = s_src[wv]; int* skey = s_key[wv]; int* sval = s_val[wv]; short* sas = s_assign[wv]; int* spos = s_pos[wv]; int* spd = s_pd[wv]; int* se0 = s_e0[wv]; int* misc = s_misc[wv]; int* send = s_end[wv];
  const int64_t wb = (int64_t)w * 512;
  if (p1 == 0) {                                                // (window numbers beyond what the greedy pass made: the kernel never looks at them)
    if (lane < 16) misc_out[(int64_t)w * 16 + lane] = 0;
    return;
  }
  for (int i = lane; i < 512; i += 64) { ssrc[i] = -1; eid_out[wb + i] = 0; lrow_out[wb + i] = 0; }
  for (int i = lane; i < 1024; i += 64) skey[i] = -1;
  for (int i = lane; i < 8 * WN_GPS; i += 64) tgt_out[(int64_t)w * (8 * WN_GPS) + i] = 0;
  if (lane < 32) send[lane] = 0;
  // the window's targets (the greedy pass gave each its stream and first slot), compacted in order: position, first edge, padded size
  if (lane == 0) misc[9] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int pb = p0; pb < p1; pb += 64) {
    const int p = pb + lane;
    const int a = p < p1 ? assign[p] : -1;
    const bool mine = a >= 0 && segbase[p / WN_SEG] + (a >> 16) == w;
    const unsigned long long m = __ballot(mine);
    const int base = misc[9];
    const int j = base + __popcll(m & ((1ull << lane) - 1ull));
    if (mine) {
      if (j < WN_TMAX) { spos[j] = p; spd[j] = pdeg[p]; se0[j] = rowptr[p]; sas[j] = (short)(a & 0xffff); }
      else left[atomicAdd(leftcnt, 1)] = p;                     // (a window holds at most 512 / WN_PAD targets: cannot happen)
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) misc[9] = min(base + __popcll(m), WN_TMAX);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  const int cnt = misc[9];
  if (lane < 8) misc[lane] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < cnt; i += 64) atomicMax(&misc[sas[i] >> 6], (sas[i] & 63) + spd[i]);    // slots used per stream
  // the slots of every placed target: one lane per target
  for (int i = lane; i < cnt; i += 64) {
    const int at = sas[i], p = spos[i], e0 = se0[i], pd = spd[i];
    const int d = rowptr[p + 1] - e0;
    for (int q = 0; q < pd; q++) {
      const int e = e0 + min(q, d - 1);
      eid_out[wb + at + q] = e;
      ssrc[at + q] = src[e];
    }
    const int gi = ((at & 63) + pd - WN_PAD) / WN_PAD, b = at >> 6;          // the group that ends this segment
    atomicOr(&send[b * 4 + gi / WN_GPT], 1 << (gi % WN_GPT));
    tgt_out[(int64_t)w * (8 * WN_GPS) + b * WN_GPS + gi] = order ? order[p] : p;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // distinct sources: open addressing in LDS, ids handed out by one counter
  if (lane == 0) misc[8] = 0;
  for (int i = lane; i < 512; i += 64) {
    const int s_ = ssrc[i];
    if (s_ < 0) continue;
    unsigned h = ((unsigned)s_ * 2654435761u) >> 22;
    for (;;) {
      const int prev = atomicCAS(&skey[h], -1, s_);
      if (prev == -1 || prev == s_) break;
      h = (h + 1) & 1023;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < 1024; i += 64)
    if (skey[i] >= 0) sval[i] = atomicAdd(&misc[8], 1);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int nU = misc[8];
  if (nU > WN_UMAX) {                                           // (more distinct rows than a stage holds: the whole window goes per target)
    for (int i = lane; i < cnt; i += 64) left[atomicAdd(leftcnt, 1)] = spos[i];
    if (lane < 16) misc_out[(int64_t)w * 16 + lane] = 0;
    return;
  }
  // (ids in table order: sorting them by source id so that the eight rows of one request are neighbours in Q measured nothing)
  for (int i = lane; i < 1024; i += 64)
    if (skey[i] >= 0) urow_out[(int64_t)w * WN_UMAX + sval[i]] = skey[i];
  for (int i = lane; i < 512; i += 64) {
    const int s_ = ssrc[i];
    if (s_ < 0) continue;
    unsigned h = ((unsigned)s_ * 2654435761u) >> 22;
    while (skey[h] != s_) h = (h + 1) & 1023;
    lrow_out[wb + i] = (uint8_t)sval[h];
  }
  // (the kernel requests the rows eight at a time: ids up to the next multiple of 8 must be real rows)
  if (lane < 8 && nU + lane < ((nU + 7) & ~7)) urow_out[(int64_t)w * WN_UMAX + nU + lane] = 0;
  uint8_t* m8 = (uint8_t*)(misc_out + (int64_t)w * 16);
  if (lane == 0) misc_out[(int64_t)w * 16] = nU;
  if (lane < 8) m8[4 + lane] = (uint8_t)((misc[lane] + 15) >> 4);
  if (lane < 32) m8[12 + lane] = (uint8_t)send[lane];
  if (lane < 5) m8[44 + 4 * lane] = 0, m8[45 + 4 * lane] = 0, m8[46 + 4 * lane] = 0, m8[47 + 4 * lane] = 0;
}

// the targets the windows do not take (more than 64 padded slots, or left over by a full window): one wave per target, lanes
// across channels (eight blocks of 64: d <= 512 in one pass, wider rows in several), the W_e slices in registers, the row pieces
// of TWO edges requested together; fp32 multiply-adds in the order of the per-edge kernel.  ~1 % of the edges of a k = 20 graph
// at 448 slots per window, a few per cent of a crowded cloud.
__global__ __launch_bounds__(256) void k_win_leftover(const float* __restrict__ p_bias, const float* __restrict__ Q, int64_t ldq,
                                                     const float* __restrict__ We, int ldwe, const float* __restrict__ ea, int de,
                                                     const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ order, const int32_t* __restrict__ left,
                                                     const int32_t* __restrict__ leftcnt, int d, float* __restrict__ out, int64_t ldo,
                                                     float* __restrict__ out_absmax) {
  // r05: one WORK-GROUP per target -- its four waves take 128 channels each (two blocks of 64: sixteen weight registers per lane
  // instead of sixty-four) and walk the target's edges EIGHT at a time.  A target here has 65 ... several hundred in-edges; one
  // wave per target with all 512 channels, four edges per trip, was a chain of ~20 dependent L2 round trips behind a 64-load
  // weight prologue (77 us per launch on the 100 000-point cloud for 1.7 % of its targets: 10 % of that step).
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n_left = *leftcnt;
  if (n_left == 0) return;                                      // (the usual case: nothing to do before the weights are even fetched)
  float amax = 0.f;
  for (int c0 = 0; c0 < d; c0 += 512) {
    const int cw = c0 + 128 * wv;                               // this wave's channels: cw + 64 j + lane, j = 0, 1
    if (cw >= d) continue;
    float w[2][8];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int c = cw + 64 * j + lane;
      ok[j] = c < d;
#pragma unroll
      for (int k = 0; k < 8; k++) w[j][k] = (ok[j] && k < de) ? We[(int64_t)c * ldwe + k] : 0.f;
    }
    for (int64_t i = blockIdx.x; i < n_left; i += gridDim.x) {
      const int p = left[i];
      const int e0 = rowptr[p], e1 = rowptr[p + 1];
      const int64_t node = order ? order[p] : p;
      float m[2] = {-INFINITY, -INFINITY};
      for (int eb = e0; eb < e1; eb += 64) {                    // a block of 64 edges: lane l holds edge eb + l's source and attributes
        const int el = min(eb + lane, e1 - 1);
        const int my_src = src[el];
        float my_z[8];
#pragma unroll
        for (int k = 0; k < 8; k++) my_z[k] = k < de ? ea[(int64_t)el * de + k] : 0.f;
        const int nb = min(64, e1 - eb);
        for (int q = 0; q < nb; q += 8) {                       // eight edges' row pieces requested together
          float v[8][2];
          int jj[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            jj[u] = min(q + u, nb - 1);                         // (a tail repeats its last edge)
            const int64_t r = (int64_t)__builtin_amdgcn_readlane(my_src, jj[u]) * ldq + cw + lane;
#pragma unroll
            for (int j = 0; j < 2; j++) v[u][j] = ok[j] ? Q[r + 64 * j] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            float z[8];
#pragma unroll
            for (int k = 0; k < 8; k++) z[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_z[k]), jj[u]));
#pragma unroll
            for (int j = 0; j < 2; j++) {
#pragma unroll
              for (int k = 0; k < 8; k++) v[u][j] = __builtin_fmaf(w[j][k], z[k], v[u][j]);
              m[j] = fmaxf(m[j], v[u][j]);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (!ok[j]) continue;
        const int c = cw + 64 * j + lane;
        const float o = m[j] + (p_bias ? p_bias[c] : 0.f);
        out[node * ldo + c] = o;
        amax = fmaxf(amax, fabsf(o));
      }
    }
  }
  if (out_absmax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0 && amax > 0.f)
      atomicMax((unsigned int*)out_absmax + ((blockIdx.x * 4 + wv) & (RGNN_BOUND_SLOTS - 1)), __float_as_uint(amax));
  }
}

}  // namespace

extern "C" int64_t rgnn_mpnn_win_plan_ints(int64_t n, int64_t n_edges) { return win_layout(n, n_edges).total_ints; }
// diagnostics: word offsets inside a plan of {number of targets on the per-target list, number of windows made}
extern "C" void rgnn_mpnn_win_plan_counters(int64_t n, int64_t n_edges, int64_t* leftover_word, int64_t* windows_word) {
  const WinPlanLayout L = win_layout(n, n_edges);
  *leftover_word = L.off_leftcnt;
  *windows_word = L.off_segbase + L.n_seg;
}

extern "C" int rgnn_mpnn_win_plan(const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order, int64_t n, int64_t n_edges,
                                  int32_t* plan, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(rowptr_t && plan && n >= 0 && n_edges >= 0, "bad arguments");
  RGNN_CHECK_ARG(n_edges == 0 || src_sorted, "null src_sorted");
  RGNN_CHECK_ARG((((uintptr_t)plan) & 15) == 0, "plan must be 16-byte aligned");
  if (n == 0) return RGNN_OK;
  const WinPlanLayout L = win_layout(n, n_edges);
  hipStream_t s = (hipStream_t)stream;
  hipMemsetAsync(plan + L.off_leftcnt, 0, 16, s);
  hipLaunchKernelGGL(k_win_pdeg, dim3(rgnn_blocks(n + 1 > MT_QUEUE_INTS ? n + 1 : MT_QUEUE_INTS, 256)), dim3(256), 0, s, rowptr_t, n, plan + L.off_pdeg,
                     plan + L.off_left, plan + L.off_leftcnt, plan + L.off_queue);
  hipMemsetAsync(plan + L.off_wend, 0, 4 * (size_t)(L.n_win + 1), s);
  hipLaunchKernelGGL(k_win_greedy, dim3(L.n_seg), dim3(64), 0, s, (const int32_t*)(plan + L.off_pdeg), n, plan + L.off_assign, plan + L.off_segcnt);
  hipLaunchKernelGGL(k_win_segbase, dim3(1), dim3(1024), 0, s, (const int32_t*)(plan + L.off_segcnt), L.n_seg, plan + L.off_segbase);
  hipLaunchKernelGGL(k_win_starts, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, (const int32_t*)(plan + L.off_assign),
                     (const int32_t*)(plan + L.off_segbase), n, L.n_seg, L.n_win, plan + L.off_wstart, plan + L.off_wend);
  hipLaunchKernelGGL(k_win_pack, dim3(rgnn_blocks(L.n_win, 4)), dim3(256), 0, s, rowptr_t, src_sorted, node_order,
                     (const int32_t*)(plan + L.off_pdeg), (const int32_t*)(plan + L.off_assign), (const int32_t*)(plan + L.off_segbase),
                     (const int32_t*)(plan + L.off_wstart), (const int32_t*)(plan + L.off_wend), L.n_win, plan + L.off_misc,
                     plan + L.off_tgt, plan + L.off_eid, (uint8_t*)(plan + L.off_lrow),
                     plan + L.off_urow, plan + L.off_left, plan + L.off_leftcnt);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// Operand image of the window kernel for one layer's weights: per 32-channel tile [3 bf16 terms of W_e | bias] (k_win_wplanes).  Depends on
// the weights only: a caller that keeps it per weight version (radargnn_amd/ops.py) saves the 5-us launch in front of every aggregation.
extern "C" int64_t rgnn_mpnn_win_wplanes_bytes(int32_t d) { return d < 1 ? -1 : (int64_t)((d + 31) / 32) * WN_BBUF; }

extern "C" int rgnn_mpnn_win_wplanes(const float* We, int64_t ldwe, int32_t de, int32_t d, const float* p_bias, void* planes,
                                     rgnn_stream_t stream) {
  RGNN_CHECK_ARG(planes && d >= 1 && de >= 0 && de <= 8 && (de == 0 || We) && (((uintptr_t)planes) & 15) == 0, "bad arguments");
  const int n_ct = (d + 31) / 32;
  hipLaunchKernelGGL(k_win_wplanes, dim3(rgnn_blocks(n_ct * 32, 256)), dim3(256), 0, (hipStream_t)stream, We, (int)ldwe, de, d, n_ct, p_bias,
                     (mt_u32x4*)planes);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

static int aggregate_win(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                         const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                         const int32_t* node_order, int32_t* plan, int64_t n, int64_t n_edges, int32_t d, float* out,
                         int64_t ldo, int32_t flags, float* out_absmax, const void* wplanes, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(Q && rowptr_t && plan && out && d >= 1, "bad arguments");
  RGNN_CHECK_ARG((flags & ~RGNN_MPNN_SKIP_EMPTY_ROWS) == 0, "unknown flags");
  RGNN_CHECK_ARG(de == 0 || ((We || wplanes) && edge_attr_sorted), "edge attributes given without weights");
  RGNN_CHECK_ARG((((uintptr_t)wplanes) & 15) == 0, "operand image not 16-byte aligned");
  const int64_t q_bytes = ((n - 1) * ldq + d) * 4, o_bytes = ((n - 1) * ldo + d) * 4;
  if (de > 8 || d > 2048 || n >= ((int64_t)1 << 24) || ldq * 4 >= ((int64_t)1 << 24) || q_bytes >= ((int64_t)1 << 31) ||
      o_bytes >= ((int64_t)1 << 31) || (ldq % 4) != 0 || (((uintptr_t)Q) & 15) != 0 || win_layout(n, n_edges).total_ints * 4 >= ((int64_t)1 << 31)) {
    rgnn_set_error("rgnn_mpnn_aggregate_win: shape not covered (de %d, d %d, n %lld, ldq %lld)", de, d, (long long)n, (long long)ldq);
    return RGNN_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & RGNN_MPNN_SKIP_EMPTY_ROWS))
    hipLaunchKernelGGL(k_tiles_zero_empty, dim3(rgnn_blocks(n, MT_WAVES)), dim3(256), 0, s, rowptr_t, node_order, n, d, out, ldo);
  if (n_edges > 0) {
    const WinPlanLayout L = win_layout(n, n_edges);
    WinParams p;
    p.p_bias = p_bias; p.Q = Q; p.ldq4 = (int)(ldq * 4); p.q_bytes = (int)q_bytes;
    p.wplanes = wplanes ? (const mt_u32x4*)wplanes : (const mt_u32x4*)(plan + L.off_wplanes); p.ea = edge_attr_sorted; p.de = de;
    p.ea_vec = (de == 8 && (((uintptr_t)edge_attr_sorted) & 15) == 0) ? 1 : 0;
    p.n_win = L.n_win; p.n_win_dev = plan + L.off_segbase + L.n_seg;
    p.plan = plan; p.plan_bytes = (int)(L.total_ints * 4);
    p.off_misc = (int)(L.off_misc * 4); p.off_eid = (int)(L.off_eid * 4); p.off_lrow = (int)(L.off_lrow * 4); p.off_urow = (int)(L.off_urow * 4); p.off_tgt = (int)(L.off_tgt * 4);
    p.queue = plan + L.off_queue;
    p.d = d; p.n_ct = (d + 31) / 32; p.out = out; p.ldo4 = (int)(ldo * 4); p.o_bytes = (int)o_bytes; p.out_absmax = out_absmax;
    p.abl = RGNN_ENV("RGNN_MPNN_WIN_ABL") ? atoi(RGNN_ENV("RGNN_MPNN_WIN_ABL")) : 0;
    if (!wplanes)
      hipLaunchKernelGGL(k_win_wplanes, dim3(rgnn_blocks(p.n_ct * 32, 256)), dim3(256), 0, s, We, (int)ldwe, de, d, p.n_ct, p_bias,
                         (mt_u32x4*)(plan + L.off_wplanes));
    const size_t lds = WN_LDS;
    static RgnnOncePerDevice attr_once;                    // (per kernel and device: common.h)
    if (attr_once.first()) {
      hipFuncSetAttribute((const void*)k_mpnn_win<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute((const void*)k_mpnn_win<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    static const int per_cu = RGNN_ENV("RGNN_MPNN_WIN_WG_PER_CU") ? atoi(RGNN_ENV("RGNN_MPNN_WIN_WG_PER_CU")) : 3;
    int64_t blocks = L.n_win;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    rgnn_prof_begin(s);
    if (out_absmax) hipLaunchKernelGGL((k_mpnn_win<true>), dim3((unsigned)blocks), dim3(WN_THREADS), lds, s, p);
    else hipLaunchKernelGGL((k_mpnn_win<false>), dim3((unsigned)blocks), dim3(WN_THREADS), lds, s, p);
    rgnn_prof_end(s);
    hipLaunchKernelGGL(k_win_leftover, dim3(2048), dim3(256), 0, s, p_bias, Q, ldq, We, (int)ldwe, edge_attr_sorted, de, rowptr_t, src_sorted,
                       node_order, (const int32_t*)(plan + L.off_left), (const int32_t*)(plan + L.off_leftcnt), d, out, ldo, out_absmax);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_mpnn_aggregate_win(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                       const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t, const int32_t* src_sorted,
                                       const int32_t* node_order, int32_t* plan, int64_t n, int64_t n_edges, int32_t d, float* out,
                                       int64_t ldo, int32_t flags, float* out_absmax, rgnn_stream_t stream) {
  return aggregate_win(p_bias, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, plan, n, n_edges, d, out, ldo, flags,
                       out_absmax, nullptr, stream);
}

// ... with the operand image built ahead (rgnn_mpnn_win_wplanes from the SAME We / p_bias, which the per-target kernel still reads)
extern "C" int rgnn_mpnn_aggregate_win_planes(const float* p_bias, const float* Q, int64_t ldq, const float* We, int64_t ldwe,
                                              const float* edge_attr_sorted, int32_t de, const int32_t* rowptr_t,
                                              const int32_t* src_sorted, const int32_t* node_order, int32_t* plan, int64_t n,
                                              int64_t n_edges, int32_t d, float* out, int64_t ldo, int32_t flags, float* out_absmax,
                                              const void* wplanes, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(wplanes != nullptr, "null operand image");
  return aggregate_win(p_bias, Q, ldq, We, ldwe, edge_attr_sorted, de, rowptr_t, src_sorted, node_order, plan, n, n_edges, d, out, ldo, flags,
                       out_absmax, wplanes, stream);
}
