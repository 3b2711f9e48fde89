// Dense layer  out = act([A1|A2] * W^T + bias)  on the bf16 matrix pipe (six-term split product, see linear.hip), with the
// operands staged by LDS-DMA.  Replaces ATen addmm behind torch_geometric's dense Linear (gnn/gnn_models.py:137-178,
// gnn/mpnn_layers.py:64-74,89-90) for the wide layers of the model (n > 64, K a multiple of 16).
//
// Why a second kernel.  The register-staged kernel (k_linear_x3) requests a k-step's operands into VGPRs, splits them and
// writes them to LDS; its ISA shows that hipcc re-uses the fragment registers as load destinations and sinks the loads
// behind most of the step's MFMAs, so every k-step opens with `s_waitcnt vmcnt(0)` on loads issued a few hundred cycles
// earlier -- the HBM latency is paid once per step (r01: 0.31 of the bf16 peak, 37 % matrix-pipe busy).  Here
//   * every operand byte travels HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write), issued from
//     inline asm so that hipcc neither counts nor drains them, two (weights) and three (activations) k-steps ahead of the
//     MFMAs that use them, into rings of three / four LDS stages; one counted `s_waitcnt vmcnt(N)` + one barrier per step;
//   * the activation tile lies in LDS as raw fp32 and is split into its three bf16 terms AFTER the fragment read, by the
//     one wave that owns those rows: the eight waves of a work-group are stacked along M (32 rows x BN columns each), so
//     no activation element is split twice and the 32 x 16 fp32 fragment of a wave is two ds_read_b128 per lane.  A wave
//     also REQUESTS its own rows, so its activation path needs no barrier, and it splits the fragment of step g + 1
//     during the MFMAs of step g;
//   * the weight planes (pre-split, [k/16][3][n][16] bf16) are shared by all eight waves: 3 x TN ds_read_b128 per wave and
//     k-step feed 6 x TN MFMAs.
// The accumulation order of every output element (k-steps of 16 ascending; per step l h', h l', m m', m h', h m', h h') is the
// same as in k_linear_x3, so the two kernels agree bit for bit (tests/test_gpu_gnn.py).
//
// LDS stage (k-step of 16):  A: [256 rows][64 B]   chunk c (4 floats) of row r at 16-byte position c ^ ((r >> 2) & 3)
//                            W: [3][BN rows][32 B] chunk c (8 bf16)  of row r at position          c ^ ((r >> 3) & 1)
// -- ds_read_b128 serves the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) one at a time; with these swizzles the
// 16 rows of a group fall on 16 distinct slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0).  LDS-DMA writes
// M0 + 16 * lane, so the swizzle is applied to the per-lane SOURCE address (the 4 / 2 lanes of a row still cover one
// contiguous 64 / 32-byte run of global memory).
#include "linear_common.h"

#ifndef RGNN_DMA_SPREAD
#define RGNN_DMA_SPREAD 1   // issue the DMA pieces of a step between its MFMA groups instead of back to back behind the barrier
#endif
#ifndef RGNN_DMA_SK_MAX_FILL
#define RGNN_DMA_SK_MAX_FILL 88   // stream-K only when the static schedule's tile rounds would be less than 88 % full
#endif
#ifndef RGNN_DMA_PIN
#define RGNN_DMA_PIN 1
#endif
#ifndef RGNN_DMA_TRACK
#define RGNN_DMA_TRACK 1    // the epilogue keeps max |out| per lane (one v_max per element; the atomic only when out_absmax is given)
#endif
#ifndef RGNN_DMA_POST_EPI_WAIT
#define RGNN_DMA_POST_EPI_WAIT 0
#endif
#ifndef RGNN_DMA_PP
#define RGNN_DMA_PP 0       // ping-pong (see the k-loop): 1 = waves 4-7 run half a k-step behind waves 0-3; 2 = ... and at s_setprio 1
#endif
#ifndef RGNN_DMA_KS
#define RGNN_DMA_KS 2       // 16-k sub-stages per k-step (one counted wait + one barrier + one request round per step) where the LDS allows it: see dma_ks
#endif
#ifndef RGNN_DMA_KS_MAX_TN
#define RGNN_DMA_KS_MAX_TN 8
#endif
#ifndef RGNN_DMA_ABL
#define RGNN_DMA_ABL 0      // experiments only: 1 no epilogue, 4 no MFMAs, 8 no DMA, 16 no activation split, 32 no barrier, 64 no DMA wait, 128 no weight-fragment LDS reads, 256 no weight DMA pieces, 512 no activation DMA pieces (results are wrong by construction)
#endif

#ifndef RGNN_DMA_TIMING
#define RGNN_DMA_TIMING 0   // experiments only (tools/x3_bench): per-wave s_memtime sums of the k-step's parts, read back through rgnn_debug_dma_timing
#endif
#if RGNN_DMA_TIMING
__device__ unsigned long long g_dma_t[2048 * 8];
extern "C" int rgnn_debug_dma_timing(unsigned long long* host, int clear) {
  if (clear) { static unsigned long long z[2048 * 8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dma_t), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dma_t), sizeof(g_dma_t));
}
#define TSTAMP() ({ __builtin_amdgcn_sched_barrier(0); unsigned long long _t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); _t; })
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rsrc` at byte offset voff (per lane) + soff (uniform) to LDS
// [lds_base, lds_base + 1024).  Out-of-range offsets deliver zeros.  Invisible to hipcc's s_waitcnt bookkeeping (that is the
// point): completion is awaited by the counted dma_wait<N>() below.  M0 is written and not restored: hipcc emits no M0
// user of its own in these kernels (checked in the ISA: every m0 access sits inside an asm block); the nop is the wait
// state between the M0 write and the DMA.  The scalar operands are written by SALU / readfirstlane several instructions
// earlier (req_begin), so no further wait states are needed in front of the load.
__device__ __forceinline__ void dma16(i32x4 rsrc, int voff, int soff, unsigned lds_base) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %2 offen lds"
      :
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, int bytes) {
  const uint64_t a = (uint64_t)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000;
  return r;
}

constexpr int DMA_BM_MAX = 256;  // rows per work-group tile: 8 waves x 32 (4-wave work-groups: 128)
constexpr int DMA_BK = 16;       // k per step = one MFMA k-extent
// Prefetch depth DW: weights are requested DW k-steps ahead of the MFMAs that use them (ring of DW + 1 stages), activations
// DW + 1 ahead (ring of DW + 2: they are split one step ahead); the counted wait of a step lets DW - 1 groups of requests stay
// outstanding.  DW = 2.  DW = 3 (RGNN_DMA_DEPTH=3: f16x2 form, TN <= 7, +30 KB of LDS; NOT with a1_scale_shift at TN = 7, where
// the table no longer fits) was built to test whether the k-loop's operand rate is (bytes in flight) / latency: bit-identical
// and no faster (profiles/r03_x3_bench_dma_depth.txt) -- it is not.
#ifndef RGNN_DMA_DEPTH
#define RGNN_DMA_DEPTH 2
#endif
// DOUBLE STEPS (r06, dma_ks = 2: f16x2 form, eight waves, column tiles of up to 160).  A k-step is TWO 16-k sub-stages behind ONE
// counted wait, ONE barrier and ONE round of request bookkeeping: the s_memtime build (profiles/r06_dense_kloop_probes.txt) put 181
// + 591 cycles of a 2 660-cycle step into the wait and the barrier, and the scalar bookkeeping of a request round into the 679
// cycles of its load half, against 480 cycles of MFMA issue per wave.  The sub-stages keep their layout (a stage is two of them
// back to back), every accumulator still sees its 16-k products in ascending order (bit-identical results), and the fragment of
// the NEXT sub-stage is split during the MFMAs of the current one, so the register budget is the single-step kernel's.  Rings:
// weights one double step ahead (ring of two), activations THREE double stages for a lead of two steps: the slot of A(g) is
// free once its second half has been read in the first half of step g, and A(g + 3) is requested into it in the second half.
__host__ __device__ constexpr int dma_ks(int tn, int npl, int wv) {
  return (RGNN_DMA_KS == 2 && !RGNN_DMA_TIMING && !RGNN_DMA_PP && npl == 2 && wv == 8 && tn <= RGNN_DMA_KS_MAX_TN) ? 2 : 1;
}
// ... column tiles wider than 160 have no room for three double stages of activations (224 columns: 96 + 56 KB of rings + 21.5 KB
// of statistics exchange): TWO stages there.  A(g + 2) is still requested in the second half of step g, into the slot whose second
// half was read in the first; what changes is when it must have landed: A(g + 1) is awaited in the MIDDLE of step g (its first half
// is read in the second half of the step), behind the weight pieces of W(g + 1) that went out in between -- a lead of one double
// step, the weights' own.
__host__ __device__ constexpr int dma_a_ring(int tn, int npl, int wv) { return dma_ks(tn, npl, wv) == 2 ? (tn <= 5 ? 3 : 2) : 0; }
__host__ __device__ constexpr int dma_depth(int tn, int npl, int wv = 8) {
  return dma_ks(tn, npl, wv) == 2 ? 1 : (RGNN_DMA_DEPTH >= 3 && npl == 2 && tn <= 7) ? 3 : 2;
}
// (wv = waves per work-group: 8 -- one work-group of 256 rows per CU -- or 4: two work-groups of 128 rows per CU, which drift
//  apart so that one of them loads and multiplies while the other stores its tile; see launch_dma)
__host__ __device__ constexpr int dma_w_pieces(int bn, int npl = 3, int wv = 8) { return (npl * bn * 2 + 64 * wv - 1) / (64 * wv); }   // 16-B chunks / threads
__host__ __device__ constexpr int dma_w_stage(int bn, int npl = 3, int wv = 8) { return dma_w_pieces(bn, npl, wv) * 64 * wv * 16; }
__host__ __device__ constexpr bool dma_pp(int wv, int npl) { return RGNN_DMA_PP != 0 && wv == 8 && npl == 2; }   // ping-pong schedule (k-loop): f16x2 form, 8 waves
__host__ __device__ constexpr int dma_lds_bytes(int bn, int npl = 3, int wv = 8) {
  const int ks = dma_ks(bn / 32, npl, wv), dw = dma_depth(bn / 32, npl, wv);
  // (double steps: the weight stage is packed -- both sub-stages' chunks back to back, no padding to whole pieces: a wave whose
  //  part of the last piece lies beyond the stage does not issue it)
  const int w_stage = ks == 2 ? ks * npl * bn * 2 * 16 : dma_w_stage(bn, npl, wv);
  const int a_ring = ks == 2 ? dma_a_ring(bn / 32, npl, wv) : dw + 2;
  return a_ring * ks * (32 * wv * DMA_BK * 4) + (dw + 1 + (dma_pp(wv, npl) ? 1 : 0)) * w_stage +
         stat_lds_floats(wv, bn) * 4 + 32 * wv * 4;
}

typedef short raw16x8 __attribute__((ext_vector_type(8)));   // eight 16-bit operand words (bf16 or f16) as they lie in LDS

// FMT 0: three bf16 terms per operand, six MFMA products per fp32 product (the form described above).
// FMT 1 (r03): TWO f16 terms per operand, THREE products -- l h', h l', h h' on v_mfma_f32_32x32x16_f16, which runs at the
//   bf16 rate.  a = h + l carries 22 significand bits (|a - h - l| <= 2^-23 |a|), the dropped l l' is <= 2^-22 |a a'|: the same
//   error class as the fp32 MFMA path (tests/test_gpu_gnn.py measures both against float64).  f16 has 5 exponent bits, so both
//   operands are pre-scaled by exact powers of two: the weight by 2^sw at split time (|W| 2^sw < 2^15, footer of the plane
//   buffer), the activations by 2^sa derived HERE from a device word that bounds them (a1_bound / a2_bound, written by the
//   kernels that produced them: |A| 2^sa < 2^15); the epilogue multiplies the accumulator by 2^-(sa + sw).  With the bound
//   at 2^15 an element keeps all 22 bits while it is >= 2^-3 (its l is a normal f16 number), smaller ones are off by at most
//   2^-25 in the scaled domain = 2^-40 of the bound -- a bound 2^19 above the tensor's typical magnitude still costs nothing.
template <int TN, bool IDX, int FMT, int WV = 8>
__global__ __launch_bounds__(64 * WV) void k_linear_dma(const LinParams p) {
  constexpr int BN = 32 * TN;
  constexpr int DMA_BM = 32 * WV;                  // rows per work-group tile
  constexpr int DMA_THREADS = 64 * WV;
  constexpr int KS = dma_ks(TN, FMT ? 2 : 3, WV);  // 16-k sub-stages per k-step (2: double steps, see dma_ks)
  constexpr int DMA_A_SUB = DMA_BM * DMA_BK * 4;   // one sub-stage of raw fp32 activations
  constexpr int DMA_A_STAGE = KS * DMA_A_SUB;
  constexpr int DMA_SK_SLOT_BYTES = 8 * 16 * DMA_THREADS * 4;   // accumulators of one work-group at the widest tile (TN = 8)
  constexpr int NPL = FMT ? 2 : 3;                 // weight planes (terms per operand)
  constexpr int W_PLANE = BN * 32;                 // bytes of one weight plane of a stage
  constexpr int NWQ = NPL * BN * 2;                // 16-byte chunks of the weight tile
  // weight pieces per thread and k-step.  Single steps: the last one may be partly beyond the tile (killed lanes write zeros into
  // the stage's padding).  Double steps: the stage is packed (chunk q of the step = sub-stage q / NWQ, chunk q % NWQ of it, at
  // byte 16 q), and a wave whose lanes of the last piece all lie beyond it does not issue that piece (NWQ is a multiple of 64).
  constexpr int NW = KS == 2 ? (KS * NWQ + DMA_THREADS - 1) / DMA_THREADS : dma_w_pieces(BN, NPL, WV);
  constexpr int NW_ALL = KS == 2 ? (KS * NWQ) / DMA_THREADS : NW;   // ... of which every wave issues the first NW_ALL
  constexpr int NA = 2;                            // a wave's own 32 rows x 4 chunks / 64 lanes (per sub-stage)
  constexpr int NLD = NW + KS * NA;                // DMA pieces per thread and k-step: NW weight pieces, then KS NA activation pieces
  constexpr int W_SUB = KS == 2 ? NWQ * 16 : dma_w_stage(BN, NPL, WV);   // one sub-stage of weight planes
  constexpr int W_STAGE = KS * W_SUB;
  constexpr int DW = dma_depth(TN, NPL, WV);       // prefetch depth of the weight stream (activations: DW + 1; double steps: DW + 2)
  constexpr bool PP = dma_pp(WV, NPL);                  // ping-pong schedule of the two waves of a SIMD (k-loop)
  constexpr int DMA_A_RING = KS == 2 ? dma_a_ring(TN, NPL, WV) : DW + 2, DMA_W_RING = DW + 1 + (PP ? 1 : 0);
  constexpr bool AR2 = KS == 2 && DMA_A_RING == 2;   // two activation stages: A(g + 1) is awaited in the middle of step g
  static_assert(KS == 1 || (DW == 1 && !PP), "double steps: weights one step ahead, activations two");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = (char*)smem;
  char* const lds_w = lds + DMA_A_RING * DMA_A_STAGE;
  float* const stat_lds = (float*)(lds_w + DMA_W_RING * W_STAGE);        // stat_lds_floats(WV, BN) floats (epilogue, stats.h)
  int* const row_tab = (int*)(stat_lds + stat_lds_floats(WV, BN));                   // [DMA_BM] (row-subset epilogue)
  float* const aff_lds = (float*)(row_tab + DMA_BM);                     // [1 or 2 tables][RGNN_AFFINE_ROWS][k1]: BatchNorm-apply of the A1 operand (optional)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  const int64_t M = (IDX && p.m_dev) ? *p.m_dev : p.m;
  const int mt = (int)((M + DMA_BM - 1) / DMA_BM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, g8 = gridDim.x >> 3;
  const int my_panels = (mt > xcd) ? (mt - xcd + 7) / 8 : 0;
  const int n_items = my_panels * p.nt;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int K = p.k1 + p.k2;
  const int nk = (K + KS * DMA_BK - 1) / (KS * DMA_BK);   // (the dispatcher guarantees K % 16 == 0 and k1 % 16 == 0; double steps: the last
                                                          //  one may have an empty second half -- zero activations against zero-padded planes)

  // ---- which (tile, k-step) units this work-group does.  Item i of this XCD = (row panel xcd + 8 (i / nt), column tile
  // i % nt): the column tiles of a panel are neighbours, so its rows leave HBM once per XCD.
  //   static:   items slot, slot + g8, ... whole, in ascending order (round robin over the XCD's work-groups);
  //   stream-K: the XCD's units u = i nk + kt are cut into g8 EQUAL ranges, one per work-group -- no partial tile rounds
  //     (522 tiles on 256 CUs are 3 rounds of which the third is 4 % full) and the work-groups drift apart, so their
  //     epilogue write bursts no longer coincide.  A work-group walks its range from the HIGHEST item down (k ascending
  //     inside an item).  An item cut between work-groups s and s + 1 is then started by s as the FIRST thing it does (k
  //     from 0; accumulators -> workspace slot s, flag s) and finished by s + 1 as the LAST thing it does: it loads those
  //     accumulators instead of zeros and carries on with the next k-step -- every output element sees exactly the
  //     accumulation sequence of the undivided tile (bit-identical), and a work-group only ever waits for a LOWER-numbered
  //     one that produced its part long before (no wait in practice, none that dispatch order could turn into a deadlock).
  // (worth it when the static deal would leave the last round of tiles mostly idle and the tiles are long enough to pay for
  // one 256 KiB hand-over per work-group)
  //   Only for layers of ONE column tile, and only where the idle part of the static deal's last round outweighs the hand-over
  //     (r03, profiles/r03_x3_bench_stream_k_teams.txt).  With nt > 1 a work-group that walks a contiguous range does a panel's
  //     column tiles one after the other, and by the time it returns to the panel's rows the XCD's other work-groups have pushed
  //     them out of L2 (PMC: 313 MB fetched for 120 MB of rows at nt = 3), where the static deal has NEIGHBOURS do them at the
  //     same time; and a hand-over moves the accumulators twice, 2 BN / (K + BN) of a tile's own traffic -- 0.8 tiles at
  //     K = 224, N = 464, more than the 0.875 idle rounds it removes there.  Static: 144 us, stream-K: 165 us; dealing the
  //     ranges to TEAMS of nt work-groups that walk the same panels in step was built too and measured 166 us (bit-identical).
  const int sk_rounds = (n_items + g8 - 1) / g8;
  const bool sk = p.sk_ws != nullptr && p.nt == 1 && n_items >= g8 && nk >= 8 &&
                  n_items * 100 < sk_rounds * g8 * RGNN_DMA_SK_MAX_FILL &&
                  (float)n_items / (float)g8 + 2.f * BN / (float)(K + BN) + 0.25f < (float)sk_rounds;
  //   parallel split-K (few items: one frame is 12 row panels, the XCD's work-groups would mostly idle and the layer's
  //     latency is one work-group's whole k-loop): every item is cut into S k-ranges done by S work-groups at the same time,
  //     each from zero; the first S - 1 leave their accumulators in the workspace, the one with the highest k-range adds
  //     them to its own in the fixed order own + range 0 + range 1 + ... and runs the epilogue.  Deterministic, but the
  //     rounding differs from the undivided k-loop in the last bit (same error class).
  int psk_S = 1;
  // (narrow column tiles only -- the ones the dispatcher picks for few rows: the wide instances have no registers to spare)
  // S comes from the LARGEST item count of an XCD, so every row sees the same k-ranges whichever panel it falls into (the
  // result of a row must not depend on the order of a row list)
  const int max_items = ((mt + 7) / 8) * p.nt;
  if (TN <= 4 && !sk && p.sk_ws != nullptr && n_items > 0 && 2 * max_items <= g8 && nk >= 8 && !p.no_split_k) {
    psk_S = g8 / max_items;
    if (psk_S > 8) psk_S = 8;
    if (psk_S > nk / 4) psk_S = nk / 4;
  }
  const bool psk = psk_S > 1;
  int w_base, w_stride, w_count, w_kb_last, w_ke_first;      // items w_base + j w_stride, j < w_count; sub-ranges of the ends
  if (psk) {
    if (slot >= n_items * psk_S) return;
    const int piece = slot % psk_S;
    w_base = slot / psk_S; w_stride = 1; w_count = 1;
    w_kb_last = nk * piece / psk_S;
    w_ke_first = nk * (piece + 1) / psk_S;
  } else if (sk) {
    const int64_t U = (int64_t)n_items * nk;
    const int64_t u_lo = U * slot / g8, u_hi = U * (slot + 1) / g8;
    const int i_lo = (int)(u_lo / nk), i_hi = (int)((u_hi - 1) / nk);
    w_base = i_hi; w_stride = -1; w_count = i_hi - i_lo + 1;
    w_kb_last = (int)(u_lo - (int64_t)i_lo * nk);
    w_ke_first = (int)(u_hi - (int64_t)i_hi * nk);
  } else {
    if (slot >= n_items) return;
    w_base = slot; w_stride = g8; w_count = (n_items - slot + g8 - 1) / g8;
    w_kb_last = 0; w_ke_first = nk;
  }
  if (p.stagger > 0 && !sk && !psk && w_count >= 2) {
    // START STAGGER (r05).  The 256 work-groups of a static launch start together and their tiles take the same time, so the whole
    // chip alternates between k-loops that only read (at the rate one tile's operand ring can pull) and epilogues that only write
    // (all 256 at once: a burst at the HBM write rate) -- r03's ablations saw the two ADD inside a work-group; across the chip they
    // add because they COINCIDE.  Work-group `slot` of an XCD therefore waits (slot & 31) / 32 of ~0.65 tile periods before its
    // first request: the phases stay spread for the rest of the launch (every tile takes the same time), reads and writes of
    // different CUs overlap, and what a work-group idles at the start (half the spread on average) comes back several times --
    // captured steps, same box, alternating: C2 2.176 -> 2.128 ms, C3 3.404 -> 3.224, C4 4.364 -> 4.223, C5 4.54 -> 4.31
    // (RGNN_DMA_STAGGER = per cent of the default spread, 0 switches it off; results are bit-identical either way).  The spread
    // follows the tile period: k-steps x (0.5 + 0.1 TN) us + 2.2 TN us of epilogue, in units of 64 clocks.
    const float period_us = (float)(K / DMA_BK) * (0.5f + 0.1f * TN) + 2.2f * TN;
    const int unit = (int)(0.8f * period_us * (float)p.stagger * 0.01f);   // s_sleep(1) steps per phase: 0.65 period / 31 phases at ~27 ns a step
    const int units = (slot & 31) * (unit > 0 ? unit : 1);
    for (int u = 0; u < units; u++) __builtin_amdgcn_s_sleep(1);
  }
  const int sk_idx = xcd * g8 + slot;              // workspace slot / flag of this work-group (its predecessor: sk_idx - 1)
  struct Cursor { int j, item, kt, kend; };
  auto cursor_begin = [&]() {
    Cursor c;
    c.j = 0; c.item = w_base; c.kt = (w_count == 1) ? w_kb_last : 0; c.kend = w_ke_first;
    return c;
  };
  auto cursor_next = [&](Cursor& c) -> bool {       // moves to the next k-step; true when it entered a new item
    if (++c.kt < c.kend) return false;
    c.j++;
    c.item += w_stride;
    c.kt = (c.j == w_count - 1) ? w_kb_last : 0;
    c.kend = nk;
    return true;
  };

  const i32x4 ra1_d = make_rsrc(p.A1, p.ext_a1);
  const i32x4 ra2_d = make_rsrc(p.A2 ? p.A2 : p.A1, p.A2 ? p.ext_a2 : 0);
  const i32x4 rw_d = make_rsrc(p.Wp, p.ext_wp);

  // ---- two load streams walk the (tile, k-step) sequence of the MFMAs: weights two steps ahead, activations three.
  // Activations: a wave requests ITS OWN 32 rows (piece s: rows 32 wave + 16 s + (lane >> 2), LDS position lane & 3 holds
  // logical chunk (lane & 3) ^ ((row >> 2) & 3)), so nobody else ever touches them: no barrier between the DMA and the
  // fragment read, only the wave's own vmcnt.  Weights: piece s covers LDS chunks 512 s + t of the stage, all waves read all.
  int va1[NA], va2[NA], vw[NW];
  Cursor ca = cursor_begin(), cw = cursor_begin();   // load streams (activations, weights)
  auto a_offsets = [&](int it) {
    const int64_t m0 = (int64_t)(xcd + 8 * (it / p.nt)) * DMA_BM + wave * 32;
#pragma unroll
    for (int s = 0; s < NA; s++) {
      const int r = 16 * s + (lane >> 2);
      const int64_t gm = m0 + r;
      int64_t row = -1;
      if (gm < M) row = (IDX && !(RGNN_DMA_ABL & 1024)) ? (int64_t)p.row_index[gm] : gm;   // (1024: experiment, no index loads)
      if ((RGNN_DMA_ABL & 2048) && row >= 0) row &= 1023;                                 // (2048: experiment, activations from cache-resident rows)
      const int c = ((lane & 3) ^ ((r >> 2) & 3)) * 16;
      va1[s] = (row >= 0) ? (int)(row * p.lda1 * 4) + c : OOB;
      va2[s] = (row >= 0) ? (int)(row * p.lda2 * 4) + c : OOB;
    }
  };
  auto w_offsets = [&](int it) {
    const int n0 = (it % p.nt) * BN;
#pragma unroll
    for (int s = 0; s < NW; s++) {
      const int q = t + DMA_THREADS * s;            // chunk -> (plane, row, position)
      const int plane = q / (2 * BN), row = (q >> 1) % BN, c = (q & 1) ^ ((row >> 3) & 1);
      const int gn = n0 + row;
      vw[s] = (q < NWQ && gn < p.n) ? (int)(((int64_t)plane * p.n + gn) * 32) + c * 16 : OOB;
      if constexpr (KS == 2) {                      // packed double stage: chunk q = sub-stage q / NWQ (one 16-k block of planes further on), chunk q % NWQ of it
        const int hh = q / NWQ, qq = q - hh * NWQ;
        const int plane2 = qq / (2 * BN), row2 = (qq >> 1) % BN, c2 = (qq & 1) ^ ((row2 >> 3) & 1);
        const int gn2 = n0 + row2;
        vw[s] = (hh < KS && gn2 < p.n) ? (int)((((int64_t)hh * NPL + plane2) * p.n + gn2) * 32) + c2 * 16 : OOB;
      }
    }
  };
  int a_ring = 0, w_ring = 0;                       // ring slots the streams fill next
  // One k-step's request = NW weight pieces then NA activation pieces (a stream that has run out keeps issuing killed
  // pieces, so the counts stay uniform).  begin() fixes the step's uniform operands, piece(i) issues piece i, end() moves
  // the streams on; the main loop spreads the pieces over the step's MFMA groups (an LDS-DMA piece costs the issuing wave
  // 60 - 180 cycles; back to back behind the barrier all eight waves pay that at the same time while the matrix pipe idles).
  struct Req { i32x4 ra_d[KS]; int a_kill[KS], w_kill, a_soff[KS], w_soff; unsigned a_base, w_base; bool use1[KS]; } rq;
  auto req_begin = [&]() {
#pragma unroll
    for (int h = 0; h < KS; h++) {                  // sub-stage h of the activation step: its own operand block, its own kill
      const int k0 = (ca.kt * KS + h) * DMA_BK;
      rq.use1[h] = k0 < p.k1;
      rq.a_kill[h] = (ca.j < w_count && k0 < K) ? 0 : OOB;
      rq.ra_d[h] = rq.use1[h] ? ra1_d : ra2_d;
      rq.a_soff[h] = __builtin_amdgcn_readfirstlane(rq.use1[h] ? k0 * 4 : (k0 - p.k1) * 4);
    }
    rq.w_soff = __builtin_amdgcn_readfirstlane(cw.kt * KS * NPL * p.n * 32);   // (planes are zero-padded to a multiple of 32 k)
    rq.a_base = __builtin_amdgcn_readfirstlane(lds0 + a_ring * DMA_A_STAGE + wave * 2048);
    rq.w_kill = (cw.j < w_count) ? 0 : OOB;
    rq.w_base = __builtin_amdgcn_readfirstlane(lds0 + DMA_A_RING * DMA_A_STAGE + w_ring * W_STAGE + wave * 1024);
  };
  auto req_piece = [&](int i) {                     // i is a compile-time constant at every call site
    if (RGNN_DMA_ABL & 8) return;
    if ((RGNN_DMA_ABL & 256) && i < NW) return;       // experiment: no weight pieces
    if ((RGNN_DMA_ABL & 512) && i >= NW) return;      // experiment: no activation pieces
    if (i < NW) {
      if (i < NW_ALL || i * DMA_THREADS + wave * 64 < KS * NWQ)    // (wave-uniform: the last piece of a packed double stage)
        dma16(rw_d, vw[i] | rq.w_kill, rq.w_soff, rq.w_base + i * (DMA_THREADS * 16));
    } else {
      const int h = (i - NW) / NA, ss = (i - NW) % NA;
      dma16(rq.ra_d[h], (rq.use1[h] ? va1[ss] : va2[ss]) | rq.a_kill[h], rq.a_soff[h], rq.a_base + h * DMA_A_SUB + ss * 1024);
    }
  };
  auto advance_a = [&]() {
    a_ring = (a_ring == DMA_A_RING - 1) ? 0 : a_ring + 1;
    if (ca.j < w_count && cursor_next(ca) && ca.j < w_count) a_offsets(ca.item);
  };
  auto advance_w = [&]() {
    w_ring = (w_ring == DMA_W_RING - 1) ? 0 : w_ring + 1;
    if (cw.j < w_count && cursor_next(cw) && cw.j < w_count) w_offsets(cw.item);
  };
  auto req_end = [&]() { advance_a(); advance_w(); };
  auto issue_a = [&]() {                            // prologue only: the activation half of a request
    req_begin();
#pragma unroll
    for (int i = NW; i < NLD; i++) req_piece(i);
    advance_a();
  };
  auto issue_w = [&]() {
    req_begin();
#pragma unroll
    for (int i = 0; i < NW; i++) req_piece(i);
    advance_w();
  };

  // ---- fragments
  // activation: row 32 wave + (lane & 31) of the tile = row (lane & 31) of the wave's own block; k = 8 (lane >> 5) .. + 7 =
  // logical chunks 2 (lane >> 5) and + 1
  const int ga = (lane >> 2) & 3;
  const int a_off0 = wave * 2048 + (lane & 31) * 64 + (((2 * (lane >> 5)) ^ ga) * 16);
  const int a_off1 = wave * 2048 + (lane & 31) * 64 + (((2 * (lane >> 5) + 1) ^ ga) * 16);
  // weight: row j * 32 + (lane & 31), k = 8 (lane >> 5) .. + 7 = chunk lane >> 5
  const int b_off = (lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) * 16);
  struct Planes { raw16x8 h, m, l; };
  // operand pre-scale of the f16x2 form (wave-uniform): a_mul = 2^sa with bound * 2^sa < 2^15; out_mul = 2^-(sa + sw)
  float a_mul = 1.f, out_mul = 1.f;
  if constexpr (FMT == 1) {
    float bound = bound_read(p.a1_bound);
    if (p.k2 > 0) bound = fmaxf(bound, bound_read(p.a2_bound));
    bound = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bound)));
    const int be = (int)((__float_as_uint(bound) >> 23) & 255u);       // bound < 2^(be - 126)
    int se = 268 - be;                                                  // 2^(se - 127) * bound < 2^15
    se = se > 253 ? 253 : se;
    a_mul = __uint_as_float((unsigned)se << 23);
    const float* foot = (const float*)((const char*)p.Wp + (size_t)2 * p.n * p.kp * 2);   // {absmax, scale, 1 / scale, 0}
    out_mul = __uint_as_float((unsigned)(254 - se) << 23) * foot[2];
  }
  // Optional BatchNorm-apply of the A1 operand (a1_aff): x := max(x * scale[k] + shift[k], lo) on the fragment, scale / shift
  // of the step's 16 k-values from an LDS table -- 8 FMAs + 8 max + 4 ds_read_b128 per step and wave, which the probe build
  // measured as free (+-1 %, profiles/r02_x3_bench_affine_probe.txt) while the separate pass it replaces costs 39 us per layer.
  const bool aff_on = p.a1_aff != nullptr;
  const float aff_lo = p.a1_relu ? 0.f : -INFINITY;
  // Per-segment tables (a1_aff_panel: a batch whose frames are normalised with their own statistics): the row list keeps a
  // segment inside 256-row tiles of its own, so a tile has ONE table.  Two are resident: the fragment split runs one k-step ahead
  // of the MFMAs and crosses into the next item during an item's last step.  Item j of this work-group reads table slot j & 1.
  const bool aff_seg = IDX && aff_on && p.a1_aff_panel != nullptr;
  auto load_aff = [&](int item, int slot_) {
    const float* src_ = p.a1_aff + (aff_seg ? (int64_t)p.a1_aff_panel[xcd + 8 * (item / p.nt)] * RGNN_AFFINE_ROWS * p.k1 : 0);
    for (int i = t; i < RGNN_AFFINE_ROWS * p.k1; i += DMA_THREADS) aff_lds[slot_ * RGNN_AFFINE_ROWS * p.k1 + i] = i < p.k1 ? src_[i] : src_[i] * a_mul;
  };
  if (aff_on) {                                     // (rows mean | g | t of rgnn.h; f16x2: g and t carry the pre-scale -- fma(x - mean, g 2^sa, t 2^sa) = 2^sa fma(x - mean, g, t) exactly)
    load_aff(w_base, 0);
    if (aff_seg && w_count > 1) load_aff(w_base + w_stride, 1);
    __syncthreads();
  }
  struct RawA { float4 x0, x1; };
  auto load_a = [&](int ring, int h = 0) -> RawA {    // the wave's fp32 fragment of sub-stage h of a stage: two ds_read_b128
    const char* st = lds + ring * DMA_A_STAGE + h * DMA_A_SUB;
    RawA r;
    r.x0 = *(const float4*)(st + a_off0);
    r.x1 = *(const float4*)(st + a_off1);
    return r;
  };
  auto split_a = [&](const RawA& raw, int kt, int aslot = 0) -> Planes {   // fp32 fragment of the 16-k sub-stage kt -> its operand terms (aslot: table slot of its item)
    float4 x0 = raw.x0, x1 = raw.x1;
    if (aff_on && kt * DMA_BK < p.k1) {               // (wave-uniform: the step lies in the A1 part)
      const float* mu = aff_lds + aslot * RGNN_AFFINE_ROWS * p.k1 + kt * DMA_BK + 8 * (lane >> 5);
      const float* sc = mu + p.k1;
      const float* sh = sc + p.k1;
      // (x - mean) g + t on channel PAIRS: v_pk_add_f32 + v_pk_fma_f32 are one issue slot per two elements (eight scalar
      // subtracts more per step than the r03 form x scale + shift cost 4 - 5 us per launch; packed, the count is r03's again)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float4 m0 = *(const float4*)mu, m1 = *(const float4*)(mu + 4);
      const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4), t0 = *(const float4*)sh, t1 = *(const float4*)(sh + 4);
      const f32x2 y0 = __builtin_elementwise_fma(f32x2{x0.x, x0.y} - f32x2{m0.x, m0.y}, f32x2{s0.x, s0.y}, f32x2{t0.x, t0.y});
      const f32x2 y1 = __builtin_elementwise_fma(f32x2{x0.z, x0.w} - f32x2{m0.z, m0.w}, f32x2{s0.z, s0.w}, f32x2{t0.z, t0.w});
      const f32x2 y2 = __builtin_elementwise_fma(f32x2{x1.x, x1.y} - f32x2{m1.x, m1.y}, f32x2{s1.x, s1.y}, f32x2{t1.x, t1.y});
      const f32x2 y3 = __builtin_elementwise_fma(f32x2{x1.z, x1.w} - f32x2{m1.z, m1.w}, f32x2{s1.z, s1.w}, f32x2{t1.z, t1.w});
      x0.x = fmaxf(y0.x, aff_lo); x0.y = fmaxf(y0.y, aff_lo); x0.z = fmaxf(y1.x, aff_lo); x0.w = fmaxf(y1.y, aff_lo);
      x1.x = fmaxf(y2.x, aff_lo); x1.y = fmaxf(y2.y, aff_lo); x1.z = fmaxf(y3.x, aff_lo); x1.w = fmaxf(y3.y, aff_lo);
    } else if constexpr (FMT == 1) {
      x0.x *= a_mul; x0.y *= a_mul; x0.z *= a_mul; x0.w *= a_mul;
      x1.x *= a_mul; x1.y *= a_mul; x1.z *= a_mul; x1.w *= a_mul;
    }
    Planes r;
    if constexpr (FMT == 1) {
      f16x4_t h0, l0, h1, l1;
      split2(x0, h0, l0);
      split2(x1, h1, l1);
      r.h = __builtin_bit_cast(raw16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      r.l = __builtin_bit_cast(raw16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
      r.m = r.l;                                     // (unused)
    } else {
      bf16x4_t h0, m0, l0, h1, m1, l1;
      split3(x0, h0, m0, l0);
      split3(x1, h1, m1, l1);
      r.h = __builtin_bit_cast(raw16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      r.m = __builtin_bit_cast(raw16x8, __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7));
      r.l = __builtin_bit_cast(raw16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    return r;
  };
  auto read_a = [&](int ring, int kt, int aslot = 0) -> Planes { return split_a(load_a(ring), kt * KS, aslot); };

  f32x16 acc[1][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[0][j][r] = 0.f;
  };
  // ---- stream-K hand-over of an item's accumulators (layout: float4 (j, q) of thread t at ((4 j + q) 512 + t) 16 bytes)
  // (buffer descriptors: lane offset t * 16 in ONE VGPR, the float4's offset in an SGPR -- with plain pointers hipcc keeps
  // dozens of precomputed 64-bit addresses in VGPRs across the whole kernel and the main loop spills)
  const __amdgpu_buffer_rsrc_t sk_mine = __builtin_amdgcn_make_buffer_rsrc(
      (char*)p.sk_ws + (size_t)sk_idx * DMA_SK_SLOT_BYTES, (short)0, p.sk_ws ? DMA_SK_SLOT_BYTES : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t sk_prev = __builtin_amdgcn_make_buffer_rsrc(
      (char*)p.sk_ws + (size_t)(sk_idx > 0 ? sk_idx - 1 : 0) * DMA_SK_SLOT_BYTES, (short)0, p.sk_ws ? DMA_SK_SLOT_BYTES : 0, 0x00020000);
  auto store_partial = [&]() {
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4v v;
        v.x = acc[0][j][4 * q]; v.y = acc[0][j][4 * q + 1]; v.z = acc[0][j][4 * q + 2]; v.w = acc[0][j][4 * q + 3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sk_mine, t * 16, (j * 4 + q) * DMA_THREADS * 16, 0);
      }
    __syncthreads();
    if (t == 0) {                                   // publish: agent-scope release, then the flag (cdna guide, Guideline 16)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __hip_atomic_store(p.sk_flags + sk_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto load_partial = [&]() {
    if (t == 0) {
      // (bounded: the producer is a lower-numbered work-group that wrote its part as the first thing it did; if it has not
      // after ~1 s something else is wrong, and a wrong tile is a better failure than a device that never comes back)
      int spin = 0;
      for (; spin < (1 << 22) && __hip_atomic_load(p.sk_flags + sk_idx - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; spin++)
        __builtin_amdgcn_s_sleep(8);
      if (spin == (1 << 22)) atomicAdd(p.sk_flags + RGNN_SPLITK_TIMEOUT_WORD, 1);   // (the tile is wrong: say so -- rgnn.h)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 v = buf_load16(sk_prev, t * 16, (j * 4 + q) * DMA_THREADS * 16);
        acc[0][j][4 * q] = v.x; acc[0][j][4 * q + 1] = v.y; acc[0][j][4 * q + 2] = v.z; acc[0][j][4 * q + 3] = v.w;
      }
#if defined(__HIP_DEVICE_COMPILE__)
    // (settle these loads HERE: otherwise hipcc carries "accumulator load possibly pending" into the k-loop and guards the
    // first MFMA on every accumulator with a vmcnt wait -- down to vmcnt(0) -- on every step, draining the DMA prefetches)
#pragma unroll
    for (int j = 0; j < TN; j++) asm volatile("" : "+v"(acc[0][j]));
#endif
    if (t == 0) __hip_atomic_store(p.sk_flags + sk_idx - 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  };
  // parallel split-K: the finishing work-group (slot = item S + S - 1) adds the parts of slots item S .. item S + S - 2
  const __amdgpu_buffer_rsrc_t sk_parts = __builtin_amdgcn_make_buffer_rsrc(
      (char*)p.sk_ws + (size_t)(psk ? sk_idx - (slot % psk_S) : 0) * DMA_SK_SLOT_BYTES, (short)0,
      psk ? psk_S * DMA_SK_SLOT_BYTES : 0, 0x00020000);
  auto combine = [&]() {
    if (TN > 4) return;
    const int first = sk_idx - (psk_S - 1);
    for (int s = 0; s < psk_S - 1; s++) {
      if (t == 0) {
        int spin = 0;
        for (; spin < (1 << 22) && __hip_atomic_load(p.sk_flags + first + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; spin++)
          __builtin_amdgcn_s_sleep(2);
        if (spin == (1 << 22)) atomicAdd(p.sk_flags + RGNN_SPLITK_TIMEOUT_WORD, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(p.sk_flags + first + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      __syncthreads();
      const int soff = __builtin_amdgcn_readfirstlane(s * DMA_SK_SLOT_BYTES);
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 v = buf_load16(sk_parts, t * 16, soff + (j * 4 + q) * DMA_THREADS * 16);
          acc[0][j][4 * q] += v.x; acc[0][j][4 * q + 1] += v.y; acc[0][j][4 * q + 2] += v.z; acc[0][j][4 * q + 3] += v.w;
        }
    }
  };
  float amax = 0.f;                                 // |out| seen by this lane (out_absmax)
#ifndef RGNN_DMA_PRIO_YOUNG
#define RGNN_DMA_PRIO_YOUNG 0
#endif
  if (((PP && RGNN_DMA_PP == 2) || RGNN_DMA_PRIO_YOUNG) && WV == 8 && (wave >> 2) == 1) __builtin_amdgcn_s_setprio(1);   // (static priority for the younger half: MI355X_MICROARCH.md item 4)
  Cursor cc = cursor_begin();                       // compute stream
  int ca_ring = 0, cw_ring = 0, ca_cur = 0;         // ... and the ring slots it reads next (ca_cur: double steps, the slot of the CURRENT step's activations)
  a_offsets(w_base);
  w_offsets(w_base);
  issue_a();                                        // A(0)
  if constexpr (KS == 2) {                          // double steps: A(1), W(0), A(2) -- step g then requests W(g + 1) and A(g + 3)
    // (the counted waits rely on the order inside a step: weight pieces first, then activation pieces -- the youngest requests are
    //  the ones that may stay outstanding)
    if constexpr (AR2) { issue_w(); issue_a(); }     // W(0), A(1)
    else { issue_a(); issue_w(); issue_a(); }        // A(1), W(0), A(2)
    a_ring = 0;                                     // (every stage was filled: the ring pointer is back at A(0)'s slot, which step 0 refills)
    dma_wait<NW_ALL + (AR2 ? 1 : 2) * KS * NA>();   // A(0) is in (waves that issue the partial last weight piece: that one too)
  } else {
#pragma unroll
    for (int d = 0; d < DW; d++) { issue_w(); issue_a(); }   // W(0), A(1); W(1), A(2); (W(2), A(3))
    dma_wait<DW * NLD>();                   // A(0) is in
  }
  Planes cur = read_a(0, cc.kt);
  ca_ring = 1;

  // this lane's bias values of the current column tile (column (lane & 31) of each 32-wide group), reloaded only when the column
  // tile changes: the epilogue then has no global load of its own (each one would be followed by a full vmcnt(0) drain of the
  // pieces prefetched for the next tile -- TN of them per tile, one after the other)
  float bias_r[TN];
  int bias_ct = -1;
  auto load_bias = [&](int ct) {
    if (ct == bias_ct) return;
    bias_ct = ct;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int gn = ct * BN + j * 32 + (lane & 31);
      float b = 0.f;
      if (gn < p.n && !(RGNN_DMA_ABL & 1024)) {
        const float* bp = (gn < p.w_split) ? p.bias1 : p.bias2;
        if (bp) b = bp[(gn < p.w_split) ? gn : gn - p.w_split];
      }
      bias_r[j] = b;
    }
  };
  load_bias(w_base % p.nt);
  int fresh = 0;                                    // k-steps left whose operands were requested before the last epilogue's stores
#if RGNN_DMA_TIMING
  unsigned long long tm_wait = 0, tm_bar = 0, tm_p1 = 0, tm_p2 = 0, tm_epi = 0, tm_steps = 0, tm_drain = 0, tm_own = 0;
  const unsigned long long tm_begin = TSTAMP();
#endif
  for (;;) {                                        // items of this work-group
    // accumulators: zeros, or -- last item of a stream-K range whose lower k-steps another work-group did -- its hand-over
    if (!psk && cc.j == w_count - 1 && w_kb_last > 0) load_partial(); else zero_acc();
    const int item = cc.item;
    if (p.nt > 1) load_bias(item % p.nt);
    const bool head_only = cc.j == 0 && cc.kend < nk;   // the item's upper k-range belongs to the next work-group
    // PING-PONG (r06).  A k-step of a wave is a LOAD half P1 (scalar bookkeeping, its two activation pieces, the fragment read and
    // split of the next step, the first weight-fragment reads) and a MATRIX half P2 (the MFMAs with the remaining fragment reads and
    // the weight pieces between them).  With one barrier per step the two waves of a SIMD (w and w + 4) run the same half at the
    // same time: the matrix pipe idles through both P1s and is then asked for both P2s.  With a second barrier between the halves
    // and waves 4-7 entering the item one barrier LATE (waves 0-3 leave it one barrier late), every barrier interval pairs one
    // wave's P1 with its SIMD partner's P2 (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory is the pairing that
    // nets).  The barrier protocol still holds: a wave waits for its own pieces of W(g) before ITS barrier B1(g), and both of
    // anybody's P2(g) lie behind everybody's B1(g).  ALL pieces of a step go out in its load half -- an LDS-DMA piece costs the
    // issuing wave ~110 cycles when eight waves issue at once (the CU accepts one per ~20 cycles), and in the matrix half those
    // cycles stop the wave's MFMA issue: measured with s_memtime, 450 of a matrix half's 1 200 cycles -- so the weight ring
    // has one stage more: waves 0-3 request W(g + 2) while waves 4-7 still multiply with W(g - 1); the slot they fill held W(g - 2).
    // Activation stages are private to a wave.
    if (PP && (wave >> 2) == 1) __builtin_amdgcn_s_barrier();
    for (;;) {                                      // k-steps
      // step g.  In flight: W(g), A(g+1) (requested during step g-2) and W(g+1), A(g+2) (step g-1).
      // (RGNN_DMA_POST_EPI_WAIT: the first DW steps behind an epilogue need pieces requested BEFORE its 16 TN stores, and the
      //  counter retires in order -- those stores may stay in flight, up to the counter's 63)
#if RGNN_DMA_TIMING
      const unsigned long long ts0 = TSTAMP();
#endif
      if (RGNN_DMA_POST_EPI_WAIT && fresh > 0) {
        fresh--;
        dma_wait<((DW - 1) * NLD + 16 * TN < 63) ? (DW - 1) * NLD + 16 * TN : 63>();
      } else if (!(RGNN_DMA_ABL & 64)) dma_wait<KS == 2 ? KS * NA : (DW - 1) * NLD>();   // this wave's pieces of W(g) and its A(g+1) have landed (double steps: only A(g+2) may be outstanding)
#if RGNN_DMA_TIMING
      const unsigned long long ts1 = TSTAMP();
#endif
      if (!(RGNN_DMA_ABL & 32)) __builtin_amdgcn_s_barrier();   // ... and everybody's W(g); nobody still reads the weight stage refilled next
#if RGNN_DMA_TIMING
      const unsigned long long ts2 = TSTAMP();
#endif
      req_begin();                                    // W(g+2), A(g+3) (its slot held A(g-1), split by this wave during step g-2)
      if (PP || !RGNN_DMA_SPREAD) {                   // (ping-pong: every piece goes out in the load half)
  #pragma unroll
        for (int i = 0; i < NLD; i++) req_piece(i);
      }
      // k-step of the NEXT compute step (the fragment split now): the next one of this item, or the first of the next item
      const int kt_nxt = (cc.kt + 1 < cc.kend) ? cc.kt + 1 : ((cc.j + 1 == w_count - 1) ? w_kb_last : 0);
      const int aslot_nxt = aff_seg ? (((cc.kt + 1 < cc.kend) ? cc.j : cc.j + 1) & 1) : 0;
      // (ping-pong: only the two LDS reads belong to the load half; the split's ~30 VALU instructions ride between the MFMAs)
      RawA raw_nxt;
      Planes nxt;
      if constexpr (KS == 1) {
        raw_nxt = load_a(ca_ring);
        if (!PP) nxt = (RGNN_DMA_ABL & 16) ? cur : split_a(raw_nxt, kt_nxt, aslot_nxt);   // split for the NEXT step: overlaps this step's MFMAs
        ca_ring = (ca_ring == DMA_A_RING - 1) ? 0 : ca_ring + 1;
      }
      const char* st = lds_w + cw_ring * W_STAGE + b_off;
      cw_ring = (cw_ring == DMA_W_RING - 1) ? 0 : cw_ring + 1;
      auto read_b = [&](int j, raw16x8 (&b)[NPL]) {
  #pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
          if (RGNN_DMA_ABL & 128) { b[pl] = cur.h; continue; }       // experiment: no weight-fragment reads from LDS
          b[pl] = *(const raw16x8*)(st + j * 32 * 32 + pl * W_PLANE);
        }
      };
      // one MFMA product of two operand terms, fp32 accumulate (bf16 or f16 words according to FMT)
      auto mm = [&](const raw16x8& a, const raw16x8& b, const f32x16& c) -> f32x16 {
        if constexpr (FMT == 1)
          return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
        else
          return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
      };
      auto mul = [&](f32x16& c, const Planes& a, const raw16x8 (&b)[NPL]) {   // smallest terms first: l h', h l', m m', m h', h m', h h'
        if (RGNN_DMA_ABL & 4) {
  #if defined(__HIP_DEVICE_COMPILE__)
          asm volatile("" :: "v"(b[0]), "v"(b[1]), "v"(b[NPL - 1]), "v"(a.h), "v"(a.m), "v"(a.l));
  #endif
          return;
        }
        if constexpr (FMT == 1) {
          c = mm(a.l, b[0], c);
          c = mm(a.h, b[1], c);
          c = mm(a.h, b[0], c);
        } else {
          c = mm(a.l, b[0], c);
          c = mm(a.h, b[NPL - 1], c);
          c = mm(a.m, b[1], c);
          c = mm(a.m, b[0], c);
          c = mm(a.h, b[1], c);
          c = mm(a.h, b[0], c);
        }
      };
      // Column groups go two at a time and their twelve MFMAs alternate between the two accumulators: whatever hipcc slots in
      // between (fragment reads, DMA pieces, the split of the next activation fragment, scalar bookkeeping) then never sits
      // between two MFMAs on the SAME accumulator -- that position costs ~43 cycles per instruction, any other ~6
      // (MI355X_MICROARCH.md, per-instruction constants).  Each accumulator still sees its k-steps and terms in the same order.
      auto mul2 = [&](f32x16& c0, f32x16& c1, const Planes& a, const raw16x8 (&b0)[NPL], const raw16x8 (&b1)[NPL]) {
        if (RGNN_DMA_ABL & 4) { mul(c0, a, b0); mul(c1, a, b1); return; }
        if constexpr (FMT == 1) {
          c0 = mm(a.l, b0[0], c0); c1 = mm(a.l, b1[0], c1);
          c0 = mm(a.h, b0[1], c0); c1 = mm(a.h, b1[1], c1);
          c0 = mm(a.h, b0[0], c0); c1 = mm(a.h, b1[0], c1);
        } else {
          c0 = mm(a.l, b0[0], c0); c1 = mm(a.l, b1[0], c1);
          c0 = mm(a.h, b0[NPL - 1], c0); c1 = mm(a.h, b1[NPL - 1], c1);
          c0 = mm(a.m, b0[1], c0); c1 = mm(a.m, b1[1], c1);
          c0 = mm(a.m, b0[0], c0); c1 = mm(a.m, b1[0], c1);
          c0 = mm(a.h, b0[1], c0); c1 = mm(a.h, b1[1], c1);
          c0 = mm(a.h, b0[0], c0); c1 = mm(a.h, b1[0], c1);
        }
      };
      constexpr int NP = (TN + 1) / 2;                  // units: pairs of column groups (the last one is a single group when TN is odd)
      auto read_unit = [&](int q, raw16x8 (&b)[2][NPL]) {
        read_b(2 * q, b[0]);
        if (2 * q + 1 < TN) read_b(2 * q + 1, b[1]);
      };
      raw16x8 bq[2][2][NPL];                             // fragments of a unit, double-buffered: unit q + 1 is read before unit q multiplies
      if constexpr (KS == 2) {
        // DOUBLE STEP: sub-stage h multiplies with `cur` while the fragment of the next sub-stage -- the second half of this step,
        // then the first half of the next step (maybe of the next item) -- is read and split.  The weight pieces of W(g + 1) go out
        // between the MFMA groups of the first half, the activation pieces of A(g + 3) in the second: their slot held A(g), whose
        // second half was read (and consumed by the split) in the first half.
        const char* const stw = st;
#pragma unroll
        for (int h = 0; h < KS; h++) {
          if (AR2 && h == 1 && !(RGNN_DMA_ABL & 64)) dma_wait<NW_ALL>();   // two stages: A(g + 1) has landed (behind it only W(g + 1))
          const RawA raw = (h == 0) ? load_a(ca_cur, 1) : load_a(ca_ring, 0);
          const int ks_n = (h == 0) ? cc.kt * KS + 1 : kt_nxt * KS;
          const int asl = (h == 0) ? (aff_seg ? (cc.j & 1) : 0) : aslot_nxt;
          nxt = (RGNN_DMA_ABL & 16) ? cur : split_a(raw, ks_n, asl);
          st = stw + h * W_SUB;
          read_unit(0, bq[0]);
          int piece = (h == 0) ? 0 : NW;                  // (compile-time after unrolling)
          const int piece_end = (h == 0) ? NW : NLD;
          const int per = ((h == 0 ? NW : KS * NA) + NP - 1) / NP;
#pragma unroll
          for (int q = 0; q < NP; q++) {
            const int cb = q & 1, nb = cb ^ 1;
            if (q + 1 < NP) read_unit(q + 1, bq[nb]);
#pragma unroll
            for (int u = 0; u < per; u++)
              if (piece < piece_end) req_piece(piece++);
            if (2 * q + 1 < TN) mul2(acc[0][2 * q], acc[0][2 * q + 1], cur, bq[cb][0], bq[cb][1]);
            else mul(acc[0][2 * q], cur, bq[cb][0]);
          }
#if defined(__HIP_DEVICE_COMPILE__)
          if (RGNN_DMA_PIN) asm volatile("" :: "v"(nxt.h), "v"(nxt.l));
#endif
          cur = nxt;
        }
        ca_cur = ca_ring;
        ca_ring = (ca_ring == DMA_A_RING - 1) ? 0 : ca_ring + 1;
        req_end();
        if (cursor_next(cc)) break;
        continue;
      }
      if (!PP) read_unit(0, bq[0]);
      bool item_done = false;
      if (PP) {                                         // ---- end of the load half
#if RGNN_DMA_TIMING
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "v"(raw_nxt.x0.x), "v"(raw_nxt.x0.w), "v"(raw_nxt.x1.x), "v"(raw_nxt.x1.w));
#endif
        tm_own += TSTAMP() - ts2;
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_unit(0, bq[0]);                            // (W(g): waves 4-7 wait for their pieces of it only in front of THIS barrier)
        nxt = (RGNN_DMA_ABL & 16) ? cur : split_a(raw_nxt, kt_nxt, aslot_nxt);
        req_end();
        item_done = cursor_next(cc);
      }
#if RGNN_DMA_TIMING
      const unsigned long long ts3 = TSTAMP();
#endif
      int piece = 0;                                    // (compile-time after unrolling)
  #pragma unroll
      for (int q = 0; q < NP; q++) {
        const int cb = q & 1, nb = cb ^ 1;
        if (q + 1 < NP) read_unit(q + 1, bq[nb]);
        if (RGNN_DMA_SPREAD && !PP) {
  #pragma unroll
          for (int u = 0; u < (NLD + NP - 1) / NP; u++)
            if (piece < NLD) req_piece(piece++);
        }
        if (2 * q + 1 < TN) mul2(acc[0][2 * q], acc[0][2 * q + 1], cur, bq[cb][0], bq[cb][1]);
        else mul(acc[0][2 * q], cur, bq[cb][0]);
      }
#if defined(__HIP_DEVICE_COMPILE__)
      // (keeps the split of the next step's activation fragment inside the MFMA block: left alone, hipcc sinks its ~50 VALU
      // instructions into the loop latch, behind all MFMAs, where both waves of a SIMD run them with the matrix pipe idle)
      if (RGNN_DMA_PIN) asm volatile("" :: "v"(nxt.h), "v"(nxt.m), "v"(nxt.l));
#endif
      if (!PP) req_end();
#if RGNN_DMA_TIMING
      {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < TN; j++) asm volatile("" : "+v"(acc[0][j]));
#endif
        const unsigned long long ts4 = TSTAMP();
        tm_wait += ts1 - ts0; tm_bar += ts2 - ts1; tm_p1 += ts3 - ts2; tm_p2 += ts4 - ts3; tm_steps++;
      }
#endif
      cur = nxt;
      if (PP ? item_done : cursor_next(cc)) break;  // the item's (sub-)range is complete
    }
    if (PP && (wave >> 2) == 0) __builtin_amdgcn_s_barrier();
#if RGNN_DMA_TIMING
    unsigned long long te0_keep = 0;
#endif
    if (head_only) {
      store_partial();
    } else {
      if (psk) combine();
      else fresh = DW;
      const int panel = xcd + 8 * (item / p.nt);
#if RGNN_DMA_TIMING
      te0_keep = TSTAMP();
#endif
      if (!(RGNN_DMA_ABL & 1))
        amax = direct_epilogue<BN, WV, 1, 1, TN, DMA_BM, IDX, RGNN_DMA_TRACK != 0>(p, acc, (int64_t)panel * DMA_BM, (item % p.nt) * BN, panel, M,
                                                                                  stat_lds, row_tab, bias_r, FMT == 1 ? out_mul : 1.f, amax);
#if defined(__HIP_DEVICE_COMPILE__)
      else {
#pragma unroll
        for (int j = 0; j < TN; j++) asm volatile("" :: "v"(acc[0][j]));
      }
#endif
    }
#if RGNN_DMA_TIMING
    if (!head_only) {
      const unsigned long long te1 = TSTAMP();
      tm_epi += te1 - te0_keep;
      if (RGNN_DMA_TIMING == 2) {                          // (the drain of the stores -- and of the pieces prefetched for the next item)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tm_drain += TSTAMP() - te1;
      }
    }
#endif
    if (cc.j >= w_count) break;
    if (aff_seg) {                                  // item cc.j - 1 is finished: its table slot takes the item after the next one
      __syncthreads();
      if (cc.j + 1 < w_count) load_aff(cc.item + w_stride, (cc.j + 1) & 1);
      __syncthreads();
    }
  }
#if RGNN_DMA_TIMING
  if (lane == 0) {
    unsigned long long* o = g_dma_t + ((blockIdx.x & 255) * 8 + (wave & 7)) * 8;
    o[0] = tm_wait; o[1] = tm_bar; o[2] = tm_p1; o[3] = tm_p2; o[4] = tm_epi; o[5] = tm_steps; o[6] = TSTAMP() - tm_begin; o[7] = RGNN_DMA_PP ? tm_own : tm_drain;
  }
#endif
  dma_wait<0>();                                    // (killed pieces of the exhausted streams)
  if (p.out_absmax) {                               // one atomic per work-group, into the slot of this work-group
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __syncthreads();                                // (stat_lds: nobody reads the last epilogue's partials any more)
    if (lane == 0) stat_lds[wave] = amax;
    __syncthreads();
    if (t == 0) {
      float m = stat_lds[0];
#pragma unroll
      for (int w = 1; w < WV; w++) m = fmaxf(m, stat_lds[w]);
      bound_raise(p.out_absmax, blockIdx.x, m);
    }
  }
}

template <int TN, bool IDX, int FMT, int WV = 8>
void launch_dma(LinParams p, hipStream_t s) {
  constexpr int BN = 32 * TN;
  constexpr int NPL = FMT ? 2 : 3;
  constexpr int DMA_BM = 32 * WV, DMA_THREADS = 64 * WV;
  const size_t lds = (size_t)dma_lds_bytes(BN, NPL, WV) + (p.a1_aff ? (size_t)(p.a1_aff_panel ? 2 : 1) * RGNN_AFFINE_ROWS * 4 * p.k1 : 0);
  p.nt = (p.n + BN - 1) / BN;
  p.mt = (int)((p.m + DMA_BM - 1) / DMA_BM);
  const int64_t tiles = (int64_t)p.mt * p.nt;
  int64_t grid = 256 * (8 / WV);                   // one 8-wave work-group per CU, or two of four waves
  if (grid > tiles && (TN > 4 || p.sk_ws == nullptr || p.no_split_k || 2 * tiles > grid)) grid = tiles;   // (else: parallel split-K)
  grid = (grid + 7) / 8 * 8;
  static RgnnOncePerDevice attr_once;
  if (attr_once.first()) {                                  // (room for the optional scale / shift table of the A1 operand: up to 1 024 columns)
    const int most = dma_lds_bytes(BN, NPL, WV) + 12288 < 160 * 1024 ? dma_lds_bytes(BN, NPL, WV) + 12288 : 160 * 1024;
    hipFuncSetAttribute((const void*)k_linear_dma<TN, IDX, FMT, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
  }
  hipLaunchKernelGGL((k_linear_dma<TN, IDX, FMT, WV>), dim3((unsigned)grid), dim3(DMA_THREADS), lds, s, p);
}

}  // namespace

// Column-tile width (in 32-column MFMA tiles).  Enough row panels to fill the chip: the width that pads the fewest columns,
// ties to the wider tile.  Few row panels (one frame: M = 3 000 -> 12 panels): narrower tiles, so that more work-groups share
// the layer -- a k-step costs a work-group about 0.35 us + 0.2 us per 32 columns (two waves per SIMD share the matrix pipe),
// and the launch takes ceil(tiles / 256) rounds of them.  The accumulation order of an output element does not depend on
// the tile width, so the result is the same bit for bit either way.
// (npl = operand planes: 3 = bf16x3, 2 = f16x2.  The 256-column tile of the bf16x3 form no longer fits the LDS since the
//  statistics exchange carries a pivot per column (r04: 164 896 bytes): that form stops at 224 columns.)
static int dma_pick_tn(int n, int64_t m, int npl) {
  const int top = npl == 3 ? 7 : 8;
  const char* e = RGNN_ENV("RGNN_DMA_TN");
  if (e) { const int v = atoi(e); if (v >= 2 && v <= top) return v; }
  int best = 2;
  if (n > 64 && n <= 96) best = 3;
  if (n > 96) {
    best = top;
    int best_pad = (n + 32 * top - 1) / (32 * top) * (32 * top);
    for (int tn = top - 1; tn >= 3; tn--) {        // (n = 272: three tiles of 96 pad 288 and measure 61 us against 72 us for two of 160)
      const int w = 32 * tn, pad = (n + w - 1) / w * w;
      if (pad < best_pad) { best_pad = pad; best = tn; }
    }
  }
  const int64_t mt = (m + DMA_BM_MAX - 1) / DMA_BM_MAX;
  if (mt * ((n + 32 * best - 1) / (32 * best)) >= 192 || RGNN_ENV("RGNN_DMA_NO_SMALL_M")) return best;
  double best_t = 1e30;
  int pick = best;
  for (int tn = 2; tn <= top; tn++) {
    const int64_t tiles = mt * ((n + 32 * tn - 1) / (32 * tn));
    const double t = (double)((tiles + 255) / 256) * (0.35 + 0.2 * tn);
    if (t < best_t - 1e-9) { best_t = t; pick = tn; }
  }
  return pick;
}

// Two 4-wave work-groups per CU (128-row tiles) instead of one of eight waves?  They drift apart, so one of them loads and
// multiplies while the other stores its tile -- the phases that ADD in the 8-wave form (profiles/r03_x3_bench_f16x2_ablations.txt)
// -- at the price of every weight tile being fetched into LDS twice per CU.
static bool dma_four_waves(const LinParams& p, int tn, int waves_env) {
  if (waves_env == 4) return true;
  if (waves_env == 8) return false;
  (void)tn;
  return false;
}

// LDS bytes the kernel instance for (n, m) needs without the optional A1 scale / shift table (linear.hip: does the table fit?)
// (f16_form: the launch carries f16 weight planes and bounds -- the f16x2 form's weight stages are smaller than the bf16x3 form's)
int rgnn_linear_dma_lds_bytes(int n, int64_t m, int f16_form) {
  const int npl = f16_form ? 2 : 3;
  return dma_lds_bytes(32 * dma_pick_tn(n, m, npl), npl);
}

// Called by rgnn_linear_fwd (linear.hip) once it has decided that the layer qualifies (bf16 planes given, buffer-descriptor
// operands, n > 64, K and k1 multiples of 16, no residual / accumulate / gather_only).  `subset`: row_index launch.
int rgnn_linear_dma_launch(const void* params, int subset, hipStream_t s) {
  const LinParams& p = *(const LinParams*)params;
  const int tn = dma_pick_tn(p.n, p.m, p.fmt == 1 ? 2 : 3);
  // two 4-wave work-groups per CU instead of one of eight (f16x2 form): RGNN_DMA_WAVES = 4 forces it, 8 forbids it
  const char* waves_e = RGNN_ENV("RGNN_DMA_WAVES");          // (read per call: tools/x3_bench switches it between variants)
  const int waves_env = waves_e ? atoi(waves_e) : 0;
  const bool four = p.fmt == 1 && p.a1_aff_panel == nullptr && dma_four_waves(p, tn, waves_env);   // (the panel map counts 256-row tiles)
#define RGNN_DMA(TN)                                                                                     \
  case TN:                                                                                               \
    if (four) { if (subset) launch_dma<TN, true, 1, 4>(p, s); else launch_dma<TN, false, 1, 4>(p, s); }  \
    else if (p.fmt == 1) { if (subset) launch_dma<TN, true, 1>(p, s); else launch_dma<TN, false, 1>(p, s); }  \
    else { if (subset) launch_dma<TN, true, 0>(p, s); else launch_dma<TN, false, 0>(p, s); }             \
    break
  switch (tn) {
    RGNN_DMA(2); RGNN_DMA(3); RGNN_DMA(4); RGNN_DMA(5); RGNN_DMA(6); RGNN_DMA(7);
    default:
      if (four) { if (subset) launch_dma<8, true, 1, 4>(p, s); else launch_dma<8, false, 1, 4>(p, s); }
      else if (p.fmt == 1) { if (subset) launch_dma<8, true, 1>(p, s); else launch_dma<8, false, 1>(p, s); }
      else { if (subset) launch_dma<8, true, 0>(p, s); else launch_dma<8, false, 0>(p, s); }
  }
#undef RGNN_DMA
  return 0;
}
