// Collation of graphs that already live in HBM into one PyG-style batch (the reference does this on the
// host: torch_geometric DataLoader -> Batch.from_data_list, utils/data_handling.py:30, then one H2D copy per batch,
// postprocessor/inference.py:57).  The whole processed dataset is kept resident as ONE set of concatenated tensors (288 GB of
// HBM hold far more than a RadarScenes split); a batch is a list of graph ids, and building it is a segmented copy:
//
//     rows   : out[dst_ptr[s] + r, :] = src[src_row[s] + r, :]           for every selected graph s, r < rows of s
//     edges  : out[:, dst_eptr[s] + e] = src_ei[:, src_edge[s] + e] + node_shift[s]     (PyG: edge_index += cumulative nodes)
//     batch  : batch[dst_ptr[s] + r] = s
//
// Pure HBM-bound data movement: one thread per 4-byte word (rows) / per edge (both index rows), the owning segment found by
// a branch-free binary search over the B + 1 destination offsets (they sit in L1/L2).  Bit-exact by construction.
#include "common.h"
#include <string.h>

namespace {

// largest s in [0, n_seg) with ptr[s] <= i   (ptr non-decreasing, ptr[0] = 0, i < ptr[n_seg]); empty segments are skipped
__device__ __forceinline__ int find_segment(const int64_t* __restrict__ ptr, int n_seg, int64_t i) {
  int lo = 0, hi = n_seg;                       // invariant: ptr[lo] <= i < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    const bool right = ptr[mid] <= i;
    lo = right ? mid : lo;
    hi = right ? hi : mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_collate_rows(const uint32_t* __restrict__ src, int64_t ld_src, int width,
                                                     const int64_t* __restrict__ seg_src_row,
                                                     const int64_t* __restrict__ seg_dst_ptr, int n_seg, int64_t n_rows,
                                                     uint32_t* __restrict__ out, int64_t ld_out,
                                                     int64_t* __restrict__ batch_out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * width) return;
  const int64_t r = idx / width;
  const int c = (int)(idx - r * width);
  const int s = find_segment(seg_dst_ptr, n_seg, r);
  const int64_t sr = seg_src_row[s] + (r - seg_dst_ptr[s]);
  out[r * ld_out + c] = src[sr * ld_src + c];
  if (batch_out && c == 0) batch_out[r] = s;
}

__global__ __launch_bounds__(256) void k_collate_edges(const int64_t* __restrict__ src_ei, int64_t ld_src,
                                                      const int64_t* __restrict__ seg_src_edge,
                                                      const int64_t* __restrict__ seg_dst_eptr,
                                                      const int64_t* __restrict__ seg_node_shift, int n_seg,
                                                      int64_t n_edges, int64_t* __restrict__ out, int64_t ld_out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int s = find_segment(seg_dst_eptr, n_seg, e);
  const int64_t se = seg_src_edge[s] + (e - seg_dst_eptr[s]);
  const int64_t shift = seg_node_shift[s];
  out[e] = src_ei[se] + shift;
  out[ld_out + e] = src_ei[ld_src + se] + shift;
}

}  // namespace

extern "C" int rgnn_collate_rows(const void* src, int64_t ld_src, int32_t width, const int64_t* seg_src_row,
                                 const int64_t* seg_dst_ptr, int32_t n_seg, int64_t n_rows, void* out, int64_t ld_out,
                                 int64_t* batch_out, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_rows >= 0 && width >= 1 && n_seg >= 0, "bad sizes");
  if (n_rows == 0) return RGNN_OK;
  RGNN_CHECK_ARG(n_seg > 0 && seg_src_row && seg_dst_ptr && src && out, "null pointers");
  hipLaunchKernelGGL(k_collate_rows, dim3(rgnn_blocks(n_rows * width, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)src, ld_src, width, seg_src_row, seg_dst_ptr, n_seg, n_rows, (uint32_t*)out, ld_out,
                     batch_out);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_collate_edges(const int64_t* src_edge_index, int64_t ld_src, const int64_t* seg_src_edge,
                                  const int64_t* seg_dst_eptr, const int64_t* seg_node_shift, int32_t n_seg, int64_t n_edges,
                                  int64_t* out, int64_t ld_out, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n_edges >= 0 && n_seg >= 0, "bad sizes");
  if (n_edges == 0) return RGNN_OK;
  RGNN_CHECK_ARG(n_seg > 0 && src_edge_index && seg_src_edge && seg_dst_eptr && seg_node_shift && out, "null pointers");
  hipLaunchKernelGGL(k_collate_edges, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, (hipStream_t)stream, src_edge_index,
                     ld_src, seg_src_edge, seg_dst_eptr, seg_node_shift, n_seg, n_edges, out, ld_out);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// Host side of a streamed batch (frames.FrameStreamer; the reference collates on the host too: utils/data_handling.py:30): the
// point arrays of n_frames frames, wherever they lie in host memory, laid back to back into ONE block -- X [n, 2], V [n, 2],
// rcs [n], timestamp [n] (float64), frame_ptr [n_frames + 1] (int64), in this order, n = sum of n_points -- so that a batch goes up
// in one copy.  No device work; called without the interpreter lock (ctypes), so a loader thread's ~10 MB of copies per batch do not
// stand between the launching thread and the interpreter (r05: 256 numpy copies per batch, each a lock hand-over, cost the
// launching thread 0.2 ms per batch).  addr: [n_frames][4] pointers {X, V, rcs, timestamp} (contiguous float64 arrays).
extern "C" int rgnn_stage_frames(int64_t n_frames, const void* const* addr, const int64_t* n_points, void* block, int64_t block_bytes) {
  RGNN_CHECK_ARG(n_frames >= 0 && block && (n_frames == 0 || (addr && n_points)), "bad arguments");
  int64_t n = 0;
  for (int64_t f = 0; f < n_frames; f++) {
    RGNN_CHECK_ARG(n_points[f] >= 0, "negative frame size");
    n += n_points[f];
  }
  RGNN_CHECK_ARG(block_bytes >= 48 * n + 8 * (n_frames + 1), "block too small");
  char* base = (char*)block;
  double* parts[4] = {(double*)base, (double*)(base + 16 * n), (double*)(base + 32 * n), (double*)(base + 40 * n)};
  int64_t* ptr = (int64_t*)(base + 48 * n);
  const int width[4] = {2, 2, 1, 1};
  int64_t at = 0;
  for (int64_t f = 0; f < n_frames; f++) {
    ptr[f] = at;
    const int64_t m = n_points[f];
    for (int a = 0; a < 4; a++) {
      if (m == 0) continue;
      RGNN_CHECK_ARG(addr[4 * f + a] != nullptr, "null frame array");
      memcpy(parts[a] + at * width[a], addr[4 * f + a], (size_t)m * width[a] * sizeof(double));
    }
    at += m;
  }
  ptr[n_frames] = at;
  return RGNN_OK;
}
