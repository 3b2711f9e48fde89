// Grid-hash neighbour search for batches of radar frames (gfx950).
//
// Replaces sklearn KDTree64.query / query_radius + scipy toarray()/nonzero() behind
// graph_constructor/graph.py:52-82, and the networkx degree pass of graph.py:93-96.
//
// Layout in HBM (one opaque workspace, see GridView):
//   frames[B]          per-frame grid: origin, cell edge h, gx*gy cells, first global cell id
//   cell_count[C+1]    points per cell (C = 2n + 64B is a static upper bound -> no host sync)
//   cell_start[C+1]    exclusive scan of cell_count
//   sorted_idx[n]      original (global) point row of the p-th point in cell order
//   sorted_frame[n]    its frame
//   sorted_cell[n]     its global cell id
//   sorted_pos[n*dim]  its coordinates (AoS, dim doubles) -> candidates of a cell are contiguous
// Queries are processed in cell order, so the 64 lanes of a wave walk the same 3 cell rows and the
// candidate loads are wave-coalesced / L1-resident.
//
// All distance arithmetic is float64 without FMA contraction (this file is compiled with
// -ffp-contract=off): d2 = ((0 + t0*t0) + t1*t1) ... exactly as the KD-tree's rdist.
#include "common.h"
#include <math.h>

namespace {

struct FrameGrid {
  double x0, y0, h;
  int32_t gx, gy;
  int32_t cell_base;
  int32_t n_pts;
  int32_t tx, pad_;  // tiles of 8 x 8 cells per grid row (cells are numbered tile by tile, see cell_id)
};

// Cells are numbered tile by tile (8 x 8 cells per tile, tiles row-major, cells row-major inside a tile) instead of
// plain row-major: the cell order of the points is then spatially compact in BOTH directions, so the 35 points of a
// cluster sit next to each other in `sorted_idx` -- which is also the order the message-passing kernel visits the
// targets in (rgnn_grid_cell_order): their gathered source rows stay L1/L2 resident.
__device__ __forceinline__ int cell_id(const FrameGrid& g, int cx, int cy) {
  return g.cell_base + (((cy >> 3) * g.tx + (cx >> 3)) << 6) + ((cy & 7) << 3) + (cx & 7);
}
__device__ __forceinline__ void cell_xy(const FrameGrid& g, int cell, int& cx, int& cy) {
  const int local = cell - g.cell_base, tile = local >> 6, in = local & 63;
  cy = (tile / g.tx) * 8 + (in >> 3);
  cx = (tile % g.tx) * 8 + (in & 7);
}

struct GridView {
  FrameGrid* frames;
  int32_t* cell_count;
  int32_t* cell_start;
  int32_t* sorted_idx;
  int32_t* sorted_frame;
  int32_t* sorted_cell;
  double* sorted_pos;
  int32_t* point_cell;  // cell of original point i (binning pass)
  int32_t* point_frame;
  int32_t* nbr_cache;   // radius search: the first RADIUS_CACHE neighbours the count pass found for point i
  int32_t* point_rank;  // position of original point i in cell order (inverse of sorted_idx)
  void* scan_tmp;
  int64_t n_cells;
  int64_t total_bytes;
};

#ifndef RGNN_RADIUS_ROUND
#define RGNN_RADIUS_ROUND 4
#endif
constexpr int RADIUS_ROUND = RGNN_RADIUS_ROUND;
#ifndef RGNN_RADIUS_CACHE
#define RGNN_RADIUS_CACHE 48   // (a 35-point cluster's rows hold 34 neighbours: with 32 slots all of them were searched twice; fill pass 27 -> 13 us)
#endif
constexpr int RADIUS_CACHE = RGNN_RADIUS_CACHE;   // (radar frames at r = 1 m: 4 neighbours on average, clusters ~34; denser rows are searched again)
constexpr int CELLS_PER_POINT = 2;
constexpr int CELLS_PER_FRAME = 64;

inline int64_t grid_cells(int64_t n, int64_t n_frames) { return CELLS_PER_POINT * n + CELLS_PER_FRAME * n_frames; }

GridView make_view(void* ws, int64_t n, int64_t B, int dim) {
  GridView v;
  char* p = (char*)ws;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += rgnn_align_up(bytes, 256);
    return r;
  };
  v.n_cells = grid_cells(n, B);
  v.frames = (FrameGrid*)take(sizeof(FrameGrid) * (B > 0 ? B : 1));
  v.cell_count = (int32_t*)take(4 * (v.n_cells + 1));
  v.cell_start = (int32_t*)take(4 * (v.n_cells + 1));
  v.sorted_idx = (int32_t*)take(4 * n);
  v.sorted_frame = (int32_t*)take(4 * n);
  v.sorted_cell = (int32_t*)take(4 * n);
  v.sorted_pos = (double*)take(8 * n * dim);
  v.point_cell = (int32_t*)take(4 * n);
  v.point_frame = (int32_t*)take(4 * n);
  v.nbr_cache = (int32_t*)take(4 * n * RADIUS_CACHE);
  v.point_rank = (int32_t*)take(4 * n);
  v.scan_tmp = take(rgnn_scan_tmp_bytes(v.n_cells + 1));
  v.total_bytes = p - (char*)ws;
  return v;
}

// ------------------------------------------------------------------------------------------------
// per-frame bounding box and grid geometry: one block per frame
// ------------------------------------------------------------------------------------------------
constexpr int FG_THREADS = 1024;   // (one block per frame: a 100 000-point frame took 198 us with 256 threads)
__global__ __launch_bounds__(FG_THREADS) void k_frame_grid(const double* __restrict__ X, int dim,
                                                   const int64_t* __restrict__ frame_ptr, FrameGrid* __restrict__ frames,
                                                   double cell_size, double pts_per_cell, int32_t* __restrict__ cell_count,
                                                   int64_t n_cells) {
  const int f = blockIdx.x;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  {  // the frame's cell counters start at zero (its share of the static cell table; frame 0 also clears the end marker)
    const int64_t c0 = CELLS_PER_POINT * beg + CELLS_PER_FRAME * (int64_t)f;
    const int64_t c1 = CELLS_PER_POINT * end + CELLS_PER_FRAME * (int64_t)(f + 1);
    for (int64_t c = c0 + threadIdx.x; c < c1; c += blockDim.x) cell_count[c] = 0;
    if (f == 0 && threadIdx.x == 0) cell_count[n_cells] = 0;
  }
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  // (four points per thread and trip, their loads in flight together: one block walks the whole frame, and a dependent load per trip
  //  made the 100 000-point cloud's box 98 memory latencies long -- 64 us)
  for (int64_t i = beg + threadIdx.x; i < end; i += 4 * (int64_t)blockDim.x) {
    double x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t j = i + u * (int64_t)blockDim.x;
      const int64_t jj = j < end ? j : i;                            // (past the end: the trip's first point again)
      x[u] = X[jj * dim]; y[u] = X[jj * dim + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      xmin = fmin(xmin, x[u]); xmax = fmax(xmax, x[u]);
      ymin = fmin(ymin, y[u]); ymax = fmax(ymax, y[u]);
    }
  }
  __shared__ double red[4][FG_THREADS];
  red[0][threadIdx.x] = xmin; red[1][threadIdx.x] = xmax; red[2][threadIdx.x] = ymin; red[3][threadIdx.x] = ymax;
  __syncthreads();
  for (int s = FG_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] = fmin(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + s]);
      red[2][threadIdx.x] = fmin(red[2][threadIdx.x], red[2][threadIdx.x + s]);
      red[3][threadIdx.x] = fmax(red[3][threadIdx.x], red[3][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    FrameGrid g;
    const int64_t nf = end - beg;
    g.n_pts = (int32_t)nf;
    g.cell_base = (int32_t)(CELLS_PER_POINT * beg + CELLS_PER_FRAME * (int64_t)f);
    const int64_t cap = CELLS_PER_POINT * nf + CELLS_PER_FRAME;
    if (nf == 0) {
      g.x0 = 0; g.y0 = 0; g.h = 1; g.gx = 1; g.gy = 1;
    } else {
      xmin = red[0][0]; xmax = red[1][0]; ymin = red[2][0]; ymax = red[3][0];
      const double ex = xmax - xmin, ey = ymax - ymin;
      double h;
      if (cell_size > 0) {
        // radius mode: h slightly larger than r so that |dx| <= r implies |cell difference| <= 1 even
        // after the rounding of (x - x0) / h
        h = cell_size * (1.0 + 9.5367431640625e-07);
      } else {
        // kNN mode: about pts_per_cell points per cell over the bounding box
        const double area = ex * ey;
        if (area > 0) h = sqrt(area * pts_per_cell / (double)nf);
        else if (ex + ey > 0) h = (ex + ey) * pts_per_cell / (double)nf;
        else h = 1.0;
        // cells that are much finer than the box are pointless
        const double hmin = fmax(ex, ey) * 1e-6;
        if (h < hmin) h = hmin;
        if (!(h > 0)) h = 1.0;
      }
      int64_t gx, gy;
      for (;;) {
        gx = (int64_t)floor(ex / h) + 1;
        gy = (int64_t)floor(ey / h) + 1;
        if (((gx + 7) / 8) * ((gy + 7) / 8) * 64 <= cap) break;  // cells incl. the padding of partial 8x8 tiles
        h *= 1.5;
      }
      g.x0 = xmin; g.y0 = ymin; g.h = h; g.gx = (int32_t)gx; g.gy = (int32_t)gy;
    }
    g.tx = (g.gx + 7) / 8;
    g.pad_ = 0;
    frames[f] = g;
  }
}

__device__ __forceinline__ int find_frame(const int64_t* __restrict__ frame_ptr, int n_frames, int64_t i) {
  int lo = 0, hi = n_frames;  // frame f has frame_ptr[f] <= i < frame_ptr[f+1]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (frame_ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_bin_count(const double* __restrict__ X, int dim, int64_t n,
                                                  const int64_t* __restrict__ frame_ptr, int n_frames,
                                                  const FrameGrid* __restrict__ frames, int32_t* __restrict__ point_cell,
                                                  int32_t* __restrict__ point_frame, int32_t* __restrict__ cell_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = find_frame(frame_ptr, n_frames, i);
  const FrameGrid g = frames[f];
  int cx = (int)floor((X[i * dim] - g.x0) / g.h);
  int cy = (int)floor((X[i * dim + 1] - g.y0) / g.h);
  cx = min(max(cx, 0), g.gx - 1);
  cy = min(max(cy, 0), g.gy - 1);
  const int c = cell_id(g, cx, cy);
  point_cell[i] = c;
  point_frame[i] = f;
  atomicAdd(&cell_count[c], 1);
}

// Points into cell order, general path, in two launches: the atomics hand out the places of a cell in ARRIVAL order (not
// reproducible), so they only fill a scratch list (sorted_cell, overwritten by the second launch); the second launch gives every
// point its cell's start plus the number of points of the cell with a smaller index (see k_grid_frame).
constexpr int GRID_ORDERED_CELL_MAX = 2048;   // cells with more points keep the atomics' arrival order: ranking by counting is quadratic
                                              // in the cell size (10^5 coincident points would be 10^10 reads); such a cell is a degenerate
                                              // input, and its order is then the one thing in the path that is not reproducible
__global__ __launch_bounds__(256) void k_bin_scatter(int64_t n, const int32_t* __restrict__ point_cell, const int32_t* __restrict__ cell_start,
                                                    int32_t* __restrict__ cell_count, int32_t* __restrict__ scratch,
                                                    int32_t* __restrict__ point_rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = point_cell[i];
  const int at = cell_start[c] + atomicSub(&cell_count[c], 1) - 1;
  scratch[at] = (int32_t)i;
  point_rank[i] = at;                                  // (arrival place: what a crowded cell keeps, k_bin_place)
}

template <int DIM>
__global__ __launch_bounds__(256) void k_bin_place(const double* __restrict__ X, int64_t n,
                                                  const int32_t* __restrict__ point_cell,
                                                  const int32_t* __restrict__ point_frame,
                                                  const int32_t* __restrict__ cell_start, const int32_t* __restrict__ scratch,
                                                  int32_t* __restrict__ sorted_idx, int32_t* __restrict__ sorted_frame,
                                                  double* __restrict__ sorted_pos, int32_t* __restrict__ point_rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = point_cell[i];
  const int s0 = cell_start[c], s1 = cell_start[c + 1];
  int p = s0;
  if (s1 - s0 > GRID_ORDERED_CELL_MAX) p = point_rank[i];
  else for (int j = s0; j < s1; j++) p += scratch[j] < (int32_t)i ? 1 : 0;
  sorted_idx[p] = (int32_t)i;
  point_rank[i] = p;
  sorted_frame[p] = point_frame[i];
#pragma unroll
  for (int d = 0; d < DIM; d++) sorted_pos[(int64_t)p * DIM + d] = X[i * DIM + d];
}

__global__ __launch_bounds__(256) void k_bin_cells(int64_t n, const int32_t* __restrict__ point_cell, const int32_t* __restrict__ point_rank,
                                                  int32_t* __restrict__ sorted_cell) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sorted_cell[point_rank[i]] = point_cell[i];
}

// ------------------------------------------------------------------------------------------------
// The whole binning of a frame in ONE block (r03): bounding box -> grid geometry -> cell of every point -> cell histogram ->
// exclusive scan -> points laid out in cell order.  The cells of a frame are a contiguous slice of the cell table and its
// points a contiguous slice of the point arrays, so the scan is local: cell_start = frame_ptr[f] + local prefix -- no
// device-wide scan, no second and third pass over all points (five launches before: k_frame_grid, k_bin_count, two scan
// kernels, k_bin_fill; 48 us on the C2 batch, most of it launch latency of a dependent chain).  The histogram / cursors live
// in LDS when the frame's cells fit (GF_LDS_CELLS), else in the global cell table.  Arithmetic (grid geometry, cell of a
// point) is the code of k_frame_grid / k_bin_count: same cells, hence the same candidate sets; the points INSIDE a cell are
// in ascending index order on either path (r05: reproducible from run to run).
// ------------------------------------------------------------------------------------------------
constexpr int GF_THREADS = 1024;
constexpr int GF_LDS_CELLS = 32 * 1024;            // 128 KB of LDS counters: frames of up to ~16 000 points
__device__ __forceinline__ double gf_shfl_xor(double v, int o) {
  int2 t = __builtin_bit_cast(int2, v);
  t.x = __shfl_xor(t.x, o, 64); t.y = __shfl_xor(t.y, o, 64);
  return __builtin_bit_cast(double, t);
}
template <int DIM>
__device__ __forceinline__ void gf_block(const double* __restrict__ X, const int64_t* __restrict__ frame_ptr,
                                                         int n_frames, FrameGrid* __restrict__ frames, double cell_size,
                                                         double pts_per_cell, int32_t* __restrict__ cell_count,
                                                         int32_t* __restrict__ cell_start, int64_t n_cells, int lds_cells,
                                                         int32_t* __restrict__ sorted_idx, int32_t* __restrict__ sorted_frame,
                                                         int32_t* __restrict__ sorted_cell, double* __restrict__ sorted_pos,
                                                         int32_t* __restrict__ point_cell, int32_t* __restrict__ point_frame,
                                                         int32_t* __restrict__ point_rank) {
  extern __shared__ int32_t gf_cnt[];
  __shared__ double red[4][GF_THREADS / 64];
  __shared__ FrameGrid sg;
  __shared__ int wsum[GF_THREADS / 64];
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  const int64_t nf = end - beg;
  const int64_t c0 = CELLS_PER_POINT * beg + CELLS_PER_FRAME * (int64_t)f;
  const int64_t cap = CELLS_PER_POINT * nf + CELLS_PER_FRAME;
  // ---- bounding box
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  for (int64_t i = beg + t; i < end; i += GF_THREADS) {
    const double x = X[i * DIM], y = X[i * DIM + 1];
    xmin = fmin(xmin, x); xmax = fmax(xmax, x);
    ymin = fmin(ymin, y); ymax = fmax(ymax, y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = fmin(xmin, gf_shfl_xor(xmin, o)); xmax = fmax(xmax, gf_shfl_xor(xmax, o));
    ymin = fmin(ymin, gf_shfl_xor(ymin, o)); ymax = fmax(ymax, gf_shfl_xor(ymax, o));
  }
  if (lane == 0) { red[0][w] = xmin; red[1][w] = xmax; red[2][w] = ymin; red[3][w] = ymax; }
  __syncthreads();
  if (t == 0) {
    for (int i = 1; i < GF_THREADS / 64; i++) {
      xmin = fmin(xmin, red[0][i]); xmax = fmax(xmax, red[1][i]); ymin = fmin(ymin, red[2][i]); ymax = fmax(ymax, red[3][i]);
    }
    FrameGrid g;
    g.n_pts = (int32_t)nf;
    g.cell_base = (int32_t)c0;
    if (nf == 0) {
      g.x0 = 0; g.y0 = 0; g.h = 1; g.gx = 1; g.gy = 1;
    } else {                                            // (the geometry rules of k_frame_grid, word for word)
      const double ex = xmax - xmin, ey = ymax - ymin;
      double h;
      if (cell_size > 0) {
        h = cell_size * (1.0 + 9.5367431640625e-07);
      } else {
        const double area = ex * ey;
        if (area > 0) h = sqrt(area * pts_per_cell / (double)nf);
        else if (ex + ey > 0) h = (ex + ey) * pts_per_cell / (double)nf;
        else h = 1.0;
        const double hmin = fmax(ex, ey) * 1e-6;
        if (h < hmin) h = hmin;
        if (!(h > 0)) h = 1.0;
      }
      int64_t gx, gy;
      for (;;) {
        gx = (int64_t)floor(ex / h) + 1;
        gy = (int64_t)floor(ey / h) + 1;
        if (((gx + 7) / 8) * ((gy + 7) / 8) * 64 <= cap) break;
        h *= 1.5;
      }
      g.x0 = xmin; g.y0 = ymin; g.h = h; g.gx = (int32_t)gx; g.gy = (int32_t)gy;
    }
    g.tx = (g.gx + 7) / 8;
    g.pad_ = 0;
    frames[f] = g;
    sg = g;
    if (f == n_frames - 1) cell_start[n_cells] = (int32_t)end;     // end marker of the cell table
  }
  __syncthreads();
  const FrameGrid g = sg;
  const int cells = ((g.gx + 7) / 8) * ((g.gy + 7) / 8) * 64;       // cells in use (incl. the padding of partial tiles) <= cap
  int32_t* cnt = (cells <= lds_cells) ? gf_cnt : cell_count + c0;   // histogram, then cursors
  for (int c = t; c < cells; c += GF_THREADS) cnt[c] = 0;
  __syncthreads();
  // ---- cell of every point + histogram
  for (int64_t i = beg + t; i < end; i += GF_THREADS) {
    int cx = (int)floor((X[i * DIM] - g.x0) / g.h);
    int cy = (int)floor((X[i * DIM + 1] - g.y0) / g.h);
    cx = min(max(cx, 0), g.gx - 1);
    cy = min(max(cy, 0), g.gy - 1);
    const int c = cell_id(g, cx, cy);
    point_cell[i] = c;
    point_frame[i] = f;
    atomicAdd(&cnt[c - g.cell_base], 1);
  }
  __syncthreads();
  // ---- exclusive scan over the cells: thread t owns `per` consecutive cells
  const int per = (cells + GF_THREADS - 1) / GF_THREADS;
  const int lo = min(t * per, cells), hi = min(lo + per, cells);
  int sum = 0;
  for (int c = lo; c < hi; c++) sum += cnt[c];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int before = 0;
  for (int i = 0; i < w; i++) before += wsum[i];
  int run = (int)beg + before + inc - sum;                           // global position of the first point of cell `lo`
  for (int c = lo; c < hi; c++) {
    const int k = cnt[c];
    cnt[c] = run;                                                    // cursor of the fill phase
    cell_start[c0 + c] = run;
    run += k;
  }
  for (int64_t c = cells + t; c < cap; c += GF_THREADS) cell_start[c0 + c] = (int32_t)end;   // cells the grid does not use
  __syncthreads();
  // ---- points into cell order.  The atomics hand out the places of a cell in ARRIVAL order, which varies from run to run (and
  // with whatever else runs on the device); everything downstream follows the order of the points inside a cell -- the visiting
  // order of the conv layers, the window plan, which targets a window hands to the per-target kernel -- and with it the last bits of
  // the outputs (r05: two threads driving two models got results 1e-7 apart from the same calls alone, tests/test_gpu_threads.py).
  // So the arrival order only fills a scratch list (sorted_cell, overwritten below); a point's place is its cell's start plus the
  // number of points of the cell with a SMALLER index: ascending index inside every cell, whatever the atomics did.
  for (int64_t i = beg + t; i < end; i += GF_THREADS) {
    const int c = point_cell[i];
    const int at = atomicAdd(&cnt[c - g.cell_base], 1);
    sorted_cell[at] = (int32_t)i;
    point_rank[i] = at;                                // (arrival place: what a crowded cell keeps, GRID_ORDERED_CELL_MAX)
  }
  __syncthreads();
  for (int64_t i = beg + t; i < end; i += GF_THREADS) {
    const int c = point_cell[i];
    const int s0 = cell_start[c0 + (c - g.cell_base)], s1 = cnt[c - g.cell_base];      // (the cursor stands at the cell's end now)
    int p = s0;
    if (s1 - s0 > GRID_ORDERED_CELL_MAX) p = point_rank[i];
    else for (int j = s0; j < s1; j++) p += sorted_cell[j] < (int32_t)i ? 1 : 0;
    sorted_idx[p] = (int32_t)i;
    point_rank[i] = p;
    sorted_frame[p] = f;
#pragma unroll
    for (int d = 0; d < DIM; d++) sorted_pos[(int64_t)p * DIM + d] = X[i * DIM + d];
  }
  __syncthreads();
  for (int64_t i = beg + t; i < end; i += GF_THREADS) sorted_cell[point_rank[i]] = point_cell[i];
}

template <int DIM>
__global__ __launch_bounds__(GF_THREADS) void k_grid_frame(const double* __restrict__ X, const int64_t* __restrict__ frame_ptr,
                                                         int n_frames, FrameGrid* __restrict__ frames, double cell_size,
                                                         double pts_per_cell, int32_t* __restrict__ cell_count,
                                                         int32_t* __restrict__ cell_start, int64_t n_cells, int lds_cells,
                                                         int32_t* __restrict__ sorted_idx, int32_t* __restrict__ sorted_frame,
                                                         int32_t* __restrict__ sorted_cell, double* __restrict__ sorted_pos,
                                                         int32_t* __restrict__ point_cell, int32_t* __restrict__ point_frame,
                                                         int32_t* __restrict__ point_rank) {
  gf_block<DIM>(X, frame_ptr, n_frames, frames, cell_size, pts_per_cell, cell_count, cell_start, n_cells, lds_cells, sorted_idx, sorted_frame,
                sorted_cell, sorted_pos, point_cell, point_frame, point_rank);
}

// The same for frames of up to GFR_PPT x 1024 points with a two-column basis (r06; every RadarScenes- / nuScenes-shaped frame): a
// thread keeps ITS points -- coordinates, cell, arrival place -- in registers from the first phase to the last, the arrival lists and
// the cells' first places live in LDS beside the counters.  k_grid_frame reads every point's coordinates three times, its cell three
// times and the arrival lists from global memory, a dependent round trip per phase (seven phases: 24 - 32 us per batch whatever the
// frames hold); here global memory is read once and written once.  Same arithmetic, same cells, same order inside a cell: the
// outputs are identical.  A frame beyond the promise (more points than the registers hold, or more cells than the LDS table)
// takes the general block code.
constexpr int GFR_PPT = 4;
__global__ __launch_bounds__(GF_THREADS) void k_grid_frame_reg(const double* __restrict__ X, const int64_t* __restrict__ frame_ptr,
                                                             int n_frames, FrameGrid* __restrict__ frames, double cell_size,
                                                             double pts_per_cell, int32_t* __restrict__ cell_count,
                                                             int32_t* __restrict__ cell_start, int64_t n_cells, int lds_cells,
                                                             int32_t* __restrict__ sorted_idx, int32_t* __restrict__ sorted_frame,
                                                             int32_t* __restrict__ sorted_cell, double* __restrict__ sorted_pos,
                                                             int32_t* __restrict__ point_cell, int32_t* __restrict__ point_frame,
                                                             int32_t* __restrict__ point_rank) {
  extern __shared__ int32_t gf_cnt[];                 // [lds_cells] counters, then cursors | [lds_cells] first places | [GFR_PPT x 1024] arrival lists
  __shared__ double red[4][GF_THREADS / 64];
  __shared__ FrameGrid sg;
  __shared__ int wsum[GF_THREADS / 64];
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  const int64_t nf = end - beg;
  const int64_t c0 = CELLS_PER_POINT * beg + CELLS_PER_FRAME * (int64_t)f;
  const int64_t cap = CELLS_PER_POINT * nf + CELLS_PER_FRAME;
  if (nf > GFR_PPT * GF_THREADS || cap > lds_cells) {      // (block-uniform: beyond the caller's promise)
    gf_block<2>(X, frame_ptr, n_frames, frames, cell_size, pts_per_cell, cell_count, cell_start, n_cells, lds_cells, sorted_idx, sorted_frame,
                sorted_cell, sorted_pos, point_cell, point_frame, point_rank);
    return;
  }
  int32_t* const cst = gf_cnt + lds_cells;
  int32_t* const arr = cst + lds_cells;
  // ---- this thread's points (point k of thread t: beg + t + k x 1024) and the bounding box
  double px[GFR_PPT], py[GFR_PPT];
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
#pragma unroll
  for (int k = 0; k < GFR_PPT; k++) {
    const int64_t i = beg + t + k * GF_THREADS;
    px[k] = 0.0; py[k] = 0.0;
    if (i < end) {
      const double2 v = *(const double2*)(X + i * 2);
      px[k] = v.x; py[k] = v.y;
      xmin = fmin(xmin, v.x); xmax = fmax(xmax, v.x);
      ymin = fmin(ymin, v.y); ymax = fmax(ymax, v.y);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    xmin = fmin(xmin, gf_shfl_xor(xmin, o)); xmax = fmax(xmax, gf_shfl_xor(xmax, o));
    ymin = fmin(ymin, gf_shfl_xor(ymin, o)); ymax = fmax(ymax, gf_shfl_xor(ymax, o));
  }
  if (lane == 0) { red[0][w] = xmin; red[1][w] = xmax; red[2][w] = ymin; red[3][w] = ymax; }
  __syncthreads();
  if (t == 0) {
    for (int i = 1; i < GF_THREADS / 64; i++) {
      xmin = fmin(xmin, red[0][i]); xmax = fmax(xmax, red[1][i]); ymin = fmin(ymin, red[2][i]); ymax = fmax(ymax, red[3][i]);
    }
    FrameGrid g;
    g.n_pts = (int32_t)nf;
    g.cell_base = (int32_t)c0;
    if (nf == 0) {
      g.x0 = 0; g.y0 = 0; g.h = 1; g.gx = 1; g.gy = 1;
    } else {                                            // (the geometry rules of k_frame_grid / gf_block, word for word)
      const double ex = xmax - xmin, ey = ymax - ymin;
      double h;
      if (cell_size > 0) {
        h = cell_size * (1.0 + 9.5367431640625e-07);
      } else {
        const double area = ex * ey;
        if (area > 0) h = sqrt(area * pts_per_cell / (double)nf);
        else if (ex + ey > 0) h = (ex + ey) * pts_per_cell / (double)nf;
        else h = 1.0;
        const double hmin = fmax(ex, ey) * 1e-6;
        if (h < hmin) h = hmin;
        if (!(h > 0)) h = 1.0;
      }
      int64_t gx, gy;
      for (;;) {
        gx = (int64_t)floor(ex / h) + 1;
        gy = (int64_t)floor(ey / h) + 1;
        if (((gx + 7) / 8) * ((gy + 7) / 8) * 64 <= cap) break;
        h *= 1.5;
      }
      g.x0 = xmin; g.y0 = ymin; g.h = h; g.gx = (int32_t)gx; g.gy = (int32_t)gy;
    }
    g.tx = (g.gx + 7) / 8;
    g.pad_ = 0;
    frames[f] = g;
    sg = g;
    if (f == n_frames - 1) cell_start[n_cells] = (int32_t)end;     // end marker of the cell table
  }
  __syncthreads();
  const FrameGrid g = sg;
  const int cells = ((g.gx + 7) / 8) * ((g.gy + 7) / 8) * 64;       // cells in use (incl. the padding of partial tiles) <= cap <= lds_cells
  for (int c = t; c < cells; c += GF_THREADS) gf_cnt[c] = 0;
  __syncthreads();
  // ---- cell of every point + histogram
  int pc[GFR_PPT];
#pragma unroll
  for (int k = 0; k < GFR_PPT; k++) {
    const int64_t i = beg + t + k * GF_THREADS;
    pc[k] = 0;
    if (i < end) {
      int cx = (int)floor((px[k] - g.x0) / g.h);
      int cy = (int)floor((py[k] - g.y0) / g.h);
      cx = min(max(cx, 0), g.gx - 1);
      cy = min(max(cy, 0), g.gy - 1);
      pc[k] = cell_id(g, cx, cy);
      point_cell[i] = pc[k];
      point_frame[i] = f;
      atomicAdd(&gf_cnt[pc[k] - g.cell_base], 1);
    }
  }
  __syncthreads();
  // ---- exclusive scan over the cells: thread t owns `per` consecutive cells
  const int per = (cells + GF_THREADS - 1) / GF_THREADS;
  const int lo = min(t * per, cells), hi = min(lo + per, cells);
  int sum = 0;
  for (int c = lo; c < hi; c++) sum += gf_cnt[c];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int before = 0;
  for (int i = 0; i < w; i++) before += wsum[i];
  int run = (int)beg + before + inc - sum;                           // global position of the first point of cell `lo`
  for (int c = lo; c < hi; c++) {
    const int k = gf_cnt[c];
    gf_cnt[c] = run;                                                 // cursor of the fill phase
    cst[c] = run;
    cell_start[c0 + c] = run;
    run += k;
  }
  for (int64_t c = cells + t; c < cap; c += GF_THREADS) cell_start[c0 + c] = (int32_t)end;   // cells the grid does not use
  __syncthreads();
  // ---- arrival places (the atomics' order varies from run to run: it only fills the lists the ranking below reads)
  int at[GFR_PPT];
#pragma unroll
  for (int k = 0; k < GFR_PPT; k++) {
    const int64_t i = beg + t + k * GF_THREADS;
    at[k] = 0;
    if (i < end) {
      at[k] = atomicAdd(&gf_cnt[pc[k] - g.cell_base], 1);
      arr[at[k] - (int)beg] = (int32_t)i;
    }
  }
  __syncthreads();
  // ---- a point's place: its cell's first place + the points of the cell with a smaller index; everything written from registers
#pragma unroll
  for (int k = 0; k < GFR_PPT; k++) {
    const int64_t i = beg + t + k * GF_THREADS;
    if (i < end) {
      const int cl = pc[k] - g.cell_base;
      const int s0 = cst[cl], s1 = gf_cnt[cl];                       // (the cursor stands at the cell's end now)
      int p = s0;
      if (s1 - s0 > GRID_ORDERED_CELL_MAX) p = at[k];
      else for (int j = s0; j < s1; j++) p += arr[j - (int)beg] < (int32_t)i ? 1 : 0;
      sorted_idx[p] = (int32_t)i;
      point_rank[i] = p;
      sorted_frame[p] = f;
      sorted_cell[p] = pc[k];
      *(double2*)(sorted_pos + (int64_t)p * 2) = make_double2(px[k], py[k]);
    }
  }
}

template <int DIM>
__device__ __forceinline__ double dist2(const double* __restrict__ q, const double* __restrict__ c) {
  double d2 = 0.0;
#pragma unroll
  for (int d = 0; d < DIM; d++) {
    const double t = q[d] - c[d];
    d2 = d2 + t * t;  // no FMA: file is built with -ffp-contract=off
  }
  return d2;
}

// ------------------------------------------------------------------------------------------------
// radius graph: count pass and fill pass share the traversal
// ------------------------------------------------------------------------------------------------
template <int DIM, bool FILL>
__global__ __launch_bounds__(256) void k_radius(int64_t n, const double* __restrict__ X,
                                               const int32_t* __restrict__ point_cell,
                                               const int32_t* __restrict__ point_frame,
                                               const FrameGrid* __restrict__ frames,
                                               const int32_t* __restrict__ cell_start,
                                               const int32_t* __restrict__ sorted_idx,
                                               const int32_t* __restrict__ sorted_frame,
                                               const int32_t* __restrict__ sorted_cell,
                                               const double* __restrict__ sorted_pos, double r2,
                                               int32_t* __restrict__ deg, const int32_t* __restrict__ rowptr,
                                               int32_t* __restrict__ col, int32_t* __restrict__ row_tmp,
                                               int32_t* __restrict__ nbr_cache, int64_t n_edges_expected,
                                               int32_t* __restrict__ status) {
  // count pass: one thread per point in CELL order (coherent candidate reads, scattered 4-B result); it also leaves the
  // first RADIUS_CACHE neighbours of every point in `nbr_cache`.
  // fill pass: one thread per point in INDEX order -- its writes (col / edge_index rows) are then contiguous across
  // the wave, which is what that pass is bound by.  Rows that fit the cache are copied from it; only denser rows repeat
  // the search (the query's cell then comes from the binning arrays).
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  if (FILL && status != nullptr && rowptr[n] != n_edges_expected) {
    // guarded fill (a captured step replayed on data whose edge count is no longer the one the buffers were sized
    // for): write nothing, flag it
    if (tid == 0) atomicOr(status, RGNN_STATUS_EDGE_COUNT_CHANGED);
    return;
  }
  if (FILL) {
    const int beg = rowptr[tid], d = rowptr[tid + 1] - beg;
    if (d <= RADIUS_CACHE) {
      for (int c = 0; c < d; c++) {
        col[beg + c] = nbr_cache[tid * RADIUS_CACHE + c];
        row_tmp[beg + c] = (int32_t)tid;
      }
      return;
    }
  }
  int i;
  int64_t p;
  int cell, frame;
  double q[DIM];
  if (FILL) {
    i = (int)tid;
    p = -1;
    cell = point_cell[i];
    frame = point_frame[i];
#pragma unroll
    for (int d = 0; d < DIM; d++) q[d] = X[(int64_t)i * DIM + d];
  } else {
    p = tid;
    i = sorted_idx[p];
    cell = sorted_cell[p];
    frame = sorted_frame[p];
#pragma unroll
    for (int d = 0; d < DIM; d++) q[d] = sorted_pos[p * DIM + d];
  }
  const FrameGrid g = frames[frame];
  int cx, cy;
  cell_xy(g, cell, cx, cy);
  int cnt = 0;
  int64_t out = FILL ? (int64_t)rowptr[i] : 0;
  // the bounds of all nine cells first, in one round of loads (cells outside the grid become empty ranges): walking the
  // cells one after the other pays the load latency nine times in a row for rows that hold four neighbours on average
  int rb[9], re[9];
#pragma unroll
  for (int u = 0; u < 9; u++) {
    const int xx = cx + (u % 3) - 1, yy = cy + (u / 3) - 1;
    const bool in = xx >= 0 && xx < g.gx && yy >= 0 && yy < g.gy;
    const int c = in ? cell_id(g, xx, yy) : g.cell_base;
    const int b = cell_start[c], e = cell_start[c + 1];
    rb[u] = b;
    re[u] = in ? e : b;
  }
  auto take = [&](int idx, double d2) {
    if (idx == i || !(d2 <= r2)) return;  // include_self=False: by identity, not by distance (duplicates stay neighbours)
    if (FILL) {
      col[out + cnt] = idx;      // unsorted (col = scratch here); k_rank_rows orders the row afterwards
      row_tmp[out + cnt] = i;
    } else if (cnt < RADIUS_CACHE) {
      nbr_cache[(int64_t)i * RADIUS_CACHE + cnt] = idx;
    }
    cnt++;
  };
#pragma unroll
  for (int u = 0; u < 9; u++) {
    // RADIUS_ROUND candidates per round, their loads issued together (a cluster cell holds dozens of points: one candidate per
    // round trip is one load latency each; measured 59 -> 36 us for the count pass of the C2 batch; rounds of eight: 38 us)
    const int last = re[u] - 1;
    for (int pp = rb[u]; pp <= last; pp += RADIUS_ROUND) {
      int id[RADIUS_ROUND];
      double d2[RADIUS_ROUND];
#pragma unroll
      for (int v = 0; v < RADIUS_ROUND; v++) {       // (positions past the range re-read its last point and are dropped below)
        const int ps = min(pp + v, last);
        id[v] = sorted_idx[ps];
        d2[v] = dist2<DIM>(q, sorted_pos + (int64_t)ps * DIM);
      }
#pragma unroll
      for (int v = 0; v < RADIUS_ROUND; v++)
        if (pp + v <= last) take(id[v], d2[v]);
    }
  }
  if (!FILL) deg[i] = cnt;
}

// Edge-parallel row sort: entry e of row i moves to rowptr[i] + (number of smaller entries of the row).  Ids inside a
// row are unique, so the ranks are a permutation; every compare is independent (no per-thread serial sort whose
// run time is set by the densest cluster), neighbouring lanes read the same row (L1 hits).
__global__ __launch_bounds__(256) void k_rank_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ row_of,
                                                  const int32_t* __restrict__ in, int64_t n_edges,
                                                  int32_t* __restrict__ out, int64_t* __restrict__ edge_index,
                                                  int64_t n, int guarded) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  if (guarded && rowptr[n] != n_edges) return;   // (see k_radius: the previous contents stay)
  const int i = row_of[e];
  const int beg = rowptr[i], last = rowptr[i + 1] - 1;
  const int32_t v = in[e];
  int rank = 0;
  for (int b = beg; b <= last; b += 4) {         // four entries per round, their loads issued together
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) c[u] = in[min(b + u, last)];
#pragma unroll
    for (int u = 0; u < 4; u++) rank += (b + u <= last && c[u] < v) ? 1 : 0;
  }
  out[beg + rank] = v;
  if (edge_index) {
    edge_index[beg + rank] = i;            // E[:,0] = query point (graph.py:61)
    edge_index[n_edges + beg + rank] = v;  // E[:,1] = neighbour   (graph.py:62)
  }
}

// Fill pass in ONE launch (r03): a team of 16 lanes per point writes its finished row -- neighbour ids ascending in `col`, both
// rows of edge_index, and (optionally) the relative_position edge attributes, for which the query's coordinates are at hand
// and the neighbours' one gather away.  Replaces k_radius<., true> (unsorted rows + row ids to scratch), k_rank_rows
// (edge-parallel sort from that scratch) and k_edge_relative_position (re-gathers X[i], X[j] through edge_index): three launches,
// 31 us and ~120 MB of traffic on the C2 batch for 16 MB of results.  Rows of up to RADIUS_CACHE neighbours come from the count
// pass's cache and are ranked with shuffles inside the team; denser rows repeat the search (the team's lanes test 16
// candidates at a time, ballot + popcount place the hits in `tmp`) and rank from there.  Same rows, same order, same values.
constexpr int ROWS_LDS = 512;                          // dense rows of up to this many neighbours are ranked from LDS (32 KB per block)
template <int DIM>
__global__ __launch_bounds__(256) void k_radius_rows(int64_t n, const double* __restrict__ X,
                                                    const int32_t* __restrict__ point_cell, const int32_t* __restrict__ point_frame,
                                                    const FrameGrid* __restrict__ frames, const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ sorted_idx, const double* __restrict__ sorted_pos,
                                                    double r2, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ nbr_cache, int32_t* __restrict__ tmp,
                                                    int64_t n_edges, int guarded, int32_t* __restrict__ col,
                                                    int64_t* __restrict__ edge_index, float* __restrict__ rel_pos,
                                                    int rel_undirected, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (guarded && (int64_t)rowptr[n] != n_edges) {       // (replayed step on modified points: previous contents stay)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, RGNN_STATUS_EDGE_COUNT_CHANGED);
    return;
  }
  if (i >= n) return;
  const int beg = rowptr[i], d = rowptr[i + 1] - beg;
  if (d == 0) return;
  double q[DIM];
#pragma unroll
  for (int k = 0; k < DIM; k++) q[k] = X[i * DIM + k];
  auto emit = [&](int v, int rank) {
    const int64_t pos = (int64_t)beg + rank;
    col[pos] = v;
    if (edge_index) {
      edge_index[pos] = i;                               // E[:,0] = query point (graph.py:61)
      edge_index[n_edges + pos] = v;                     // E[:,1] = neighbour   (graph.py:62)
    }
    if (rel_pos) {
      double dx = q[0] - X[(int64_t)v * DIM], dy = q[1] - X[(int64_t)v * DIM + 1];      // graph.py:199-200
      if (rel_undirected) { dx = fabs(dx); dy = fabs(dy); }
      *(float2*)(rel_pos + pos * 2) = make_float2((float)dx, (float)dy);
    }
  };
  if (d <= RADIUS_CACHE) {
    constexpr int SLOTS = (RADIUS_CACHE + 15) / 16;
    int e[SLOTS], rk[SLOTS];
#pragma unroll
    for (int s_ = 0; s_ < SLOTS; s_++) {
      e[s_] = (s_ * 16 + l < d) ? nbr_cache[i * RADIUS_CACHE + s_ * 16 + l] : 0x7fffffff;
      rk[s_] = 0;
    }
#pragma unroll
    for (int s_ = 0; s_ < SLOTS; s_++) {
      if (s_ * 16 >= d) break;
      for (int c = 0; c < 16 && s_ * 16 + c < d; c++) {
        const int v = __shfl(e[s_], c, 16);
#pragma unroll
        for (int u = 0; u < SLOTS; u++) rk[u] += (v < e[u]) ? 1 : 0;
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < SLOTS; s_++)
      if (s_ * 16 + l < d) emit(e[s_], rk[s_]);
    return;
  }
  // ---- a dense row: search again, 16 candidates at a time; hits are staged in LDS (rows of up to ROWS_LDS neighbours: a dense
  // cluster, a 100 000-point cloud at r = 1 m with ~34 neighbours per point) or, beyond that, in the global scratch
  __shared__ int32_t stage_lds[16][ROWS_LDS];
  int32_t* const stg = stage_lds[threadIdx.x >> 4];
  const bool in_lds = d <= ROWS_LDS;
  const FrameGrid g = frames[point_frame[i]];
  int cx, cy;
  cell_xy(g, point_cell[i], cx, cy);
  const int team_shift = (threadIdx.x & 63) & ~15;       // this team's bits in a wave-wide ballot
  int cnt = 0;
  for (int u = 0; u < 9; u++) {
    const int xx = cx + (u % 3) - 1, yy = cy + (u / 3) - 1;
    if (xx < 0 || xx >= g.gx || yy < 0 || yy >= g.gy) continue;
    const int c = cell_id(g, xx, yy);
    const int b = cell_start[c], en = cell_start[c + 1];
    for (int pp = b; pp < en; pp += 16) {
      const int ps = pp + l;
      bool hit = false;
      int idx = 0;
      if (ps < en) {
        idx = sorted_idx[ps];
        const double d2 = dist2<DIM>(q, sorted_pos + (int64_t)ps * DIM);
        hit = idx != (int)i && d2 <= r2;
      }
      const unsigned m = (unsigned)((__ballot(hit) >> team_shift) & 0xffffu);
      if (hit) {
        const int at = cnt + __popc(m & ((1u << l) - 1u));
        if (in_lds) stg[at] = idx; else tmp[beg + at] = idx;
      }
      cnt += __popc(m);
    }
  }
  if (in_lds) {
    // (the team's lanes are lanes of one wave: its LDS writes are ordered before the reads below by the wave's own program order
    //  plus an LDS wait -- no block-wide barrier, which teams with cached rows have already left)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c = l; c < d; c += 16) {
      const int v = ((volatile int32_t*)stg)[c];
      int rank = 0;
      for (int o = 0; o < d; o++) rank += (((volatile int32_t*)stg)[o] < v) ? 1 : 0;
      emit(v, rank);
    }
    return;
  }
  __threadfence();                                        // the staged row is read back by the other lanes of the team (L2)
  for (int c = l; c < d; c += 16) {
    const int v = __hip_atomic_load(tmp + beg + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int rank = 0;
    for (int o = 0; o < d; o++) rank += (__hip_atomic_load(tmp + beg + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) ? 1 : 0;
    emit(v, rank);
  }
}

// Search + fill in ONE launch for REPLAYED steps (r06, rgnn_radius_graph_rows_direct).  A captured step already knows the rows of its
// graph: the committed rowptr of the last replay whose graph matched (rgnn_radius_rows_commit).  A point's team searches its 3 x 3 cells
// once, and if the row it finds has the committed length it is ranked and written at the committed place -- no count pass, no scan, no
// commit copy, no neighbour cache (k_radius + two scan kernels + k_rows_commit + k_radius_rows: five launches, ~94 us on the C2 batch).
// A row of another length means the points changed under the captured graph: nothing of that row is written (the previous contents
// stay, as with the guarded fill) and STATUS_EDGE_COUNT_CHANGED is raised -- per row, which is stricter than the total the guarded fill
// compares.  The nine cell ranges are fetched in one round (lane u < 9 takes cell u) and walked as ONE list, sixteen candidates a trip.
template <int DIM>
__global__ __launch_bounds__(256) void k_radius_rows_direct(int64_t n, const double* __restrict__ X,
                                                           const int32_t* __restrict__ point_cell, const int32_t* __restrict__ point_frame,
                                                           const FrameGrid* __restrict__ frames, const int32_t* __restrict__ cell_start,
                                                           const int32_t* __restrict__ sorted_idx, const double* __restrict__ sorted_pos,
                                                           double r2, const int32_t* __restrict__ rowptr, int32_t* __restrict__ tmp,
                                                           int64_t n_edges, int32_t* __restrict__ col, int64_t* __restrict__ edge_index,
                                                           float* __restrict__ rel_pos, int rel_undirected, int32_t* __restrict__ status) {
  // (rows of up to ROWS_LDS_DIRECT neighbours are staged in LDS: 8 KB per block, so that the CU's wave slots -- not its LDS -- bound
  //  the occupancy of this latency-bound kernel; k_radius_rows' 32 KB allow five blocks per CU)
  constexpr int ROWS_LDS_DIRECT = 128;
  __shared__ int32_t stage_lds[16][ROWS_LDS_DIRECT];
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (i >= n) return;
  if (i == 0 && l == 0 && (int64_t)rowptr[n] != n_edges) atomicOr(status, RGNN_STATUS_EDGE_COUNT_CHANGED);   // (the buffers were sized for n_edges)
  const int beg = rowptr[i], d = rowptr[i + 1] - beg;
  int32_t* const stg = stage_lds[threadIdx.x >> 4];
  const bool in_lds = d <= ROWS_LDS_DIRECT;
  double q[DIM];
#pragma unroll
  for (int k = 0; k < DIM; k++) q[k] = X[i * DIM + k];
  const FrameGrid g = frames[point_frame[i]];
  int cx, cy;
  cell_xy(g, point_cell[i], cx, cy);
  // lane u < 9: the range of cell u (empty outside the grid); then a running sum over the nine lanes: the cells as one list
  int b = 0, len = 0;
  if (l < 9) {
    const int xx = cx + (l % 3) - 1, yy = cy + (l / 3) - 1;
    if (xx >= 0 && xx < g.gx && yy >= 0 && yy < g.gy) {
      const int c = cell_id(g, xx, yy);
      b = cell_start[c];
      len = cell_start[c + 1] - b;
    }
  }
  int incl = len;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(incl, o, 16); if (l >= o) incl += u; }
  const int total = __shfl(incl, 15, 16);
  const int team_shift = (threadIdx.x & 63) & ~15;       // this team's bits in a wave-wide ballot
  int cnt = 0;
  for (int v0 = 0; v0 < total; v0 += 16) {
    const int v = v0 + l;
    bool hit = false;
    int idx = 0;
    // candidate v of the list lies in the first cell whose running sum exceeds v (nine shuffles, no memory)
    int ps = -1;
#pragma unroll
    for (int u = 0; u < 9; u++) {
      const int iu = __shfl(incl, u, 16), bu = __shfl(b, u, 16), lu = __shfl(len, u, 16);
      if (ps < 0 && v < iu) ps = bu + (v - (iu - lu));
    }
    if (v < total) {
      idx = sorted_idx[ps];
      const double d2 = dist2<DIM>(q, sorted_pos + (int64_t)ps * DIM);
      hit = idx != (int)i && d2 <= r2;
    }
    const unsigned m = (unsigned)((__ballot(hit) >> team_shift) & 0xffffu);
    if (hit) {
      const int at = cnt + __popc(m & ((1u << l) - 1u));
      if (at < d) { if (in_lds) stg[at] = idx; else tmp[beg + at] = idx; }     // (never beyond the committed row)
    }
    cnt += __popc(m);
  }
  if (cnt != d) {                                         // the points changed under the captured graph: this row keeps its contents
    if (l == 0) atomicOr(status, RGNN_STATUS_EDGE_COUNT_CHANGED);
    return;
  }
  if (d == 0) return;
  auto emit = [&](int v, int rank) {
    const int64_t pos = (int64_t)beg + rank;
    col[pos] = v;
    if (edge_index) {
      edge_index[pos] = i;                               // E[:,0] = query point (graph.py:61)
      edge_index[n_edges + pos] = v;                     // E[:,1] = neighbour   (graph.py:62)
    }
    if (rel_pos) {
      double dx = q[0] - X[(int64_t)v * DIM], dy = q[1] - X[(int64_t)v * DIM + 1];      // graph.py:199-200
      if (rel_undirected) { dx = fabs(dx); dy = fabs(dy); }
      *(float2*)(rel_pos + pos * 2) = make_float2((float)dx, (float)dy);
    }
  };
  if (in_lds) {
    // (the team's lanes are lanes of one wave: its LDS writes are ordered before the reads below by the wave's own program order
    //  plus an LDS wait)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c = l; c < d; c += 16) {
      const int v = ((volatile int32_t*)stg)[c];
      int rank = 0;
      for (int o = 0; o < d; o++) rank += (((volatile int32_t*)stg)[o] < v) ? 1 : 0;
      emit(v, rank);
    }
    return;
  }
  __threadfence();                                        // the staged row is read back by the other lanes of the team (L2)
  for (int c = l; c < d; c += 16) {
    const int v = __hip_atomic_load(tmp + beg + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int rank = 0;
    for (int o = 0; o < d; o++) rank += (__hip_atomic_load(tmp + beg + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) ? 1 : 0;
    emit(v, rank);
  }
}

// ------------------------------------------------------------------------------------------------
// kNN: ring expansion over the grid, k best candidates per thread kept in LDS
// ------------------------------------------------------------------------------------------------
constexpr int KNN_THREADS = 128;

template <int DIM>
__global__ __launch_bounds__(KNN_THREADS) void k_knn(int64_t n, int k, const FrameGrid* __restrict__ frames,
                                                    const int32_t* __restrict__ cell_start,
                                                    const int32_t* __restrict__ sorted_idx,
                                                    const int32_t* __restrict__ sorted_frame,
                                                    const int32_t* __restrict__ sorted_cell,
                                                    const double* __restrict__ sorted_pos, int32_t* __restrict__ nbr,
                                                    int64_t* __restrict__ edge_index, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sd = (double*)smem;                          // [k][KNN_THREADS]
  int32_t* si = (int32_t*)(sd + (size_t)k * KNN_THREADS);  // [k][KNN_THREADS]
  const int t = threadIdx.x;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + t;
  if (p >= n) return;
  const FrameGrid g = frames[sorted_frame[p]];
  const int i = sorted_idx[p];
  const int64_t E = n * (int64_t)k;
  if (g.n_pts <= k) {  // sklearn: "Expected n_neighbors < n_samples_fit"
    atomicOr(status, RGNN_STATUS_KNN_TOO_FEW_POINTS);
    for (int a = 0; a < k; a++) {
      nbr[(int64_t)i * k + a] = -1;
      if (edge_index) { edge_index[(int64_t)i * k + a] = i; edge_index[E + (int64_t)i * k + a] = -1; }
    }
    return;
  }
  int cx, cy;
  cell_xy(g, sorted_cell[p], cx, cy);
  double q[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) q[d] = sorted_pos[p * DIM + d];

  int count = 0, worst_slot = 0, worst_i = -1;
  double worst_d = -1.0;

  auto scan_range = [&](int beg, int end) {
    for (int pp = beg; pp < end; pp++) {
      if (pp == p) continue;
      const double d2 = dist2<DIM>(q, sorted_pos + (int64_t)pp * DIM);
      const int idx = sorted_idx[pp];
      bool rescan = false;
      if (count < k) {
        sd[count * KNN_THREADS + t] = d2;
        si[count * KNN_THREADS + t] = idx;
        count++;
        rescan = (count == k);
      } else if (d2 < worst_d || (d2 == worst_d && idx < worst_i)) {
        sd[worst_slot * KNN_THREADS + t] = d2;
        si[worst_slot * KNN_THREADS + t] = idx;
        rescan = true;
      }
      if (rescan) {  // worst = max by (distance, index)
        worst_d = -1.0; worst_i = -1;
        for (int a = 0; a < k; a++) {
          const double da = sd[a * KNN_THREADS + t];
          const int ia = si[a * KNN_THREADS + t];
          if (da > worst_d || (da == worst_d && ia > worst_i)) { worst_d = da; worst_i = ia; worst_slot = a; }
        }
      }
    }
  };

  for (int R = 0;; R++) {
    const int x0 = cx - R, x1 = cx + R, y0 = cy - R, y1 = cy + R;
    const int xa = max(x0, 0), xb = min(x1, g.gx - 1);
    for (int yy = max(y0, 0); yy <= min(y1, g.gy - 1); yy++) {
      if (yy == y0 || yy == y1) {
        for (int xx = xa; xx <= xb; xx++) {  // full ring row
          const int c = cell_id(g, xx, yy);
          scan_range(cell_start[c], cell_start[c + 1]);
        }
      } else {
        if (x0 >= 0) { const int c = cell_id(g, x0, yy); scan_range(cell_start[c], cell_start[c + 1]); }
        if (x1 < g.gx) { const int c = cell_id(g, x1, yy); scan_range(cell_start[c], cell_start[c + 1]); }
      }
    }
    if (x0 <= 0 && y0 <= 0 && x1 >= g.gx - 1 && y1 >= g.gy - 1) break;  // whole frame visited
    if (count == k) {
      // every unvisited point lies outside the visited block of cells: lower bound of its distance
      double lb = INFINITY;
      if (x0 > 0) lb = fmin(lb, q[0] - (g.x0 + (double)x0 * g.h));
      if (x1 < g.gx - 1) lb = fmin(lb, (g.x0 + (double)(x1 + 1) * g.h) - q[0]);
      if (y0 > 0) lb = fmin(lb, q[1] - (g.y0 + (double)y0 * g.h));
      if (y1 < g.gy - 1) lb = fmin(lb, (g.y0 + (double)(y1 + 1) * g.h) - q[1]);
      lb -= 1e-9 * g.h;  // slack for the rounding of the binning division
      if (lb > 0 && worst_d < lb * lb) break;
    }
  }
  // ascending (distance, index): selection sort in LDS, k is small
  for (int a = 0; a < k; a++) {
    int best = a;
    double bd = sd[a * KNN_THREADS + t];
    int bi = si[a * KNN_THREADS + t];
    for (int b = a + 1; b < k; b++) {
      const double db = sd[b * KNN_THREADS + t];
      const int ib = si[b * KNN_THREADS + t];
      if (db < bd || (db == bd && ib < bi)) { bd = db; bi = ib; best = b; }
    }
    if (best != a) {
      sd[best * KNN_THREADS + t] = sd[a * KNN_THREADS + t];
      si[best * KNN_THREADS + t] = si[a * KNN_THREADS + t];
      sd[a * KNN_THREADS + t] = bd;
      si[a * KNN_THREADS + t] = bi;
    }
    const int64_t e = (int64_t)i * k + a;
    nbr[e] = bi;
    if (edge_index) { edge_index[e] = i; edge_index[E + e] = bi; }
  }
}

// ------------------------------------------------------------------------------------------------
// kNN, a TEAM of lanes per query (k <= KNN_CAP - TEAM).  The lanes of a team evaluate TEAM candidates of a run of cells
// at once; candidates that beat the current k-th best key (distance, index) are appended to a 128-entry LDS buffer, and
// when the buffer would overflow -- and at the end of every ring -- it is pruned to the k smallest keys by RANK COUNTING
// (every lane counts, for its KNN_CAP / TEAM entries, the entries with a smaller key: keys are unique, so the ranks are the
// sorted positions and the survivors land in ascending order; no serial insertion, no sort).  Same traversal, same
// termination rule and the same float64 distance as k_knn, so the rows are identical; a frame of 3 000 points keeps
// 3 000 teams busy instead of 47 waves of serial scans (C1: 292 -> 19 us), and at full batches the scan work per query
// drops by the team width (64 x 3 000, k = 20: 1 234 -> 391 us).
// ------------------------------------------------------------------------------------------------
constexpr int KNN_CAP = 128;
struct __attribute__((aligned(16))) KnnKey { double d; int32_t i; int32_t pad; };

template <int DIM, int TEAM>
__global__ __launch_bounds__(256) void k_knn_team(int64_t n, int k, const FrameGrid* __restrict__ frames,
                                                 const int32_t* __restrict__ cell_start,
                                                 const int32_t* __restrict__ sorted_idx,
                                                 const int32_t* __restrict__ sorted_frame,
                                                 const int32_t* __restrict__ sorted_cell,
                                                 const double* __restrict__ sorted_pos, int32_t* __restrict__ nbr,
                                                 int64_t* __restrict__ edge_index, int32_t* __restrict__ status,
                                                 const double* __restrict__ X = nullptr, float* __restrict__ rel_pos = nullptr,
                                                 int rel_undirected = 0, int32_t* __restrict__ degree_init = nullptr) {
  constexpr int NE = KNN_CAP / TEAM, TEAMS = 256 / TEAM;
  __shared__ KnnKey buf[TEAMS][2][KNN_CAP];
  const int team = threadIdx.x / TEAM, lane = threadIdx.x % TEAM;
  const int64_t p = (int64_t)blockIdx.x * TEAMS + team;
  if (p >= n) return;
  const FrameGrid g = frames[sorted_frame[p]];
  const int i = sorted_idx[p];
  const int64_t E = n * (int64_t)k;
  if (g.n_pts <= k) {  // sklearn: "Expected n_neighbors < n_samples_fit"
    if (lane == 0) atomicOr(status, RGNN_STATUS_KNN_TOO_FEW_POINTS);
    for (int a = lane; a < k; a += TEAM) {
      nbr[(int64_t)i * k + a] = -1;
      if (edge_index) { edge_index[(int64_t)i * k + a] = i; edge_index[E + (int64_t)i * k + a] = -1; }
    }
    return;
  }
  int cx, cy;
  cell_xy(g, sorted_cell[p], cx, cy);
  double q[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) q[d] = sorted_pos[p * DIM + d];

  int cnt = 0, sel = 0;
  bool dirty = false;
  double thr_d = INFINITY;            // key of the k-th best entry once k entries are known (else +inf)
  int thr_i = 0x7fffffff;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  auto team_ballot = [&](bool v) -> unsigned long long {
    const unsigned long long b = __ballot(v);
    if (TEAM == 64) return b;
    return (b >> ((threadIdx.x & 63) / TEAM * TEAM)) & ((1ull << (TEAM & 63)) - 1ull);
  };
  auto lds_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };

  auto prune = [&]() {
    KnnKey* A = buf[team][sel];
    KnnKey* B = buf[team][sel ^ 1];
    lds_sync();
    double od[NE];
    int oi[NE], rk[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const int e = lane + TEAM * j;
      od[j] = INFINITY; oi[j] = 0x7fffffff; rk[j] = 0;
      if (e < cnt) { od[j] = A[e].d; oi[j] = A[e].i; }
    }
    for (int f = 0; f < cnt; f++) {
      const double fd = A[f].d;
      const int fi = A[f].i;
#pragma unroll
      for (int j = 0; j < NE; j++) rk[j] += (fd < od[j] || (fd == od[j] && fi < oi[j])) ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const int e = lane + TEAM * j;
      if (e < cnt && rk[j] < k) { B[rk[j]].d = od[j]; B[rk[j]].i = oi[j]; }
    }
    sel ^= 1;
    cnt = min(cnt, k);
    dirty = false;
    lds_sync();
    if (cnt == k) { thr_d = B[k - 1].d; thr_i = B[k - 1].i; }
  };

  auto scan_range = [&](int beg, int end) {
    for (int base = beg; base < end; base += TEAM) {
      const int pp = base + lane;
      bool pass = false;
      double d2 = 0.0;
      int idx = 0;
      if (pp < end && pp != p) {
        d2 = dist2<DIM>(q, sorted_pos + (int64_t)pp * DIM);
        idx = sorted_idx[pp];
        pass = d2 < thr_d || (d2 == thr_d && idx < thr_i);
      }
      unsigned long long mask = team_ballot(pass);
      if (!mask) continue;
      int np = __popcll(mask);
      if (cnt + np > KNN_CAP) {
        prune();
        pass = pass && (d2 < thr_d || (d2 == thr_d && idx < thr_i));
        mask = team_ballot(pass);
        np = __popcll(mask);
      }
      if (pass) {
        KnnKey* A = buf[team][sel];
        const int pos = cnt + __popcll(mask & lt_mask);
        A[pos].d = d2;
        A[pos].i = idx;
      }
      cnt += np;
      dirty = dirty || np > 0;
    }
  };
  // cells (xa..xb, yy): cells of one 8-wide tile row have consecutive ids, so their points are ONE contiguous run
  auto scan_cells = [&](int xa, int xb, int yy) {
    for (int xx = xa; xx <= xb;) {
      const int xe = min(xb, xx | 7);
      const int c0 = cell_id(g, xx, yy);
      scan_range(cell_start[c0], cell_start[c0 + (xe - xx) + 1]);
      xx = xe + 1;
    }
  };

  for (int R = 1;; R++) {
    const int x0 = cx - R, x1 = cx + R, y0 = cy - R, y1 = cy + R;
    const int xa = max(x0, 0), xb = min(x1, g.gx - 1);
    if constexpr (TEAM == 64) {
      // The ring's cells one per lane (the 3 x 3 block: 9; ring R >= 2: 8 R, 64 at a time): every lane fetches the bounds of
      // ITS cell -- one round of loads for the whole ring instead of a dependent pair per cell -- and the wave then walks
      // the NON-EMPTY cells only (sparse clutter has rings of dozens of empty cells), merging cells whose point ranges are
      // adjacent in the cell-ordered array (the cells of an 8-wide tile row) into one run.
      const int ncell = (R == 1) ? 9 : 8 * R;
      for (int c0 = 0; c0 < ncell; c0 += 64) {
        const int c = c0 + lane;
        int xx = 0, yy = 0;
        bool in = c < ncell;
        if (R == 1) { xx = cx + c % 3 - 1; yy = cy + c / 3 - 1; }
        else {
          const int w = 2 * R + 1, hgt = 2 * R - 1;
          if (c < w) { xx = x0 + c; yy = y0; }
          else if (c < 2 * w) { xx = x0 + (c - w); yy = y1; }
          else if (c < 2 * w + hgt) { xx = x0; yy = y0 + 1 + (c - 2 * w); }
          else { xx = x1; yy = y0 + 1 + (c - 2 * w - hgt); }
        }
        in = in && xx >= 0 && xx < g.gx && yy >= 0 && yy < g.gy;
        int cb = 0, ce = 0;
        if (in) { const int id = cell_id(g, xx, yy); cb = cell_start[id]; ce = cell_start[id + 1]; }
        unsigned long long live = __ballot(ce > cb);
        while (live) {
          const int j = __builtin_ctzll(live);
          live &= live - 1;
          const int rb = __builtin_amdgcn_readlane(cb, j);
          int re = __builtin_amdgcn_readlane(ce, j);
          while (live) {                             // following non-empty cells that continue this run
            const int j2 = __builtin_ctzll(live);
            if (__builtin_amdgcn_readlane(cb, j2) != re) break;
            re = __builtin_amdgcn_readlane(ce, j2);
            live &= live - 1;
          }
          scan_range(rb, re);
        }
      }
    } else {
      for (int yy = max(y0, 0); yy <= min(y1, g.gy - 1); yy++) {
        if (R == 1 || yy == y0 || yy == y1) {
          scan_cells(xa, xb, yy);  // the 3 x 3 block first (rings 0 and 1), then full ring rows
        } else {
          if (x0 >= 0) { const int c = cell_id(g, x0, yy); scan_range(cell_start[c], cell_start[c + 1]); }
          if (x1 < g.gx) { const int c = cell_id(g, x1, yy); scan_range(cell_start[c], cell_start[c + 1]); }
        }
      }
    }
    if (cnt >= k && dirty) prune();
    if (x0 <= 0 && y0 <= 0 && x1 >= g.gx - 1 && y1 >= g.gy - 1) break;  // whole frame visited
    if (cnt == k) {
      // every unvisited point lies outside the visited block of cells: lower bound of its distance
      double lb = INFINITY;
      if (x0 > 0) lb = fmin(lb, q[0] - (g.x0 + (double)x0 * g.h));
      if (x1 < g.gx - 1) lb = fmin(lb, (g.x0 + (double)(x1 + 1) * g.h) - q[0]);
      if (y0 > 0) lb = fmin(lb, q[1] - (g.y0 + (double)y0 * g.h));
      if (y1 < g.gy - 1) lb = fmin(lb, (g.y0 + (double)(y1 + 1) * g.h) - q[1]);
      lb -= 1e-9 * g.h;  // slack for the rounding of the binning division
      if (lb > 0 && thr_d < lb * lb) break;
    }
  }
  // the last prune left the k best in ascending (distance, index) order
  const KnnKey* S = buf[team][sel];
  if (degree_init && lane == 0) degree_init[i] = k;        // out-degree: what rgnn_undirected_degree_preset starts from
  for (int a = lane; a < k; a += TEAM) {
    const int bi = S[a].i;
    const int64_t e = (int64_t)i * k + a;
    nbr[e] = bi;
    if (edge_index) { edge_index[e] = i; edge_index[E + e] = bi; }
    if (rel_pos) {                                         // relative_position of edge (i -> bi): graph.py:199-200
      double dx = X[(int64_t)i * DIM] - X[(int64_t)bi * DIM], dy = X[(int64_t)i * DIM + 1] - X[(int64_t)bi * DIM + 1];
      if (rel_undirected) { dx = fabs(dx); dy = fabs(dy); }
      *(float2*)(rel_pos + e * 2) = make_float2((float)dx, (float)dy);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kNN on SMALL frames (r05): brute force per frame, one wave per query, k-th smallest by a radix search on the distance bits.
//
// A nuScenes-shaped sweep has ~300 points: the whole frame is 5 candidates per lane, and k_knn_team's ring walk over the grid
// (cell bounds, run merging, ~900 scalar instructions per query) and its rank-counting prune (128 buffered keys x ~15
// instructions, at least once per query) cost more than evaluating every point of the frame (C3: 297 us per batch, 0.02 of HBM).
// Here a wave keeps the frame's points in registers (lane l holds points l, l + 64, ...: NV per lane), walks QPW queries of the
// frame and for each of them
//   * evaluates all distances with the KD-tree's arithmetic (dist2<DIM>: float64, dimension order, no FMA);
//   * finds the k-th smallest distance by a binary search over the BIT PATTERN of the keys from the top bit down (non-negative
//     doubles order like their bit patterns): count(key < X) is NV compares + ballots; the search stops at the first X that
//     separates exactly k keys (typically ~20 steps: the exponent bits, then until the k-th and (k + 1)-th distances differ).
//     Equal distances at the k-th place are broken by index exactly like every other kernel here (distance asc, index asc);
//   * compacts the k selected keys into LDS, ranks them there (k x k comparisons: the ranks are the sorted positions) and
//     writes the row -- same outputs as k_knn_team, bit-identical rows (tests/test_gpu_graph.py).
// ------------------------------------------------------------------------------------------------
constexpr int KF_QPW = 8;            // queries per wave (the candidates in registers are loaded once for all of them)
constexpr int KF_MAXK = 32;

template <int DIM, int NV>
__global__ __launch_bounds__(256) void k_knn_frame(int64_t n, int k, int n_frames, int waves_per_frame, const FrameGrid* __restrict__ frames,
                                                  const int64_t* __restrict__ frame_ptr, const int32_t* __restrict__ sorted_idx,
                                                  const double* __restrict__ sorted_pos, int32_t* __restrict__ nbr,
                                                  int64_t* __restrict__ edge_index, int32_t* __restrict__ status,
                                                  const double* __restrict__ X, float* __restrict__ rel_pos, int rel_undirected,
                                                  int32_t* __restrict__ degree_init) {
  __shared__ double s_d[4][KF_MAXK];
  __shared__ double s_x[4][KF_MAXK], s_y[4][KF_MAXK];      // the selected neighbours' first two coordinates (relative_position)
  __shared__ int32_t s_i[4][KF_MAXK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * 4 + wv;
  const int f = (int)(w / waves_per_frame);
  if (f >= n_frames) return;
  const int64_t beg = frame_ptr[f];
  const int nf = (int)(frame_ptr[f + 1] - beg);
  const int qs = (int)(w % waves_per_frame) * KF_QPW, qe = min(nf, qs + KF_QPW);
  if (qs >= nf) return;
  const int64_t E = n * (int64_t)k;
  if (nf <= k) {  // sklearn: "Expected n_neighbors < n_samples_fit"
    if (lane == 0) atomicOr(status, RGNN_STATUS_KNN_TOO_FEW_POINTS);
    for (int q = qs; q < qe; q++) {
      const int i = sorted_idx[beg + q];
      for (int a = lane; a < k; a += 64) {
        nbr[(int64_t)i * k + a] = -1;
        if (edge_index) { edge_index[(int64_t)i * k + a] = i; edge_index[E + (int64_t)i * k + a] = -1; }
      }
    }
    return;
  }
  // this lane's candidates: points lane, lane + 64, ... of the frame (cell order)
  double cp[NV][DIM];
  int ci[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const int j = lane + 64 * v;
    const int64_t pj = beg + min(j, nf - 1);
#pragma unroll
    for (int d = 0; d < DIM; d++) cp[v][d] = sorted_pos[pj * DIM + d];
    ci[v] = sorted_idx[pj];
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  double* ld = s_d[wv];
  double* lx = s_x[wv];
  double* ly = s_y[wv];
  int32_t* li = s_i[wv];
  // the query's data one query ahead (a wave walks its queries serially: without this every query opens with a global round trip)
  int i_next = sorted_idx[beg + qs];
  double qn[DIM];
#pragma unroll
  for (int d = 0; d < DIM; d++) qn[d] = sorted_pos[(beg + qs) * DIM + d];
  for (int q = qs; q < qe; q++) {
    const int i = i_next;
    double qp[DIM];
#pragma unroll
    for (int d = 0; d < DIM; d++) qp[d] = qn[d];
    {
      const int64_t pn = beg + min(q + 1, qe - 1);
      i_next = sorted_idx[pn];
#pragma unroll
      for (int d = 0; d < DIM; d++) qn[d] = sorted_pos[pn * DIM + d];
    }
    double dv[NV];
    unsigned long long key[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int j = lane + 64 * v;
      dv[v] = dist2<DIM>(qp, cp[v]);
      key[v] = (j < nf && j != q) ? (unsigned long long)__double_as_longlong(dv[v]) : ~0ull;      // (d2 >= 0: bit order = value order)
    }
    // the k-th smallest key: binary search over the bit pattern, 32 bits at a time (64-bit integer compares run at a quarter of the
    // rate: for the upper half key < X is a compare of the high words alone; the lower half is only reached when k-th and
    // (k + 1)-th distance agree in their exponent and top 20 mantissa bits, and then counts among the keys with that high word)
    unsigned long long T = 0;
    bool exact = false;
    {
      unsigned hi[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) hi[v] = (unsigned)(key[v] >> 32);
      unsigned th = 0;
      for (int b = 30; b >= 0; b--) {
        const unsigned xb = th | (1u << b);
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < NV; v++) cnt += __popcll(__ballot(hi[v] < xb));
        if (cnt == k) { th = xb; exact = true; break; }
        if (cnt < k) th = xb;
      }
      T = (unsigned long long)th << 32;
      if (!exact) {
        int below = 0;
        unsigned long long eqm[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) { below += __popcll(__ballot(hi[v] < th)); eqm[v] = __ballot(hi[v] == th); }
        unsigned tl = 0;
        for (int b = 31; b >= 0; b--) {
          const unsigned xb = tl | (1u << b);
          int cnt = below;
#pragma unroll
          for (int v = 0; v < NV; v++) cnt += __popcll(__ballot((unsigned)key[v] < xb) & eqm[v]);
          if (cnt == k) { tl = xb; exact = true; break; }
          if (cnt < k) tl = xb;
        }
        T |= tl;
      }
    }
    bool sel[NV];
    if (exact) {
#pragma unroll
      for (int v = 0; v < NV; v++) sel[v] = key[v] < T;
    } else {
      // T is the k-th smallest key itself and it occurs more than once among the candidates: all keys below it, and of the keys
      // equal to it the ones with the smallest indices
      int below = 0;
#pragma unroll
      for (int v = 0; v < NV; v++) { sel[v] = key[v] < T; below += __popcll(__ballot(sel[v])); }
      bool tie[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) tie[v] = key[v] == T;
      for (int r = below; r < k; r++) {
        int best = 0x7fffffff;
#pragma unroll
        for (int v = 0; v < NV; v++) if (tie[v]) best = min(best, ci[v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
#pragma unroll
        for (int v = 0; v < NV; v++) if (tie[v] && ci[v] == best) { tie[v] = false; sel[v] = true; }
      }
    }
    // compact the k selected keys into LDS, rank them (keys are unique: the ranks are the sorted positions), write the row
    int base = 0;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const unsigned long long m = __ballot(sel[v]);
      if (sel[v]) { const int at = base + __popcll(m & lt_mask); ld[at] = dv[v]; li[at] = ci[v]; lx[at] = cp[v][0]; ly[at] = cp[v][1]; }
      base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (entry a in lane a; the other entries come out of the registers by v_readlane -- a loop over LDS reads waited for the LDS
    //  twice per entry: 2 800 of a query's ~5 000 cycles)
    const int la = min(lane, k - 1);
    const double da = ld[la];
    const int ia = li[la];
    int rank = 0;
    {
      const int dlo = (int)(unsigned)__double_as_longlong(da), dhi = (int)(unsigned)(__double_as_longlong(da) >> 32);
#pragma unroll 4
      for (int o = 0; o < k; o++) {
        const unsigned blo = (unsigned)__builtin_amdgcn_readlane(dlo, o), bhi = (unsigned)__builtin_amdgcn_readlane(dhi, o);
        const double db = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
        const int ib = __builtin_amdgcn_readlane(ia, o);
        rank += (db < da || (db == da && ib < ia)) ? 1 : 0;
      }
    }
    if (lane < k) {
      const int64_t e = (int64_t)i * k + rank;
      nbr[e] = ia;
      if (edge_index) { edge_index[e] = i; edge_index[E + e] = ia; }
      if (rel_pos) {                                         // relative_position of edge (i -> ia): graph.py:199-200
        // (the cell-ordered copies ARE the rows of X: same float64 values, so the same differences as k_knn_team's X[i] - X[ia])
        double dx = qp[0] - lx[lane], dy = qp[1] - ly[lane];
        if (rel_undirected) { dx = fabs(dx); dy = fabs(dy); }
        *(float2*)(rel_pos + e * 2) = make_float2((float)dx, (float)dy);
      }
    }
    if (degree_init && lane == 0) degree_init[i] = k;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// undirected degree and CSR-by-target
// ------------------------------------------------------------------------------------------------
// |out U in| = |out| + |in| - |out n in|, two launches and no scratch: k_degree_init writes the out-degree, then a team of
// 16 lanes per row i walks its edges (i -> j): +1 for j (an in-edge of j), and -1 for i when row j holds i as well (the
// pair would otherwise count twice).  Integer atomics: the result does not depend on their order.  (Looking a mutual pair
// up from one end only and correcting both ends halves the row scans but adds an atomic per pair: measured slower.)
__global__ __launch_bounds__(256) void k_degree_init(const int32_t* __restrict__ rowptr, int64_t n,
                                                    int32_t* __restrict__ degree) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) degree[i] = rowptr[i + 1] - rowptr[i];
}

__global__ __launch_bounds__(256) void k_degree_edges(const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, int64_t n,
                                                     int32_t* __restrict__ degree) {
  // team of 16 lanes per row i, every lane with edges of its own (i -> j): lane t takes the row's entries t, t + 16, ... and
  // scans row j for i, eight entries per round with their loads issued together.  (One edge after the other with the team
  // searching row j together was a chain of three dependent loads per edge, twenty edges deep for a k = 20 row: 213 us on a
  // 64-frame batch.)
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  if (i >= n) return;                                        // (whole teams leave: 16 divides the block size)
  const int beg = rowptr[i], end = rowptr[i + 1];
  int mutual = 0;
  for (int e = beg + lane; e < end; e += 16) {
    const int j = col[e];
    atomicAdd(&degree[j], 1);
    const int jb = rowptr[j], last = rowptr[j + 1] - 1;
    bool found = false;
    for (int f = jb; f <= last && !found; f += 8) {
      int c[8];
#pragma unroll
      for (int v = 0; v < 8; v++) c[v] = col[min(f + v, last)];   // (past the row: its last entry again)
#pragma unroll
      for (int v = 0; v < 8; v++) found |= c[v] == (int)i;
    }
    mutual += found ? 1 : 0;
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) mutual += __shfl_xor(mutual, o, 16);
  if (lane == 0 && mutual) atomicSub(&degree[i], mutual);
}

// Undirected degree of a kNN graph from its CSR by target (r04): |N(i) u N^-1(i)| = k + (in-edges of i) - (in-edges whose source i
// lists itself), graph.py:93-96.  A team of 16 lanes per target: the target's own row nbr[i, 0..k) goes to LDS once and serves all
// its in-edges -- k_degree_edges walks the edges by SOURCE and fetches a different row of nbr for every one of them (169 us on a
// 64 x 3000, k = 20 batch).  No atomics: the in-degree is the segment length.
__global__ __launch_bounds__(256) void k_degree_knn_csr(const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ order, const int32_t* __restrict__ nbr, int64_t n, int k,
                                                       int32_t* __restrict__ degree) {
  __shared__ int row[16][64];
  const int team = threadIdx.x >> 4, t = threadIdx.x & 15;
  const int64_t p = (int64_t)blockIdx.x * 16 + team;
  if (p >= n) return;                                            // (whole teams leave)
  const int64_t i = order ? order[p] : p;
  for (int q = t; q < k; q += 16) row[team][q] = nbr[i * k + q];
  const int e0 = rowptr_t[p], e1 = rowptr_t[p + 1];
  int mutual = 0;
  for (int e = e0 + t; e < e1; e += 16) {                        // (the team's lanes read row[] they wrote themselves: same wave, in order)
    const int s_ = src[e];
    bool found = false;
    for (int q = 0; q < k; q++) found |= row[team][q] == s_;
    mutual += found ? 1 : 0;
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) mutual += __shfl_xor(mutual, o, 16);
  if (t == 0) degree[i] = k + (e1 - e0) - mutual;
}

__global__ __launch_bounds__(256) void k_count_i64(const int64_t* __restrict__ keys, int64_t n_keys,
                                                  const int32_t* __restrict__ rank, int32_t* __restrict__ counts) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_keys) atomicAdd(&counts[rank ? (int64_t)rank[keys[e]] : keys[e]], 1);
}

__global__ __launch_bounds__(256) void k_invert_permutation(const int32_t* __restrict__ order, int64_t n,
                                                           int32_t* __restrict__ rank) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) rank[order[p]] = (int32_t)p;
}

__global__ __launch_bounds__(256) void k_csr_fill(const int64_t* __restrict__ tgt, int64_t n_edges,
                                                 const int32_t* __restrict__ rank,
                                                 const int32_t* __restrict__ rowptr_t, int32_t* __restrict__ cursor,
                                                 int32_t* __restrict__ perm, int32_t* __restrict__ src_sorted = nullptr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t t = rank ? (int64_t)rank[tgt[e]] : tgt[e];
  const int slot = atomicSub(&cursor[t], 1) - 1;  // cursor = the segment's edge count (k_count_i64): counts down, no second memset
  perm[rowptr_t[t] + slot] = (int32_t)e;
  if (src_sorted != nullptr) src_sorted[rowptr_t[t] + slot] = (int32_t)tgt[e - n_edges];   // (unordered build: row 0 of edge_index lies n_edges entries in front of row 1)
}

// Edge-parallel stable ordering of the CSR-by-target segments: entry v (an edge id) of segment t moves to
// rowptr_t[t] + (number of smaller edge ids in the segment).  Balanced regardless of the in-degree distribution.
__global__ __launch_bounds__(256) void k_csr_rank(const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                 const int32_t* __restrict__ rank, const int32_t* __restrict__ rowptr_t,
                                                 const int32_t* __restrict__ perm_in, int32_t* __restrict__ perm,
                                                 int32_t* __restrict__ src_sorted) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int32_t v = perm_in[e];
  const int64_t tgt = edge_index[n_edges + v];
  const int64_t t = rank ? (int64_t)rank[tgt] : tgt;
  const int beg = rowptr_t[t], last = rowptr_t[t + 1] - 1;
  int r = 0;
  for (int b = beg; b <= last; b += 4) {         // four entries per round, their loads issued together
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) c[u] = perm_in[min(b + u, last)];
#pragma unroll
    for (int u = 0; u < 4; u++) r += (b + u <= last && c[u] < v) ? 1 : 0;
  }
  perm[beg + r] = v;
  src_sorted[beg + r] = (int32_t)edge_index[v];
}

// CSR by target of a batch of frames with UNIFORM out-degree (kNN graphs: edge e = i k + j has source i), ONE block per frame
// (r03): a frame's edges [k beg, k end) are a contiguous slice and all their targets lie in the frame, so its in-degree histogram
// (LDS atomics), the scan (rowptr_t = k beg + local prefix: no device-wide scan), the fill and the stable ordering inside every
// segment (edge ids ascending: rank counting, like k_csr_rank) are local.  Replaces two memsets, k_count_i64, one or two scan
// kernels, k_csr_fill and k_csr_rank; also leaves the in-degree per node and the number of nodes with incoming edges per frame
// behind (k_split_frames then builds the row lists of the conv layers in one more launch instead of three).  Same arrays.
constexpr int CF_THREADS = 1024;
constexpr int CF_LDS_NODES = 24 * 1024;             // 96 KB of LDS counters
__global__ __launch_bounds__(CF_THREADS) void k_csr_frames(const int64_t* __restrict__ edge_index, int64_t n_edges, int64_t k,
                                                         const int64_t* __restrict__ frame_ptr, int n_frames, int64_t n,
                                                         const int32_t* __restrict__ rank, int32_t* __restrict__ rowptr_t,
                                                         int32_t* __restrict__ perm_tmp, int32_t* __restrict__ perm,
                                                         int32_t* __restrict__ src_sorted, int32_t* __restrict__ in_degree,
                                                         int32_t* __restrict__ frame_nonempty) {
  extern __shared__ int32_t cf_cnt[];
  __shared__ int wsum[CF_THREADS / 64];
  __shared__ int nz_total;
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t beg = frame_ptr[f], end = frame_ptr[f + 1];
  const int nf = (int)(end - beg);
  const int64_t e0 = k * beg, e1 = k * end;
  const int64_t* tgt = edge_index + n_edges;
  for (int c = t; c < nf; c += CF_THREADS) cf_cnt[c] = 0;
  if (t == 0) nz_total = 0;
  __syncthreads();
  // ---- in-degree histogram over the segments (visiting order: segment of node v = rank[v], which lies in [beg, end))
  for (int64_t e = e0 + t; e < e1; e += CF_THREADS) {
    const int64_t v = tgt[e];
    atomicAdd(&cf_cnt[(rank ? (int64_t)rank[v] : v) - beg], 1);
  }
  __syncthreads();
  // ---- per node (node numbering) its in-degree + the frame's count of nodes with incoming edges
  int nz = 0;
  for (int64_t i = beg + t; i < end; i += CF_THREADS) {
    const int d = cf_cnt[(rank ? (int64_t)rank[i] : i) - beg];
    if (in_degree) in_degree[i] = d;
    nz += d > 0 ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o, 64);
  if (lane == 0 && nz) atomicAdd(&nz_total, nz);
  // ---- exclusive scan over the segments: thread t owns `per` consecutive ones
  const int per = (nf + CF_THREADS - 1) / CF_THREADS;
  const int lo = min(t * per, nf), hi = min(lo + per, nf);
  int sum = 0;
  for (int c = lo; c < hi; c++) sum += cf_cnt[c];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  if (t == 0 && frame_nonempty) frame_nonempty[f] = nz_total;
  int before = 0;
  for (int i = 0; i < w; i++) before += wsum[i];
  int run = (int)e0 + before + inc - sum;
  for (int c = lo; c < hi; c++) {
    const int d = cf_cnt[c];
    cf_cnt[c] = run;                                     // cursor of the fill phase
    rowptr_t[beg + c] = run;
    run += d;
  }
  if (f == n_frames - 1 && t == 0) rowptr_t[n] = (int32_t)n_edges;
  __syncthreads();
  // ---- fill (arbitrary order inside a segment) ...
  for (int64_t e = e0 + t; e < e1; e += CF_THREADS) {
    const int64_t v = tgt[e];
    const int at = atomicAdd(&cf_cnt[(rank ? (int64_t)rank[v] : v) - beg], 1);
    perm_tmp[at] = (int32_t)e;
  }
  __threadfence_block();
  __syncthreads();
  // ---- ... then edge ids ascending inside every segment (after the fill cf_cnt[c] is the END of segment c)
  for (int64_t q = e0 + t; q < e1; q += CF_THREADS) {
    const int32_t v = perm_tmp[q];
    const int64_t tv = tgt[v];
    const int c = (int)((rank ? (int64_t)rank[tv] : tv) - beg);
    const int sb = (c == 0) ? (int)e0 : cf_cnt[c - 1], last = cf_cnt[c] - 1;
    int r = 0;
    for (int b = sb; b <= last; b += 4) {
      int cc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) cc[u] = perm_tmp[min(b + u, last)];
#pragma unroll
      for (int u = 0; u < 4; u++) r += (b + u <= last && cc[u] < v) ? 1 : 0;
    }
    perm[sb + r] = v;
    src_sorted[sb + r] = (int32_t)edge_index[v];
  }
}

// CSR by target of a SYMMETRIC graph whose edges are grouped by their source (rows of a radius search: edge ids ascending
// with the source, neighbour ids ascending inside a row): node t's in-edges are the twins of its out-edges, so its in-degree
// is its row length (no histogram), and the edge (i -> t) sits in t's segment at the position of i in row t -- its rank
// among t's neighbours, counted like k_rank_rows does (edge ids ascend with the source, so this IS the stable order).
__global__ __launch_bounds__(256) void k_sym_degree(const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ rank,
                                                   int64_t n, int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n) return;
  if (t == n) { cnt[n] = 0; return; }
  cnt[rank ? rank[t] : t] = rowptr_src[t + 1] - rowptr_src[t];
}

__global__ __launch_bounds__(256) void k_sym_csr(const int64_t* __restrict__ edge_index, int64_t n_edges,
                                                const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ rank,
                                                const int32_t* __restrict__ rowptr_t, int32_t* __restrict__ perm,
                                                int32_t* __restrict__ src_sorted, int32_t* __restrict__ status,
                                                int32_t* __restrict__ own = nullptr) {
  // One thread per OUT-edge e = (t -> i) of row t; it places the twin (i -> t) in t's segment.  The sources of t's in-edges
  // ascend exactly like t's row does, so the twin's slot is rowptr_t[.] + (e - rowptr_src[t]): the lanes of a wave (edges of
  // one row) write neighbouring slots.  The twin's edge id is rowptr_src[i] + (rank of t in row i): a binary search (rows
  // ascend; most edges of a radar frame sit in clusters whose rows hold dozens of entries, and every probe is a dependent
  // L2 access -- the linear count measured 52 us on the C2 batch).
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t t = edge_index[e], i = edge_index[n_edges + e];
  if (e < rowptr_src[t] || e >= rowptr_src[t + 1] || rowptr_src[0] != 0 ||
      (int64_t)rowptr_t[rank ? rank[t] : t] + (e - rowptr_src[t]) >= n_edges) {    // the edge list is not grouped by these rows (e.g. a stale list under a
    if (status) atomicOr(status, RGNN_STATUS_NOT_SYMMETRIC);   // replayed step whose search found a different graph): write nothing
    return;
  }
  const int slot = rowptr_t[rank ? rank[t] : t] + (int)(e - rowptr_src[t]);
  if (perm == nullptr) {
    // (r03) the caller does not need the twin's edge id -- it reads the attributes of the OWN edge e = (t -> i) at the slot of
    // its twin (i -> t), e.g. because they are antisymmetric (relative_position, directed) -- so no search: the slot's source
    // and the own edge, 25 -> 8 us on the C2 batch.  (The symmetry claim is then the caller's: nothing here looks for the twin.)
    src_sorted[slot] = (int32_t)i;
    own[slot] = (int32_t)e;
    return;
  }
  const int beg = rowptr_src[i], end = rowptr_src[i + 1];
  int lo = beg, hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (edge_index[n_edges + mid] < t) lo = mid + 1; else hi = mid;
  }
  if (!(lo < end && edge_index[n_edges + lo] == t)) {   // (i -> t) is missing: the caller's symmetry claim is wrong
    if (status) atomicOr(status, RGNN_STATUS_NOT_SYMMETRIC);
    return;
  }
  perm[slot] = lo;
  src_sorted[slot] = (int32_t)i;
}

}  // namespace

// ================================================================================================
extern "C" int64_t rgnn_grid_workspace_bytes(int64_t n, int64_t n_frames, int32_t dim) {
  if (n < 0 || n_frames < 0 || (dim != 2 && dim != 4 && dim != 8)) return -1;
  return make_view(nullptr, n, n_frames, dim).total_bytes;
}

static int check_grid(const rgnn_grid* g) {
  RGNN_CHECK_ARG(g != nullptr, "null grid");
  RGNN_CHECK_ARG(g->dim == 2 || g->dim == 4 || g->dim == 8, "dim must be 2, 4 or 8");
  RGNN_CHECK_ARG(g->n >= 0 && g->n_frames >= 0, "negative sizes");
  RGNN_CHECK_ARG(g->n < (int64_t)1 << 30, "n too large for int32 cell ids");
  RGNN_CHECK_ARG(g->n == 0 || (g->X && g->frame_ptr && g->ws), "null pointers");
  RGNN_CHECK_ARG(g->ws_bytes >= rgnn_grid_workspace_bytes(g->n, g->n_frames, g->dim), "workspace too small");
  return RGNN_OK;
}

extern "C" int rgnn_grid_build(const rgnn_grid* g, double cell_size, double pts_per_cell, rgnn_stream_t stream) {
  return rgnn_grid_build_frames(g, cell_size, pts_per_cell, 0, stream);
}

extern "C" int rgnn_grid_build_frames(const rgnn_grid* g, double cell_size, double pts_per_cell, int64_t max_frame_points,
                                      rgnn_stream_t stream) {
  int rc = check_grid(g);
  if (rc) return rc;
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(cell_size > 0 || pts_per_cell > 0, "need cell_size > 0 or pts_per_cell > 0");
  hipStream_t s = (hipStream_t)stream;
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  // one block per frame does the whole binning (k_grid_frame) when the caller vouches for moderately sized frames; a frame
  // beyond the promise still works (its block falls back to the global cell table), it just takes that block longer
  // (frames whose cells do not fit the LDS table -- one 100 000-point cloud -- stay on the general path: a single block walking
  //  global counters would be slower than five launches that use the whole chip)
  if (max_frame_points > 0 && CELLS_PER_POINT * max_frame_points + CELLS_PER_FRAME <= (int64_t)GF_LDS_CELLS &&
      RGNN_ENV("RGNN_GRID_SPLIT") == nullptr) {
    const int64_t want = CELLS_PER_POINT * max_frame_points + CELLS_PER_FRAME;
    const int lds_cells = (int)(want < GF_LDS_CELLS ? want : GF_LDS_CELLS);
    static RgnnOncePerDevice attr_once;                     // (per kernel and device: common.h)
    if (attr_once.first()) {
      hipFuncSetAttribute((const void*)k_grid_frame<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS_CELLS * 4);
      hipFuncSetAttribute((const void*)k_grid_frame<4>, hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS_CELLS * 4);
      hipFuncSetAttribute((const void*)k_grid_frame<8>, hipFuncAttributeMaxDynamicSharedMemorySize, GF_LDS_CELLS * 4);
    }
    // frames whose points fit a thread's registers (GFR_PPT x 1024), two-column basis: the register-resident block (k_grid_frame_reg)
    const bool reg = g->dim == 2 && max_frame_points <= (int64_t)GFR_PPT * GF_THREADS && ((uintptr_t)g->X & 15) == 0 &&
                     (2 * (size_t)lds_cells + GFR_PPT * GF_THREADS) * 4 <= 150 * 1024 && RGNN_ENV("RGNN_GRID_NO_REG") == nullptr;
    if (reg) {
      static RgnnOncePerDevice reg_once;
      if (reg_once.first())
        hipFuncSetAttribute((const void*)k_grid_frame_reg, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      hipLaunchKernelGGL(k_grid_frame_reg, dim3((unsigned)g->n_frames), dim3(GF_THREADS), (2 * (size_t)lds_cells + GFR_PPT * GF_THREADS) * 4, s,
                         g->X, g->frame_ptr, (int)g->n_frames, v.frames, cell_size, pts_per_cell, v.cell_count, v.cell_start, v.n_cells,
                         lds_cells, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, v.point_cell, v.point_frame, v.point_rank);
    } else if (g->dim == 2)
      hipLaunchKernelGGL(k_grid_frame<2>, dim3((unsigned)g->n_frames), dim3(GF_THREADS), (size_t)lds_cells * 4, s, g->X, g->frame_ptr,
                         (int)g->n_frames, v.frames, cell_size, pts_per_cell, v.cell_count, v.cell_start, v.n_cells, lds_cells,
                         v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, v.point_cell, v.point_frame, v.point_rank);
    else if (g->dim == 4)
      hipLaunchKernelGGL(k_grid_frame<4>, dim3((unsigned)g->n_frames), dim3(GF_THREADS), (size_t)lds_cells * 4, s, g->X, g->frame_ptr,
                         (int)g->n_frames, v.frames, cell_size, pts_per_cell, v.cell_count, v.cell_start, v.n_cells, lds_cells,
                         v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, v.point_cell, v.point_frame, v.point_rank);
    else
      hipLaunchKernelGGL(k_grid_frame<8>, dim3((unsigned)g->n_frames), dim3(GF_THREADS), (size_t)lds_cells * 4, s, g->X, g->frame_ptr,
                         (int)g->n_frames, v.frames, cell_size, pts_per_cell, v.cell_count, v.cell_start, v.n_cells, lds_cells,
                         v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, v.point_cell, v.point_frame, v.point_rank);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  hipLaunchKernelGGL(k_frame_grid, dim3((unsigned)g->n_frames), dim3(FG_THREADS), 0, s, g->X, g->dim, g->frame_ptr, v.frames,
                     cell_size, pts_per_cell, v.cell_count, v.n_cells);
  hipLaunchKernelGGL(k_bin_count, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->X, g->dim, g->n, g->frame_ptr,
                     (int)g->n_frames, v.frames, v.point_cell, v.point_frame, v.cell_count);
  rc = rgnn_exclusive_scan_i32(v.cell_count, v.cell_start, v.n_cells, v.scan_tmp, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bin_scatter, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->n, v.point_cell, v.cell_start, v.cell_count, v.sorted_cell,
                     v.point_rank);
  if (g->dim == 2)
    hipLaunchKernelGGL(k_bin_place<2>, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->X, g->n, v.point_cell, v.point_frame, v.cell_start,
                       (const int32_t*)v.sorted_cell, v.sorted_idx, v.sorted_frame, v.sorted_pos, v.point_rank);
  else if (g->dim == 4)
    hipLaunchKernelGGL(k_bin_place<4>, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->X, g->n, v.point_cell, v.point_frame, v.cell_start,
                       (const int32_t*)v.sorted_cell, v.sorted_idx, v.sorted_frame, v.sorted_pos, v.point_rank);
  else
    hipLaunchKernelGGL(k_bin_place<8>, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->X, g->n, v.point_cell, v.point_frame, v.cell_start,
                       (const int32_t*)v.sorted_cell, v.sorted_idx, v.sorted_frame, v.sorted_pos, v.point_rank);
  hipLaunchKernelGGL(k_bin_cells, dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->n, v.point_cell, v.point_rank, v.sorted_cell);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

template <bool FILL>
static int launch_radius(const rgnn_grid* g, double r, int32_t* deg, const int32_t* rowptr, int32_t* col,
                         int64_t* edge_index, int64_t n_edges, int32_t* tmp, int32_t* status, rgnn_stream_t stream) {
  int rc = check_grid(g);
  if (rc) return rc;
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(r >= 0, "negative radius");
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  const double r2 = r * r;  // sklearn: reduced radius = r ** 2
  hipStream_t s = (hipStream_t)stream;
  int32_t* unsorted = FILL ? tmp : nullptr;
  int32_t* row_tmp = FILL ? tmp + n_edges : nullptr;
  if (g->dim == 2)
    hipLaunchKernelGGL((k_radius<2, FILL>), dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->n, g->X, v.point_cell,
                       v.point_frame, v.frames, v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, r2,
                       deg, rowptr, unsorted, row_tmp, v.nbr_cache, n_edges, status);
  else if (g->dim == 4)
    hipLaunchKernelGGL((k_radius<4, FILL>), dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->n, g->X, v.point_cell,
                       v.point_frame, v.frames, v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, r2,
                       deg, rowptr, unsorted, row_tmp, v.nbr_cache, n_edges, status);
  else
    hipLaunchKernelGGL((k_radius<8, FILL>), dim3(rgnn_blocks(g->n, 256)), dim3(256), 0, s, g->n, g->X, v.point_cell,
                       v.point_frame, v.frames, v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, r2,
                       deg, rowptr, unsorted, row_tmp, v.nbr_cache, n_edges, status);
  if (FILL)
    hipLaunchKernelGGL(k_rank_rows, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, rowptr, row_tmp, unsorted, n_edges,
                       col, edge_index, g->n, status != nullptr ? 1 : 0);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_radius_graph_count(const rgnn_grid* g, double r, int32_t* deg, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(g && (g->n == 0 || deg), "null deg");
  return launch_radius<false>(g, r, deg, nullptr, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int rgnn_radius_graph_fill(const rgnn_grid* g, double r, const int32_t* rowptr, int32_t* col,
                                      int64_t* edge_index, int64_t n_edges, int32_t* tmp, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(g && (g->n == 0 || (rowptr && (col || n_edges == 0))), "null rowptr/col");
  if (n_edges == 0) return RGNN_OK;
  RGNN_CHECK_ARG(tmp != nullptr, "null tmp (int32 [2 * n_edges])");
  return launch_radius<true>(g, r, nullptr, rowptr, col, edge_index, n_edges, tmp, nullptr, stream);
}

extern "C" int rgnn_radius_graph_fill_checked(const rgnn_grid* g, double r, const int32_t* rowptr, int32_t* col,
                                              int64_t* edge_index, int64_t n_edges, int32_t* tmp, int32_t* status,
                                              rgnn_stream_t stream) {
  RGNN_CHECK_ARG(g && (g->n == 0 || (rowptr && (col || n_edges == 0))), "null rowptr/col");
  RGNN_CHECK_ARG(status != nullptr, "null status");
  if (n_edges == 0 || g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(tmp != nullptr, "null tmp (int32 [2 * n_edges])");
  return launch_radius<true>(g, r, nullptr, rowptr, col, edge_index, n_edges, tmp, status, stream);
}

namespace {
__global__ __launch_bounds__(256) void k_rows_commit(const int32_t* __restrict__ rowptr_new, int64_t n, int64_t n_edges,
                                                    int32_t* __restrict__ committed, int32_t* __restrict__ status,
                                                    const int32_t* __restrict__ deg_new, int32_t* __restrict__ deg_committed) {
  if ((int64_t)rowptr_new[n] != n_edges) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, RGNN_STATUS_EDGE_COUNT_CHANGED);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    committed[i] = rowptr_new[i];
    if (deg_committed != nullptr && i < n) deg_committed[i] = deg_new[i];
  }
}
}  // namespace

extern "C" int rgnn_radius_graph_rows(const rgnn_grid* g, double r, const int32_t* rowptr, int32_t* col, int64_t* edge_index,
                                      int64_t n_edges, int32_t* tmp, int32_t* status, float* relative_position,
                                      int32_t undirected, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(g && (g->n == 0 || (rowptr && (col || n_edges == 0))), "null rowptr/col");
  int rc = check_grid(g);
  if (rc) return rc;
  if (g->n == 0 || (n_edges == 0 && status == nullptr)) return RGNN_OK;
  RGNN_CHECK_ARG(n_edges == 0 || tmp != nullptr, "null tmp (int32 [n_edges])");
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  hipStream_t s = (hipStream_t)stream;
  const double r2 = r * r;
  if (g->dim == 2)
    hipLaunchKernelGGL(k_radius_rows<2>, dim3(rgnn_blocks(g->n, 16)), dim3(256), 0, s, g->n, g->X, v.point_cell, v.point_frame, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_pos, r2, rowptr, v.nbr_cache, tmp, n_edges, status != nullptr ? 1 : 0, col,
                       edge_index, relative_position, undirected, status);
  else if (g->dim == 4)
    hipLaunchKernelGGL(k_radius_rows<4>, dim3(rgnn_blocks(g->n, 16)), dim3(256), 0, s, g->n, g->X, v.point_cell, v.point_frame, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_pos, r2, rowptr, v.nbr_cache, tmp, n_edges, status != nullptr ? 1 : 0, col,
                       edge_index, relative_position, undirected, status);
  else
    hipLaunchKernelGGL(k_radius_rows<8>, dim3(rgnn_blocks(g->n, 16)), dim3(256), 0, s, g->n, g->X, v.point_cell, v.point_frame, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_pos, r2, rowptr, v.nbr_cache, tmp, n_edges, status != nullptr ? 1 : 0, col,
                       edge_index, relative_position, undirected, status);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_radius_graph_rows_direct(const rgnn_grid* g, double r, const int32_t* rowptr_committed, int32_t* col,
                                             int64_t* edge_index, int64_t n_edges, int32_t* tmp, int32_t* status,
                                             float* relative_position, int32_t undirected, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(g && status && (g->n == 0 || (rowptr_committed && (col || n_edges == 0))), "null rowptr / col / status");
  int rc = check_grid(g);
  if (rc) return rc;
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(n_edges == 0 || tmp != nullptr, "null tmp (int32 [n_edges])");
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  hipStream_t s = (hipStream_t)stream;
  const double r2 = r * r;
#define RGNN_RD(D)                                                                                                                 \
  hipLaunchKernelGGL(k_radius_rows_direct<D>, dim3(rgnn_blocks(g->n, 16)), dim3(256), 0, s, g->n, g->X, v.point_cell, v.point_frame,   \
                     v.frames, v.cell_start, v.sorted_idx, v.sorted_pos, r2, rowptr_committed, tmp, n_edges, col, edge_index,      \
                     relative_position, undirected, status)
  if (g->dim == 2) RGNN_RD(2); else if (g->dim == 4) RGNN_RD(4); else RGNN_RD(8);
#undef RGNN_RD
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_radius_rows_commit(const int32_t* rowptr_new, int64_t n, int64_t n_edges, int32_t* rowptr_committed,
                                       int32_t* status, const int32_t* deg_new, int32_t* deg_committed, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && rowptr_new && rowptr_committed && status, "null pointers");
  RGNN_CHECK_ARG((deg_new == nullptr) == (deg_committed == nullptr), "deg_new and deg_committed go together");
  const int64_t nb = rgnn_blocks(n + 1, 256);
  hipLaunchKernelGGL(k_rows_commit, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, rowptr_new, n,
                     n_edges, rowptr_committed, status, deg_new, deg_committed);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_knn_graph(const rgnn_grid* g, int32_t k, int32_t* nbr, int64_t* edge_index, int32_t* status,
                              rgnn_stream_t stream) {
  return rgnn_knn_graph_attrs(g, k, nbr, edge_index, status, nullptr, 0, nullptr, stream);
}

extern "C" int rgnn_knn_graph_frames(const rgnn_grid* g, int32_t k, int64_t max_frame_points, int32_t* nbr, int64_t* edge_index,
                                     int32_t* status, float* relative_position, int32_t undirected, int32_t* degree_init,
                                     rgnn_stream_t stream) {
  int rc = check_grid(g);
  if (rc) return rc;
  RGNN_CHECK_ARG(k >= 1, "k must be >= 1");
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(nbr && status, "null outputs");
  static const int brute_max = RGNN_ENV("RGNN_KNN_FRAME_MAX") ? atoi(RGNN_ENV("RGNN_KNN_FRAME_MAX")) : 512;   // (1 000-point frames: 455 us against the grid walk's 280)
  // small frames (the grid was built by rgnn_grid_build_frames: frames are contiguous slices of the cell-ordered arrays): brute
  // force per frame, one wave per query (k_knn_frame); anything else: the grid walk
  if (max_frame_points < 1 || max_frame_points > brute_max || max_frame_points > 1024 || k > KF_MAXK || k < 3 || (g->dim != 2 && g->dim != 4))
    return rgnn_knn_graph_attrs(g, k, nbr, edge_index, status, relative_position, undirected, degree_init, stream);
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  hipStream_t s = (hipStream_t)stream;
  const int wpf = (int)((max_frame_points + KF_QPW - 1) / KF_QPW);
  const int64_t waves = g->n_frames * (int64_t)wpf;
#define RGNN_KNN_FRAME_GO(DIM, NV)                                                                                          \
  hipLaunchKernelGGL((k_knn_frame<DIM, NV>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, g->n, k, (int)g->n_frames, wpf,  \
                     v.frames, (const int64_t*)g->frame_ptr, v.sorted_idx, v.sorted_pos, nbr, edge_index, status,               \
                     (const double*)g->X, relative_position, (int)undirected, degree_init)
  if (g->dim == 2) {
    if (max_frame_points <= 320) RGNN_KNN_FRAME_GO(2, 5); else if (max_frame_points <= 512) RGNN_KNN_FRAME_GO(2, 8); else RGNN_KNN_FRAME_GO(2, 16);
  } else {
    if (max_frame_points <= 320) RGNN_KNN_FRAME_GO(4, 5); else if (max_frame_points <= 512) RGNN_KNN_FRAME_GO(4, 8); else RGNN_KNN_FRAME_GO(4, 16);
  }
#undef RGNN_KNN_FRAME_GO
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_knn_graph_attrs(const rgnn_grid* g, int32_t k, int32_t* nbr, int64_t* edge_index, int32_t* status,
                                    float* relative_position, int32_t undirected, int32_t* degree_init, rgnn_stream_t stream) {
  int rc = check_grid(g);
  if (rc) return rc;
  RGNN_CHECK_ARG(k >= 1, "k must be >= 1");
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(nbr && status, "null outputs");
  const size_t lds = (size_t)k * KNN_THREADS * 12;
  if (lds > 160 * 1024) {
    rgnn_set_error("rgnn_knn_graph: k = %d needs %zu B of LDS per block (max 163840)", k, lds);
    return RGNN_ERR_UNSUPPORTED;
  }
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  hipStream_t s = (hipStream_t)stream;
  // A team of 64 lanes per query (k_knn_team) unless k is too large for its buffer, or k <= 2 on a large batch (measured,
  // 192 k points: k = 1 86 us with one thread per query against 232 us; k = 20 1 234 against 391 us; k = 40 4 863 against
  // 601 us; one 3 000-point frame, k = 10: 292 against 19 us).  RGNN_KNN_TEAM = 16 / 32 / 64 / 0 overrides (tools/knn_bench.py).
  const char* team_e = RGNN_ENV("RGNN_KNN_TEAM");
  const int team_env = team_e ? atoi(team_e) : ((k <= 2 && g->n > 32768) ? 0 : 64);
  int team = (team_env == 16 || team_env == 32 || team_env == 64) && k <= KNN_CAP - team_env ? team_env : 0;
  if ((relative_position || degree_init) && team == 0) {     // (the extra outputs are written by the team kernel only)
    RGNN_CHECK_ARG(k <= KNN_CAP - 64, "relative_position / degree_init need k <= KNN_CAP - 64 (the team kernel)");
    team = 64;
  }
  if (team) {
#define RGNN_KNN_TEAM_GO(DIM, TEAM)                                                                                   \
  hipLaunchKernelGGL((k_knn_team<DIM, TEAM>), dim3(rgnn_blocks(g->n, 256 / TEAM)), dim3(256), 0, s, g->n, k, v.frames, \
                     v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, nbr, edge_index, status,      \
                     (const double*)g->X, relative_position, (int)undirected, degree_init)
    if (g->dim == 2) {
      if (team == 16) RGNN_KNN_TEAM_GO(2, 16); else if (team == 32) RGNN_KNN_TEAM_GO(2, 32); else RGNN_KNN_TEAM_GO(2, 64);
    } else if (g->dim == 4) {
      if (team == 16) RGNN_KNN_TEAM_GO(4, 16); else if (team == 32) RGNN_KNN_TEAM_GO(4, 32); else RGNN_KNN_TEAM_GO(4, 64);
    } else {
      if (team == 16) RGNN_KNN_TEAM_GO(8, 16); else if (team == 32) RGNN_KNN_TEAM_GO(8, 32); else RGNN_KNN_TEAM_GO(8, 64);
    }
#undef RGNN_KNN_TEAM_GO
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  if (g->dim == 2) {
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_knn<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_knn<2>, dim3(rgnn_blocks(g->n, KNN_THREADS)), dim3(KNN_THREADS), lds, s, g->n, k, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, nbr, edge_index, status);
  } else if (g->dim == 4) {
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_knn<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_knn<4>, dim3(rgnn_blocks(g->n, KNN_THREADS)), dim3(KNN_THREADS), lds, s, g->n, k, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, nbr, edge_index, status);
  } else {
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)k_knn<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_knn<8>, dim3(rgnn_blocks(g->n, KNN_THREADS)), dim3(KNN_THREADS), lds, s, g->n, k, v.frames,
                       v.cell_start, v.sorted_idx, v.sorted_frame, v.sorted_cell, v.sorted_pos, nbr, edge_index, status);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_grid_cell_order(const rgnn_grid* g, int32_t* order, rgnn_stream_t stream) {
  int rc = check_grid(g);
  if (rc) return rc;
  if (g->n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(order, "null order");
  GridView v = make_view(g->ws, g->n, g->n_frames, g->dim);
  hipMemcpyAsync(order, v.sorted_idx, 4 * g->n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return RGNN_OK;
}

extern "C" int rgnn_undirected_degree(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* in_deg_tmp,
                                      int32_t* degree_out, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0, "negative n");
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rowptr && degree_out, "null pointers");
  // row-parallel, so E (= rowptr[n], device-side) is never needed on the host; in_deg_tmp is no longer used
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_degree_init, dim3(rgnn_blocks(n, 256)), dim3(256), 0, s, rowptr, n, degree_out);
  hipLaunchKernelGGL(k_degree_edges, dim3(rgnn_blocks(n * 16, 256)), dim3(256), 0, s, rowptr, col, n, degree_out);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_undirected_degree_preset(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* degree,
                                             rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0, "negative n");
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rowptr && col && degree, "null pointers");
  hipLaunchKernelGGL(k_degree_edges, dim3(rgnn_blocks(n * 16, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, n, degree);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_knn_degree_from_csr(const int32_t* rowptr_t, const int32_t* src_sorted, const int32_t* node_order, const int32_t* nbr,
                                        int64_t n, int32_t k, int32_t* degree, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && k >= 1 && k <= 64, "bad sizes (k <= 64)");
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(rowptr_t && src_sorted && nbr && degree, "null pointers");
  hipLaunchKernelGGL(k_degree_knn_csr, dim3(rgnn_blocks(n, 16)), dim3(256), 0, (hipStream_t)stream, rowptr_t, src_sorted, node_order, nbr, n,
                     (int)k, degree);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int64_t rgnn_csr_by_target_tmp_bytes(int64_t n, int64_t n_edges) {
  return rgnn_align_up(4 * (n + 1), 256) + rgnn_align_up(rgnn_scan_tmp_bytes(n + 1), 256) + rgnn_align_up(4 * n_edges, 256);
}

extern "C" int rgnn_source_rowptr(const int64_t* edge_index, int64_t n, int64_t n_edges, const int32_t* rank,
                                  int32_t* rowptr_s, void* tmp, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_edges >= 0 && n_edges < ((int64_t)1 << 31), "bad sizes");
  RGNN_CHECK_ARG(rowptr_s && tmp && (n_edges == 0 || edge_index), "null pointers");
  hipStream_t s = (hipStream_t)stream;
  int32_t* cnt = (int32_t*)tmp;
  void* scan_tmp = (char*)tmp + rgnn_align_up(4 * (n + 1), 256);
  hipMemsetAsync(cnt, 0, rgnn_align_up(4 * (n + 1), 256), s);   // (the aligned size of the region: ONE fill launch, not body + tail)
  if (n_edges > 0)
    hipLaunchKernelGGL(k_count_i64, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index, n_edges, rank, cnt);
  return rgnn_exclusive_scan_i32(cnt, rowptr_s, n, scan_tmp, stream);
}

extern "C" int rgnn_invert_permutation(const int32_t* order, int64_t n, int32_t* rank, rgnn_stream_t stream) {
  if (n == 0) return RGNN_OK;
  RGNN_CHECK_ARG(order && rank, "null pointers");
  hipLaunchKernelGGL(k_invert_permutation, dim3(rgnn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, order, n, rank);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_csr_by_target_symmetric(const int64_t* edge_index, const int32_t* rowptr_src, int64_t n,
                                            int64_t n_edges, const int32_t* target_rank, int32_t* rowptr_t,
                                            int32_t* src_sorted, int32_t* perm, void* tmp, int32_t* status,
                                            rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_edges >= 0 && n_edges < ((int64_t)1 << 31), "bad sizes");
  RGNN_CHECK_ARG(rowptr_t && tmp && rowptr_src, "null pointers");
  hipStream_t s = (hipStream_t)stream;
  int32_t* cnt = (int32_t*)tmp;
  void* scan_tmp = (char*)tmp + rgnn_align_up(4 * (n + 1), 256);
  hipLaunchKernelGGL(k_sym_degree, dim3(rgnn_blocks(n + 1, 256)), dim3(256), 0, s, rowptr_src, target_rank, n, cnt);
  int rc = rgnn_exclusive_scan_i32(cnt, rowptr_t, n, scan_tmp, stream);
  if (rc) return rc;
  if (n_edges > 0) {
    RGNN_CHECK_ARG(edge_index && src_sorted && perm, "null edge arrays");
    hipLaunchKernelGGL(k_sym_csr, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index, n_edges, rowptr_src,
                       target_rank, rowptr_t, perm, src_sorted, status);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_csr_by_target_symmetric_own(const int64_t* edge_index, const int32_t* rowptr_src, int64_t n, int64_t n_edges,
                                                const int32_t* target_rank, int32_t* rowptr_t, int32_t* src_sorted,
                                                int32_t* own_edge, void* tmp, int32_t* status, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && n_edges >= 0 && n_edges < ((int64_t)1 << 31), "bad sizes");
  RGNN_CHECK_ARG(rowptr_t && tmp && rowptr_src, "null pointers");
  hipStream_t s = (hipStream_t)stream;
  int32_t* cnt = (int32_t*)tmp;
  void* scan_tmp = (char*)tmp + rgnn_align_up(4 * (n + 1), 256);
  hipLaunchKernelGGL(k_sym_degree, dim3(rgnn_blocks(n + 1, 256)), dim3(256), 0, s, rowptr_src, target_rank, n, cnt);
  int rc = rgnn_exclusive_scan_i32(cnt, rowptr_t, n, scan_tmp, stream);
  if (rc) return rc;
  if (n_edges > 0) {
    RGNN_CHECK_ARG(edge_index && src_sorted && own_edge, "null edge arrays");
    hipLaunchKernelGGL(k_sym_csr, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index, n_edges, rowptr_src, target_rank,
                       rowptr_t, (int32_t*)nullptr, src_sorted, status, own_edge);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

extern "C" int rgnn_csr_by_target_frames(const int64_t* edge_index, int64_t n, int64_t n_edges, int64_t k,
                                         const int64_t* frame_ptr, int64_t n_frames, int64_t max_frame_points,
                                         const int32_t* target_rank, int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm,
                                         int32_t* perm_tmp, int32_t* in_degree, int32_t* frame_nonempty, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && k >= 1 && n_edges == n * k && n_edges < ((int64_t)1 << 31), "edges must be n * k (uniform out-degree)");
  RGNN_CHECK_ARG(n_frames >= 1 && max_frame_points >= 1 && max_frame_points <= CF_LDS_NODES, "frames too large for the LDS histogram");
  RGNN_CHECK_ARG(edge_index && frame_ptr && rowptr_t && src_sorted && perm && perm_tmp, "null pointers");
  static RgnnOncePerDevice attr_once;                       // (per kernel and device: common.h)
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)k_csr_frames, hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS_NODES * 4);
  }
  hipLaunchKernelGGL(k_csr_frames, dim3((unsigned)n_frames), dim3(CF_THREADS), (size_t)max_frame_points * 4, (hipStream_t)stream,
                     edge_index, n_edges, k, frame_ptr, (int)n_frames, n, target_rank, rowptr_t, perm_tmp, perm, src_sorted, in_degree,
                     frame_nonempty);
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

static int csr_by_target_impl(const int64_t* edge_index, int64_t n, int64_t n_edges, const int32_t* target_rank,
                              int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, void* tmp, rgnn_stream_t stream, bool ordered);
extern "C" int rgnn_csr_by_target(const int64_t* edge_index, int64_t n, int64_t n_edges, const int32_t* target_rank,
                                  int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, void* tmp,
                                  rgnn_stream_t stream) {
  return csr_by_target_impl(edge_index, n, n_edges, target_rank, rowptr_t, src_sorted, perm, tmp, stream, true);
}
// ... without the stable ordering inside the segments (r06): the places the fill's atomics hand out ARE the result -- for reductions that
// do not depend on the order of a segment's edges (max), whose callers save the ranking pass (59 us of a k = 20 batch of 64 x 3000 points).
extern "C" int rgnn_csr_by_target_unordered(const int64_t* edge_index, int64_t n, int64_t n_edges, const int32_t* target_rank,
                                            int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, void* tmp,
                                            rgnn_stream_t stream) {
  return csr_by_target_impl(edge_index, n, n_edges, target_rank, rowptr_t, src_sorted, perm, tmp, stream, false);
}
static int csr_by_target_impl(const int64_t* edge_index, int64_t n, int64_t n_edges, const int32_t* target_rank,
                              int32_t* rowptr_t, int32_t* src_sorted, int32_t* perm, void* tmp, rgnn_stream_t stream, bool ordered) {
  RGNN_CHECK_ARG(n >= 0 && n_edges >= 0 && n_edges < ((int64_t)1 << 31), "bad sizes");
  RGNN_CHECK_ARG(rowptr_t && tmp, "null pointers");
  hipStream_t s = (hipStream_t)stream;
  int32_t* cnt = (int32_t*)tmp;
  void* scan_tmp = (char*)tmp + rgnn_align_up(4 * (n + 1), 256);
  int32_t* perm_unsorted = (int32_t*)((char*)scan_tmp + rgnn_align_up(rgnn_scan_tmp_bytes(n + 1), 256));
  hipMemsetAsync(cnt, 0, rgnn_align_up(4 * (n + 1), 256), s);   // (the aligned size of the region: ONE fill launch, not body + tail)
  if (n_edges > 0) {
    RGNN_CHECK_ARG(edge_index && src_sorted && perm, "null edge arrays");
    hipLaunchKernelGGL(k_count_i64, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index + n_edges, n_edges,
                       target_rank, cnt);
  }
  int rc = rgnn_exclusive_scan_i32(cnt, rowptr_t, n, scan_tmp, stream);
  if (rc) return rc;
  if (n_edges > 0) {
    if (!ordered) {
      hipLaunchKernelGGL(k_csr_fill, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index + n_edges, n_edges,
                         target_rank, rowptr_t, cnt, perm, src_sorted);
    } else {
      hipLaunchKernelGGL(k_csr_fill, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index + n_edges, n_edges,
                         target_rank, rowptr_t, cnt, perm_unsorted, (int32_t*)nullptr);
      hipLaunchKernelGGL(k_csr_rank, dim3(rgnn_blocks(n_edges, 256)), dim3(256), 0, s, edge_index, n_edges, target_rank,
                         rowptr_t, perm_unsorted, perm, src_sorted);
    }
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}

// Byte offsets of the cell order (sorted_idx) and of its inverse (point_rank) inside the grid workspace: callers that keep the
// workspace alive read them in place instead of copying (rgnn_grid_cell_order) / inverting (rgnn_invert_permutation) them.
extern "C" int rgnn_grid_order_offsets(int64_t n, int64_t n_frames, int32_t dim, int64_t* order_offset, int64_t* rank_offset) {
  RGNN_CHECK_ARG(n >= 0 && n_frames >= 0 && (dim == 2 || dim == 4 || dim == 8) && order_offset && rank_offset, "bad arguments");
  GridView v = make_view(nullptr, n, n_frames, dim);
  *order_offset = (char*)v.sorted_idx - (char*)nullptr;
  *rank_offset = (char*)v.point_rank - (char*)nullptr;
  return RGNN_OK;
}
