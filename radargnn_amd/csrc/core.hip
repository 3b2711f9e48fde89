// Version / error plumbing and the device-wide exclusive scan.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void rgnn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

extern "C" void rgnn_profile_next_launch(void* ev_start, void* ev_stop) {
  g_ev_start = (hipEvent_t)ev_start;
  g_ev_stop = (hipEvent_t)ev_stop;
}
void rgnn_prof_begin(hipStream_t s) {
  if (g_ev_start) { hipEventRecord(g_ev_start, s); g_ev_start = nullptr; }
}
void rgnn_prof_end(hipStream_t s) {
  if (g_ev_stop) { hipEventRecord(g_ev_stop, s); g_ev_stop = nullptr; }
}

// Side streams of the host layer (ops.independent_stream): plain non-blocking HIP streams on the current device, created HERE rather
// than taken from torch's pool of 32 (a pool stream cannot be given back, and candidates that turn out to share a hardware queue
// with a stream already in use would eat the pool until its streams repeat).
extern "C" int rgnn_stream_create(rgnn_stream_t* out) {
  RGNN_CHECK_ARG(out != nullptr, "null pointer");
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { rgnn_set_error("hipStreamCreateWithFlags: %s", hipGetErrorString(e)); return RGNN_ERR_LAUNCH; }
  *out = (rgnn_stream_t)s;
  return RGNN_OK;
}
extern "C" int rgnn_stream_destroy(rgnn_stream_t stream) {
  if (stream == nullptr) return RGNN_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { rgnn_set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return RGNN_ERR_LAUNCH; }
  return RGNN_OK;
}

extern "C" const char* rgnn_version(void) { return "rgnn 0.1 (gfx950)"; }
extern "C" const char* rgnn_last_error(void) { return g_err; }
std::atomic<int> g_rgnn_env_epoch{0};
extern "C" void rgnn_env_reload(void) { g_rgnn_env_epoch.fetch_add(1, std::memory_order_acq_rel); }

// ------------------------------------------------------------------------------------------------
// Exclusive scan of int32, reduce-then-scan in three launches.  2048 items per 256-thread block;
// each thread owns 8 consecutive items (two int4 loads), wave-level prefix via __shfl_up (64 lanes).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// block-wide inclusive scan of one value per thread; returns inclusive prefix, *total = block sum
__device__ __forceinline__ int block_inclusive_scan(int v, int* total) {
  __shared__ int wave_sums[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (lane == 63) wave_sums[w] = inc;
  __syncthreads();
  int add = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; i++) {
    int s = wave_sums[i];
    if (i < w) add += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return inc + add;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int32_t* __restrict__ in, int64_t n,
                                                                 int32_t* __restrict__ block_sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) s += in[base + i];
  int tot;
  block_inclusive_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of the block sums in place (loops with a carry for > 2048 blocks)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_offsets(int32_t* __restrict__ block_sums, int64_t nb) {
  int carry = 0;
  for (int64_t start = 0; start < nb; start += SCAN_THREADS) {
    int64_t i = start + threadIdx.x;
    int v = (i < nb) ? block_sums[i] : 0;
    int tot;
    int inc = block_inclusive_scan(v, &tot);
    if (i < nb) block_sums[i] = carry + inc - v;
    carry += tot;
  }
}

// RAW: block_offsets holds the plain block sums (no k_scan_offsets pass): every block adds up the sums before it itself --
// for the few hundred blocks of a batch that is cheaper than a third launch in the chain.
template <bool RAW>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                            int64_t n, const int32_t* __restrict__ block_offsets) {
  int before = 0;
  if (RAW) {
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += block_offsets[b];
    block_inclusive_scan(part, &before);
  }
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int tot;
  int inc = block_inclusive_scan(s, &tot);
  int run = (RAW ? before : block_offsets[blockIdx.x]) + inc - s;  // exclusive prefix of this thread's first item
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    run += v[i];
    if (base + i < n) out[base + i + 1] = run;  // out[j+1] = inclusive(j) -> out is the exclusive scan, out[n] = total
  }
}
// short inputs (one frame: 3 000 rows, 6 000 cells): ONE block walks the tiles with a carry -- one launch instead of two
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_small(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
  int carry = 0;
  if (threadIdx.x == 0) out[0] = 0;
  for (int64_t tile = 0; tile < n; tile += SCAN_TILE) {
    const int64_t base = tile + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      v[i] = (base + i < n) ? in[base + i] : 0;
      s += v[i];
    }
    int tot;
    const int inc = block_inclusive_scan(s, &tot);
    int run = carry + inc - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
      run += v[i];
      if (base + i < n) out[base + i + 1] = run;
    }
    carry += tot;
  }
}
}  // namespace

extern "C" int64_t rgnn_scan_tmp_bytes(int64_t n) {
  int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  return rgnn_align_up((nb + 1) * 4, 256);
}

extern "C" int rgnn_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* tmp, rgnn_stream_t stream) {
  RGNN_CHECK_ARG(n >= 0 && out != nullptr, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipMemsetAsync(out, 0, 4, s);
    return RGNN_OK;
  }
  RGNN_CHECK_ARG(in != nullptr && tmp != nullptr, "null input/tmp");
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb <= 8) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, n);
    RGNN_CHECK_LAUNCH();
    return RGNN_OK;
  }
  int32_t* sums = (int32_t*)tmp;
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, n, sums);
  if (nb <= 2048) {
    hipLaunchKernelGGL(k_scan_final<true>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, n, sums);
  } else {
    hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(SCAN_THREADS), 0, s, sums, nb);
    hipLaunchKernelGGL(k_scan_final<false>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, n, sums);
  }
  RGNN_CHECK_LAUNCH();
  return RGNN_OK;
}
