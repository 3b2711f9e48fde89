// Shared helpers of librgnn (gfx950 only).  No torch, no STL containers across the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>

#include "../../include/rgnn.h"

#define RGNN_WAVE 64
// rows per column-statistics panel: what the dense epilogues write (linear_common.h) and the BatchNorm kernels read (norm.hip)
#define RGNN_STAT_PANEL_ROWS 128

void rgnn_set_error(const char* fmt, ...);

#define RGNN_CHECK_ARG(cond, msg)                      \
  do {                                                 \
    if (!(cond)) {                                     \
      rgnn_set_error("%s: %s", __func__, msg);        \
      return RGNN_ERR_INVALID_ARGUMENT;                \
    }                                                  \
  } while (0)

#define RGNN_CHECK_LAUNCH()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      rgnn_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return RGNN_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

static inline int64_t rgnn_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline unsigned rgnn_blocks(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// Environment switches (experiments, A/B tools): read ONCE per call site and cached -- a dense launch used to call getenv a dozen
// times.  rgnn_env_reload() (rgnn.h) bumps the epoch: every site reads its variable again on its next use, so tools
// that flip a switch inside one process (tools/x3_bench, bench.py's fp32 line) call it after changing the environment.
extern std::atomic<int> g_rgnn_env_epoch;
#define RGNN_ENV(NAME)                                                                     \
  ([]() -> const char* {                                                                   \
    static std::atomic<int> ep__{-1};                                                      \
    static std::atomic<const char*> v__{nullptr};                                          \
    const int e__ = g_rgnn_env_epoch.load(std::memory_order_acquire);                      \
    if (ep__.load(std::memory_order_acquire) != e__) {                                     \
      v__.store(getenv(NAME), std::memory_order_relaxed);                                  \
      ep__.store(e__, std::memory_order_release);                                          \
    }                                                                                      \
    return v__.load(std::memory_order_relaxed);                                            \
  }())

// Profiling hook (bench.py): rgnn_profile_next_launch() arms a pair of HIP events that the next instrumented entry
// point (rgnn_linear_fwd, rgnn_mpnn_aggregate) records immediately around its kernel launch, on the launch stream.
void rgnn_prof_begin(hipStream_t s);
void rgnn_prof_end(hipStream_t s);

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) is per DEVICE: a "done" flag per kernel and device, so that a process that drives
// several GPUs sets it on each of them (r04 kept one flag per process -- right for the one-process-per-GPU launch contract only).
#include <atomic>
struct RgnnOncePerDevice {
  std::atomic<unsigned long long> seen{0};
  bool first() {                                  // true exactly once per (this object, current device); devices >= 64 always true
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const unsigned long long bit = 1ull << dev;
    return !(seen.fetch_or(bit, std::memory_order_relaxed) & bit);
  }
};
