// Calibration of rocprofv3's WRITE_SIZE (and FETCH_SIZE) on gfx950 against KNOWN byte counts, in the store patterns librgnn uses
// (MI355X_MICROARCH.md, HBM section: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 tools/attic/write_size_probe.hip -o /tmp/write_size_probe
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/wsp_w -o w -- /tmp/write_size_probe
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/wsp_f -o f -- /tmp/write_size_probe
// Every kernel moves exactly BYTES = 512 MiB (larger than L2 + Infinity Cache), once.
//   k_store16       float4 stores, lane-contiguous (the dense epilogue's row pieces when a lane holds 4 columns)
//   k_store4        4-byte stores, lane-contiguous: 256 B per wave instruction
//   k_store4_rows   4-byte stores, 32 lanes x 4 B into one row, the other half-wave into another row 1 856 B away (the dense
//                   epilogue's accumulator layout: one column per lane, rows of 464 floats)
//   k_store16_nt    float4 streaming (nt) stores (the edge kernel's aggregated rows)
//   k_store_rows464 1 856-byte rows written as 116 float4 per row by 116 lanes of two waves (row-subset epilogue pattern)
//   k_load16        float4 loads of the same bytes (FETCH_SIZE reference point: reported = 1/2 of the bytes on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr long BYTES = 512l << 20;

__global__ __launch_bounds__(256) void k_store16(float4* p, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_store4(float* p, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) p[i] = (float)i;
}
__global__ __launch_bounds__(256) void k_store4_rows(float* p, long rows) {       // rows of 464 floats; a wave writes 32 columns of 2 rows per store
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r2 = wave; r2 < rows / 2; r2 += waves)
    for (int c = 0; c < 464; c += 32)
      if (c + (lane & 31) < 464) p[(2 * r2 + (lane >> 5)) * 464 + c + (lane & 31)] = (float)c;
}
__global__ __launch_bounds__(256) void k_store16_nt(float* p, long n4) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n4; i += (long)gridDim.x * blockDim.x * 4) {
    __builtin_nontemporal_store(1.f, p + i); __builtin_nontemporal_store(2.f, p + i + 1);
    __builtin_nontemporal_store(3.f, p + i + 2); __builtin_nontemporal_store((float)i, p + i + 3);
  }
}
__global__ __launch_bounds__(128) void k_store_rows464(float4* p, long rows) {    // one row of 116 float4 per 128-thread block pass
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    if (threadIdx.x < 116) p[r * 116 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, (float)r);
}
__global__ __launch_bounds__(256) void k_load16(const float4* p, long n16, float* sink) {
  float a = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 1.2345e-30f) sink[0] = a;
}

int main() {
  float* buf; float* sink;
  CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 64));
  const long rows = BYTES / (464 * 4);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_store16, dim3(4096), dim3(256), 0, 0, (float4*)buf, BYTES / 16);
    hipLaunchKernelGGL(k_store4, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4);
    hipLaunchKernelGGL(k_store4_rows, dim3(4096), dim3(256), 0, 0, buf, rows);
    hipLaunchKernelGGL(k_store16_nt, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4);
    hipLaunchKernelGGL(k_store_rows464, dim3(8192), dim3(128), 0, 0, (float4*)buf, rows);
    hipLaunchKernelGGL(k_load16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, BYTES / 16, sink);
    CK(hipDeviceSynchronize());
  }
  printf("bytes per kernel: %ld (k_store4_rows / k_store_rows464: %ld)\n", BYTES, rows * 464 * 4);
  return 0;
}
