// FETCH_SIZE calibration for K1's access pattern: random 512-byte rows of a
// 4 GiB buffer, each row read exactly once, by wave-instructions that take 64-byte, 128-byte or
// whole-row (512-byte) contiguous pieces per row; plus a coalesced streaming read as the control.
// Every kernel reads exactly `bytes` = 4 GiB.  Run each under `rocprofv3 --pmc FETCH_SIZE` and
// compare the counter (KiB) with the known byte count:
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int64_t kRows = int64_t{1} << 23;  // x 512 B = 4 GiB
constexpr int kRowF4 = 32;                    // float4 per row

__device__ __forceinline__ int64_t perm(int64_t i) { return (i * 2654435761ll + 12345) & (kRows - 1); }

// LPR lanes share one row per instruction: piece = LPR * 16 bytes; 64 / LPR rows per instruction
template <int LPR>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ buf, float* out) {
  constexpr int G = 64 / LPR, CH = kRowF4 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  const int64_t wave = (int64_t{blockIdx.x} * 256 + threadIdx.x) / 64;
  const int64_t nwaves = int64_t{gridDim.x} * 4;
  float acc = 0.f;
  for (int64_t r0 = wave * G; r0 < kRows; r0 += nwaves * G) {
    const float4* row = buf + perm(r0 + grp) * kRowF4 + lig;
    float4 v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = row[i * LPR];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc += v[i].x + v[i].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, float* out) {
  const int64_t n = kRows * kRowF4, stride = int64_t{gridDim.x} * 256;
  float acc = 0.f;
  for (int64_t i = int64_t{blockIdx.x} * 256 + threadIdx.x; i < n; i += stride) acc += buf[i].x;
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  float4* buf;
  float* out;
  const size_t bytes = size_t(kRows) * kRowF4 * 16;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto timed = [&](const char* name, auto launch) {
    launch();  // warm-up (also a profiled dispatch: every dispatch reads the same 4 GiB)
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.3f ms  %7.1f GB/s  (%zu bytes per dispatch)\n", name, ms, bytes / ms / 1e6, bytes);
  };
  timed("stream_16B_per_lane", [&] { hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, buf, out); });
  timed("gather_rows_64B_pieces", [&] { hipLaunchKernelGGL(gather_kernel<4>, dim3(4096), dim3(256), 0, 0, buf, out); });
  timed("gather_rows_128B_pieces", [&] { hipLaunchKernelGGL(gather_kernel<8>, dim3(4096), dim3(256), 0, 0, buf, out); });
  timed("gather_rows_512B_pieces", [&] { hipLaunchKernelGGL(gather_kernel<32>, dim3(4096), dim3(256), 0, 0, buf, out); });
  hipDeviceSynchronize();
  return 0;
}
