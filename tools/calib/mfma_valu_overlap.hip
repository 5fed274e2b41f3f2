// Do fp32-input MFMAs (v_mfma_f32_32x32x2_f32) and ordinary vector ALU instructions of the SAME
// SIMD overlap?  One wave per SIMD (and, second run, two): (a) a dependent chain of MFMAs, (b) a
// stream of independent v_cmp + v_addc pairs, (c) both interleaved in program order (1 MFMA, 10 VALU).
// If the matrix pipe and the vector ALU are separate resources, (c) ~ max(a, b); if not, (c) ~ a + b.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kIters = 4096;

template <int MODE, int NACC = 1>  // 1 = mfma only, 2 = valu only, 3 = interleaved
__global__ __launch_bounds__(512) void k(float* out, long long* cycles, float x, float y) {
  unsigned cc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  unsigned c0 = 0, c1 = 0;
  float s = x + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 4
  for (int it = 0; it < kIters; ++it) {
    if (MODE & 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < 10 && NACC == 9; ++j)   // sign-bit form: subtract, shift the sign in
        asm volatile("v_sub_f32 %1, %2, %3\n\tv_alignbit_b32 %0, %0, %1, 31"
                     : "+v"(cc[0]), "=&v"(cc[1 + j % 4]) : "v"(s), "v"(y));
#pragma unroll
      for (int j = 0; j < 10 && NACC != 9; ++j)   // exactly the evaluator's pair: compare -> VCC, add with carry
        asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                     : "+v"(cc[j % (NACC == 9 ? 1 : NACC)]) : "v"(s), "v"(y) : "vcc");
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = clock64();
  if (threadIdx.x % 64 == 0) cycles[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
  float r = c0 + c1;
  for (int i = 0; i < 8; ++i) r += cc[i];
  for (int i = 0; i < 16; ++i) r += acc[i];
  if (r == 1234.5f) out[0] = r;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8 * 8);
  for (int threads : {256, 512}) {
    long long h[3][8];
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[0], cyc, 64, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[1], cyc, 64, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[2], cyc, 64, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD: per iteration (clock64 ticks): mfma only %.1f, 10 cmp+addc only %.1f, interleaved %.1f\n",
           threads / 256, double(h[0][0]) / kIters, double(h[1][0]) / kIters, double(h[2][0]) / kIters);
    hipLaunchKernelGGL((k<2, 5>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[1], cyc, 64, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<3, 5>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[2], cyc, 64, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k<2, 9>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[0], cyc, 64, hipMemcpyDeviceToHost);
    printf("   10 x (v_sub_f32 + v_alignbit_b32) only: %.1f\n", double(h[0][0]) / kIters);
    hipLaunchKernelGGL((k<3, 9>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, 2.f); hipMemcpy(h[0], cyc, 64, hipMemcpyDeviceToHost);
    printf("   mfma + 10 x (v_sub_f32 + v_alignbit_b32): %.1f\n", double(h[0][0]) / kIters);
    printf("   with 5 independent accumulators: 10 cmp+addc only %.1f, interleaved %.1f\n",
           double(h[1][0]) / kIters, double(h[2][0]) / kIters);
  }
  return 0;
}
