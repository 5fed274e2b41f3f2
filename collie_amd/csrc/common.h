// Shared host/device helpers for libcollie_hip.so (gfx950 only: wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/collie_hip.h"

namespace collie {

constexpr int kWave = 64;

// ---- error reporting (thread-local text behind collie_last_error) -------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define COLLIE_CHECK_HIP(expr)                                  \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return ::collie::hip_fail(_e, #expr); \
  } while (0)

#define COLLIE_REQUIRE(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      ::collie::set_error(__VA_ARGS__); \
      return -1;                       \
    }                                  \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return 0;
}

// ---- workspace carving ----------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ int64_t load_idx(const void* p, int dtype, int64_t i) {
  return dtype == COLLIE_I64 ? static_cast<const int64_t*>(p)[i]
                             : static_cast<int64_t>(static_cast<const int32_t*>(p)[i]);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// xor-butterfly sum over `width` consecutive lanes (width power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<kWave>(v); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}

// Philox4x32-10 (Salmon et al., SC'11): the counter-based generator of the device sampler and of
// the dropout masks; known-answer vectors are checked through oracle/sampler_oracle.py.
struct Philox {
  uint32_t c[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    c1 = static_cast<uint32_t>(p1);
    c3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}

// torch.sigmoid in fp32: 1 / (1 + exp(-x))
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace collie
