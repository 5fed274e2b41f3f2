// Hybrid head, backward on the rows that can carry a gradient (hybrid.py, _compact_rows): under the
// one-negative losses (adaptive BPR / hinge, WARP) the score gradient of a batch of B samples and K
// negatives is zero outside the B positives and ONE negative per sample.  Two small kernels replace
// the torch glue that found those 2 B rows and gathered the saved activations down to them
// (arange / abs / argmax / mul / add / stack / four index_selects: ~17 launches, ~250 us per step
// at B = 65,536): which rows, in one launch; the rows of up to four saved tensors, in one launch.
#include "common.h"

namespace collie {
namespace {

constexpr int kBlock = 256;

// sample b: k* = the first k with the largest |g[B + k B + b]| (torch.argmax: first maximum; all
// zero -> k = 0, a zero row).  Rows 2b, 2b + 1 of the compact batch = the positive and that negative.
__global__ __launch_bounds__(kBlock) void active_rows_kernel(
    const float* __restrict__ g, int64_t B, int K, const int64_t* __restrict__ users,
    const int64_t* __restrict__ items, int64_t* __restrict__ act, float* __restrict__ g_out,
    int64_t* __restrict__ users_out, int64_t* __restrict__ items_out) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  int best = 0;
  float best_abs = fabsf(g[B + b]);
  for (int k = 1; k < K; ++k) {
    const float a = fabsf(g[B + static_cast<int64_t>(k) * B + b]);
    if (a > best_abs) {
      best_abs = a;
      best = k;
    }
  }
  const int64_t neg_row = B + static_cast<int64_t>(best) * B + b;
  act[2 * b] = b;
  act[2 * b + 1] = neg_row;
  g_out[2 * b] = g[b];
  g_out[2 * b + 1] = g[neg_row];
  users_out[2 * b] = users[b];
  users_out[2 * b + 1] = users[neg_row];
  items_out[2 * b] = items[b];
  items_out[2 * b + 1] = items[neg_row];
}

// calculate_loss's expansion of a batch into (1 + K) B (user, item) pairs (base_pipeline.py:603-622:
// users.repeat(K + 1), cat(pos, neg.T.flatten())): pair e = k B + b is (users[b], k == 0 ? pos[b] :
// neg[b, k - 1]), ids widened to int64.
__global__ __launch_bounds__(kBlock) void pair_ids_kernel(const collie_mf_batch b,
                                                          int64_t* __restrict__ all_users,
                                                          int64_t* __restrict__ all_items) {
  const int64_t B = b.batch;
  const int64_t m = B * (1 + b.num_neg);
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= m) return;
  const int64_t k = e / B, s = e - k * B;
  all_users[e] = load_idx(b.users, b.users_dtype, s);
  all_items[e] = k == 0 ? load_idx(b.pos_items, b.pos_dtype, s)
                        : load_idx(b.neg_items, b.neg_dtype, s * b.neg_stride_b + (k - 1) * b.neg_stride_k);
}

struct GatherSet {
  int n;
  const float* src[4];
  float* out[4];
  int64_t ld[4];
  int width[4];
};

// 16 lanes per output row; every tensor of the set in turn (16-byte pieces when VEC)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const GatherSet s,
                                                             const int64_t* __restrict__ idx,
                                                             int64_t m_out, int64_t m_src,
                                                             int32_t* __restrict__ status) {
  constexpr int LPR = 16;
  const int lig = threadIdx.x % LPR;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * (kBlock / LPR);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * (kBlock / LPR) + threadIdx.x / LPR; j < m_out;
       j += ngroups) {
    int64_t r = idx[j];
    const bool ok = r >= 0 && r < m_src;
    if (!ok) {
      if (lig == 0 && status) atomicOr(status, 1);
      r = 0;
    }
    for (int t = 0; t < s.n; ++t) {
      const float* src = s.src[t] + r * s.ld[t];
      float* dst = s.out[t] + j * s.width[t];
      if (VEC) {
        for (int c = lig; c < s.width[t] / 4; c += LPR) {
          float4 v = reinterpret_cast<const float4*>(src)[c];
          if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          reinterpret_cast<float4*>(dst)[c] = v;
        }
      } else {
        for (int c = lig; c < s.width[t]; c += LPR) dst[c] = ok ? src[c] : 0.0f;
      }
    }
  }
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_active_rows(const float* g, int64_t batch, int32_t num_neg,
                                  const int64_t* users, const int64_t* items, int64_t* act,
                                  float* g_out, int64_t* users_out, int64_t* items_out,
                                  collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(batch >= 0 && num_neg >= 1, "active_rows: bad sizes");
  if (batch == 0) return 0;
  COLLIE_REQUIRE(g && users && items && act && g_out && users_out && items_out,
                 "active_rows: NULL pointer");
  hipLaunchKernelGGL(active_rows_kernel, dim3(static_cast<unsigned>((batch + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, g, batch, num_neg, users, items, act, g_out, users_out,
                     items_out);
  return check_launch("active_rows_kernel");
}

extern "C" int collie_gather_rows(int32_t n, const float* const* srcs, const int64_t* lds,
                                  const int32_t* widths, float* const* outs, const int64_t* idx,
                                  int64_t m_out, int64_t m_src, int32_t* status,
                                  collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(n >= 1 && n <= 4, "gather_rows: 1..4 tensors per call");
  COLLIE_REQUIRE(m_out >= 0 && m_src >= 1, "gather_rows: bad sizes");
  if (m_out == 0) return 0;
  COLLIE_REQUIRE(srcs && lds && widths && outs && idx, "gather_rows: NULL pointer");
  GatherSet s;
  s.n = n;
  bool vec = true;
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  for (int t = 0; t < 4; ++t) {
    const bool on = t < n;
    s.src[t] = on ? srcs[t] : nullptr;
    s.out[t] = on ? outs[t] : nullptr;
    s.ld[t] = on ? lds[t] : 0;
    s.width[t] = on ? widths[t] : 0;
    if (!on) continue;
    COLLIE_REQUIRE(s.src[t] && s.out[t] && s.width[t] >= 1 && s.ld[t] >= s.width[t],
                   "gather_rows: tensor %d: bad pointer / width / ld", t);
    vec = vec && s.width[t] % 4 == 0 && s.ld[t] % 4 == 0 && al16(s.src[t]) && al16(s.out[t]);
  }
  int64_t blocks = (m_out + kBlock / 16 - 1) / (kBlock / 16);
  if (blocks > 32768) blocks = 32768;
  if (vec)
    hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, s, idx, m_out, m_src, status);
  else
    hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, s, idx, m_out, m_src, status);
  return check_launch("gather_rows_kernel");
}

namespace collie {
int validate_batch(const collie_mf_batch* b);
}

extern "C" int collie_pair_ids(const collie_mf_batch* b, int64_t* all_users, int64_t* all_items,
                               collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_batch(b)) return -1;
  COLLIE_REQUIRE(all_users && all_items, "pair_ids: NULL output");
  const int64_t m = b->batch * (1 + static_cast<int64_t>(b->num_neg));
  hipLaunchKernelGGL(pair_ids_kernel, dim3(static_cast<unsigned>((m + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, *b, all_users, all_items);
  return check_launch("pair_ids_kernel");
}
