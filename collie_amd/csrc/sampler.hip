// Device-side batch assembly + negative sampling: produces the loaders' batch
// ((users int32 [B], pos int32 [B]), neg int64 [B, K]) in HBM, so that an epoch of the training
// hot path never waits for the host.  Semantics of Interactions.__getitem__ / _negative_sample
// (collie/interactions/datasets.py:375-445): approximate mode = K uniform draws per sample
// (:439-443); exact mode = sequential rejection of candidates the user has interacted with or
// the sample already holds, a per-sample budget of rejections, and the reference's fill rule once
// the budget is spent (:407-425).  Random stream: Philox4x32-10, counter = (global sample position,
// draw number), key = seed -- a batch does not depend on the grid or on how an epoch is cut into
// batches.
#include "common.h"

namespace collie {
namespace {

constexpr int kBlock = 256;

// uniform item id in [0, num_items): high 64 bits of (64 random bits) x num_items
__device__ __forceinline__ int64_t draw_item(uint64_t seed, uint64_t position, uint32_t draw,
                                             uint64_t num_items) {
  const Philox r = philox4x32_10(static_cast<uint32_t>(position), static_cast<uint32_t>(position >> 32),
                                 draw, 0u, static_cast<uint32_t>(seed),
                                 static_cast<uint32_t>(seed >> 32));
  const uint64_t r64 = static_cast<uint64_t>(r.c[0]) | (static_cast<uint64_t>(r.c[1]) << 32);
  return static_cast<int64_t>(__umul64hi(r64, num_items));
}

// is `item` one of the user's positives (CSR row sorted ascending)?
__device__ __forceinline__ bool csr_contains(const int32_t* __restrict__ indices, int64_t lo,
                                             int64_t hi, int64_t item) {
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (indices[mid] < item) lo = mid + 1; else hi = mid;
  }
  return lo < end && indices[lo] == item;
}

struct SampleParams {
  const int32_t* rows;
  const int32_t* cols;
  const int64_t* order;
  int64_t start, batch;
  const int64_t* indptr;
  const int32_t* indices;
  int64_t num_items;
  int32_t num_neg, max_tries;
  uint64_t seed, position_base;
  int32_t* users_out;
  int32_t* pos_out;
  int64_t* neg_out;
};

// The exact rule, one thread, one sample: the definition every other path must reproduce.
__device__ void sample_exact_sequential(const SampleParams& p, int64_t lo, int64_t hi,
                                        uint64_t position, int64_t* neg) {
  const uint64_t N = static_cast<uint64_t>(p.num_items);
  const int K = p.num_neg;
  uint32_t d = 0;
  int got = 0, checked = 0;
  while (got < K) {
    int64_t cand = draw_item(p.seed, position, d++, N);
    for (;;) {
      bool clash = csr_contains(p.indices, lo, hi, cand);
      for (int j = 0; j < got && !clash; ++j) clash = neg[j] == cand;
      if (!clash) break;
      if (checked >= p.max_tries) {  // budget spent: datasets.py:411-419
        const int left = K - got - 1;
        for (int j = 0; j < left; ++j) neg[got++] = draw_item(p.seed, position, d++, N);
        break;
      }
      cand = draw_item(p.seed, position, d++, N);
      ++checked;
    }
    neg[got++] = cand;
  }
}

// approximate mode (datasets.py:439-443): one thread per (sample, k), coalesced stores
__global__ __launch_bounds__(kBlock) void sample_approx_kernel(const SampleParams p) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int K = p.num_neg;
  if (g < p.batch) {
    const int64_t idx = p.order ? p.order[p.start + g] : p.start + g;
    p.users_out[g] = p.rows[idx];
    p.pos_out[g] = p.cols[idx];
  }
  if (g >= p.batch * K) return;
  const int64_t t = g / K;
  const uint32_t k = static_cast<uint32_t>(g - t * K);
  const uint64_t position = p.position_base + static_cast<uint64_t>(p.start + t);
  p.neg_out[g] = draw_item(p.seed, position, k, static_cast<uint64_t>(p.num_items));
}

// exact mode: kLps lanes per sample evaluate kLps consecutive candidates of the sample's stream at
// once (one binary search each, in parallel).  Until the rejection budget is spent the sequential
// rule is "the first K candidates of the stream that are neither positives of the user nor equal
// to an earlier non-positive candidate", which a window can decide with a ballot and a prefix
// count; a sample whose rejections exceed the budget is redone by one lane with the sequential
// rule, so the output equals sample_exact_sequential() bit for bit in every case.
constexpr int kLps = 16;
constexpr int kGroups = kBlock / kLps;
constexpr int kMaxNegLds = 64;  // accepted negatives kept in LDS (larger K: sequential rule)

__global__ __launch_bounds__(kBlock) void sample_exact_kernel(const SampleParams p) {
  __shared__ int64_t accepted[kGroups][kMaxNegLds];
  const int lane = lane_id();
  const int grp_in_block = threadIdx.x / kLps;
  const int lig = lane % kLps;
  const int gshift = (lane / kLps) * kLps;  // first lane of this group inside the wave
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kGroups + grp_in_block;
  const bool on = t < p.batch;              // uniform inside a lane group
  const int K = p.num_neg;
  const uint64_t N = static_cast<uint64_t>(p.num_items);
  int64_t lo = 0, hi = 0;
  uint64_t position = 0;
  int64_t* neg = p.neg_out;
  if (on) {
    const int64_t idx = p.order ? p.order[p.start + t] : p.start + t;
    const int32_t u = p.rows[idx];
    if (lig == 0) {
      p.users_out[t] = u;
      p.pos_out[t] = p.cols[idx];
    }
    lo = p.indptr[u];
    hi = p.indptr[u + 1];
    position = p.position_base + static_cast<uint64_t>(p.start + t);
    neg = p.neg_out + t * K;
  }
  int got = 0, rejected = 0;
  bool fallback = on && K > kMaxNegLds;
  uint32_t base_d = 0;
  int64_t* mine = accepted[grp_in_block];
  // every lane of the wave stays in the loop until all four groups are done (ballots are wave-wide)
  while (__any(on && !fallback && got < K)) {
    const bool active = on && !fallback && got < K;
    const int64_t cand = draw_item(p.seed, position, base_d + lig, N);
    bool bad = active ? csr_contains(p.indices, lo, hi, cand) : true;
    const bool nonpos = !bad;
    // duplicate of an accepted negative or of an earlier non-positive candidate of this window
    if (active && !bad)
      for (int j = 0; j < got; ++j) bad |= mine[j] == cand;
#pragma unroll
    for (int j = 0; j < kLps - 1; ++j) {
      const int64_t oc = __shfl(cand, gshift + j, kWave);
      const bool onp = __shfl(nonpos ? 1 : 0, gshift + j, kWave) != 0;
      if (j < lig && onp && oc == cand) bad = true;
    }
    const unsigned long long wave_ok = __ballot(active && !bad);
    const uint32_t ok = static_cast<uint32_t>(wave_ok >> gshift) & 0xffffu;  // this group's lanes
    const int before = __popc(ok & ((1u << lig) - 1u));
    const int need = K - got;
    const int n_ok = __popc(ok);
    // candidates consumed by the sequential rule: up to and including the need-th valid one
    int consumed = kLps;
    if (n_ok >= need) {
      uint32_t m = ok;
      for (int i = 1; i < need; ++i) m &= m - 1;  // drop the need-1 lowest set bits
      consumed = __ffs(static_cast<int>(m));      // 1-based position of the need-th valid lane
    }
    if (active) {
      const int rej_here = consumed - min(n_ok, need);
      rejected += rej_here;
      if (rejected > p.max_tries) {
        fallback = true;  // the budget rule decides this sample: redo it sequentially
      } else {
        if (!bad && before < need) {
          mine[got + before] = cand;
          neg[got + before] = cand;
        }
        got += min(n_ok, need);
        base_d += kLps;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the LDS list is read by the next window
  }
  if (fallback && lig == 0) sample_exact_sequential(p, lo, hi, position, neg);
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_sample_batch(const int32_t* rows, const int32_t* cols, int64_t nnz,
                                   const int64_t* order, int64_t start, int64_t batch,
                                   const int64_t* indptr, const int32_t* indices,
                                   int64_t num_items, int32_t num_neg, int32_t max_tries,
                                   uint64_t seed, uint64_t position_base, int32_t* users_out,
                                   int32_t* pos_out, int64_t* neg_out, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(rows && cols && users_out && pos_out && neg_out, "sample_batch: NULL pointer");
  COLLIE_REQUIRE(nnz >= 1 && batch >= 1 && start >= 0 && start + batch <= nnz,
                 "sample_batch: batch [%lld, %lld) is outside the %lld interactions",
                 static_cast<long long>(start), static_cast<long long>(start + batch),
                 static_cast<long long>(nnz));
  COLLIE_REQUIRE(num_items >= 1 && num_neg >= 1, "sample_batch: bad num_items / num_negative_samples");
  COLLIE_REQUIRE((indptr == nullptr) == (indices == nullptr),
                 "sample_batch: indptr and indices must be given together");
  SampleParams p;
  p.rows = rows;
  p.cols = cols;
  p.order = order;
  p.start = start;
  p.batch = batch;
  p.indptr = indptr;
  p.indices = indices;
  p.num_items = num_items;
  p.num_neg = num_neg;
  p.max_tries = max_tries;
  p.seed = seed;
  p.position_base = position_base;
  p.users_out = users_out;
  p.pos_out = pos_out;
  p.neg_out = neg_out;
  if (indptr == nullptr || max_tries <= 0) {
    const int64_t threads = batch * num_neg;
    hipLaunchKernelGGL(sample_approx_kernel,
                       dim3(static_cast<unsigned>((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       stream, p);
  } else {
    hipLaunchKernelGGL(sample_exact_kernel,
                       dim3(static_cast<unsigned>((batch + kGroups - 1) / kGroups)), dim3(kBlock), 0,
                       stream, p);
  }
  return check_launch("sample_batch_kernel");
}
