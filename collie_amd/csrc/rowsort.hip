// K3a: deterministic "coalesce" of the looked-up row ids (torch/optim/_functional.py:44).
//
// Stable LSD radix sort (8 bits per pass, only as many passes as the table's row count needs) of
// (row id -> contribution code) pairs, followed by segment-head extraction.  The stable order
// inside a segment is the order of the batch, so the floating-point sums the optimizer kernels
// form over a segment are bit-reproducible run to run.  Integer atomics (LDS/global counters)
// are used for histograms only; their results do not depend on arrival order.
#include "train_ws.h"

namespace collie {
namespace {

constexpr int kBlock = 256;
constexpr int kItems = kSortTile / kBlock;  // 8 keys per thread
static_assert(kItems * kBlock == kSortTile, "tile shape");

struct KeySrc {
  const void* a;  // first `na` keys (users, or positive items)
  const void* b;  // remaining keys: negatives, element (i, k) at b[i * sb + k * sk]
  int64_t na;
  int64_t sb, sk;
  int64_t rows;  // ids are clamped into [0, rows): a bad id can no longer address memory
  int dta, dtb;
  int K;
  int contig;  // negatives are a contiguous [B, K] block
};

__device__ __forceinline__ uint32_t first_key(const KeySrc& s, int64_t j) {
  int64_t id;
  if (j < s.na) {
    id = load_idx(s.a, s.dta, j);
  } else {
    const int64_t jj = j - s.na;
    if (s.contig) {
      id = load_idx(s.b, s.dtb, jj);
    } else {
      const uint32_t q = static_cast<uint32_t>(jj) / static_cast<uint32_t>(s.K);
      const uint32_t r = static_cast<uint32_t>(jj) - q * static_cast<uint32_t>(s.K);
      id = load_idx(s.b, s.dtb, static_cast<int64_t>(q) * s.sb + static_cast<int64_t>(r) * s.sk);
    }
  }
  if (id < 0) id = 0;
  if (id >= s.rows) id = s.rows - 1;
  return static_cast<uint32_t>(id);
}

template <bool FIRST>
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const KeySrc src,
                                                            const uint32_t* __restrict__ keys_in,
                                                            int64_t m, int shift,
                                                            uint32_t* __restrict__ hist,
                                                            int64_t ntiles,
                                                            uint32_t* __restrict__ totals) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = base + i * kBlock + threadIdx.x;
    if (idx < m) {
      const uint32_t key = FIRST ? first_key(src, idx) : keys_in[idx];
      atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  const uint32_t c = h[threadIdx.x];
  hist[static_cast<int64_t>(threadIdx.x) * ntiles + blockIdx.x] = c;
  if (c) atomicAdd(&totals[threadIdx.x], c);
}

// one workgroup per digit: exclusive scan of that digit's per-tile counts, offset by the number
// of keys with a smaller digit.
__global__ __launch_bounds__(kBlock) void radix_scan_kernel(uint32_t* __restrict__ hist,
                                                            int64_t ntiles,
                                                            const uint32_t* __restrict__ totals) {
  __shared__ uint32_t sh[kBlock];
  const int d = blockIdx.x;
  const int tid = threadIdx.x;
  sh[tid] = tid < d ? totals[tid] : 0u;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  const uint32_t digit_base = sh[0];
  __syncthreads();

  uint32_t* row = hist + static_cast<int64_t>(d) * ntiles;
  const int64_t per = (ntiles + kBlock - 1) / kBlock;
  const int64_t lo = tid * per;
  const int64_t hi = lo + per < ntiles ? lo + per : ntiles;
  uint32_t local = 0;
  for (int64_t i = lo; i < hi; ++i) local += row[i];
  sh[tid] = local;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 thread sums
  for (int o = 1; o < kBlock; o <<= 1) {
    const uint32_t add = tid >= o ? sh[tid - o] : 0u;
    __syncthreads();
    sh[tid] += add;
    __syncthreads();
  }
  uint32_t running = digit_base + sh[tid] - local;
  for (int64_t i = lo; i < hi; ++i) {
    const uint32_t c = row[i];
    row[i] = running;
    running += c;
  }
}

template <bool FIRST>
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(
    const KeySrc src, const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    int64_t m, int shift, const uint32_t* __restrict__ hist, int64_t ntiles,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ uint32_t wcount[kWaves][256];
  __shared__ uint32_t gbase[256];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid / kWave;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) wcount[i][tid] = 0;
  __syncthreads();

  // wave-striped tile: wave w owns keys [w*512, (w+1)*512) of the tile, iteration-major
  const int64_t wbase = static_cast<int64_t>(blockIdx.x) * kSortTile + w * (kWave * kItems);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t key[kItems], val[kItems], rnk[kItems];
  volatile uint32_t* mycount = wcount[w];
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = wbase + i * kWave + lane;
    const bool valid = idx < m;
    key[i] = valid ? (FIRST ? first_key(src, idx) : keys_in[idx]) : 0xffffffffu;
    val[i] = valid ? vals_in[idx] : 0u;
    const uint32_t digit = (key[i] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (digit >> bit) & 1u;
      const unsigned long long bm = __ballot(one);
      peers &= one ? bm : ~bm;
    }
    const uint32_t before = __popcll(peers & lt_mask);
    uint32_t old = 0;
    if (valid) old = mycount[digit];
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) mycount[digit] = old + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rnk[i] = old + before;
  }
  __syncthreads();
  {
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
      const uint32_t c = wcount[i][tid];
      wcount[i][tid] = run;
      run += c;
    }
    gbase[tid] = hist[static_cast<int64_t>(tid) * ntiles + blockIdx.x];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = wbase + i * kWave + lane;
    if (idx < m) {
      const uint32_t digit = (key[i] >> shift) & 255u;
      const uint32_t pos = gbase[digit] + wcount[w][digit] + rnk[i];
      keys_out[pos] = key[i];
      vals_out[pos] = val[i];
    }
  }
}

// ---- segment heads ------------------------------------------------------------------------------
__device__ __forceinline__ bool is_head(const uint32_t* keys, int64_t idx, int64_t m) {
  return idx < m && (idx == 0 || keys[idx] != keys[idx - 1]);
}

__global__ __launch_bounds__(kBlock) void seg_count_kernel(const uint32_t* __restrict__ keys,
                                                           int64_t m,
                                                           uint32_t* __restrict__ blk_heads) {
  __shared__ uint32_t sh[kBlock / kWave];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
  uint32_t cnt = 0;
#pragma unroll
  for (int i = 0; i < kItems; ++i)
    cnt += __popcll(__ballot(is_head(keys, base + i * kBlock + threadIdx.x, m)));
  if (lane_id() == 0) sh[threadIdx.x / kWave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int i = 0; i < kBlock / kWave; ++i) t += sh[i];
    blk_heads[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void seg_scan_kernel(uint32_t* __restrict__ blk_heads,
                                                          int64_t ntiles, int64_t m,
                                                          uint32_t* __restrict__ nseg,
                                                          uint32_t* __restrict__ seg_start) {
  __shared__ uint32_t sh[kBlock];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < ntiles; c0 += kBlock) {
    const int64_t i = c0 + tid;
    const uint32_t v = i < ntiles ? blk_heads[i] : 0u;
    sh[tid] = v;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
      const uint32_t add = tid >= o ? sh[tid - o] : 0u;
      __syncthreads();
      sh[tid] += add;
      __syncthreads();
    }
    if (i < ntiles) blk_heads[i] = carry + sh[tid] - v;
    __syncthreads();
    if (tid == kBlock - 1) carry += sh[tid];
    __syncthreads();
  }
  if (tid == 0) {
    *nseg = carry;
    seg_start[carry] = static_cast<uint32_t>(m);
  }
}

__global__ __launch_bounds__(kBlock) void seg_write_kernel(const uint32_t* __restrict__ keys,
                                                           int64_t m,
                                                           const uint32_t* __restrict__ blk_heads,
                                                           uint32_t* __restrict__ seg_start) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ uint32_t counts[kItems * kWaves];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
  const int lane = lane_id();
  const int w = threadIdx.x / kWave;
  unsigned long long ball[kItems];
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    ball[i] = __ballot(is_head(keys, base + i * kBlock + threadIdx.x, m));
    if (lane == 0) counts[i * kWaves + w] = __popcll(ball[i]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = blk_heads[blockIdx.x];
    for (int i = 0; i < kItems * kWaves; ++i) {
      const uint32_t c = counts[i];
      counts[i] = run;
      run += c;
    }
  }
  __syncthreads();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    if ((ball[i] >> lane) & 1ull) {
      const uint32_t pos = counts[i * kWaves + w] + __popcll(ball[i] & lt_mask);
      seg_start[pos] = static_cast<uint32_t>(base + i * kBlock + threadIdx.x);
    }
  }
}

}  // namespace

int validate_tables(const collie_mf_tables* t);
int validate_batch(const collie_mf_batch* b);

}  // namespace collie

using namespace collie;

extern "C" int collie_mf_train_build_segments(const collie_mf_tables* t, const collie_mf_batch* b,
                                              void* workspace, size_t workspace_bytes,
                                              int32_t side, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(side == COLLIE_SIDE_USER || side == COLLIE_SIDE_ITEM, "bad side %d", side);
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  const SideWs& sw = ws.side[side];

  KeySrc src;
  memset(&src, 0, sizeof(src));
  if (side == COLLIE_SIDE_USER) {
    src.a = b->users;
    src.dta = b->users_dtype;
    src.na = b->batch;
    src.rows = t->num_users;
    src.K = 1;
    src.contig = 1;
  } else {
    src.a = b->pos_items;
    src.dta = b->pos_dtype;
    src.na = b->batch;
    src.b = b->neg_items;
    src.dtb = b->neg_dtype;
    src.sb = b->neg_stride_b;
    src.sk = b->neg_stride_k;
    src.K = b->num_neg;
    src.contig = (b->neg_stride_k == 1 && b->neg_stride_b == b->num_neg) ? 1 : 0;
    src.rows = t->num_items;
  }
  const int npass = radix_passes(src.rows);
  COLLIE_CHECK_HIP(hipMemsetAsync(sw.totals, 0, 4 * 256 * sizeof(uint32_t), stream));
  const dim3 grid(static_cast<unsigned>(sw.ntiles));
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = 8 * pass;
    uint32_t* totals = sw.totals + 256 * pass;
    uint32_t* kout = sw.keys[pass & 1];
    uint32_t* vout = sw.vals[pass & 1];
    if (pass == 0) {
      hipLaunchKernelGGL(radix_hist_kernel<true>, grid, dim3(kBlock), 0, stream, src, nullptr,
                         sw.m, shift, sw.hist, sw.ntiles, totals);
      hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(kBlock), 0, stream, sw.hist,
                         sw.ntiles, totals);
      hipLaunchKernelGGL(radix_scatter_kernel<true>, grid, dim3(kBlock), 0, stream, src, nullptr,
                         sw.codes, sw.m, shift, sw.hist, sw.ntiles, kout, vout);
    } else {
      const uint32_t* kin = sw.keys[(pass - 1) & 1];
      const uint32_t* vin = sw.vals[(pass - 1) & 1];
      hipLaunchKernelGGL(radix_hist_kernel<false>, grid, dim3(kBlock), 0, stream, src, kin, sw.m,
                         shift, sw.hist, sw.ntiles, totals);
      hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(kBlock), 0, stream, sw.hist,
                         sw.ntiles, totals);
      hipLaunchKernelGGL(radix_scatter_kernel<false>, grid, dim3(kBlock), 0, stream, src, kin,
                         vin, sw.m, shift, sw.hist, sw.ntiles, kout, vout);
    }
  }
  const uint32_t* sorted = sw.keys[(npass - 1) & 1];
  hipLaunchKernelGGL(seg_count_kernel, grid, dim3(kBlock), 0, stream, sorted, sw.m, sw.blk_heads);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(kBlock), 0, stream, sw.blk_heads, sw.ntiles,
                     sw.m, sw.nseg, sw.seg_start);
  hipLaunchKernelGGL(seg_write_kernel, grid, dim3(kBlock), 0, stream, sorted, sw.m, sw.blk_heads,
                     sw.seg_start);
  return check_launch("build_segments");
}
