// K3a: deterministic "coalesce" of looked-up row ids (torch/optim/_functional.py:44).
//
// Stable LSD radix sort (8..11 bits per pass; only as many passes as the key range needs) of
// (row key -> contribution code) pairs, followed by segment-head extraction.  The stable order
// inside a segment is the order of the batch, so the floating-point sums the optimizer kernels
// form over a segment are bit-reproducible run to run.  Integer atomics (LDS/global counters)
// are used for histograms only; their results do not depend on arrival order.
#include "rowsort.h"

namespace collie {
namespace {

constexpr int kBlock = 256;
constexpr int kItems = kSortTile / kBlock;  // 8 entries per thread
static_assert(kItems * kBlock == kSortTile, "tile shape");

template <int BITS>
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys_in,
                                                            int64_t m, int shift,
                                                            uint32_t* __restrict__ hist,
                                                            int64_t ntiles,
                                                            uint32_t* __restrict__ totals) {
  constexpr int RADIX = 1 << BITS;
  __shared__ uint32_t h[RADIX];
  for (int i = threadIdx.x; i < RADIX; i += kBlock) h[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kSortTile;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = base + i * kBlock + threadIdx.x;
    if (idx < m) atomicAdd(&h[(keys_in[idx] >> shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RADIX; d += kBlock)
    hist[static_cast<int64_t>(d) * ntiles + blockIdx.x] = h[d];
  (void)totals;
}

// one workgroup per digit: exclusive scan of that digit's per-tile counts (positions inside the
// digit's bucket) and the bucket's size.  Where a bucket starts -- the keys with a smaller digit --
// is added by the scatter kernel from the RADIX sizes (one small scan per workgroup there instead of
// one more launch per pass here).  (The first version had every tile add its counts to totals[]
// with global atomics: ntiles x RADIX atomics on RADIX addresses cost more than the histogram.)
__global__ __launch_bounds__(kBlock) void radix_scan_kernel(uint32_t* __restrict__ hist,
                                                            int64_t ntiles,
                                                            uint32_t* __restrict__ totals) {
  __shared__ uint32_t sh[kBlock];
  const int d = blockIdx.x;
  const int tid = threadIdx.x;
  uint32_t* row = hist + static_cast<int64_t>(d) * ntiles;
  const int64_t per = (ntiles + kBlock - 1) / kBlock;
  const int64_t lo = tid * per;
  const int64_t hi = lo + per < ntiles ? lo + per : ntiles;
  uint32_t local = 0;
  for (int64_t i = lo; i < hi; ++i) local += row[i];
  sh[tid] = local;
  __syncthreads();
  for (int o = 1; o < kBlock; o <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread sums
    const uint32_t add = tid >= o ? sh[tid - o] : 0u;
    __syncthreads();
    sh[tid] += add;
    __syncthreads();
  }
  if (tid == kBlock - 1) totals[d] = sh[tid];
  uint32_t running = sh[tid] - local;
  for (int64_t i = lo; i < hi; ++i) {
    const uint32_t c = row[i];
    row[i] = running;
    running += c;
  }
}

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  A tile's run inside a
// bucket is only a few entries long and the neighbouring runs of the same output line belong to the
// neighbouring tiles: with tile = blockIdx those come from 8 different L2s and every line is written
// back 8 times as a partial line.  This bijection gives each XCD one contiguous range of tiles.
__device__ __forceinline__ int64_t xcd_tile(int64_t b, int64_t ntiles) {
  const int64_t q = ntiles / 8, r = ntiles % 8;
  const int64_t x = b % 8, j = b / 8;
  return x * q + (x < r ? x : r) + j;
}

// The entries of a tile are first brought into digit order inside LDS (their rank among the tile's
// entries of the same digit is known from the ballots), then written out in that order: consecutive
// lanes then write consecutive addresses of a bucket's run instead of 64 scattered 4-byte stores per
// instruction (ML-20M shape, 3 M entries: 36.5 -> see DESIGN.md K3a).
template <int BITS>
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t m,
    int shift, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, int64_t ntiles,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  constexpr int RADIX = 1 << BITS;
  constexpr int kWaves = kBlock / kWave;
  constexpr int DPT = RADIX / kBlock;          // digits per thread in the tile-local scan
  static_assert(DPT * kBlock == RADIX, "radix is a multiple of the workgroup size");
  __shared__ uint32_t wcount[kWaves][RADIX];
  __shared__ uint32_t gbase[RADIX];            // global position of the tile's first entry of a digit
  __shared__ uint32_t lstart[RADIX];           // tile-local position of it
  __shared__ uint32_t skey[kSortTile], sval[kSortTile];
  __shared__ uint32_t wave_tot[kWaves], wave_big[kWaves];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int w = tid / kWave;
  for (int i = tid; i < kWaves * RADIX; i += kBlock) (&wcount[0][0])[i] = 0;
  __syncthreads();

  // wave-striped tile: wave w owns entries [w*512, (w+1)*512) of the tile, iteration-major
  const int64_t tile = xcd_tile(blockIdx.x, ntiles);
  const int64_t tbase = tile * kSortTile;
  const int64_t wbase = tbase + w * (kWave * kItems);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t key[kItems], val[kItems], rnk[kItems];
  volatile uint32_t* mycount = wcount[w];
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = wbase + i * kWave + lane;
    const bool valid = idx < m;
    key[i] = valid ? keys_in[idx] : 0xffffffffu;
    val[i] = valid ? vals_in[idx] : 0u;
    const uint32_t digit = (key[i] >> shift) & (RADIX - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
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
  // per digit: the waves' counts become offsets inside the digit's run of this tile; per thread
  // DPT consecutive digits, then one exclusive scan over the workgroup for the runs' starts
  // (the same scan, over the bucket sizes of the whole input, gives where every bucket starts)
  uint32_t dcount[DPT], mine = 0, big[DPT], big_mine = 0, in_bucket[DPT];
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    const int d = tid * DPT + j;
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
      const uint32_t c = wcount[i][d];
      wcount[i][d] = run;
      run += c;
    }
    dcount[j] = run;
    mine += run;
    big[j] = totals[d];
    big_mine += big[j];
    in_bucket[j] = hist[static_cast<int64_t>(d) * ntiles + tile];
  }
  uint32_t inc = mine, big_inc = big_mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o, kWave);
    const uint32_t big_up = __shfl_up(big_inc, o, kWave);
    if (lane >= o) {
      inc += up;
      big_inc += big_up;
    }
  }
  if (lane == kWave - 1) {
    wave_tot[w] = inc;
    wave_big[w] = big_inc;
  }
  __syncthreads();
  uint32_t before_me = inc - mine, big_before = big_inc - big_mine;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) {
    before_me += i < w ? wave_tot[i] : 0u;
    big_before += i < w ? wave_big[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < DPT; ++j) {
    lstart[tid * DPT + j] = before_me;
    before_me += dcount[j];
    gbase[tid * DPT + j] = big_before + in_bucket[j];
    big_before += big[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int64_t idx = wbase + i * kWave + lane;
    if (idx < m) {
      const uint32_t digit = (key[i] >> shift) & (RADIX - 1);
      const uint32_t lp = lstart[digit] + wcount[w][digit] + rnk[i];
      skey[lp] = key[i];
      sval[lp] = val[i];
    }
  }
  __syncthreads();
  const int64_t left = m - tbase;
  const int n_here = left < kSortTile ? static_cast<int>(left) : kSortTile;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int j = i * kBlock + tid;
    if (j < n_here) {
      const uint32_t k = skey[j];
      const uint32_t digit = (k >> shift) & (RADIX - 1);
      const uint32_t pos = gbase[digit] + (static_cast<uint32_t>(j) - lstart[digit]);
      keys_out[pos] = k;
      vals_out[pos] = sval[j];
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
    nseg[0] = carry;
    nseg[1] = carry;  // overwritten by seg_write_kernel if a key >= split exists
    nseg[2] = 0;      // pieces of long segments (piece_plan_kernel)
    seg_start[carry] = static_cast<uint32_t>(m);
  }
}

__global__ __launch_bounds__(kBlock) void seg_write_kernel(const uint32_t* __restrict__ keys,
                                                           int64_t m, uint32_t split,
                                                           const uint32_t* __restrict__ blk_heads,
                                                           uint32_t* __restrict__ seg_start,
                                                           uint32_t* __restrict__ nseg,
                                                           uint32_t* __restrict__ seg_of) {
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
  if (seg_of) {  // segment index of EVERY entry (heads at or before it, minus one)
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      const int64_t idx = base + i * kBlock + threadIdx.x;
      if (idx < m)
        seg_of[idx] = counts[i * kWaves + w] + __popcll(ball[i] & (lt_mask | (1ull << lane))) - 1u;
    }
  }
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    if ((ball[i] >> lane) & 1ull) {
      const int64_t idx = base + i * kBlock + threadIdx.x;
      const uint32_t pos = counts[i * kWaves + w] + __popcll(ball[i] & lt_mask);
      seg_start[pos] = static_cast<uint32_t>(idx);
      // the first segment of the second table (keys are sorted: exactly one such head)
      if (keys[idx] >= split && (idx == 0 || keys[idx - 1] < split)) nseg[1] = pos;
    }
  }
}

// one thread per segment: segments longer than kPieceLen get ceil(len / kPieceLen) consecutive
// pieces.  The piece numbers come from an integer atomic (their values vary run to run); what a
// piece sums and the order in which a segment's pieces are combined do not.
__global__ __launch_bounds__(kBlock) void piece_plan_kernel(const uint32_t* __restrict__ seg_start,
                                                            uint32_t* __restrict__ nseg,
                                                            uint32_t* __restrict__ long_base,
                                                            uint32_t* __restrict__ piece_start,
                                                            uint32_t* __restrict__ piece_end) {
  const uint32_t n = nseg[0];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t seg = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; seg < n;
       seg += stride) {
    const uint32_t start = seg_start[seg], end = seg_start[seg + 1];
    const uint32_t len = end - start;
    if (len <= kPieceLen) {
      long_base[seg] = kNoPiece;
      continue;
    }
    const uint32_t np = (len + kPieceLen - 1) / kPieceLen;
    const uint32_t base = atomicAdd(&nseg[2], np);
    long_base[seg] = base;
    for (uint32_t i = 0; i < np; ++i) {
      piece_start[base + i] = start + i * kPieceLen;
      const uint32_t pe = start + (i + 1) * kPieceLen;
      piece_end[base + i] = pe < end ? pe : end;
    }
  }
}

// ---- small batches: the whole coalesce in ONE workgroup ---------------------------------------------
// Up to kSmallMax entries (B <= 1365 at 3 entries per sample, i.e. the reference's default batch
// size of 1,024): the ~11 launches of the general path cost more than their work.  Entries live in
// LDS; the sort is a stable LSD radix sort, 6 bits per pass: wave w owns entries [256 w, 256 w + 256)
// of the current order and ranks them among its own entries of the same digit (ballot matching, as
// radix_scatter_kernel does); the 64 x 16 (digit, wave) counts are one per thread in digit-major
// order, so ONE workgroup-wide exclusive scan turns them into output positions.  Then segment heads
// and the split index, all written in the general layout.  No segment is cut into pieces here
// (long_base = kNoPiece, nseg[2] = 0): with at most 4,096 entries the row-update kernels sum even a
// degenerate segment in entry order themselves, and the two piece kernels are not launched
// (small_coalesce() tells the callers).
constexpr int kSmallBlock = 1024;
constexpr int kSmallPer = 4;
static_assert(kSmallBlock * kSmallPer == kSmallMax, "small-sort shape");
constexpr int kSmallBits = 6;
constexpr int kSmallRadix = 1 << kSmallBits;
constexpr int kSmallWaves = kSmallBlock / kWave;
static_assert(kSmallRadix * kSmallWaves == kSmallBlock, "one (digit, wave) counter per thread");

// exclusive prefix sum of one value per thread over the workgroup; total to every thread
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_tot, uint32_t& total) {
  const int lane = lane_id();
  const int w = threadIdx.x / kWave;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t up = __shfl_up(inc, o, kWave);
    if (lane >= o) inc += up;
  }
  __syncthreads();  // wave_tot from the previous call has been read by everyone
  if (lane == kWave - 1) wave_tot[w] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < kSmallWaves; ++i) {
    const uint32_t t = wave_tot[i];
    before += i < w ? t : 0u;
    all += t;
  }
  total = all;
  return before + inc - v;
}

__global__ __launch_bounds__(kSmallBlock) void small_sort_segment_kernel(
    const uint32_t* __restrict__ ekeys, const uint32_t* __restrict__ ecodes, int m, int nbits,
    uint32_t split, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
    uint32_t* __restrict__ seg_start, uint32_t* __restrict__ nseg, uint32_t* __restrict__ long_base,
    uint32_t* __restrict__ seg_of) {
  __shared__ uint32_t kbuf[2][kSmallMax];
  __shared__ uint32_t vbuf[2][kSmallMax];
  __shared__ uint32_t wcount[kSmallBlock];   // [wave][digit]: a wave's lanes hit distinct banks
  __shared__ uint32_t wave_tot[kSmallWaves];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int w = tid / kWave;
  for (int i = tid; i < kSmallMax; i += kSmallBlock) {
    kbuf[0][i] = i < m ? ekeys[i] : 0xffffffffu;  // padding sorts last and keeps its place
    vbuf[0][i] = i < m ? ecodes[i] : 0u;
  }
  int cur = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int shift = 0; shift < nbits; shift += kSmallBits) {
    wcount[tid] = 0;
    __syncthreads();               // also: kbuf[cur] / vbuf[cur] are complete
    uint32_t key[kSmallPer], val[kSmallPer], rnk[kSmallPer];
    volatile uint32_t* mycount = wcount;
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
      const int idx = w * (kWave * kSmallPer) + i * kWave + lane;
      key[i] = kbuf[cur][idx];
      val[i] = vbuf[cur][idx];
      const uint32_t digit = (key[i] >> shift) & (kSmallRadix - 1);
      unsigned long long peers = ~0ull;
#pragma unroll
      for (int bit = 0; bit < kSmallBits; ++bit) {
        const bool one = (digit >> bit) & 1u;
        const unsigned long long bm = __ballot(one);
        peers &= one ? bm : ~bm;
      }
      const uint32_t before = __popcll(peers & lt_mask);
      const uint32_t old = mycount[w * kSmallRadix + digit];
      __builtin_amdgcn_wave_barrier();
      if (before == 0) mycount[w * kSmallRadix + digit] = old + __popcll(peers);
      __builtin_amdgcn_wave_barrier();
      rnk[i] = old + before;
    }
    __syncthreads();
    // thread t scans the counter of (digit t / 16, wave t % 16): digit-major order = output order
    const int mine_at = (tid % kSmallWaves) * kSmallRadix + tid / kSmallWaves;
    const uint32_t mine = wcount[mine_at];
    uint32_t total;
    const uint32_t base = block_excl_scan(mine, wave_tot, total);
    wcount[mine_at] = base;        // every thread read its own counter before the scan's barriers
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
      const uint32_t digit = (key[i] >> shift) & (kSmallRadix - 1);
      const uint32_t pos = wcount[w * kSmallRadix + digit] + rnk[i];
      kbuf[cur ^ 1][pos] = key[i];
      vbuf[cur ^ 1][pos] = val[i];
    }
    cur ^= 1;
    __syncthreads();               // the positions in wcount have been used by everyone
  }
  // ---- sorted entries out + segment heads ----------------------------------------------------------
  uint32_t nh = 0;
  bool head[kSmallPer];
#pragma unroll
  for (int j = 0; j < kSmallPer; ++j) {
    const int idx = tid * kSmallPer + j;
    const uint32_t key = kbuf[cur][idx];
    if (idx < m) {
      keys_out[idx] = key;
      vals_out[idx] = vbuf[cur][idx];
    }
    head[j] = idx < m && (idx == 0 || key != kbuf[cur][idx - 1]);
    nh += head[j] ? 1u : 0u;
  }
  uint32_t total_heads;
  uint32_t hbefore = block_excl_scan(nh, wave_tot, total_heads);
  uint32_t first_split = 0xffffffffu;
#pragma unroll
  for (int j = 0; j < kSmallPer; ++j) {
    const int idx = tid * kSmallPer + j;
    if (seg_of && idx < m) seg_of[idx] = hbefore + (head[j] ? 1u : 0u) - 1u;
    if (head[j]) {
      seg_start[hbefore] = idx;
      long_base[hbefore] = kNoPiece;
      const uint32_t key = kbuf[cur][idx];
      if (key >= split && (idx == 0 || kbuf[cur][idx - 1] < split)) first_split = hbefore;
      ++hbefore;
    }
  }
  if (tid == 0) {
    seg_start[total_heads] = m;
    nseg[0] = total_heads;
    nseg[2] = 0;
  }
  // the first segment of the second table: exactly one head qualifies, if any
  const bool any_split = __syncthreads_or(first_split != 0xffffffffu);
  if (first_split != 0xffffffffu) nseg[1] = first_split;
  if (!any_split && tid == 0) nseg[1] = total_heads;
}

template <int BITS>
int run_pass(const SortWs& s, const uint32_t* kin, const uint32_t* vin, uint32_t* kout,
             uint32_t* vout, int shift, uint32_t* totals, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(s.ntiles));
  hipLaunchKernelGGL(radix_hist_kernel<BITS>, grid, dim3(kBlock), 0, stream, kin, s.m, shift,
                     s.hist, s.ntiles, totals);
  hipLaunchKernelGGL(radix_scan_kernel, dim3(1u << BITS), dim3(kBlock), 0, stream, s.hist,
                     s.ntiles, totals);
  hipLaunchKernelGGL((radix_scatter_kernel<BITS>), grid, dim3(kBlock), 0, stream, kin, vin, s.m,
                     shift, s.hist, totals, s.ntiles, kout, vout);
  return check_launch("radix pass");
}

}  // namespace

int sort_and_segment(const SortWs& s, int64_t nkeys, uint32_t split, hipStream_t stream,
                     uint32_t* seg_of) {
  const RadixPlan plan = radix_plan(nkeys);
  COLLIE_REQUIRE(plan.passes <= 4, "key range too large for the radix plan");
  if (small_coalesce(s.m)) {
    int nbits = 1;
    while ((int64_t{1} << nbits) < nkeys) ++nbits;
    const int out = (plan.passes - 1) & 1;  // where sorted_keys() / sorted_vals() look
    hipLaunchKernelGGL(small_sort_segment_kernel, dim3(1), dim3(kSmallBlock), 0, stream, s.ekeys,
                       s.ecodes, static_cast<int>(s.m), nbits, split, s.keys[out], s.vals[out],
                       s.seg_start, s.nseg, s.long_base, seg_of);
    return check_launch("small_sort_segment_kernel");
  }
  for (int pass = 0; pass < plan.passes; ++pass) {
    const uint32_t* kin = pass == 0 ? s.ekeys : s.keys[(pass - 1) & 1];
    const uint32_t* vin = pass == 0 ? s.ecodes : s.vals[(pass - 1) & 1];
    uint32_t* totals = s.totals + (size_t{1} << kMaxRadixBits) * pass;
    int rc;
    switch (plan.bits) {
      case 8: rc = run_pass<8>(s, kin, vin, s.keys[pass & 1], s.vals[pass & 1], 8 * pass, totals, stream); break;
      case 9: rc = run_pass<9>(s, kin, vin, s.keys[pass & 1], s.vals[pass & 1], 9 * pass, totals, stream); break;
      case 10: rc = run_pass<10>(s, kin, vin, s.keys[pass & 1], s.vals[pass & 1], 10 * pass, totals, stream); break;
      default: rc = run_pass<11>(s, kin, vin, s.keys[pass & 1], s.vals[pass & 1], 11 * pass, totals, stream); break;
    }
    if (rc) return rc;
  }
  const uint32_t* sorted = sorted_keys(s, nkeys);
  const dim3 grid(static_cast<unsigned>(s.ntiles));
  hipLaunchKernelGGL(seg_count_kernel, grid, dim3(kBlock), 0, stream, sorted, s.m, s.blk_heads);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(kBlock), 0, stream, s.blk_heads, s.ntiles, s.m,
                     s.nseg, s.seg_start);
  hipLaunchKernelGGL(seg_write_kernel, grid, dim3(kBlock), 0, stream, sorted, s.m, split,
                     s.blk_heads, s.seg_start, s.nseg, seg_of);
  int64_t pblocks = (s.m + kBlock - 1) / kBlock;
  if (pblocks > 1024) pblocks = 1024;
  hipLaunchKernelGGL(piece_plan_kernel, dim3(static_cast<unsigned>(pblocks)), dim3(kBlock), 0,
                     stream, s.seg_start, s.nseg, s.long_base, s.piece_start, s.piece_end);
  return check_launch("segment heads");
}

int validate_tables(const collie_mf_tables* t);
int validate_batch(const collie_mf_batch* b);

}  // namespace collie

using namespace collie;

extern "C" int collie_mf_train_build_segments(const collie_mf_tables* t, const collie_mf_batch* b,
                                              void* workspace, size_t workspace_bytes,
                                              collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  return sort_and_segment(ws.sort, t->num_users + t->num_items,
                          static_cast<uint32_t>(t->num_users), stream);
}
