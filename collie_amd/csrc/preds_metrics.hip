// Metric reductions over a MATERIALISED score matrix preds [n_rows, n_cols]: the generic route of
// collie/metrics.py, where a metric callable receives `preds` from get_preds (metrics.py:77-107)
// for ANY model (hybrid head, custom models) and any catalogue size:
//   * preds.topk(k) (metrics.py:171) for any k <= n_cols: per row, rounds of up to 1024 entries;
//     a round is a 4-pass MSB radix select of the round's last key among the entries that come
//     after the previous round's last entry, a collection pass (ties in ascending id order) and an
//     exact rank-by-counting sort in LDS.  Order: descending score, ascending id on ties
//     (-0.0 == +0.0; NaN sorts above +inf, as torch.topk treats it).
//   * the rank counters of mrr / auc (metrics.py:224-240, 266-282) against the positives'
//     thresholds, same definitions as the MFMA evaluator's (collie_mf_eval_rank_counts).
// HBM-bound streaming kernels; integer atomics only.
#include "common.h"

namespace collie {

int launch_pos_thresholds(const int64_t* indptr, const int64_t* indices, int64_t n_users,
                          int64_t nnz, float* pos_scores, float* thr_lo, float* thr_hi,
                          float* best_score, int64_t* best_id, hipStream_t stream);

namespace {

constexpr int kSelBlock = 1024;
constexpr int kSelWaves = kSelBlock / kWave;
constexpr int kRound = 1024;  // entries extracted, in final order, per round

// monotone float -> uint key; s + 0.0f turns -0.0 into +0.0 so that the two compare equal
__device__ __forceinline__ uint32_t okey(float s) {
  const uint32_t u = __float_as_uint(s + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(kSelBlock) void preds_topk_kernel(const float* __restrict__ preds,
                                                               int64_t n_cols, int64_t ld, int k,
                                                               float* __restrict__ out_s,
                                                               int64_t* __restrict__ out_i) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wave_ties[kSelWaves];
  __shared__ float cs[kRound];
  __shared__ int ci[kRound];
  __shared__ uint32_t sh_prefix, sh_remaining, sh_ngt, sh_cut_key;
  __shared__ int sh_cut_id;

  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const float* row = preds + static_cast<int64_t>(blockIdx.x) * ld;
  // wave-contiguous column segments (multiples of 64): lane order inside a wave-iteration,
  // iteration order inside a wave and wave order together are ascending column order
  const int64_t per_wave = ((n_cols + kSelWaves - 1) / kSelWaves + kWave - 1) / kWave * kWave;
  const int64_t w_begin = wave * per_wave;
  const int64_t w_end = w_begin + per_wave < n_cols ? w_begin + per_wave : n_cols;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  bool has_cut = false;
  uint32_t ckey = 0;
  int cid = -1;
  for (int done = 0; done < k; done += kRound) {
    const int kr = k - done < kRound ? k - done : kRound;
    // ---- radix select: key T of the round's last entry among the eligible entries ----------
    uint32_t prefix = 0, mask = 0, remaining = static_cast<uint32_t>(kr);
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (threadIdx.x < 256) hist[threadIdx.x] = 0;
      __syncthreads();
      for (int64_t j = w_begin + lane; j < w_end; j += kWave) {
        const uint32_t key = okey(row[j]);
        const bool elig = !has_cut || key < ckey || (key == ckey && j > cid);
        if (elig && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t cum = 0;
        int b = 255;
        for (; b > 0; --b) {
          if (cum + hist[b] >= remaining) break;
          cum += hist[b];
        }
        sh_prefix = prefix | (static_cast<uint32_t>(b) << shift);
        sh_remaining = remaining - cum;
        sh_ngt = 0;
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_remaining;
      mask |= 0xffu << shift;
    }
    const uint32_t T = prefix;
    const uint32_t ties_needed = remaining;                       // >= 1
    const uint32_t ngt = static_cast<uint32_t>(kr) - ties_needed;  // eligible entries above T

    // ---- ties per wave segment -> base index of each wave's ties (ascending column order) ----
    uint32_t cnt = 0;
    for (int64_t j0 = w_begin; j0 < w_end; j0 += kWave) {
      const int64_t j = j0 + lane;
      bool tie = false;
      if (j < w_end) {
        const uint32_t key = okey(row[j]);
        const bool elig = !has_cut || key < ckey || (key == ckey && j > cid);
        tie = elig && key == T;
      }
      cnt += __popcll(__ballot(tie));
    }
    if (lane == 0) wave_ties[wave] = cnt;
    __syncthreads();
    uint32_t tie_base = 0;
    for (int w = 0; w < wave; ++w) tie_base += wave_ties[w];

    // ---- collect: entries above T in any order, then the first ties_needed ties ---------------
    for (int64_t j0 = w_begin; j0 < w_end; j0 += kWave) {
      const int64_t j = j0 + lane;
      bool tie = false, gt = false;
      float s = 0.0f;
      if (j < w_end) {
        s = row[j];
        const uint32_t key = okey(s);
        const bool elig = !has_cut || key < ckey || (key == ckey && j > cid);
        tie = elig && key == T;
        gt = elig && key > T;
      }
      if (gt) {
        const uint32_t slot = atomicAdd(&sh_ngt, 1u);
        cs[slot] = s;
        ci[slot] = static_cast<int>(j);
      }
      const unsigned long long tm = __ballot(tie);
      if (tie) {
        const uint32_t idx = tie_base + __popcll(tm & lt_mask);
        if (idx < ties_needed) {
          cs[ngt + idx] = s;
          ci[ngt + idx] = static_cast<int>(j);
        }
      }
      tie_base += __popcll(tm);
    }
    __syncthreads();

    // ---- exact order inside the round: rank by counting (score desc, id asc) ------------------
    for (int i = threadIdx.x; i < kr; i += kSelBlock) {
      const uint32_t ki = okey(cs[i]);
      const int ii = ci[i];
      int rank = 0;
      for (int j = 0; j < kr; ++j) {
        const uint32_t kj = okey(cs[j]);
        rank += (kj > ki || (kj == ki && ci[j] < ii)) ? 1 : 0;
      }
      const int64_t o = static_cast<int64_t>(blockIdx.x) * k + done + rank;
      out_s[o] = cs[i];
      out_i[o] = ii;
      if (rank == kr - 1) {
        sh_cut_key = ki;
        sh_cut_id = ii;
      }
    }
    __syncthreads();
    ckey = sh_cut_key;
    cid = sh_cut_id;
    has_cut = true;
    __syncthreads();
  }
}

// pos_scores[e] = preds[row of e, indices[e]]
__global__ __launch_bounds__(256) void preds_gather_pos_kernel(
    const float* __restrict__ preds, int64_t n_rows, int64_t n_cols, int64_t ld,
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
    float* __restrict__ pos_scores, int32_t* __restrict__ status) {
  const int64_t nnz = indptr[n_rows];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < nnz; e += stride) {
    int64_t lo = 0, hi = n_rows;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) / 2;
      if (indptr[mid] <= e) lo = mid; else hi = mid;
    }
    const int64_t c = indices[e];
    if (c < 0 || c >= n_cols) {
      if (status) atomicOr(status, 1);
      pos_scores[e] = 0.0f;
    } else {
      pos_scores[e] = preds[lo * ld + c];
    }
  }
}

constexpr int kThrChunk = 1024;

// counts[row] += (#above best, sum_e #(s < lo_e), sum_e #(s <= hi_e)) over this block's columns
__global__ __launch_bounds__(256) void preds_rank_counts_kernel(
    const float* __restrict__ preds, int64_t n_cols, int64_t ld, int64_t cols_per_block,
    const int64_t* __restrict__ indptr, const float* __restrict__ thr_lo,
    const float* __restrict__ thr_hi, const float* __restrict__ best_score,
    const int64_t* __restrict__ best_id, unsigned long long* __restrict__ counts) {
  __shared__ float lo[kThrChunk], hi[kThrChunk];
  __shared__ unsigned long long red[3][4];
  const int64_t u = blockIdx.x;
  const float* row = preds + u * ld;
  const int64_t e_begin = indptr[u], e_end = indptr[u + 1];
  if (e_begin == e_end) return;  // no positives: rr = auc = 0 whatever the counts (uniform exit)
  const float bscore = best_score[u];
  const int64_t bid = best_id[u];
  const int64_t c_begin = static_cast<int64_t>(blockIdx.y) * cols_per_block;
  const int64_t c_end = c_begin + cols_per_block < n_cols ? c_begin + cols_per_block : n_cols;
  unsigned long long c_above = 0, c_lt = 0, c_le = 0;
  for (int64_t tb = e_begin; tb < e_end; tb += kThrChunk) {
    const int n = e_end - tb < kThrChunk ? static_cast<int>(e_end - tb) : kThrChunk;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      lo[i] = thr_lo[tb + i];
      hi[i] = thr_hi[tb + i];
    }
    __syncthreads();
    for (int64_t j = c_begin + threadIdx.x; j < c_end; j += 256) {
      const float s = row[j];
      if (tb == e_begin) c_above += (s > bscore || (s == bscore && j < bid)) ? 1ull : 0ull;
      uint32_t a = 0, b = 0;
      for (int e = 0; e < n; ++e) {
        a += s < lo[e] ? 1u : 0u;
        b += s <= hi[e] ? 1u : 0u;
      }
      c_lt += a;
      c_le += b;
    }
  }
  for (int o = kWave / 2; o > 0; o >>= 1) {
    c_above += __shfl_xor(c_above, o, kWave);
    c_lt += __shfl_xor(c_lt, o, kWave);
    c_le += __shfl_xor(c_le, o, kWave);
  }
  const int wave = threadIdx.x / kWave;
  __syncthreads();
  if (lane_id() == 0) {
    red[0][wave] = c_above;
    red[1][wave] = c_lt;
    red[2][wave] = c_le;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long v =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(&counts[u * 3 + threadIdx.x], v);
  }
}

int check_preds(const float* preds, int64_t n_rows, int64_t n_cols, int64_t ld) {
  COLLIE_REQUIRE(preds != nullptr, "preds is NULL");
  COLLIE_REQUIRE(n_rows >= 1 && n_cols >= 1 && ld >= n_cols, "preds: bad shape [%lld, %lld] ld %lld",
                 static_cast<long long>(n_rows), static_cast<long long>(n_cols),
                 static_cast<long long>(ld));
  COLLIE_REQUIRE(n_cols < (int64_t{1} << 31), "preds: more than 2^31-1 columns is not supported");
  COLLIE_REQUIRE(n_rows < (int64_t{1} << 31), "preds: more than 2^31-1 rows is not supported");
  return 0;
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_preds_topk(const float* preds, int64_t n_rows, int64_t n_cols, int64_t ld,
                                 int32_t k, float* topk_scores, int64_t* topk_ids,
                                 collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_preds(preds, n_rows, n_cols, ld)) return -1;
  COLLIE_REQUIRE(topk_scores && topk_ids, "preds_topk: NULL output");
  COLLIE_REQUIRE(k >= 1 && k <= n_cols, "preds_topk: k (%d) must be in [1, number of items]", k);
  hipLaunchKernelGGL(preds_topk_kernel, dim3(static_cast<unsigned>(n_rows)), dim3(kSelBlock), 0,
                     stream, preds, n_cols, ld, k, topk_scores, topk_ids);
  return check_launch("preds_topk_kernel");
}

extern "C" int collie_preds_pos_scores(const float* preds, int64_t n_rows, int64_t n_cols,
                                       int64_t ld, const int64_t* indptr, const int64_t* indices,
                                       int64_t nnz, float* pos_scores, float* thr_lo, float* thr_hi,
                                       float* best_score, int64_t* best_id, int32_t* status,
                                       collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_preds(preds, n_rows, n_cols, ld)) return -1;
  COLLIE_REQUIRE(indptr && indices && pos_scores, "preds_pos_scores: NULL pointer");
  COLLIE_REQUIRE(nnz >= 0, "preds_pos_scores: bad nnz");
  if (nnz > 0) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(preds_gather_pos_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       stream, preds, n_rows, n_cols, ld, indptr, indices, pos_scores, status);
    if (check_launch("preds_gather_pos_kernel")) return -2;
  }
  return launch_pos_thresholds(indptr, indices, n_rows, nnz, pos_scores, thr_lo, thr_hi, best_score,
                               best_id, stream);
}

extern "C" int collie_preds_rank_counts(const float* preds, int64_t n_rows, int64_t n_cols,
                                        int64_t ld, const int64_t* indptr, const float* thr_lo,
                                        const float* thr_hi, const float* best_score,
                                        const int64_t* best_id, uint64_t* counts,
                                        collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_preds(preds, n_rows, n_cols, ld)) return -1;
  COLLIE_REQUIRE(indptr && thr_lo && thr_hi && best_score && best_id && counts,
                 "preds_rank_counts: NULL pointer");
  // enough blocks to fill the chip when there are few rows, at least 16 K columns per block
  int64_t nchunk = (2048 + n_rows - 1) / n_rows;
  const int64_t max_chunk = (n_cols + 16383) / 16384;
  if (nchunk > max_chunk) nchunk = max_chunk;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > 65535) nchunk = 65535;
  const int64_t cols_per_block = (n_cols + nchunk - 1) / nchunk;
  nchunk = (n_cols + cols_per_block - 1) / cols_per_block;
  hipLaunchKernelGGL(preds_rank_counts_kernel,
                     dim3(static_cast<unsigned>(n_rows), static_cast<unsigned>(nchunk)), dim3(256),
                     0, stream, preds, n_cols, ld, cols_per_block, indptr, thr_lo, thr_hi,
                     best_score, best_id, reinterpret_cast<unsigned long long*>(counts));
  return check_launch("preds_rank_counts_kernel");
}
