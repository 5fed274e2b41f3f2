// Generic (un-fused) pieces of the path: MatrixFactorizationModel.forward over arbitrary
// (user, item) pairs with its autograd, and the stand-alone loss callables on score tensors.
// The fused training step (mf_train.hip) and the MFMA evaluator (mf_eval.hip) are the hot paths;
// these kernels keep `model(users, items)` and `bpr_loss(pos, neg, ...)` working for every other
// caller (custom losses, get_item_predictions, ...), still without any stock torch arithmetic.
#include "common.h"

namespace collie {

int validate_tables(const collie_mf_tables* t);
int pick_vec(const collie_mf_tables* t);

namespace {

constexpr int kBlock = 256;

template <int VEC>
__device__ __forceinline__ float row_dot(const float* a, const float* b, int lig, int lpr,
                                         int nchunks) {
  float acc = 0.0f;
  for (int c = lig; c < nchunks; c += lpr) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc = fmaf(a[c * VEC + i], b[c * VEC + i], acc);
  }
  return acc;
}

struct ScoreParams {
  collie_mf_tables t;
  const void* users;
  const void* items;
  int64_t m;
  int udt, idt;
  int has_y;
  float y_lo, y_hi;
  float* out;
  const float* grad_out;
  float* gu_rows;
  float* gi_rows;
  float* gbias;
  int32_t* status;
  int nchunks;
  // nn.Dropout on both gathered rows (matrix_factorization.py:143-147), training mode only
  uint64_t drop_thr;   // an element is kept iff its 32 random bits >= drop_thr (= p * 2^32); 0 = off
  float keep_scale;    // 1 / (1 - p)
  uint64_t drop_seed, drop_call;
};

// Multipliers nn.Dropout applies to element d of the user row / item row of pair q in forward
// call `drop_call`: 0 or 1 / (1 - p), independent for the two rows (two of the four words of one
// Philox4x32-10 block: counter = (q, d, call low word), key = seed ^ call high word).
__device__ __forceinline__ void drop_mult(const ScoreParams& p, int64_t q, int d, float& mu,
                                          float& mi) {
  const Philox r = philox4x32_10(static_cast<uint32_t>(q), static_cast<uint32_t>(q >> 32),
                                 static_cast<uint32_t>(d), static_cast<uint32_t>(p.drop_call),
                                 static_cast<uint32_t>(p.drop_seed),
                                 static_cast<uint32_t>(p.drop_seed >> 32) ^
                                     static_cast<uint32_t>(p.drop_call >> 32));
  mu = static_cast<uint64_t>(r.c[0]) >= p.drop_thr ? p.keep_scale : 0.0f;
  mi = static_cast<uint64_t>(r.c[1]) >= p.drop_thr ? p.keep_scale : 0.0f;
}

// BWD = false: scores.  BWD = true: values of the uncoalesced sparse gradients.
// DROP: rows pass through nn.Dropout before the product (and its backward after it).
template <int VEC, int LPR, bool BWD, bool DROP = false>
__global__ __launch_bounds__(kBlock) void mf_score_kernel(const ScoreParams p) {
  constexpr int GPW = kWave / LPR;  // pairs per wave
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int D = p.t.dim;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) / kWave;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kBlock / kWave);
  for (int64_t q0 = wave * GPW; q0 < p.m; q0 += nwaves * GPW) {
    const int64_t q = q0 + grp;
    const bool valid = q < p.m;
    int64_t u = 0, it = 0;
    if (valid) {
      u = load_idx(p.users, p.udt, q);
      it = load_idx(p.items, p.idt, q);
    }
    const bool bad = valid && (u < 0 || u >= p.t.num_users || it < 0 || it >= p.t.num_items);
    if (bad) {
      if (lig == 0 && p.status) atomicOr(p.status, 1);
      u = 0;
      it = 0;
    }
    const float* urow = p.t.user_emb + u * D;
    const float* irow = p.t.item_emb + it * D;
    float part = 0.0f;
    if (valid && (!BWD || p.has_y)) {
      if (!DROP) {
        part = row_dot<VEC>(urow, irow, lig, LPR, p.nchunks);
      } else {
        for (int d = lig; d < D; d += LPR) {
          float mu, mi;
          drop_mult(p, q, d, mu, mi);
          part = fmaf(urow[d] * mu, irow[d] * mi, part);
        }
      }
    }
    part = group_sum<LPR>(part);
    float s = (part + p.t.user_bias[u]) + p.t.item_bias[it];
    if (!BWD) {
      if (p.has_y) s = sigmoidf_(s) * (p.y_hi - p.y_lo) + p.y_lo;
      if (valid && lig == 0) p.out[q] = bad ? 0.0f : s;
    } else {
      float g = valid ? p.grad_out[q] : 0.0f;
      if (p.has_y) {
        const float sig = sigmoidf_(s);
        g = g * (p.y_hi - p.y_lo) * (sig * (1.0f - sig));
      }
      if (bad) g = 0.0f;
      if (valid) {
        if (lig == 0) p.gbias[q] = g;
        if (DROP) {
          // d/d(row) of dropout(row): the dropped partner's value times this row's own multiplier
          for (int d = lig; d < D; d += LPR) {
            float mu, mi;
            drop_mult(p, q, d, mu, mi);
            p.gu_rows[q * D + d] = (g * (irow[d] * mi)) * mu;
            p.gi_rows[q * D + d] = (g * (urow[d] * mu)) * mi;
          }
        }
        for (int c = lig; !DROP && c < p.nchunks; c += LPR) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const int d = c * VEC + i;
            p.gu_rows[q * D + d] = g * irow[d];
            p.gi_rows[q * D + d] = g * urow[d];
          }
        }
      }
    }
  }
}

template <bool BWD>
int launch_score_drop(int lpr, dim3 grid, hipStream_t stream, const ScoreParams& p) {
  switch (lpr) {
    case 1: hipLaunchKernelGGL((mf_score_kernel<1, 1, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((mf_score_kernel<1, 2, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((mf_score_kernel<1, 4, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((mf_score_kernel<1, 8, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((mf_score_kernel<1, 16, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((mf_score_kernel<1, 32, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((mf_score_kernel<1, 64, BWD, true>), grid, dim3(kBlock), 0, stream, p); break;
  }
  return check_launch("mf_score_kernel<dropout>");
}

template <int VEC, bool BWD>
int launch_score(int lpr, dim3 grid, hipStream_t stream, const ScoreParams& p) {
  switch (lpr) {
    case 1: hipLaunchKernelGGL((mf_score_kernel<VEC, 1, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((mf_score_kernel<VEC, 2, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((mf_score_kernel<VEC, 4, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((mf_score_kernel<VEC, 8, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((mf_score_kernel<VEC, 16, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((mf_score_kernel<VEC, 32, BWD>), grid, dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((mf_score_kernel<VEC, 64, BWD>), grid, dim3(kBlock), 0, stream, p); break;
  }
  return check_launch("mf_score_kernel");
}

template <bool BWD>
int run_score(ScoreParams& p, hipStream_t stream) {
  const int vec = p.drop_thr ? 1 : pick_vec(&p.t);   // the dropout route works element-wise
  p.nchunks = (p.t.dim + vec - 1) / vec;
  int lpr = 1;
  while (lpr < p.nchunks && lpr < kWave) lpr *= 2;
  const int64_t pairs_per_block = (kBlock / kWave) * (kWave / lpr);
  int64_t blocks = (p.m + pairs_per_block - 1) / pairs_per_block;
  if (blocks > 8192) blocks = 8192;
  dim3 grid(static_cast<unsigned>(blocks));
  if (p.drop_thr) return launch_score_drop<BWD>(lpr, grid, stream, p);
  if (vec == 4) return launch_score<4, BWD>(lpr, grid, stream, p);
  if (vec == 2) return launch_score<2, BWD>(lpr, grid, stream, p);
  return launch_score<1, BWD>(lpr, grid, stream, p);
}

// ---------------------------------------------------------------------------------------------
// stand-alone losses on score tensors: one thread per sample
// ---------------------------------------------------------------------------------------------
struct LossParams {
  collie_loss_cfg cfg;
  const float* s_pos;
  const float* s_neg;  // [K, B]
  const int64_t* pos_items;
  const int64_t* neg_items;  // [K, B]
  float* grad_pos;
  float* grad_neg;
  float* partials;
  int64_t B;
  int K;
  float inv_b;
};

__device__ __forceinline__ float ideal_diff(const LossParams& p, int64_t b, int k) {
  if (p.cfg.n_meta == 0) return 1.0f;
  const int64_t pi = p.pos_items[b];
  const int64_t ni = p.neg_items[static_cast<int64_t>(k) * p.B + b];
  float frac = 0.0f;
  for (int m = 0; m < p.cfg.n_meta; ++m)
    frac += (p.cfg.meta[m][pi] == p.cfg.meta[m][ni]) ? p.cfg.meta_weight[m] : 0.0f;
  return 1.0f - frac;
}

__global__ __launch_bounds__(kBlock) void loss_kernel(const LossParams p) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int kind = p.cfg.kind;
  float l1 = 0.0f, l2 = 0.0f;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; b < p.B; b += stride) {
    const float sp = p.s_pos[b];
    int sel = 0;
    float ell = 0.0f, g = 0.0f;
    if (kind == COLLIE_LOSS_WARP) {
      for (int k = 0; k < p.K; ++k) {
        const float h = (ideal_diff(p, b, k) - sp) + p.s_neg[static_cast<int64_t>(k) * p.B + b];
        if (h > 0.0f) {
          const float w = logf(static_cast<float>(p.cfg.num_items) / static_cast<float>(k + 1));
          ell = w * h;
          g = -(w * (1.0f + 2.0f * ell)) * p.inv_b;
          sel = k;
          break;
        }
      }
    } else {
      float sn = p.s_neg[b];
      if (kind == COLLIE_LOSS_ADAPTIVE_BPR || kind == COLLIE_LOSS_ADAPTIVE_HINGE) {
        for (int k = 1; k < p.K; ++k) {
          const float v = p.s_neg[static_cast<int64_t>(k) * p.B + b];
          if (v > sn) {
            sn = v;
            sel = k;
          }
        }
      }
      const float c = ideal_diff(p, b, sel);
      const float d = sp - sn;
      if (kind == COLLIE_LOSS_BPR || kind == COLLIE_LOSS_ADAPTIVE_BPR) {
        const float sig = sigmoidf_(d);
        ell = c - sig;
        g = -(sig * (1.0f - sig)) * (1.0f + 2.0f * ell) * p.inv_b;
      } else {
        const float x = c - d;
        ell = fmaxf(x, 0.0f);
        g = x >= 0.0f ? -(1.0f + 2.0f * ell) * p.inv_b : 0.0f;
      }
    }
    l1 += ell;
    l2 += ell * ell;
    if (p.grad_pos) p.grad_pos[b] = g;
    if (p.grad_neg)
      for (int k = 0; k < p.K; ++k)
        p.grad_neg[static_cast<int64_t>(k) * p.B + b] = (k == sel) ? -g : 0.0f;
  }
  __shared__ float red[2][kBlock / kWave];
  l1 = wave_sum(l1);
  l2 = wave_sum(l2);
  if (lane_id() == 0) {
    red[0][threadIdx.x / kWave] = l1;
    red[1][threadIdx.x / kWave] = l2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c = 0.0f;
    for (int w = 0; w < kBlock / kWave; ++w) {
      a += red[0][w];
      c += red[1][w];
    }
    p.partials[2 * blockIdx.x] = a;
    p.partials[2 * blockIdx.x + 1] = c;
  }
}

__global__ __launch_bounds__(kWave) void loss_finalize2_kernel(const float* partials, int nblocks,
                                                               float inv_b, float* loss_out) {
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kWave) {
    a += partials[2 * i];
    c += partials[2 * i + 1];
  }
  for (int o = kWave / 2; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, kWave);
    c += __shfl_xor(c, o, kWave);
  }
  if (threadIdx.x == 0) *loss_out = static_cast<float>((a + c) * static_cast<double>(inv_b));
}

int64_t loss_blocks(int64_t batch) {
  int64_t blocks = (batch + kBlock - 1) / kBlock;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return blocks;
}


// cosine similarity of every row of one embedding table with row `id`
// (base_pipeline.py:854-864: rows divided by their L2 norm, then one matmul row):
//   out[j] = sum_d (E[j,d] / |E[j]|) * (E[id,d] / |E[id]|)
// 16 lanes per row: squared-norm and dot in one pass over the row, fixed butterflies.
__global__ __launch_bounds__(kBlock) void similarity_kernel(const float* __restrict__ emb,
                                                            int64_t rows, int dim, int64_t id,
                                                            float* __restrict__ out) {
  constexpr int LPR = 16;
  const int lig = lane_id() % LPR;
  const float* ref = emb + id * dim;
  float rn = 0.0f;
  for (int d = lig; d < dim; d += LPR) rn = fmaf(ref[d], ref[d], rn);
  rn = sqrtf(group_sum<LPR>(rn));
  const int64_t stride = static_cast<int64_t>(gridDim.x) * (kBlock / LPR);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * (kBlock / LPR) + threadIdx.x / LPR;
       j < rows + (kBlock / LPR); j += stride) {  // whole wave stays in step for the butterflies
    const bool on = j < rows;
    const float* row = emb + (on ? j : 0) * dim;
    float n2 = 0.0f;
    for (int d = lig; d < dim; d += LPR) n2 = fmaf(row[d], row[d], n2);
    const float nj = sqrtf(group_sum<LPR>(n2));
    float dot = 0.0f;
    for (int d = lig; d < dim; d += LPR) dot = fmaf(row[d] / nj, ref[d] / rn, dot);
    dot = group_sum<LPR>(dot);
    if (on && lig == 0) out[j] = dot;
  }
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_mf_score(const collie_mf_tables* t, const void* users, int32_t users_dtype,
                               const void* items, int32_t items_dtype, int64_t m,
                               int32_t has_y_range, float y_lo, float y_hi, float* scores,
                               int32_t* status, collie_stream_t stream) {
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(m >= 0, "negative pair count");
  if (m == 0) return 0;
  COLLIE_REQUIRE(users && items && scores, "mf_score: NULL pointer");
  ScoreParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.users = users;
  p.items = items;
  p.udt = users_dtype;
  p.idt = items_dtype;
  p.m = m;
  p.has_y = has_y_range;
  p.y_lo = y_lo;
  p.y_hi = y_hi;
  p.out = scores;
  p.status = status;
  return run_score<false>(p, static_cast<hipStream_t>(stream));
}

extern "C" int collie_mf_score_backward(const collie_mf_tables* t, const void* users,
                                        int32_t users_dtype, const void* items,
                                        int32_t items_dtype, int64_t m, int32_t has_y_range,
                                        float y_lo, float y_hi, const float* grad_out,
                                        float* grad_user_rows, float* grad_item_rows,
                                        float* grad_bias, int32_t* status,
                                        collie_stream_t stream) {
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(m >= 0, "negative pair count");
  if (m == 0) return 0;
  COLLIE_REQUIRE(users && items && grad_out && grad_user_rows && grad_item_rows && grad_bias,
                 "mf_score_backward: NULL pointer");
  ScoreParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.users = users;
  p.items = items;
  p.udt = users_dtype;
  p.idt = items_dtype;
  p.m = m;
  p.has_y = has_y_range;
  p.y_lo = y_lo;
  p.y_hi = y_hi;
  p.grad_out = grad_out;
  p.gu_rows = grad_user_rows;
  p.gi_rows = grad_item_rows;
  p.gbias = grad_bias;
  p.status = status;
  return run_score<true>(p, static_cast<hipStream_t>(stream));
}

namespace {
int set_dropout(ScoreParams& p, float drop_p, uint64_t seed, uint64_t call) {
  COLLIE_REQUIRE(drop_p > 0.0f && drop_p <= 1.0f, "dropout: p must be in (0, 1] (got %g)", drop_p);
  p.drop_thr = static_cast<uint64_t>(static_cast<double>(drop_p) * 4294967296.0);
  if (p.drop_thr == 0) p.drop_thr = 1;
  p.keep_scale = drop_p < 1.0f ? 1.0f / (1.0f - drop_p) : 0.0f;
  p.drop_seed = seed;
  p.drop_call = call;
  return 0;
}
}  // namespace

extern "C" int collie_mf_score_dropout(const collie_mf_tables* t, const void* users,
                                       int32_t users_dtype, const void* items,
                                       int32_t items_dtype, int64_t m, int32_t has_y_range,
                                       float y_lo, float y_hi, float dropout_p, uint64_t seed,
                                       uint64_t call, float* scores, int32_t* status,
                                       collie_stream_t stream) {
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(m >= 0, "negative pair count");
  if (m == 0) return 0;
  COLLIE_REQUIRE(users && items && scores, "mf_score_dropout: NULL pointer");
  ScoreParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.users = users;
  p.items = items;
  p.udt = users_dtype;
  p.idt = items_dtype;
  p.m = m;
  p.has_y = has_y_range;
  p.y_lo = y_lo;
  p.y_hi = y_hi;
  p.out = scores;
  p.status = status;
  if (set_dropout(p, dropout_p, seed, call)) return -1;
  return run_score<false>(p, static_cast<hipStream_t>(stream));
}

extern "C" int collie_mf_score_dropout_backward(
    const collie_mf_tables* t, const void* users, int32_t users_dtype, const void* items,
    int32_t items_dtype, int64_t m, int32_t has_y_range, float y_lo, float y_hi, float dropout_p,
    uint64_t seed, uint64_t call, const float* grad_out, float* grad_user_rows,
    float* grad_item_rows, float* grad_bias, int32_t* status, collie_stream_t stream) {
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(m >= 0, "negative pair count");
  if (m == 0) return 0;
  COLLIE_REQUIRE(users && items && grad_out && grad_user_rows && grad_item_rows && grad_bias,
                 "mf_score_dropout_backward: NULL pointer");
  ScoreParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.users = users;
  p.items = items;
  p.udt = users_dtype;
  p.idt = items_dtype;
  p.m = m;
  p.has_y = has_y_range;
  p.y_lo = y_lo;
  p.y_hi = y_hi;
  p.grad_out = grad_out;
  p.gu_rows = grad_user_rows;
  p.gi_rows = grad_item_rows;
  p.gbias = grad_bias;
  p.status = status;
  if (set_dropout(p, dropout_p, seed, call)) return -1;
  return run_score<true>(p, static_cast<hipStream_t>(stream));
}

extern "C" int64_t collie_loss_partials(int64_t batch) { return loss_blocks(batch); }

extern "C" int collie_loss_forward_backward(const collie_loss_cfg* cfg, const float* s_pos,
                                            const float* s_neg, int64_t batch, int32_t num_neg,
                                            const int64_t* pos_items, const int64_t* neg_items,
                                            float* loss_out, float* grad_pos, float* grad_neg,
                                            float* partials, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(cfg && s_pos && s_neg && loss_out && partials, "loss: NULL pointer");
  COLLIE_REQUIRE(batch >= 1 && num_neg >= 1, "loss: empty batch");
  COLLIE_REQUIRE(cfg->kind >= COLLIE_LOSS_BPR && cfg->kind <= COLLIE_LOSS_WARP,
                 "%d is not a valid loss function.", cfg->kind);
  COLLIE_REQUIRE(cfg->n_meta >= 0 && cfg->n_meta <= COLLIE_MAX_META,
                 "at most %d metadata arrays are supported", COLLIE_MAX_META);
  if (cfg->n_meta > 0) COLLIE_REQUIRE(pos_items && neg_items, "loss: metadata needs item ids");
  if (cfg->kind == COLLIE_LOSS_BPR || cfg->kind == COLLIE_LOSS_HINGE)
    COLLIE_REQUIRE(num_neg == 1, "non-adaptive losses take one negative score per positive");
  LossParams p;
  p.cfg = *cfg;
  p.s_pos = s_pos;
  p.s_neg = s_neg;
  p.pos_items = pos_items;
  p.neg_items = neg_items;
  p.grad_pos = grad_pos;
  p.grad_neg = grad_neg;
  p.partials = partials;
  p.B = batch;
  p.K = num_neg;
  p.inv_b = 1.0f / static_cast<float>(batch);
  const int64_t blocks = loss_blocks(batch);
  hipLaunchKernelGGL(loss_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, stream, p);
  if (check_launch("loss_kernel")) return -2;
  hipLaunchKernelGGL(loss_finalize2_kernel, dim3(1), dim3(kWave), 0, stream, partials,
                     static_cast<int>(blocks), p.inv_b, loss_out);
  return check_launch("loss_finalize2_kernel");
}

extern "C" int collie_embedding_similarity(const float* emb, int64_t rows, int32_t dim, int64_t id,
                                           float* out, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(emb && out, "embedding_similarity: NULL pointer");
  COLLIE_REQUIRE(rows >= 1 && dim >= 1, "embedding_similarity: empty table");
  COLLIE_REQUIRE(id >= 0 && id < rows, "embedding_similarity: id %lld is not in the table",
                 static_cast<long long>(id));
  int64_t blocks = (rows + (kBlock / 16) - 1) / (kBlock / 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(similarity_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                     stream, emb, rows, dim, id, out);
  return check_launch("similarity_kernel");
}
