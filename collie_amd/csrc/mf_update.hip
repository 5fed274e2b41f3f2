// K3b + K4 + K5: segmented gradient reduction fused with the optimizer row update.
//
// One wave per unique looked-up row (a "segment" of the sorted id list).  The wave adds the
// staged per-sample gradient rows that belong to the row in batch order (deterministic), then
// applies torch's update formula to that row in registers and writes p (and m, v / state_sum)
// back once.  Reference arithmetic:
//   SparseAdam  torch/optim/_functional.py:62-84     Adagrad  torch/optim/adagrad.py:361-385
//   Adam        torch/optim/adam.py:416-547          SGD      torch/optim/sgd.py (momentum 0)
// "Touched" semantics: a row that was looked up but received a zero
// gradient (non-selected negatives) is still a segment here, so SparseAdam decays its moments
// and moves the parameter exactly as torch does after coalesce().
//
// Compiled with -ffp-contract=off so that every rounding step of the torch formulas is kept.
#include "train_ws.h"

namespace collie {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kUpdateBlocks = 2048;

struct Hyper {
  int algo;
  int dense;
  float neg_lr;      // SGD: -lr | Adagrad: -clr | Adam: -lr/bc1 | SparseAdam: -lr*sqrt(bc2)/bc1
  float c1;          // 1 - beta1
  float c2;          // 1 - beta2
  float beta2;
  float eps;
  float bc2_sqrt;    // Adam: sqrt(1 - beta2^t)
  float weight_decay;
};

// One element of torch's update. g already includes grad_scale.  Returns the new parameter.
__device__ __forceinline__ float apply_update(const Hyper& h, float p, float g, float& s1,
                                              float& s2) {
  switch (h.algo) {
    case COLLIE_OPT_SGD: {
      if (h.dense && h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      return p + h.neg_lr * g;
    }
    case COLLIE_OPT_ADAGRAD: {
      if (h.dense && h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      s1 = s1 + g * g;
      const float std_ = sqrtf(s1) + h.eps;
      return p + h.neg_lr * (g / std_);
    }
    case COLLIE_OPT_SPARSE_ADAM: {
      const float um = (g - s1) * h.c1;
      const float uv = (g * g - s2) * h.c2;
      s1 = s1 + um;
      s2 = s2 + uv;
      const float denom = sqrtf(s2) + h.eps;
      return p + h.neg_lr * (s1 / denom);
    }
    default: {  // COLLIE_OPT_ADAM (dense)
      if (h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      s1 = s1 + h.c1 * (g - s1);                 // exp_avg.lerp_(grad, 1 - beta1)
      s2 = s2 * h.beta2;                         // exp_avg_sq.mul_(beta2)
      s2 = s2 + h.c2 * (g * g);                  //            .addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(s2) / h.bc2_sqrt + h.eps;
      return p + h.neg_lr * (s1 / denom);
    }
  }
}

struct UpdParams {
  float* table;      // [rows, D] (D = 1 for a bias table)
  float* s1;
  float* s2;
  int32_t* stamp;    // dense algorithms: stamp[row] = step for touched rows (else NULL)
  const float* stage;  // [B, D] staged rows (embedding) | gcoef [B] (bias)
  const uint32_t* keys;
  const uint32_t* vals;
  const uint32_t* seg_start;
  const uint32_t* nseg;
  const float* grad_scale;
  int64_t* rows_out;   // export mode
  float* vals_out;
  int32_t* count_out;
  int D;
  int step;
  int user_bias;       // user-bias table: the gradient is identically zero (see below)
  Hyper h;
};

// EXPORT = true writes the coalesced gradient instead of updating the table.
template <bool EXPORT>
__global__ __launch_bounds__(kBlock) void emb_rows_kernel(const UpdParams p) {
  const int lane = lane_id();
  const uint32_t nseg = *p.nseg;
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const int D = p.D;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  if (EXPORT && blockIdx.x == 0 && threadIdx.x == 0) *p.count_out = static_cast<int32_t>(nseg);
  for (int64_t seg = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
       seg < nseg; seg += nwaves) {
    const uint32_t start = p.seg_start[seg];
    const uint32_t end = p.seg_start[seg + 1];
    const int64_t row = p.keys[start];
    if (EXPORT && lane == 0) p.rows_out[seg] = row;
    for (int d0 = 0; d0 < D; d0 += kWave) {
      const int d = d0 + lane;
      const bool on = d < D;
      float pv = 0.0f, s1 = 0.0f, s2 = 0.0f;
      if (!EXPORT && on) {
        pv = p.table[row * D + d];
        if (p.s1) s1 = p.s1[row * D + d];
        if (p.s2) s2 = p.s2[row * D + d];
      }
      float acc = 0.0f;
      for (uint32_t e0 = start; e0 < end; e0 += kWave) {
        const uint32_t code = (e0 + lane < end) ? p.vals[e0 + lane] : kCodeNone;
        unsigned long long live = __ballot(code != kCodeNone);
        while (live) {
          // up to four staged rows in flight, added in batch order
          float x[4];
          float sg[4];
          int n = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            x[q] = 0.0f;
            sg[q] = 0.0f;
            if (live) {
              const int l = __ffsll(static_cast<long long>(live)) - 1;
              live &= live - 1;
              const uint32_t c = __shfl(code, l, kWave);
              sg[q] = (c & kCodeNeg) ? -1.0f : 1.0f;
              if (on) x[q] = p.stage[static_cast<int64_t>(c & kCodeMask) * D + d];
              ++n;
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < n) acc += sg[q] * x[q];
        }
      }
      const float g = acc * scale;
      if (EXPORT) {
        if (on) p.vals_out[seg * D + d] = g;
      } else if (on) {
        const float np = apply_update(p.h, pv, g, s1, s2);
        p.table[row * D + d] = np;
        if (p.s1) p.s1[row * D + d] = s1;
        if (p.s2) p.s2[row * D + d] = s2;
      }
    }
    if (!EXPORT && p.stamp && lane == 0) p.stamp[row] = p.step;
  }
}

// bias tables: one wave per segment, lanes stride the segment, fixed-shape butterfly sum.
// d loss / d user_bias is identically zero for every pairwise loss (the user bias cancels in
// s+ - s-); the reference's value is only the fp32 residue of (+g) + (-g) in coalesce order, so
// the user-bias gradient is produced as an exact 0 (the row is still "touched").
template <bool EXPORT>
__global__ __launch_bounds__(kBlock) void bias_rows_kernel(const UpdParams p) {
  const int lane = lane_id();
  const uint32_t nseg = *p.nseg;
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  if (EXPORT && blockIdx.x == 0 && threadIdx.x == 0) *p.count_out = static_cast<int32_t>(nseg);
  for (int64_t seg = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
       seg < nseg; seg += nwaves) {
    const uint32_t start = p.seg_start[seg];
    const uint32_t end = p.seg_start[seg + 1];
    const int64_t row = p.keys[start];
    float acc = 0.0f;
    if (!p.user_bias) {
      for (uint32_t e = start + lane; e < end; e += kWave) {
        const uint32_t c = p.vals[e];
        if (c != kCodeNone) {
          const float gv = p.stage[c & kCodeMask];
          acc += (c & kCodeNeg) ? -gv : gv;
        }
      }
      acc = wave_sum(acc);
    }
    const float g = acc * scale;
    if (lane == 0) {
      if (EXPORT) {
        p.rows_out[seg] = row;
        p.vals_out[seg] = g;
      } else {
        float s1 = p.s1 ? p.s1[row] : 0.0f;
        float s2 = p.s2 ? p.s2[row] : 0.0f;
        p.table[row] = apply_update(p.h, p.table[row], g, s1, s2);
        if (p.s1) p.s1[row] = s1;
        if (p.s2) p.s2[row] = s2;
        if (p.stamp) p.stamp[row] = p.step;
      }
    }
  }
}

// K5: dense optimizers also move every row that was NOT looked up this step (gradient 0, plus
// weight decay): Adam decays m, v and keeps stepping along m.
__global__ __launch_bounds__(kBlock) void dense_sweep_kernel(float* __restrict__ table,
                                                             float* __restrict__ s1p,
                                                             float* __restrict__ s2p,
                                                             const int32_t* __restrict__ stamp,
                                                             int64_t rows, int D, int step,
                                                             const Hyper h) {
  const int64_t total = rows * D;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += stride) {
    const int64_t row = i / D;
    if (stamp[row] == step) continue;
    float s1 = s1p ? s1p[i] : 0.0f;
    float s2 = s2p ? s2p[i] : 0.0f;
    table[i] = apply_update(h, table[i], 0.0f, s1, s2);
    if (s1p) s1p[i] = s1;
    if (s2p) s2p[i] = s2;
  }
}

}  // namespace

int validate_tables(const collie_mf_tables* t);
int validate_batch(const collie_mf_batch* b);

namespace {

int fill_common(UpdParams& p, const collie_mf_tables* t, const collie_mf_batch* b, void* workspace,
                size_t workspace_bytes, int side, int is_bias) {
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(side == COLLIE_SIDE_USER || side == COLLIE_SIDE_ITEM, "bad side %d", side);
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  const SideWs& sw = ws.side[side];
  const int64_t rows = side == COLLIE_SIDE_USER ? t->num_users : t->num_items;
  const int fin = (radix_passes(rows) - 1) & 1;
  memset(&p, 0, sizeof(p));
  p.keys = sw.keys[fin];
  p.vals = sw.vals[fin];
  p.seg_start = sw.seg_start;
  p.nseg = sw.nseg;
  if (is_bias) {
    p.table = side == COLLIE_SIDE_USER ? t->user_bias : t->item_bias;
    p.stage = ws.gcoef;
    p.D = 1;
    p.user_bias = side == COLLIE_SIDE_USER;
  } else {
    p.table = side == COLLIE_SIDE_USER ? t->user_emb : t->item_emb;
    p.stage = side == COLLIE_SIDE_USER ? ws.stage_u : ws.stage_i;
    p.D = t->dim;
  }
  return 0;
}

unsigned grid_for(int64_t ids) {
  int64_t blocks = (ids + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > kUpdateBlocks) blocks = kUpdateBlocks;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_mf_train_update(const collie_mf_tables* t, const collie_mf_batch* b,
                                      void* workspace, size_t workspace_bytes, int32_t side,
                                      int32_t is_bias, int32_t dense,
                                      const collie_optim_cfg* cfg, float* state1, float* state2,
                                      int32_t* row_stamp, const float* grad_scale,
                                      collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  UpdParams p;
  if (fill_common(p, t, b, workspace, workspace_bytes, side, is_bias)) return -1;
  COLLIE_REQUIRE(cfg != nullptr, "optimizer cfg is NULL");
  COLLIE_REQUIRE(cfg->step >= 1, "optimizer step must be >= 1 (got %d)", cfg->step);
  Hyper& h = p.h;
  h.algo = cfg->algo;
  h.dense = dense ? 1 : 0;
  h.eps = cfg->eps;
  h.weight_decay = cfg->weight_decay;
  h.beta2 = cfg->beta2;
  h.c1 = static_cast<float>(1.0 - static_cast<double>(cfg->beta1));
  h.c2 = static_cast<float>(1.0 - static_cast<double>(cfg->beta2));
  h.bc2_sqrt = 1.0f;
  const double lr = cfg->lr;
  int32_t* stamp = nullptr;
  switch (cfg->algo) {
    case COLLIE_OPT_SGD:
      h.neg_lr = static_cast<float>(-lr);
      break;
    case COLLIE_OPT_ADAGRAD:
      COLLIE_REQUIRE(state1 != nullptr, "Adagrad needs state1 = state_sum");
      h.neg_lr = static_cast<float>(-(lr / (1.0 + (cfg->step - 1) * static_cast<double>(cfg->lr_decay))));
      break;
    case COLLIE_OPT_SPARSE_ADAM: {
      COLLIE_REQUIRE(!dense, "SparseAdam does not support dense gradients, please consider Adam instead");
      COLLIE_REQUIRE(state1 && state2, "SparseAdam needs state1 = exp_avg, state2 = exp_avg_sq");
      const double bc1 = 1.0 - pow(static_cast<double>(cfg->beta1), cfg->step);
      const double bc2 = 1.0 - pow(static_cast<double>(cfg->beta2), cfg->step);
      h.neg_lr = static_cast<float>(-(lr * sqrt(bc2) / bc1));
      break;
    }
    case COLLIE_OPT_ADAM: {
      COLLIE_REQUIRE(dense, "Adam does not support sparse gradients, please consider SparseAdam instead");
      COLLIE_REQUIRE(state1 && state2, "Adam needs state1 = exp_avg, state2 = exp_avg_sq");
      const double bc1 = 1.0 - pow(static_cast<double>(cfg->beta1), cfg->step);
      const double bc2 = 1.0 - pow(static_cast<double>(cfg->beta2), cfg->step);
      h.neg_lr = static_cast<float>(-(lr / bc1));
      h.bc2_sqrt = static_cast<float>(sqrt(bc2));
      break;
    }
    default:
      COLLIE_REQUIRE(false, "%d is not a valid optimizer!", cfg->algo);
  }
  if (!dense)
    COLLIE_REQUIRE(cfg->weight_decay == 0.0f,
                   "weight_decay option is not compatible with sparse gradients");
  p.s1 = state1;
  p.s2 = state2;
  p.grad_scale = grad_scale;
  p.step = cfg->step;

  // dense algorithms whose zero-gradient update is not the identity need the untouched-row sweep
  const bool sweep = dense && (cfg->algo == COLLIE_OPT_ADAM || cfg->weight_decay != 0.0f);
  const int64_t rows = side == COLLIE_SIDE_USER ? t->num_users : t->num_items;
  if (sweep) {
    COLLIE_REQUIRE(row_stamp != nullptr,
                   "dense Adam / weight decay need row_stamp = int32[rows] scratch");
    stamp = row_stamp;
    p.stamp = stamp;
  }
  const unsigned grid = grid_for(side == COLLIE_SIDE_USER ? b->batch
                                                          : b->batch * (1 + int64_t{b->num_neg}));
  if (is_bias)
    hipLaunchKernelGGL(bias_rows_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL(emb_rows_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, p);
  if (check_launch("rows_kernel")) return -2;
  if (sweep) {
    const int64_t total = rows * p.D;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dense_sweep_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, p.table, p.s1, p.s2, stamp, rows, p.D, p.step, p.h);
    return check_launch("dense_sweep_kernel");
  }
  return 0;
}

extern "C" int collie_mf_train_export_grad(const collie_mf_tables* t, const collie_mf_batch* b,
                                           void* workspace, size_t workspace_bytes, int32_t side,
                                           int32_t is_bias, int64_t* rows_out, float* vals_out,
                                           int32_t* count_out, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  UpdParams p;
  if (fill_common(p, t, b, workspace, workspace_bytes, side, is_bias)) return -1;
  COLLIE_REQUIRE(rows_out && vals_out && count_out, "export: NULL output");
  p.rows_out = rows_out;
  p.vals_out = vals_out;
  p.count_out = count_out;
  const unsigned grid = grid_for(side == COLLIE_SIDE_USER ? b->batch
                                                          : b->batch * (1 + int64_t{b->num_neg}));
  if (is_bias)
    hipLaunchKernelGGL(bias_rows_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL(emb_rows_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, p);
  return check_launch("export_grad");
}
