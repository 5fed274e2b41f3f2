// Explicit-feedback branch of calculate_loss: (users, items, ratings) ->
// preds = forward(users, items), loss = MSELoss / L1Loss (mean) -- collie/model/base/
// base_pipeline.py:635-650, collie/model/matrix_factorization.py:143-157 (y_range sigmoid rescale),
// base_pipeline.py:287-292 (loss strings) -- fused with the score-level backward, staged in the
// same workspace layout as the implicit step so that the coalesce and optimizer kernels are shared:
//     stage_u[b] = g * I[item_b]     stage_i[b] = g * U[user_b]     gcoef[b] = g = dL/ds_b
// Entries per sample: the user row, the item row, and a third "no contribution" entry on the same
// item row (the shared sort always carries 3 entries per sample).  Unlike the pairwise losses the
// user-bias gradient is not identically zero here: the bias update is called with is_bias = 2.
#include "train_ws.h"

namespace collie {

int validate_tables(const collie_mf_tables* t);

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

struct ExplicitParams {
  collie_mf_tables t;
  const void* users;
  const void* items;
  const float* ratings;
  int users_dtype, items_dtype;
  int64_t batch;
  int kind;  // 0 = mse, 1 = mae
  int has_y;
  float y_lo, y_hi;
  TrainWs ws;
  float* preds_out;
  int32_t* status;
  int need_grad;
  int nchunks;
  float inv_b;
};

template <int VEC>
struct XVec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ XVec<VEC> xload(const float* p) {
  XVec<VEC> r;
  if (VEC == 4) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    r.v[0] = x.x; r.v[1 % VEC] = x.y; r.v[2 % VEC] = x.z; r.v[3 % VEC] = x.w;
  } else if (VEC == 2) {
    const float2 x = *reinterpret_cast<const float2*>(p);
    r.v[0] = x.x; r.v[1 % VEC] = x.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void xstore(float* p, const XVec<VEC>& r) {
  if (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
  } else if (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1 % VEC]);
  } else {
    *p = r.v[0];
  }
}

// LPR lanes own one sample (64 / LPR samples per wave); a lane strides the row's VEC-float chunks.
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void mf_explicit_fwd_kernel(const ExplicitParams p) {
  constexpr int GPW = kWave / LPR;
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x / kWave;
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int D = p.t.dim;
  const int64_t B = p.batch;
  const uint32_t key_off = static_cast<uint32_t>(p.t.num_users);
  float loss1 = 0.0f;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t b0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block) * GPW;
       b0 < B; b0 += nwaves * GPW) {
    const int64_t b = b0 + grp;
    const bool on = b < B;
    int64_t u = 0, it = 0;
    float rating = 0.0f;
    if (on) {
      u = load_idx(p.users, p.users_dtype, b);
      it = load_idx(p.items, p.items_dtype, b);
      rating = p.ratings[b];
    }
    const bool bad = on && (u < 0 || u >= p.t.num_users || it < 0 || it >= p.t.num_items);
    if (bad) {  // never touch memory through a bad id
      if (lig == 0 && p.status) atomicOr(p.status, 1);
      u = 0;
      it = 0;
    }
    const float* urow = p.t.user_emb + u * D;
    const float* irow = p.t.item_emb + it * D;
    float dot = 0.0f;
    if (on)
      for (int c = lig; c < p.nchunks; c += LPR) {
        const XVec<VEC> a = xload<VEC>(urow + c * VEC), x = xload<VEC>(irow + c * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) dot = fmaf(a.v[i], x.v[i], dot);
      }
    dot = group_sum<LPR>(dot);
    float s = 0.0f, pred = 0.0f, g = 0.0f, ell = 0.0f;
    if (on && !bad) {
      s = (dot + p.t.user_bias[u]) + p.t.item_bias[it];  // matrix_factorization.py:146-150
      float sig = 0.0f;
      pred = s;
      if (p.has_y) {                                     // :152-157
        sig = sigmoidf_(s);
        pred = sig * (p.y_hi - p.y_lo) + p.y_lo;
      }
      const float diff = pred - rating;
      float dldp;
      if (p.kind == 0) {  // torch.nn.MSELoss(reduction='mean')
        ell = diff * diff;
        dldp = (2.0f * p.inv_b) * diff;
      } else {            // torch.nn.L1Loss(reduction='mean'): sign(0) = 0
        ell = fabsf(diff);
        dldp = diff > 0.0f ? p.inv_b : (diff < 0.0f ? -p.inv_b : 0.0f);
      }
      g = p.has_y ? (dldp * (p.y_hi - p.y_lo)) * ((1.0f - sig) * sig) : dldp;
    }
    if (lig == 0 && on) {
      loss1 += ell;
      if (p.preds_out) p.preds_out[b] = pred;
    }
    if (!p.need_grad || !on) continue;
    const bool live = !bad && g != 0.0f;
    if (lig == 0) {  // entries: user row, item row, and an empty third entry on the same item row
      const uint32_t code = live ? static_cast<uint32_t>(b) : kCodeNone;
      const uint32_t ikey = key_off + static_cast<uint32_t>(it);
      p.ws.sort.ekeys[b] = static_cast<uint32_t>(u);
      p.ws.sort.ecodes[b] = code;
      p.ws.sort.ekeys[B + 2 * b] = ikey;
      p.ws.sort.ecodes[B + 2 * b] = code;
      p.ws.sort.ekeys[B + 2 * b + 1] = ikey;
      p.ws.sort.ecodes[B + 2 * b + 1] = kCodeNone;
      p.ws.gcoef[b] = live ? g : 0.0f;
    }
    if (!live) continue;
    for (int c = lig; c < p.nchunks; c += LPR) {
      const XVec<VEC> a = xload<VEC>(urow + c * VEC), x = xload<VEC>(irow + c * VEC);
      XVec<VEC> gu, gi;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        gu.v[i] = g * x.v[i];
        gi.v[i] = g * a.v[i];
      }
      xstore<VEC>(p.ws.stage_u + b * D + c * VEC, gu);
      xstore<VEC>(p.ws.stage_i + b * D + c * VEC, gi);
    }
  }
  // per-block loss partials: fixed grid, fixed butterfly -> deterministic
  loss1 = group_sum<kWave>(loss1);
  __shared__ float red[kWavesPerBlock];
  if (lane == 0) red[wave_in_block] = loss1;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f;
    for (int w = 0; w < kWavesPerBlock; ++w) a += red[w];
    p.ws.loss_partials[blockIdx.x] = a;
  }
}

// loss = sum / B  (reduction='mean'); fixed-order reduction of the per-block partials
__global__ __launch_bounds__(kBlock) void explicit_finalize_kernel(const float* partials, int nblocks,
                                                                   float inv_b, float* loss_out,
                                                                   int32_t* status) {
  __shared__ double sh[kWavesPerBlock];
  double a = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) a += partials[i];
  for (int o = kWave / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, kWave);
  if (lane_id() == 0) sh[threadIdx.x / kWave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += sh[w];
    const float loss = static_cast<float>(t * static_cast<double>(inv_b));
    *loss_out = loss;
    if (status && !isfinite(loss)) atomicOr(status, 2);  // detect_anomaly's flag (trainer.py:333)
  }
}

template <int VEC>
void launch_explicit(int lpr, dim3 grid, hipStream_t stream, const ExplicitParams& p) {
  switch (lpr) {
    case 1: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 1>), grid, dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 2>), grid, dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 4>), grid, dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 8>), grid, dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 16>), grid, dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 32>), grid, dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((mf_explicit_fwd_kernel<VEC, 64>), grid, dim3(kBlock), 0, stream, p); break;
  }
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_mf_explicit_forward(const collie_mf_tables* t, const void* users,
                                          int32_t users_dtype, const void* items,
                                          int32_t items_dtype, const float* ratings, int64_t batch,
                                          int32_t loss_kind, int32_t has_y_range, float y_lo,
                                          float y_hi, void* workspace, size_t workspace_bytes,
                                          int32_t need_grad, float* loss_out, float* preds_out,
                                          int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(users && items && ratings && loss_out, "explicit_forward: NULL pointer");
  COLLIE_REQUIRE(batch >= 1 && 3 * batch < (int64_t{1} << 30), "explicit_forward: bad batch size");
  COLLIE_REQUIRE(loss_kind == COLLIE_EXPLICIT_MSE || loss_kind == COLLIE_EXPLICIT_MAE,
                 "%d is not a valid explicit loss (0 = mse, 1 = mae)", loss_kind);
  TrainWs ws = carve_train_ws(workspace, batch, 1, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  ExplicitParams p;
  p.t = *t;
  p.users = users;
  p.items = items;
  p.ratings = ratings;
  p.users_dtype = users_dtype;
  p.items_dtype = items_dtype;
  p.batch = batch;
  p.kind = loss_kind;
  p.has_y = has_y_range;
  p.y_lo = y_lo;
  p.y_hi = y_hi;
  p.ws = ws;
  p.preds_out = preds_out;
  p.status = status;
  p.need_grad = need_grad;
  p.inv_b = 1.0f / static_cast<float>(batch);
  auto aligned = [](const void* q, size_t a) { return reinterpret_cast<uintptr_t>(q) % a == 0; };
  int vec = 1;
  if (t->dim % 4 == 0 && aligned(t->user_emb, 16) && aligned(t->item_emb, 16)) vec = 4;
  else if (t->dim % 2 == 0 && aligned(t->user_emb, 8) && aligned(t->item_emb, 8)) vec = 2;
  p.nchunks = (t->dim + vec - 1) / vec;
  int lpr = 1;
  while (lpr < p.nchunks && lpr < kWave) lpr *= 2;
  const int gpw = kWave / lpr;
  int64_t nblocks = (batch + gpw * kWavesPerBlock - 1) / (gpw * kWavesPerBlock);
  if (nblocks > kMaxLossBlocks) nblocks = kMaxLossBlocks;
  const dim3 grid(static_cast<unsigned>(nblocks));
  if (vec == 4) launch_explicit<4>(lpr, grid, stream, p);
  else if (vec == 2) launch_explicit<2>(lpr, grid, stream, p);
  else launch_explicit<1>(lpr, grid, stream, p);
  if (check_launch("mf_explicit_fwd_kernel")) return -2;
  hipLaunchKernelGGL(explicit_finalize_kernel, dim3(1), dim3(kBlock), 0, stream, ws.loss_partials,
                     static_cast<int>(nblocks), p.inv_b, loss_out, status);
  return check_launch("explicit_finalize_kernel");
}
