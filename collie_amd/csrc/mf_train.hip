// K1: fused calculate_loss + backward coefficients for the implicit MF training step.
//
// Replaces, in one launch: users.long()/neg.T/repeat/flatten copies, 8 nn.Embedding lookups,
// mul+sum+bias adds for B positive and K*B negative pairs (matrix_factorization.py:143-150,
// base_pipeline.py:610-623), the loss (collie/loss/*.py) and the score-level part of autograd.
//
// Mapping (wave64): one wave per sample b.  LPR lanes cooperate on one embedding row (VEC floats
// per lane, 16-byte loads when dim % 4 == 0), so 64/LPR item rows are gathered per wave-instruction;
// the user row is read once per sample and stays in registers for all 1+K dot products.
// After the dot products lane j (0 = positive, 1..K = negatives) owns score j, so the adaptive
// arg-max (first max wins), the WARP first-violation search and the partial-credit factor are
// single ballot/ffs steps.  All five losses give at most ONE negative a non-zero gradient, so the
// backward stage writes just two rows per sample:
//     stage_u[b] = g * I[pos] - g * I[neg_sel]      (gradient of U[user_b])
//     stage_i[b] = g * U[user_b]                    (+ for I[pos], - for I[neg_sel])
// plus g itself (bias gradients) and one 32-bit code per looked-up id that tells the
// segment-reduce/optimizer kernels which staged row, with which sign, belongs to that id.
#include <cstdlib>
#include <type_traits>

#include "train_ws.h"

namespace collie {
namespace {

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  float v[1];
};
template <>
struct alignas(8) Vec<2> {
  float v[2];
};
template <>
struct alignas(16) Vec<4> {
  float v[4];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> vload(const float* p) {
  return *reinterpret_cast<const Vec<VEC>*>(p);
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const Vec<VEC>& x) {
  *reinterpret_cast<Vec<VEC>*>(p) = x;
}
template <int VEC>
__device__ __forceinline__ Vec<VEC> vzero() {
  Vec<VEC> z;
#pragma unroll
  for (int i = 0; i < VEC; ++i) z.v[i] = 0.0f;
  return z;
}
template <int VEC>
__device__ __forceinline__ float vdot(const Vec<VEC>& a, const Vec<VEC>& b) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s = fmaf(a.v[i], b.v[i], s);
  return s;
}

struct FwdParams {
  collie_mf_tables t;
  collie_mf_batch b;
  collie_loss_cfg loss;
  TrainWs ws;
  float* s_pos_out;
  float* s_neg_out;
  int32_t* status;
  int need_grad;
  int step_id;
  const StepDyn* dyn;  // captured-step replay: the tag comes from device memory (train_ws.h)
  int nchunks;  // ceil(dim / VEC)
  float inv_b;
};

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

struct SampleIds {
  int64_t u;
  int64_t item64;
};

__device__ __forceinline__ SampleIds load_ids(const FwdParams& p, int64_t b, int lane, int K) {
  SampleIds s;
  s.u = load_idx(p.b.users, p.b.users_dtype, b);
  s.item64 = 0;
  if (lane == 0) {
    s.item64 = load_idx(p.b.pos_items, p.b.pos_dtype, b);
  } else if (lane <= K) {
    s.item64 = load_idx(p.b.neg_items, p.b.neg_dtype,
                        b * p.b.neg_stride_b + static_cast<int64_t>(lane - 1) * p.b.neg_stride_k);
  }
  return s;
}

// NPASS > 0: number of row-gather passes is a compile-time constant (all 1+K item rows of a sample
// are in flight together and stay in registers); NPASS == 0: generic loop (large K or dim).
template <int VEC, int LPR, int NPASS>
__global__ __launch_bounds__(kBlock) void mf_train_fwd_kernel(const FwdParams p) {
  constexpr int RPW = kWave / LPR;  // item rows gathered per wave-instruction
  constexpr int NP = NPASS > 0 ? NPASS : 1;
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x / kWave;
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int K = p.b.num_neg;
  const int D = p.t.dim;
  const int64_t B = p.b.batch;
  const int npass = (K + 1 + RPW - 1) / RPW;
  const bool my_chunk = lig < p.nchunks;
  const int kind = p.loss.kind;
  const uint32_t key_off = static_cast<uint32_t>(p.t.num_users);

  float loss1 = 0.0f, loss2 = 0.0f;

  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  int64_t b = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block;
  SampleIds next = {0, 0};
  if (b < B) next = load_ids(p, b, lane, K);
  for (; b < B; b += nwaves) {
    // ---- ids: lane 0 = positive item, lanes 1..K = negatives; prefetch the next sample's ----
    const SampleIds cur = next;
    if (b + nwaves < B) next = load_ids(p, b + nwaves, lane, K);
    const int64_t u = cur.u;
    const bool bad_item = lane <= K && (cur.item64 < 0 || cur.item64 >= p.t.num_items);
    const bool bad = (u < 0 || u >= p.t.num_users) || (__ballot(bad_item) != 0ull);
    if (bad) {  // wave-uniform: never touch memory through a bad id
      if (lane == 0 && p.status) atomicOr(p.status, 1);
      if (p.need_grad && lane < 3) {
        const int64_t slot = lane == 0 ? b : B + 2 * b + (lane - 1);
        p.ws.sort.ekeys[slot] = lane == 0 ? 0u : key_off;
        p.ws.sort.ecodes[slot] = kCodeNone;
        if (lane == 0) p.ws.gcoef[b] = 0.0f;
      }
      continue;
    }
    const int item = static_cast<int>(cur.item64);

    const float ub = p.t.user_bias[u];
    const float ib = lane <= K ? p.t.item_bias[item] : 0.0f;
    const float* urow = p.t.user_emb + u * D;

    // ---- 1+K dot products; lane j ends up with dot_j ---------------------------------------
    float mydot = 0.0f;
    Vec<VEC> uvec = vzero<VEC>();
    Vec<VEC> iv[NP];
    if (NPASS > 0) {
      if (my_chunk) uvec = vload<VEC>(urow + lig * VEC);
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        const int j = pass * RPW + grp;
        const int it = __shfl(item, j <= K ? j : K, kWave);
        iv[pass] = vzero<VEC>();
        if (j <= K && my_chunk)
          iv[pass] = vload<VEC>(p.t.item_emb + static_cast<int64_t>(it) * D + lig * VEC);
      }
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        const float part = group_sum<LPR>(vdot<VEC>(uvec, iv[pass]));
        const float got = __shfl(part, (lane % RPW) * LPR, kWave);
        if (lane / RPW == pass) mydot = got;
      }
    } else {
      for (int pass = 0; pass < npass; ++pass) {
        const int j = pass * RPW + grp;
        const int it = __shfl(item, j <= K ? j : K, kWave);
        const float* irow = p.t.item_emb + static_cast<int64_t>(it) * D;
        float part = 0.0f;
        if (j <= K)
          for (int c = lig; c < p.nchunks; c += LPR)
            part += vdot<VEC>(vload<VEC>(urow + c * VEC), vload<VEC>(irow + c * VEC));
        part = group_sum<LPR>(part);
        const float got = __shfl(part, (lane % RPW) * LPR, kWave);
        if (lane / RPW == pass) mydot = got;
      }
    }
    const float s = (mydot + ub) + ib;  // matrix_factorization.py:146-150

    // ---- partial credit: c_j = 1 - sum_m w_m [meta_m[pos] == meta_m[neg_j]] -----------------
    float c = 1.0f;
    if (p.loss.n_meta > 0) {
      const int pos_item = __shfl(item, 0, kWave);
      float frac = 0.0f;
      if (lane >= 1 && lane <= K) {
        for (int m = 0; m < p.loss.n_meta; ++m)
          frac += (p.loss.meta[m][pos_item] == p.loss.meta[m][item]) ? p.loss.meta_weight[m] : 0.0f;
      }
      c = 1.0f - frac;
    }

    // ---- loss, selected negative, score gradient ---------------------------------------------
    const float sp = __shfl(s, 0, kWave);
    const bool isneg = lane >= 1 && lane <= K;
    int sel = 1;
    float ell = 0.0f, g = 0.0f;
    if (kind == COLLIE_LOSS_WARP) {
      const float h = (c - sp) + s;  // warp.py:99
      const unsigned long long viol = __ballot(isneg && h > 0.0f);
      if (viol) {
        sel = __ffsll(static_cast<long long>(viol)) - 1;  // == number of tries (1-based)
        const float ht = __shfl(h, sel, kWave);
        const float w = logf(static_cast<float>(p.loss.num_items) / static_cast<float>(sel));
        ell = w * ht;
        g = -(w * (1.0f + 2.0f * ell)) * p.inv_b;
      }
    } else {
      if (kind == COLLIE_LOSS_ADAPTIVE_BPR || kind == COLLIE_LOSS_ADAPTIVE_HINGE) {
        const float top = wave_max(isneg ? s : -INFINITY);
        const unsigned long long at = __ballot(isneg && s == top);
        if (at) sel = __ffsll(static_cast<long long>(at)) - 1;  // first max, as torch.max
      }
      const float sn = __shfl(s, sel, kWave);
      const float cs = __shfl(c, sel, kWave);
      const float d = sp - sn;
      if (kind == COLLIE_LOSS_BPR || kind == COLLIE_LOSS_ADAPTIVE_BPR) {
        const float sig = sigmoidf_(d);
        ell = cs - sig;
        g = -(sig * (1.0f - sig)) * (1.0f + 2.0f * ell) * p.inv_b;
      } else {
        const float x = cs - d;
        ell = fmaxf(x, 0.0f);
        g = x >= 0.0f ? -(1.0f + 2.0f * ell) * p.inv_b : 0.0f;  // clamp passes grad at the kink
      }
    }
    loss1 += ell;
    loss2 += ell * ell;

    if (p.s_pos_out && lane == 0) p.s_pos_out[b] = s;
    if (p.s_neg_out && isneg) p.s_neg_out[static_cast<int64_t>(lane - 1) * B + b] = s;

    if (!p.need_grad) continue;

    // ---- backward staging ---------------------------------------------------------------------
    // Every looked-up item is stamped as touched (SparseAdam also moves zero-gradient rows); only
    // the user, the positive and the selected negative become sort entries.
    const bool live = g != 0.0f;
    const int sel_item = __shfl(item, sel, kWave);
    const int pos_item_b = __shfl(item, 0, kWave);
    if (lane <= K && p.ws.touch_item) p.ws.touch_item[item] = p.dyn ? p.dyn->tag : touch_tag(p.step_id);
    if (lane < 3) {  // lane 0: user entry, lane 1: positive item, lane 2: selected negative
      const int64_t slot = lane == 0 ? b : B + 2 * b + (lane - 1);
      const uint32_t key =
          lane == 0 ? static_cast<uint32_t>(u)
                    : key_off + static_cast<uint32_t>(lane == 1 ? pos_item_b : sel_item);
      p.ws.sort.ekeys[slot] = key;
      p.ws.sort.ecodes[slot] =
          live ? (static_cast<uint32_t>(b) | (lane == 2 ? kCodeNeg : 0u)) : kCodeNone;
      if (lane == 0) p.ws.gcoef[b] = g;
    }
    if (!live) continue;
    float* su = p.ws.stage_u + b * D;
    float* si = p.ws.stage_i + b * D;
    if (NPASS > 0) {
      // the selected negative's row is already in registers: pass sel / RPW, lane group sel % RPW
      Vec<VEC> pick = iv[0];
#pragma unroll
      for (int pass = 1; pass < NP; ++pass)
        if (sel / RPW == pass) pick = iv[pass];
      Vec<VEC> isel;
#pragma unroll
      for (int i = 0; i < VEC; ++i) isel.v[i] = __shfl(pick.v[i], (sel % RPW) * LPR + lig, kWave);
      if (grp == 0 && my_chunk) {
        Vec<VEC> gu, gi;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          gu.v[i] = g * iv[0].v[i] - g * isel.v[i];
          gi.v[i] = g * uvec.v[i];
        }
        vstore<VEC>(su + lig * VEC, gu);
        vstore<VEC>(si + lig * VEC, gi);
      }
    } else if (grp == 0) {
      const float* prow = p.t.item_emb + static_cast<int64_t>(pos_item_b) * D;
      const float* srow = p.t.item_emb + static_cast<int64_t>(sel_item) * D;
      for (int cidx = lig; cidx < p.nchunks; cidx += LPR) {
        const Vec<VEC> a = vload<VEC>(prow + cidx * VEC);
        const Vec<VEC> n = vload<VEC>(srow + cidx * VEC);
        const Vec<VEC> uu = vload<VEC>(urow + cidx * VEC);
        Vec<VEC> gu, gi;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          gu.v[i] = g * a.v[i] - g * n.v[i];
          gi.v[i] = g * uu.v[i];
        }
        vstore<VEC>(su + cidx * VEC, gu);
        vstore<VEC>(si + cidx * VEC, gi);
      }
    }
  }

  // ---- per-block loss partials (fixed grid -> fixed summation order -> deterministic) ---------
  __shared__ float red[2][kWavesPerBlock];
  if (lane == 0) {
    red[0][wave_in_block] = loss1;
    red[1][wave_in_block] = loss2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c2 = 0.0f;
    for (int w = 0; w < kWavesPerBlock; ++w) {
      a += red[0][w];
      c2 += red[1][w];
    }
    p.ws.loss_partials[2 * blockIdx.x] = a;
    p.ws.loss_partials[2 * blockIdx.x + 1] = c2;
  }
}

// ---- DPP reductions: one VALU instruction per butterfly step inside a 16-lane row ---------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(o);
}
// sum over LPR consecutive lanes; every lane of the group receives the total
template <int LPR>
__device__ __forceinline__ float group_sum_dpp(float v) {
  if (LPR >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LPR >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPR >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if (LPR >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, kWave);
  if (LPR >= 64) v += __shfl_xor(v, 32, kWave);
  return v;
}

struct FastIds {
  int64_t u;       // wave-uniform (scalar loads)
  int64_t pos;     // wave-uniform
  int64_t neg[4];  // per lane group: negative j - 1 of pass
};

// STD: the loaders' dtypes (users / pos int32, negatives int64) -> no dtype dispatch in the loop
template <int LPR, int NP, bool STD>
__device__ __forceinline__ FastIds load_fast_ids(const FwdParams& p, int64_t b, int grp, int K) {
  constexpr int RPW = kWave / LPR;
  FastIds s;
  // b is wave-uniform, but the prefetched ids must stay in VGPRs until they are used one
  // iteration later (an eager v_readfirstlane would put a full s_waitcnt vmcnt(0) right behind
  // the prefetch): hide the uniformity of the address from the compiler.
  int64_t zero = 0;
  asm volatile("" : "+v"(zero));
  const int64_t bv = b + zero;
  if (STD) {
    s.u = static_cast<const int32_t*>(p.b.users)[bv];
    s.pos = static_cast<const int32_t*>(p.b.pos_items)[bv];
  } else {
    s.u = load_idx(p.b.users, p.b.users_dtype, bv);
    s.pos = load_idx(p.b.pos_items, p.b.pos_dtype, bv);
  }
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    int j = pass * RPW + grp;
    j = j < 1 ? 1 : (j > K ? K : j);  // clamped: rows outside 1..K are masked later
    const int64_t off = b * p.b.neg_stride_b + static_cast<int64_t>(j - 1) * p.b.neg_stride_k;
    s.neg[pass] = STD ? static_cast<const int64_t*>(p.b.neg_items)[off]
                      : load_idx(p.b.neg_items, p.b.neg_dtype, off);
  }
  return s;
}

// Fast path of K1 (dim <= LPR * VEC, 1 + K <= NPASS * 64 / LPR): row j of a sample (0 = positive,
// 1..K = negatives) belongs to lane group j % RPW in pass j / RPW and never leaves it: every lane
// of the group holds the row's id, dot product, bias and score, so the loss selection is a few
// compare-selects per lane plus log2(RPW) cross-group steps.  All 1 + K rows are in flight together.
template <int VEC, int LPR, int NPASS, bool STD, bool META>
__global__ __launch_bounds__(kBlock) void mf_train_fwd_fast(const FwdParams p) {
  constexpr int RPW = kWave / LPR;
  const int lane = lane_id();
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // scalar
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int K = p.b.num_neg;
  const int D = p.t.dim;
  const int64_t B = p.b.batch;
  const bool my_chunk = lig < p.nchunks;
  const int kind = p.loss.kind;
  const bool warp_kind = kind == COLLIE_LOSS_WARP;
  const bool bpr_kind = kind == COLLIE_LOSS_BPR || kind == COLLIE_LOSS_ADAPTIVE_BPR;
  const uint32_t key_off = static_cast<uint32_t>(p.t.num_users);
  const int n_meta = META ? p.loss.n_meta : 0;

  float loss1 = 0.0f, loss2 = 0.0f;

  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  int64_t b = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block;
  FastIds next;
  if (b < B) next = load_fast_ids<LPR, NPASS, STD>(p, b, grp, K);
  for (; b < B; b += nwaves) {
    const FastIds cur = next;
    if (b + nwaves < B) next = load_fast_ids<LPR, NPASS, STD>(p, b + nwaves, grp, K);
    // ids are < 2^31 (validated): scalarise them now that they are needed
    const int64_t u = cur.u < 0 || cur.u > 0x7fffffff
                          ? -1
                          : __builtin_amdgcn_readfirstlane(static_cast<int>(cur.u));
    const int64_t pos64 = cur.pos < 0 || cur.pos > 0x7fffffff
                              ? -1
                              : __builtin_amdgcn_readfirstlane(static_cast<int>(cur.pos));
    bool bad_lane = false;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
      bad_lane |= cur.neg[pass] < 0 || cur.neg[pass] >= p.t.num_items;
    const bool bad = (u < 0 || u >= p.t.num_users) || (pos64 < 0 || pos64 >= p.t.num_items) ||
                     (__ballot(bad_lane) != 0ull);
    if (bad) {  // wave-uniform: never touch memory through a bad id
      if (lane == 0 && p.status) atomicOr(p.status, 1);
      if (p.need_grad && lane < 3) {
        const int64_t slot = lane == 0 ? b : B + 2 * b + (lane - 1);
        p.ws.sort.ekeys[slot] = lane == 0 ? 0u : key_off;
        p.ws.sort.ecodes[slot] = kCodeNone;
        if (lane == 0) p.ws.gcoef[b] = 0.0f;
      }
      continue;
    }

    // ---- gathers: user row once, all 1+K item rows (+ biases) in flight ---------------------
    const float ub = p.t.user_bias[u];
    Vec<VEC> uvec = vzero<VEC>();
    if (my_chunk) uvec = vload<VEC>(p.t.user_emb + u * D + lig * VEC);
    Vec<VEC> iv[NPASS];
    float ib[NPASS];
    int item[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int j = pass * RPW + grp;
      item[pass] = static_cast<int>(j == 0 ? pos64 : cur.neg[pass]);
      // rows past K re-read negative K (ids are clamped): no exec masking around the loads
      ib[pass] = p.t.item_bias[item[pass]];
      iv[pass] = vzero<VEC>();
      if (my_chunk)
        iv[pass] = vload<VEC>(p.t.item_emb + static_cast<int64_t>(item[pass]) * D + lig * VEC);
    }
    float s[NPASS];
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
      s[pass] = (group_sum_dpp<LPR>(vdot<VEC>(uvec, iv[pass])) + ub) + ib[pass];

    const float sp =  // row 0 lives on lane 0
        __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s[0])));
    const int pos_item = static_cast<int>(pos64);

    // ---- candidate selection inside the lane, then across the RPW lane groups ----------------
    // adaptive / plain: highest negative score, lowest j on ties (torch.max).
    // warp: lowest j with c_j - s+ + s_j > 0 (warp.py:143-152).
    float best_v = -INFINITY;  // negative score (adaptive) or h (warp) of the candidate
    float best_c = 1.0f;
    int best_j = 0x7fffffff;
    int best_item = 0;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int j = pass * RPW + grp;
      float c = 1.0f;
      if (META && j >= 1 && j <= K) {
        float frac = 0.0f;
        for (int m = 0; m < n_meta; ++m)
          frac += (p.loss.meta[m][pos_item] == p.loss.meta[m][item[pass]]) ? p.loss.meta_weight[m]
                                                                           : 0.0f;
        c = 1.0f - frac;
      }
      const float v = warp_kind ? (c - sp) + s[pass] : s[pass];
      const bool cand = j >= 1 && j <= K &&
                        (warp_kind ? (v > 0.0f && best_j == 0x7fffffff) : (v > best_v));
      if (cand) {
        best_v = v;
        best_c = c;
        best_j = j;
        best_item = item[pass];
      }
    }
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
      const float ov = __shfl_xor(best_v, off, kWave);
      const int oj = __shfl_xor(best_j, off, kWave);
      const int oi = __shfl_xor(best_item, off, kWave);
      float oc = 1.0f;
      if (META) oc = __shfl_xor(best_c, off, kWave);
      const bool take = warp_kind ? (oj < best_j) : (ov > best_v || (ov == best_v && oj < best_j));
      if (take) {
        best_v = ov;
        best_c = oc;
        best_j = oj;
        best_item = oi;
      }
    }
    // every lane now holds the same (best_v, best_c, best_j, best_item)
    int sel = best_j == 0x7fffffff ? 1 : best_j;
    float ell = 0.0f, g = 0.0f;
    if (warp_kind) {
      if (best_j != 0x7fffffff) {
        const float w = logf(static_cast<float>(p.loss.num_items) / static_cast<float>(sel));
        ell = w * best_v;
        g = -(w * (1.0f + 2.0f * ell)) * p.inv_b;
      }
    } else {
      const float d = sp - best_v;
      if (bpr_kind) {
        const float sig = sigmoidf_(d);
        ell = best_c - sig;
        g = -(sig * (1.0f - sig)) * (1.0f + 2.0f * ell) * p.inv_b;
      } else {
        const float x = best_c - d;
        ell = fmaxf(x, 0.0f);
        g = x >= 0.0f ? -(1.0f + 2.0f * ell) * p.inv_b : 0.0f;  // clamp passes grad at the kink
      }
    }
    loss1 += ell;
    loss2 += ell * ell;

    if (p.s_pos_out || p.s_neg_out) {
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int j = pass * RPW + grp;
        if (lig == 0 && j <= K) {
          if (j == 0) {
            if (p.s_pos_out) p.s_pos_out[b] = s[pass];
          } else if (p.s_neg_out) {
            p.s_neg_out[static_cast<int64_t>(j - 1) * B + b] = s[pass];
          }
        }
      }
    }
    if (!p.need_grad) continue;

    // ---- backward staging ---------------------------------------------------------------------
    const bool live = g != 0.0f;
    if (best_j == 0x7fffffff) {
      // no candidate (warp without violation): the "selected" row is negative 1, gradient 0
      best_item = __shfl(item[(1 / RPW) % NPASS], (1 % RPW) * LPR, kWave);
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
      if (lig == 0 && pass * RPW + grp <= K && p.ws.touch_item)
        p.ws.touch_item[item[pass]] = p.dyn ? p.dyn->tag : touch_tag(p.step_id);
    if (lane < 3) {  // lane 0: user entry, lane 1: positive item, lane 2: selected negative
      const int64_t slot = lane == 0 ? b : B + 2 * b + (lane - 1);
      const uint32_t key =
          lane == 0 ? static_cast<uint32_t>(u)
                    : key_off + static_cast<uint32_t>(lane == 1 ? pos_item : best_item);
      p.ws.sort.ekeys[slot] = key;
      p.ws.sort.ecodes[slot] =
          live ? (static_cast<uint32_t>(b) | (lane == 2 ? kCodeNeg : 0u)) : kCodeNone;
      if (lane == 0) p.ws.gcoef[b] = g;
    }
    if (!live) continue;
    // the selected negative's row is in registers: pass sel / RPW, lane group sel % RPW
    Vec<VEC> pick = iv[0];
#pragma unroll
    for (int pass = 1; pass < NPASS; ++pass)
      if (sel / RPW == pass) pick = iv[pass];
    Vec<VEC> isel;
#pragma unroll
    for (int i = 0; i < VEC; ++i) isel.v[i] = __shfl(pick.v[i], (sel % RPW) * LPR + lig, kWave);
    if (grp == 0 && my_chunk) {
      Vec<VEC> gu, gi;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        gu.v[i] = g * iv[0].v[i] - g * isel.v[i];
        gi.v[i] = g * uvec.v[i];
      }
      vstore<VEC>(p.ws.stage_u + b * D + lig * VEC, gu);
      vstore<VEC>(p.ws.stage_i + b * D + lig * VEC, gi);
    }
  }

  __shared__ float red[2][kWavesPerBlock];
  if (lane == 0) {
    red[0][wave_in_block] = loss1;
    red[1][wave_in_block] = loss2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c2 = 0.0f;
    for (int w = 0; w < kWavesPerBlock; ++w) {
      a += red[0][w];
      c2 += red[1][w];
    }
    p.ws.loss_partials[2 * blockIdx.x] = a;
    p.ws.loss_partials[2 * blockIdx.x + 1] = c2;
  }
}

// ---- K1, LDS-parked variant ------------------------------------------------------------------------
// One gather pass per sample: 64 / LPR >= 1 + K lane groups, group j owns row j (0 = positive,
// 1..K = negatives), a lane owns CH 16-byte chunks of its row: chunk c = i * LPR + lig for
// i < CH (dim == 4 * CH * LPR exactly).  The rows never pass through VGPRs on their way in: every
// row chunk is fetched by an LDS-DMA load (global_load_lds_dwordx4: per-lane source address,
// wave-linear LDS destination) into a wave-private parking area, so the gathers in flight cost no
// registers and the kernel runs at 8 waves per SIMD.  Instruction i of the item gather writes
// park[i][lane]; each lane reads back exactly the 16 bytes it asked for, and the staging step at
// the end reads the positive row, the selected negative row and the user row straight out of the
// parking area (no shuffles).  The selection itself is one wave-wide max + one ballot: groups are
// laid out in lane order, so the lowest set lane of the ballot is the lowest j (torch.max's first
// maximum; WARP's first violation).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct LdsParams {
  const float* user_emb;
  const float* item_emb;
  const float* user_bias;
  const float* item_bias;
  const int32_t* users;
  const int32_t* pos_items;
  const int64_t* neg_items;
  int64_t neg_stride_b, neg_stride_k;
  int64_t batch;
  int32_t num_users, num_items;
  int32_t num_neg, dim;
  uint8_t* touch_item;
  uint32_t* ekeys;
  uint32_t* ecodes;
  float* gcoef;
  float* stage_u;
  float* stage_i;
  float* loss_partials;
  int32_t* status;
  int32_t step_id, need_grad;
  const StepDyn* dyn;       // captured-step replay (train_ws.h), else NULL
  float inv_b, warp_items;  // warp_items = float(loss.num_items)
  const int32_t* meta[COLLIE_MAX_META];
  float meta_weight[COLLIE_MAX_META];
  int32_t n_meta;
};

// max over the wave; every lane of a 16-lane row holds the row's max after the DPP steps, the two
// row broadcasts carry it to lane 63
template <int LPR>
__device__ __forceinline__ float wave_max_from_groups(float v) {
  auto mx = [](float a, int ctrl_result) { return fmaxf(a, __int_as_float(ctrl_result)); };
  if (LPR <= 1) v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xf, 0xf, false));
  if (LPR <= 2) v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xf, 0xf, false));
  if (LPR <= 4) v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xf, 0xf, false));
  if (LPR <= 8) v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xf, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xa, 0xf, false));
  v = mx(v, __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xc, 0xf, false));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// N LDS-DMA loads of 16 bytes per lane: source src + I * SRC_BYTES, destination dst + I * DST_F4
// float4 slots (+ lane * 16 by the hardware).  The instruction's immediate offset is added to the
// global address AND to the LDS address, so the LDS base passed for load I is pulled back by it.
// Issued through inline asm: after __builtin_amdgcn_global_load_lds the compiler makes every later
// LDS read wait for ALL outstanding LDS-DMA, which would serialise the two parking areas below (the
// gather of sample b + 1 is in flight while sample b is read back).  The waits are explicit.
template <int I, int N, int SRC_BYTES, int DST_F4>
__device__ __forceinline__ void glds_chunks(const float* src, float4* dst) {
  static_assert(DST_F4 * 16 >= SRC_BYTES, "LDS base would move below the parking area");
  if constexpr (I < N) {
    const uint32_t l = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
        (lptr_t)(reinterpret_cast<char*>(dst) + I * (DST_F4 * 16 - SRC_BYTES)))));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                 ::"v"(src), "s"(l), "n"(I * SRC_BYTES)
                 : "memory", "m0");
    glds_chunks<I + 1, N, SRC_BYTES, DST_F4>(src, dst);
  }
}

template <int NP>
struct LdsIds {
  int u, pos;        // wave-uniform
  int64_t neg[NP];   // per lane group and pass
};

template <int NP>
__device__ __forceinline__ LdsIds<NP> load_lds_ids(const LdsParams& p, int64_t b,
                                                   const int64_t (&neg_lane_off)[NP]) {
  LdsIds<NP> s;
  int64_t zero = 0;
  asm volatile("" : "+v"(zero));  // keep the prefetched ids in VGPRs until they are used
  const int64_t bv = b + zero;
  s.u = p.users[bv];
  s.pos = p.pos_items[bv];
#pragma unroll
  for (int q = 0; q < NP; ++q) s.neg[q] = p.neg_items[bv * p.neg_stride_b + neg_lane_off[q]];
  return s;
}

// KIND: 0 = warp, 1 = (adaptive) bpr, 2 = (adaptive) hinge.
// NP passes of G = 64 / LPR rows: row j = q * G + grp (q = pass) belongs to lane group grp; with
// NP = 2 the lane groups are wider (LPR = 8 at K = 10), so every wave-instruction of the gather
// reads 128 contiguous bytes of each row (a full cache line) instead of 64.
template <int CH, int LPR, int NP, int KIND, bool META>
__global__ __launch_bounds__(kBlock, 4) void mf_train_fwd_lds(const LdsParams p) {
  // Small rows (<= 4 chunk instructions per sample): two parking areas per wave and a software
  // pipeline, 4 waves per SIMD (ML-20M shape: forward kernel 582 -> 554 us).  Larger rows keep one
  // area and 4 waves per SIMD: on HBM-resident tables the kernel runs at the chip's request rate
  // and the halved occupancy of a second area costs more than the overlap gives (1.79 -> 1.84 ms).
  constexpr bool PIPE = CH * NP <= 4;
  constexpr int G = kWave / LPR;
  constexpr int kPassF4 = CH * kWave;      // float4 slots of one pass of the item parking area
  constexpr int kItemF4 = NP * kPassF4;
  constexpr int kUserF4 = CH * LPR;
  // PIPE: two parking areas per wave, the gather of sample b + 1 lands while sample b is read back
  __shared__ float4 park[kWavesPerBlock][PIPE ? 2 : 1][kItemF4 + kUserF4];

  const int lane = lane_id();
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int K = p.num_neg;
  const int D = p.dim;
  const int64_t B = p.batch;
  const uint32_t key_off = static_cast<uint32_t>(p.num_users);
  const int tag = p.dyn ? p.dyn->tag : touch_tag(p.step_id);
  // no stamps for callers that find the looked-up rows themselves (COLLIE_NO_STAMPS)
  const bool stamps = p.need_grad && p.touch_item != nullptr;
  bool row_lane[NP], neg_lane[NP];
  int64_t neg_lane_off[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int j = q * G + grp;
    const int jn = j > K ? K : (j < 1 ? 1 : j);  // negative read for this row slot (1..K)
    neg_lane_off[q] = static_cast<int64_t>(jn - 1) * p.neg_stride_k;
    row_lane[q] = j <= K;
    neg_lane[q] = j >= 1 && j <= K;
  }

  float loss1 = 0.0f, loss2 = 0.0f;
  // A wave owns a contiguous range of samples, so that the per-sample sort entries and gradient
  // coefficients (wave-uniform values) can be collected one per lane and written 64 samples at a
  // time as full lines instead of as 3-lane stores every iteration.
  const int64_t total_waves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t per = (B + total_waves - 1) / total_waves;
  int64_t b = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave_in_block) * per;
  const int64_t b_end = b + per < B ? b + per : B;
  uint32_t e_uk = 0, e_uc = 0, e_pk = 0, e_pc = 0, e_nk = 0, e_nc = 0;
  float e_g = 0.0f;
  int filled = 0;  // samples collected since the last flush
  auto flush = [&](int64_t first) {
    if (lane < filled) {
      const int64_t bb = first + lane;
      p.ekeys[bb] = e_uk;
      p.ecodes[bb] = e_uc;
      p.gcoef[bb] = e_g;
      p.ekeys[B + 2 * bb] = e_pk;
      p.ekeys[B + 2 * bb + 1] = e_nk;
      p.ecodes[B + 2 * bb] = e_pc;
      p.ecodes[B + 2 * bb + 1] = e_nc;
    }
  };
  auto collect = [&](int64_t bb, uint32_t uk, uint32_t pk, uint32_t nk, bool live, float g) {
    if (lane == filled) {
      const uint32_t code = live ? static_cast<uint32_t>(bb) : kCodeNone;
      e_uk = uk;
      e_pk = pk;
      e_nk = nk;
      e_uc = code;
      e_pc = code;
      e_nc = live ? (code | kCodeNeg) : kCodeNone;
      e_g = g;
    }
    if (++filled == kWave) {
      flush(bb - (kWave - 1));
      filled = 0;
    }
  };

  // A sample in flight: its ids decoded, its gathers and scalar loads issued (nothing consumed).
  struct Flight {
    int u, pos_item;
    int item[NP];
    bool bad;
    float ub;
    float ib[NP];
    int stamp[NP];
  };
  auto launch = [&](const LdsIds<NP>& ids, int buf) {
    Flight f;
    f.u = __builtin_amdgcn_readfirstlane(ids.u);
    f.pos_item = __builtin_amdgcn_readfirstlane(ids.pos);
    bool bad_lane = false;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int64_t item64 = (q == 0 && grp == 0) ? static_cast<int64_t>(f.pos_item) : ids.neg[q];
      bad_lane |= static_cast<uint64_t>(item64) >= static_cast<uint64_t>(p.num_items);
      f.item[q] = static_cast<int>(item64);
    }
    f.bad = static_cast<uint32_t>(f.u) >= static_cast<uint32_t>(p.num_users) ||
            (__ballot(bad_lane) != 0ull);
    f.ub = 0.0f;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f.ib[q] = 0.0f;
      f.stamp[q] = 0;
    }
    if (f.bad) return f;  // wave-uniform: never touch memory through a bad id
    // ---- gathers: LDS-DMA, nothing lands in a VGPR -------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // earlier reads of this parking area are done
    float4* const il = &park[wave_in_block][buf][0];
    float4* const ul = &park[wave_in_block][buf][kItemF4];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const float* irow = p.item_emb + static_cast<int64_t>(f.item[q]) * D + lig * 4;
      glds_chunks<0, CH, LPR * 16, kWave>(irow, il + q * kPassF4);
    }
    if (grp == 0) {
      const float* urow = p.user_emb + static_cast<int64_t>(f.u) * D + lig * 4;
      glds_chunks<0, CH, LPR * 16, LPR>(urow, ul);
    }
    f.ub = p.user_bias[f.u];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f.ib[q] = p.item_bias[f.item[q]];
      // the item's stamp: most items of a large batch were already stamped by another sample of
      // this step; a stale read only costs a redundant store
      if (stamps) f.stamp[q] = p.touch_item[f.item[q]];
    }
    return f;
  };

  // Software pipeline over the wave's samples: at the top of iteration b everything issued so far
  // has landed (gather of b, its scalars, the ids of b + 1); the gather of b + 1 and the ids of
  // b + 2 are then issued BEFORE sample b is computed from its parking area.  The compiler's own
  // waits for the VGPR loads are forced to the top of the loop (the "+v" touches): a wait placed
  // later would also drain the DMA it cannot see -- vmcnt is one in-order counter.
  LdsIds<NP> ids_next;
  ids_next.u = ids_next.pos = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) ids_next.neg[q] = 0;
  Flight fl;
  fl.u = fl.pos_item = 0;
  fl.bad = true;
  fl.ub = 0.0f;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    fl.item[q] = 0;
    fl.ib[q] = 0.0f;
    fl.stamp[q] = 0;
  }
  if (b < b_end) {
    LdsIds<NP> first = load_lds_ids<NP>(p, b, neg_lane_off);
    asm volatile("" : "+v"(first.u), "+v"(first.pos));
#pragma unroll
    for (int q = 0; q < NP; ++q) asm volatile("" : "+v"(first.neg[q]));
    if (PIPE) {
      fl = launch(first, 0);
      if (b + 1 < b_end) ids_next = load_lds_ids<NP>(p, b + 1, neg_lane_off);
    } else {
      ids_next = first;
    }
  }
  int pbuf = 0;
  for (; b < b_end; ++b, pbuf ^= (PIPE ? 1 : 0)) {
    if (!PIPE) {  // one parking area: issue, then wait (other waves fill the gap)
      fl = launch(ids_next, 0);
      if (b + 1 < b_end) ids_next = load_lds_ids<NP>(p, b + 1, neg_lane_off);
    }
    asm volatile("" : "+v"(fl.ub), "+v"(ids_next.u), "+v"(ids_next.pos));
#pragma unroll
    for (int q = 0; q < NP; ++q)
      asm volatile("" : "+v"(fl.ib[q]), "+v"(fl.stamp[q]), "+v"(ids_next.neg[q]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const Flight cur = fl;
    if (PIPE && b + 1 < b_end) {
      fl = launch(ids_next, pbuf ^ 1);
      if (b + 2 < b_end) ids_next = load_lds_ids<NP>(p, b + 2, neg_lane_off);
    }
    const int u = cur.u;
    const int pos_item = cur.pos_item;
    int item[NP];
    float ib[NP];
    int stamp[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      item[q] = cur.item[q];
      ib[q] = cur.ib[q];
      stamp[q] = cur.stamp[q];
    }
    const float ub = cur.ub;
    if (cur.bad) {
      if (lane == 0 && p.status) atomicOr(p.status, 1);
      if (p.need_grad) collect(b, 0u, key_off, key_off, false, 0.0f);
      continue;
    }
    float4* const items_l = &park[wave_in_block][pbuf][0];
    float4* const user_l = &park[wave_in_block][pbuf][kItemF4];

    // ---- dot products: chunk order i, then across the LPR lanes of the row ----------------------
    float s[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      float dot = 0.0f;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float4 a = items_l[q * kPassF4 + i * kWave + lane];
        const float4 c = user_l[i * LPR + lig];
        dot = fmaf(c.x, a.x, dot);
        dot = fmaf(c.y, a.y, dot);
        dot = fmaf(c.z, a.z, dot);
        dot = fmaf(c.w, a.w, dot);
      }
      s[q] = (group_sum_dpp<LPR>(dot) + ub) + ib[q];
    }
    const float sp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s[0])));

    // ---- selection: one max + one ballot per pass (rows of pass 0 come before rows of pass 1) ---
    float c[NP], v[NP];
    uint64_t mask[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      c[q] = 1.0f;
      if (META) {
        float frac = 0.0f;
        for (int m = 0; m < p.n_meta; ++m)
          frac += (p.meta[m][pos_item] == p.meta[m][item[q]]) ? p.meta_weight[m] : 0.0f;
        c[q] = 1.0f - frac;
      }
      v[q] = KIND == 0 ? (c[q] - sp) + s[q] : s[q];
    }
    if (KIND == 0) {
#pragma unroll
      for (int q = 0; q < NP; ++q) mask[q] = __ballot(neg_lane[q] & (v[q] > 0.0f));
    } else {
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < NP; ++q) m = fmaxf(m, neg_lane[q] ? v[q] : -INFINITY);
      const float vmax = wave_max_from_groups<LPR>(m);
#pragma unroll
      for (int q = 0; q < NP; ++q) mask[q] = __ballot(neg_lane[q] & (v[q] == vmax));
    }
    int qs = 0;  // pass of the selected row (wave-uniform)
    uint64_t msel = mask[0];
    if (NP > 1 && msel == 0ull) {
      qs = 1;
      msel = mask[NP - 1];
    }
    const bool none = msel == 0ull;  // warp without a violation (or only NaN scores)
    if (none) qs = 0;
    const int wl = none ? LPR : static_cast<int>(__builtin_ctzll(msel));
    const int sel_grp = wl / LPR;
    const int sel = qs * G + sel_grp;  // 1-based index of the selected negative
    const float v_q = qs == 0 ? v[0] : v[NP - 1];
    const float c_q = qs == 0 ? c[0] : c[NP - 1];
    const int item_q = qs == 0 ? item[0] : item[NP - 1];
    const float best_v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_q), wl));
    const float best_c = META ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_q), wl)) : 1.0f;
    const int best_item = __builtin_amdgcn_readlane(item_q, wl);
    float ell = 0.0f, g = 0.0f;
    if (KIND == 0) {
      if (!none) {
        const float w = logf(p.warp_items / static_cast<float>(sel));
        ell = w * best_v;
        g = -(w * (1.0f + 2.0f * ell)) * p.inv_b;
      }
    } else if (!none) {
      const float d = sp - best_v;
      if (KIND == 1) {
        const float sig = sigmoidf_(d);
        ell = best_c - sig;
        g = -(sig * (1.0f - sig)) * (1.0f + 2.0f * ell) * p.inv_b;
      } else {
        const float x = best_c - d;
        ell = fmaxf(x, 0.0f);
        g = x >= 0.0f ? -(1.0f + 2.0f * ell) * p.inv_b : 0.0f;  // clamp passes grad at the kink
      }
    }
    loss1 += ell;
    loss2 += ell * ell;
    if (!p.need_grad) continue;

    // ---- backward staging ---------------------------------------------------------------------
    const bool live = g != 0.0f;
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (stamps && lig == 0 && row_lane[q] && stamp[q] != tag) p.touch_item[item[q]] = static_cast<uint8_t>(tag);
    if (live) {  // lane l stages 16-byte chunks l, l + 64, ... of the two rows (dim 512: two)
#pragma unroll
      for (int c0 = 0; c0 < CH * LPR; c0 += kWave) {
        const int cidx = c0 + lane;
        if (CH * LPR >= kWave || cidx < CH * LPR) {
          const int i = cidx / LPR, l2 = cidx % LPR;
          const float4 uu = user_l[cidx];
          const float4 ip = items_l[i * kWave + l2];
          const float4 is = items_l[qs * kPassF4 + i * kWave + sel_grp * LPR + l2];
          float4 gu, gi;
          gu.x = g * ip.x - g * is.x;
          gu.y = g * ip.y - g * is.y;
          gu.z = g * ip.z - g * is.z;
          gu.w = g * ip.w - g * is.w;
          gi.x = g * uu.x;
          gi.y = g * uu.y;
          gi.z = g * uu.z;
          gi.w = g * uu.w;
          typedef float nt4 __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(*reinterpret_cast<nt4*>(&gu), reinterpret_cast<nt4*>(p.stage_u + b * D + cidx * 4));
          __builtin_nontemporal_store(*reinterpret_cast<nt4*>(&gi), reinterpret_cast<nt4*>(p.stage_i + b * D + cidx * 4));
        }
      }
    }
    collect(b, static_cast<uint32_t>(u), key_off + static_cast<uint32_t>(pos_item),
            key_off + static_cast<uint32_t>(best_item), live, g);
  }
  if (p.need_grad && filled > 0) flush(b_end - filled);

  __shared__ float red[2][kWavesPerBlock];
  if (lane == 0) {
    red[0][wave_in_block] = loss1;
    red[1][wave_in_block] = loss2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, c2 = 0.0f;
    for (int w = 0; w < kWavesPerBlock; ++w) {
      a += red[0][w];
      c2 += red[1][w];
    }
    p.loss_partials[2 * blockIdx.x] = a;
    p.loss_partials[2 * blockIdx.x + 1] = c2;
  }
}

__global__ __launch_bounds__(kBlock) void loss_finalize_kernel(const float* partials, int nblocks,
                                                               float inv_b, float* loss_out,
                                                               int32_t* status) {
  finalize_loss_block(partials, nblocks, inv_b, loss_out, status);
}

template <int LPR, int NP>
void launch_fast(bool std_ids, bool meta, dim3 grid, hipStream_t stream, const FwdParams& p) {
  if (std_ids && !meta)
    hipLaunchKernelGGL((mf_train_fwd_fast<4, LPR, NP, true, false>), grid, dim3(kBlock), 0, stream, p);
  else if (std_ids)
    hipLaunchKernelGGL((mf_train_fwd_fast<4, LPR, NP, true, true>), grid, dim3(kBlock), 0, stream, p);
  else if (!meta)
    hipLaunchKernelGGL((mf_train_fwd_fast<4, LPR, NP, false, false>), grid, dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL((mf_train_fwd_fast<4, LPR, NP, false, true>), grid, dim3(kBlock), 0, stream, p);
}

template <int LPR>
void launch_fast_np(int np, bool std_ids, bool meta, dim3 grid, hipStream_t stream,
                    const FwdParams& p) {
  switch (np) {
    case 1: launch_fast<LPR, 1>(std_ids, meta, grid, stream, p); break;
    case 2: launch_fast<LPR, 2>(std_ids, meta, grid, stream, p); break;
    case 3: launch_fast<LPR, 3>(std_ids, meta, grid, stream, p); break;
    default: launch_fast<LPR, 4>(std_ids, meta, grid, stream, p); break;
  }
}

// fast path: float4 rows (dim % 4 == 0), 16 <= dim <= 256, at most 4 gather passes
int launch_fwd_fast(int lpr, int np, dim3 grid, hipStream_t stream, const FwdParams& p) {
  const bool std_ids = p.b.users_dtype == COLLIE_I32 && p.b.pos_dtype == COLLIE_I32 &&
                       p.b.neg_dtype == COLLIE_I64;
  const bool meta = p.loss.n_meta > 0;
  switch (lpr) {
    case 4: launch_fast_np<4>(np, std_ids, meta, grid, stream, p); break;
    case 8: launch_fast_np<8>(np, std_ids, meta, grid, stream, p); break;
    case 16: launch_fast_np<16>(np, std_ids, meta, grid, stream, p); break;
    case 32: launch_fast_np<32>(np, std_ids, meta, grid, stream, p); break;
    default: launch_fast_np<64>(np, std_ids, meta, grid, stream, p); break;
  }
  return check_launch("mf_train_fwd_fast");
}

template <int CH, int LPR, int NP>
void launch_lds3(int kind3, bool meta, dim3 grid, hipStream_t stream, const LdsParams& q) {
#define COLLIE_LDS_CASE(KIND_)                                                                        \
  if (meta)                                                                                           \
    hipLaunchKernelGGL((mf_train_fwd_lds<CH, LPR, NP, KIND_, true>), grid, dim3(kBlock), 0, stream, q); \
  else                                                                                                \
    hipLaunchKernelGGL((mf_train_fwd_lds<CH, LPR, NP, KIND_, false>), grid, dim3(kBlock), 0, stream, q)
  if (kind3 == 0) { COLLIE_LDS_CASE(0); }
  else if (kind3 == 1) { COLLIE_LDS_CASE(1); }
  else { COLLIE_LDS_CASE(2); }
#undef COLLIE_LDS_CASE
}

template <int CH, int LPR>
void launch_lds2(int np, int kind3, bool meta, dim3 grid, hipStream_t stream, const LdsParams& q) {
  if (np == 1) launch_lds3<CH, LPR, 1>(kind3, meta, grid, stream, q);
  else launch_lds3<CH, LPR, 2>(kind3, meta, grid, stream, q);
}

template <int LPR>
void launch_lds1(int ch, int np, int kind3, bool meta, dim3 grid, hipStream_t stream,
                 const LdsParams& q) {
  switch (ch) {
    case 1: launch_lds2<1, LPR>(np, kind3, meta, grid, stream, q); break;
    case 2: launch_lds2<2, LPR>(np, kind3, meta, grid, stream, q); break;
    case 4: launch_lds2<4, LPR>(np, kind3, meta, grid, stream, q); break;
    default: launch_lds2<8, LPR>(np, kind3, meta, grid, stream, q); break;
  }
}

// LDS-parked path.  Conditions (checked by the caller): float4 rows, dim == 16 * ch * lpr bytes
// exactly, 1 + K <= np * 64 / lpr, the loaders' id dtypes, no score outputs.
int launch_fwd_lds(int lpr, int ch, int np, dim3 grid, hipStream_t stream, const FwdParams& p) {
  LdsParams q;
  memset(&q, 0, sizeof(q));
  q.user_emb = p.t.user_emb;
  q.item_emb = p.t.item_emb;
  q.user_bias = p.t.user_bias;
  q.item_bias = p.t.item_bias;
  q.users = static_cast<const int32_t*>(p.b.users);
  q.pos_items = static_cast<const int32_t*>(p.b.pos_items);
  q.neg_items = static_cast<const int64_t*>(p.b.neg_items);
  q.neg_stride_b = p.b.neg_stride_b;
  q.neg_stride_k = p.b.neg_stride_k;
  q.batch = p.b.batch;
  q.num_users = static_cast<int32_t>(p.t.num_users);
  q.num_items = static_cast<int32_t>(p.t.num_items);
  q.num_neg = p.b.num_neg;
  q.dim = p.t.dim;
  q.touch_item = p.ws.touch_item;
  q.ekeys = p.ws.sort.ekeys;
  q.ecodes = p.ws.sort.ecodes;
  q.gcoef = p.ws.gcoef;
  q.stage_u = p.ws.stage_u;
  q.stage_i = p.ws.stage_i;
  q.loss_partials = p.ws.loss_partials;
  q.status = p.status;
  q.step_id = p.step_id;
  q.dyn = p.dyn;
  q.need_grad = p.need_grad;
  q.inv_b = p.inv_b;
  q.warp_items = static_cast<float>(p.loss.num_items);
  q.n_meta = p.loss.n_meta;
  for (int m = 0; m < COLLIE_MAX_META; ++m) {
    q.meta[m] = p.loss.meta[m];
    q.meta_weight[m] = p.loss.meta_weight[m];
  }
  const int kind = p.loss.kind;
  const int kind3 = kind == COLLIE_LOSS_WARP ? 0
                    : (kind == COLLIE_LOSS_BPR || kind == COLLIE_LOSS_ADAPTIVE_BPR) ? 1 : 2;
  const bool meta = p.loss.n_meta > 0;
  switch (lpr) {
    case 4: launch_lds1<4>(ch, np, kind3, meta, grid, stream, q); break;
    case 8: launch_lds1<8>(ch, np, kind3, meta, grid, stream, q); break;
    default: launch_lds1<16>(ch, np, kind3, meta, grid, stream, q); break;
  }
  return check_launch("mf_train_fwd_lds");
}

template <int VEC>
int launch_fwd(int lpr, dim3 grid, hipStream_t stream, const FwdParams& p) {
  switch (lpr) {
    case 1: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 1, 0>), grid, dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 2, 0>), grid, dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 4, 0>), grid, dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 8, 0>), grid, dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 16, 0>), grid, dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 32, 0>), grid, dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((mf_train_fwd_kernel<VEC, 64, 0>), grid, dim3(kBlock), 0, stream, p); break;
  }
  return check_launch("mf_train_fwd_kernel");
}

}  // namespace

int validate_tables(const collie_mf_tables* t) {
  COLLIE_REQUIRE(t != nullptr, "tables is NULL");
  COLLIE_REQUIRE(t->user_emb && t->item_emb && t->user_bias && t->item_bias,
                 "tables: NULL table pointer");
  COLLIE_REQUIRE(t->dim >= 1, "tables: dim must be >= 1 (got %d)", t->dim);
  COLLIE_REQUIRE(t->num_users >= 1 && t->num_items >= 1, "tables: empty table");
  COLLIE_REQUIRE(t->num_users < (int64_t{1} << 31) && t->num_items < (int64_t{1} << 31),
                 "tables: more than 2^31-1 rows per shard is not supported");
  return 0;
}

int validate_batch(const collie_mf_batch* b) {
  COLLIE_REQUIRE(b != nullptr, "batch is NULL");
  COLLIE_REQUIRE(b->batch >= 1, "batch: batch size must be >= 1");
  COLLIE_REQUIRE(b->users && b->pos_items && b->neg_items, "batch: NULL index pointer");
  COLLIE_REQUIRE(b->num_neg >= 1 && b->num_neg <= COLLIE_MAX_NEG,
                 "batch: num_negative_samples must be in [1, %d] (got %d)", COLLIE_MAX_NEG,
                 b->num_neg);
  COLLIE_REQUIRE(b->batch * (1 + static_cast<int64_t>(b->num_neg)) < (int64_t{1} << 30),
                 "batch: (1 + K) * B must be < 2^30");
  return 0;
}

// widest vector width every row start is aligned for
int pick_vec(const collie_mf_tables* t) {
  auto aligned = [](const void* p, size_t a) { return reinterpret_cast<uintptr_t>(p) % a == 0; };
  if (t->dim % 4 == 0 && aligned(t->user_emb, 16) && aligned(t->item_emb, 16)) return 4;
  if (t->dim % 2 == 0 && aligned(t->user_emb, 8) && aligned(t->item_emb, 8)) return 2;
  return 1;
}

}  // namespace collie

using namespace collie;

extern "C" size_t collie_mf_train_workspace_bytes(int64_t batch, int32_t num_neg, int32_t dim,
                                                  int64_t num_users, int64_t num_items) {
  if (batch < 1 || num_neg < 1 || dim < 1 || num_users < 1 || num_items < 1) return 0;
  return carve_train_ws(nullptr, batch, num_neg, dim, num_users, num_items).bytes;
}

extern "C" int collie_mf_train_forward(const collie_mf_tables* t, const collie_mf_batch* b,
                                       const collie_loss_cfg* loss, void* workspace,
                                       size_t workspace_bytes, int32_t need_grad, int32_t step_id,
                                       float* loss_out, float* s_pos_out, float* s_neg_out,
                                       int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(loss != nullptr && loss_out != nullptr, "loss cfg / loss_out is NULL");
  COLLIE_REQUIRE(loss->kind >= COLLIE_LOSS_BPR && loss->kind <= COLLIE_LOSS_WARP,
                 "%d is not a valid loss function.", loss->kind);
  COLLIE_REQUIRE(loss->n_meta >= 0 && loss->n_meta <= COLLIE_MAX_META,
                 "at most %d metadata arrays are supported", COLLIE_MAX_META);
  if (loss->kind == COLLIE_LOSS_BPR || loss->kind == COLLIE_LOSS_HINGE)
    COLLIE_REQUIRE(b->num_neg == 1, "non-adaptive losses need exactly one negative per sample");
  if (loss->kind == COLLIE_LOSS_WARP)
    COLLIE_REQUIRE(b->num_neg > 1, "Cannot use WARP loss with a single negative sample!");
  // need_grad bit 1 (COLLIE_DEFER_LOSS): loss_out is written by collie_mf_train_advance_step
  const bool defer_loss = (need_grad & COLLIE_DEFER_LOSS) != 0;
  const bool no_stamps = (need_grad & COLLIE_NO_STAMPS) != 0;
  need_grad &= 1;
  const bool from_device = step_id == COLLIE_STEP_DEVICE;
  COLLIE_REQUIRE(step_id >= 1 || from_device, "step_id must be >= 1 (got %d)", step_id);
  COLLIE_REQUIRE(!from_device || need_grad, "COLLIE_STEP_DEVICE is for steps with need_grad");
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  // the 8-bit stamp tags start a new cycle: forget the previous cycle's stamps (train_ws.h)
  // (a replayed step: collie_mf_train_advance_step does it on the device)
  if (need_grad && !from_device && !no_stamps && step_id % kTouchCycle == 0)
    COLLIE_CHECK_HIP(hipMemsetAsync(ws.touch_item, 0, static_cast<size_t>(t->num_items), stream));

  FwdParams p;
  p.t = *t;
  p.b = *b;
  p.loss = *loss;
  p.ws = ws;
  if (no_stamps) p.ws.touch_item = nullptr;
  p.s_pos_out = s_pos_out;
  p.s_neg_out = s_neg_out;
  p.status = status;
  p.need_grad = need_grad;
  p.step_id = step_id;
  p.dyn = from_device ? ws.dyn_cur : nullptr;
  p.inv_b = 1.0f / static_cast<float>(b->batch);
  const int vec = pick_vec(t);
  p.nchunks = (t->dim + vec - 1) / vec;
  int lpr = 1;
  while (lpr < p.nchunks && lpr < kWave) lpr *= 2;
  const int rpw = kWave / lpr;
  const int np = (b->num_neg + 1 + rpw - 1) / rpw;
  const bool fast = vec == 4 && lpr >= 4 && p.nchunks <= lpr && np <= 4;

  int64_t nblocks = (b->batch + kWavesPerBlock - 1) / kWavesPerBlock;
  if (nblocks > kMaxLossBlocks) nblocks = kMaxLossBlocks;
  dim3 grid(static_cast<unsigned>(nblocks));
  // LDS-parked variant: the widest lane groups that give every row of the sample its own group in
  // ONE pass; two passes only when K needs them.  Rows of >= 512 bytes (dim >= 128) are taken in
  // 128-byte pieces (8-lane groups) even when that costs a second pass: a 64-byte piece is half an
  // L2 line, and random rows gathered in half lines read 22 % slower from HBM
  // (profiles/r02_fetch_size_calibration.json; this kernel on the 100 M x 10 M shape: -4 %, the same
  // time on cache-resident tables).  At dim 64 one pass of 4-lane groups stays (two passes of
  // 8-lane groups measured 15 % slower on the cache-resident ML-20M shape).
  int lds_lpr = 16;
  while (lds_lpr > 4 && (lds_lpr > p.nchunks || kWave / lds_lpr < b->num_neg + 1)) lds_lpr /= 2;
  if (lds_lpr == 4 && p.nchunks >= 32 && b->num_neg + 1 <= 2 * (kWave / 8)) lds_lpr = 8;
  const int lds_groups = kWave / lds_lpr;
  const int lds_np = (b->num_neg + 1 + lds_groups - 1) / lds_groups;
  const int lds_ch = p.nchunks / lds_lpr;
  const bool lds_ok = vec == 4 && lds_np <= 2 && lds_groups >= 2 && lds_ch >= 1 &&
                      lds_ch * lds_lpr == p.nchunks &&
                      (lds_ch == 1 || lds_ch == 2 || lds_ch == 4 || lds_ch == 8) &&
                      b->users_dtype == COLLIE_I32 && b->pos_dtype == COLLIE_I32 &&
                      b->neg_dtype == COLLIE_I64 && !s_pos_out && !s_neg_out;
  int rc;
  if (lds_ok) rc = launch_fwd_lds(lds_lpr, lds_ch, lds_np, grid, stream, p);
  else if (fast) rc = launch_fwd_fast(lpr, np, grid, stream, p);
  else if (vec == 4) rc = launch_fwd<4>(lpr, grid, stream, p);
  else if (vec == 2) rc = launch_fwd<2>(lpr, grid, stream, p);
  else rc = launch_fwd<1>(lpr, grid, stream, p);
  if (rc || defer_loss) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(kBlock), 0, stream, ws.loss_partials,
                     static_cast<int>(nblocks), p.inv_b, loss_out, status);
  return check_launch("loss_finalize_kernel");
}

// ---- one batch into the buffers a captured step reads (collie_mf_batch_copy) ----------------------
namespace collie {
namespace {
__device__ __forceinline__ void store_idx(void* p, int dtype, int64_t i, int64_t v) {
  if (dtype == COLLIE_I64) static_cast<int64_t*>(p)[i] = v;
  else static_cast<int32_t*>(p)[i] = static_cast<int32_t>(v);
}
__global__ __launch_bounds__(kBlock) void batch_copy_kernel(const collie_mf_batch s,
                                                            const collie_mf_batch d) {
  const int64_t B = s.batch;
  const int K = s.num_neg;
  const int64_t total = B * (K + 2);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += stride) {
    if (i < B) {
      store_idx(const_cast<void*>(d.users), d.users_dtype, i, load_idx(s.users, s.users_dtype, i));
    } else if (i < 2 * B) {
      const int64_t j = i - B;
      store_idx(const_cast<void*>(d.pos_items), d.pos_dtype, j,
                load_idx(s.pos_items, s.pos_dtype, j));
    } else {
      const int64_t j = i - 2 * B, row = j / K, k = j - row * K;
      store_idx(const_cast<void*>(d.neg_items), d.neg_dtype,
                row * d.neg_stride_b + k * d.neg_stride_k,
                load_idx(s.neg_items, s.neg_dtype, row * s.neg_stride_b + k * s.neg_stride_k));
    }
  }
}
}  // namespace
}  // namespace collie

extern "C" int collie_mf_batch_copy(const collie_mf_batch* src, const collie_mf_batch* dst,
                                    collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_batch(src) || validate_batch(dst)) return -1;
  COLLIE_REQUIRE(src->batch == dst->batch && src->num_neg == dst->num_neg,
                 "batch_copy: shapes differ");
  int64_t blocks = (src->batch * (src->num_neg + 2) + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(batch_copy_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                     stream, *src, *dst);
  return check_launch("batch_copy_kernel");
}
