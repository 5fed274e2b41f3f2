// K10: the dense layers of the hybrid model's head (BASELINE configs[4]) —
// collie/model/hybrid_pretrained_matrix_factorization.py:327-464 of the reference:
//   metadata rows -> Linear + LeakyReLU stack; torch.cat((user_meta_out, U[users], I[items],
//   item_meta_out), 1) -> Linear + LeakyReLU stack -> Linear(.., 1) + user bias + item bias.
//
// Everything is exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense peak), because the parity
// bar is 1e-5 against the reference's fp32 CPU path.  Three GEMM shapes, two kernels:
//   * dense_fwd_kernel:   Y[m, n] = act(sum_k X[m, k] W[n, k] + b[n])  (+ bias-table rows)
//       X is never materialised: an operand is up to four column segments, each the rows of a
//       table selected by an index array (nn.Embedding gather, metadata[ids, :]) or the previous
//       layer's activations, optionally multiplied by LeakyReLU'(saved output) — so the same kernel
//       is the data backward (dX = dZ W, W read through swapped strides) with the activation
//       backward fused into its operand load.
//   * dense_wgrad_kernel: dW[n, k] = sum_m dZ[m, n] X[m, k], db[n] = sum_m dZ[m, n] (a ones
//       column appended to X).  The batch is cut into fixed chunks; each workgroup writes the
//       partial of its chunk and wgrad_reduce_kernel adds the partials in fixed order — no
//       floating-point atomics, deterministic.
// MFMA mapping (both kernels): a lane supplies A[i = lane & 31][kk = lane >> 5] and
// B[kk = lane >> 5][j = lane & 31]; accumulator register r of a lane is D[i = (r & 3) + 8 (r >> 2)
// + 4 (lane >> 5)][j = lane & 31].
#include "common.h"

namespace collie {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKC = 32;       // contraction columns staged per step (forward)
constexpr int kLD = kKC + 4;  // padded LDS row: conflict-free ds_read_b128 of the fragments
constexpr int kFwdBlock = 512;
constexpr int kFwdRows = 32 * (kFwdBlock / kWave);  // 256 operand rows per workgroup

struct Seg {
  const float* src;
  const void* idx;   // NULL: row m of src
  const float* act;  // NULL, or the saved activations this segment's values are back-propagated through
  int64_t ld;
  int64_t act_ld;
  int64_t rows;      // rows of src (bounds of idx)
  int idx_dtype;
  int begin;         // first operand column of the segment
  int end;
};

struct Operand {
  Seg s[COLLIE_DENSE_MAX_SEGS];
  int n_seg;
  int width;
};

__device__ __forceinline__ int64_t seg_row(const Seg& s, int64_t m, bool& ok, int32_t* status) {
  int64_t r = m;
  if (s.idx) {
    r = load_idx(s.idx, s.idx_dtype, m);
    if (r < 0 || r >= s.rows) {
      ok = false;
      if (status) atomicOr(status, 1);
      return 0;
    }
  }
  return r;
}

// operand columns k .. k+3 of row m.  VEC: every segment boundary, ld and base is float4-aligned,
// so the four columns live in one segment and load as one 16-byte access.
template <bool VEC>
__device__ __forceinline__ f32x4 load_op4(const Operand& op, int64_t m, int k, float slope,
                                          int32_t* status) {
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (VEC) {
    if (k >= op.width) return v;
#pragma unroll
    for (int si = 0; si < COLLIE_DENSE_MAX_SEGS; ++si) {
      if (si >= op.n_seg) break;
      const Seg& s = op.s[si];
      if (k < s.begin || k >= s.end) continue;
      bool ok = true;
      const int64_t r = seg_row(s, m, ok, status);
      if (!ok) return v;
      v = *reinterpret_cast<const f32x4*>(s.src + r * s.ld + (k - s.begin));
      if (s.act) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s.act + m * s.act_ld + (k - s.begin));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.0f ? v[e] : v[e] * slope;
      }
      return v;
    }
    return v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = k + e;
    if (kk >= op.width) break;
    for (int si = 0; si < op.n_seg; ++si) {
      const Seg& s = op.s[si];
      if (kk < s.begin || kk >= s.end) continue;
      bool ok = true;
      const int64_t r = seg_row(s, m, ok, status);
      if (!ok) break;
      float x = s.src[r * s.ld + (kk - s.begin)];
      if (s.act) x = s.act[m * s.act_ld + (kk - s.begin)] > 0.0f ? x : x * slope;
      v[e] = x;
      break;
    }
  }
  return v;
}

struct FwdParams {
  Operand x;
  int64_t m;
  const float* w;  // W[n, k] at w[n * w_ld_out + k * w_ld_in]
  int64_t w_ld_out, w_ld_in;
  const float* bias;  // [n_out] or NULL
  int n_in, n_out;
  int leaky;
  float slope;
  // optional: + user_bias[users[m]] + item_bias[items[m]] (hybrid_pretrained...:394-398)
  const void* users;
  const void* items;
  int users_dtype, items_dtype;
  const float* user_bias;
  const float* item_bias;
  int64_t num_users, num_items;
  float* y;
  int64_t y_ld;
  int32_t* status;
};

// 8 waves x 32 operand rows; every wave multiplies its rows by the same NTILES x 32 output columns.
template <int NTILES, bool VEC>
__global__ __launch_bounds__(kFwdBlock) void dense_fwd_kernel(const FwdParams p) {
  constexpr int NT = 32 * NTILES;
  __shared__ __attribute__((aligned(16))) float Xs[kFwdRows * kLD];
  __shared__ __attribute__((aligned(16))) float Ws[NT * kLD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kFwdRows;
  const int n0 = blockIdx.y * NT;

  f32x16 acc[NTILES];
#pragma unroll
  for (int nt = 0; nt < NTILES; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;

  for (int k0 = 0; k0 < p.n_in; k0 += kKC) {
    __syncthreads();
    // ---- operand rows: 256 x 32 floats, one 16-byte slot per thread and pass -------------------
#pragma unroll
    for (int s = tid; s < kFwdRows * (kKC / 4); s += kFwdBlock) {
      const int r = s >> 3;
      const int c4 = s & 7;
      const int64_t m = m0 + r;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (m < p.m) v = load_op4<VEC>(p.x, m, k0 + c4 * 4, p.slope, p.status);
      *reinterpret_cast<f32x4*>(&Xs[r * kLD + c4 * 4]) = v;
    }
    // ---- weights: NT x 32 floats; threads walk along the contiguous direction of W -------------
    if (p.w_ld_in == 1) {
      for (int e = tid; e < NT * kKC; e += kFwdBlock) {
        const int n = e >> 5, kk = e & 31;
        float v = 0.0f;
        if (n0 + n < p.n_out && k0 + kk < p.n_in)
          v = p.w[static_cast<int64_t>(n0 + n) * p.w_ld_out + (k0 + kk)];
        Ws[n * kLD + kk] = v;
      }
    } else {
      for (int e = tid; e < NT * kKC; e += kFwdBlock) {
        const int kk = e / NT, n = e - kk * NT;
        float v = 0.0f;
        if (n0 + n < p.n_out && k0 + kk < p.n_in)
          v = p.w[static_cast<int64_t>(n0 + n) * p.w_ld_out + static_cast<int64_t>(k0 + kk) * p.w_ld_in];
        Ws[n * kLD + kk] = v;
      }
    }
    __syncthreads();

    const float* arow = &Xs[(wave * 32 + col) * kLD + 4 * h];
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + 8 * q);
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) {
      const float* brow = &Ws[(nt * 32 + col) * kLD + 4 * h];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(brow + 8 * q);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b[0], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b[1], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b[2], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], b[3], acc[nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: a lane owns output column n and 16 rows; 32 lanes store 128 contiguous bytes ----
#pragma unroll
  for (int nt = 0; nt < NTILES; ++nt) {
    const int n = n0 + nt * 32 + col;
    if (n >= p.n_out) continue;
    const float bn = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m >= p.m) continue;
      float v = acc[nt][r] + bn;
      if (p.leaky) v = v > 0.0f ? v : v * p.slope;
      if (p.user_bias) {
        const int64_t u = load_idx(p.users, p.users_dtype, m);
        const int64_t it = load_idx(p.items, p.items_dtype, m);
        if (u < 0 || u >= p.num_users || it < 0 || it >= p.num_items) {
          if (p.status) atomicOr(p.status, 1);
        } else {
          v = (v + p.user_bias[u]) + p.item_bias[it];
        }
      }
      p.y[m * p.y_ld + n] = v;
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------
constexpr int kWgBlock = 256;
constexpr int kWgTile = 128;          // output rows (n) and columns (k) per workgroup
constexpr int kWgLD = kWgTile + 32;   // LDS row stride: the two lane halves hit disjoint banks

struct WgParams {
  Operand dz;  // [m, n_out]
  Operand x;   // [m, n_in]
  int64_t m;
  int64_t chunk;  // operand rows per partial (multiple of 32)
  int n_out, n_in;
  float slope;
  float* partial;  // [nchunks, n_out, n_in + 1]; column n_in = bias gradient
  int32_t* status;
};

template <bool VEC>
__global__ __launch_bounds__(kWgBlock) void dense_wgrad_kernel(const WgParams p) {
  __shared__ __attribute__((aligned(16))) float Zs[32 * kWgLD];
  __shared__ __attribute__((aligned(16))) float Xs[32 * kWgLD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int kb = blockIdx.y * kWgTile;
  const int nb = blockIdx.z * kWgTile;
  const int64_t mbeg = static_cast<int64_t>(blockIdx.x) * p.chunk;
  int64_t mend = mbeg + p.chunk;
  if (mend > p.m) mend = p.m;

  f32x16 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[kt][i] = 0.0f;

  for (int64_t ms = mbeg; ms < mend; ms += 32) {
    __syncthreads();
#pragma unroll
    for (int s = tid; s < 32 * (kWgTile / 4); s += kWgBlock) {
      const int r = s >> 5;
      const int c4 = s & 31;
      const int64_t m = ms + r;
      f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
      if (m < mend) {
        const int n = nb + c4 * 4;
        if (n < p.n_out) z = load_op4<VEC>(p.dz, m, n, p.slope, p.status);
        const int k = kb + c4 * 4;
        if (k < p.n_in) x = load_op4<VEC>(p.x, m, k, p.slope, p.status);
        // the ones column that turns the last output column into the bias gradient
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e == p.n_in) x[e] = 1.0f;
      }
      *reinterpret_cast<f32x4*>(&Zs[r * kWgLD + c4 * 4]) = z;
      *reinterpret_cast<f32x4*>(&Xs[r * kWgLD + c4 * 4]) = x;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = Zs[(2 * t + h) * kWgLD + wave * 32 + col];
      const float* xr = &Xs[(2 * t + h) * kWgLD + col];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[kt * 32], acc[kt], 0, 0, 0);
    }
  }

  const int ldp = p.n_in + 1;
  float* out = p.partial + static_cast<int64_t>(blockIdx.x) * p.n_out * ldp;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = kb + kt * 32 + col;
    if (k > p.n_in) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nb + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < p.n_out) out[static_cast<int64_t>(n) * ldp + k] = acc[kt][r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           int nchunks, int n_out, int n_in,
                                                           float* __restrict__ dw,
                                                           float* __restrict__ db) {
  const int ldp = n_in + 1;
  const int64_t total = static_cast<int64_t>(n_out) * ldp;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= total) return;
  // four interleaved running sums, combined in a fixed order: deterministic
  float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += partial[(c + q) * total + e];
  }
  for (int q = 0; c < nchunks; ++c, ++q) s[q] += partial[c * total + e];
  const float sum = (s[0] + s[1]) + (s[2] + s[3]);
  const int n = static_cast<int>(e / ldp);
  const int k = static_cast<int>(e - static_cast<int64_t>(n) * ldp);
  if (k < n_in) {
    if (dw) dw[static_cast<int64_t>(n) * n_in + k] = sum;
  } else if (db) {
    db[n] = sum;
  }
}

// ---- host side ------------------------------------------------------------------------------------
int build_operand(Operand& op, const collie_dense_operand* in, int64_t m, bool& vec,
                  const char* what) {
  COLLIE_REQUIRE(in != nullptr, "%s: NULL operand", what);
  COLLIE_REQUIRE(in->n_seg >= 1 && in->n_seg <= COLLIE_DENSE_MAX_SEGS, "%s: n_seg %d not in 1..%d",
                 what, in->n_seg, COLLIE_DENSE_MAX_SEGS);
  memset(&op, 0, sizeof(op));
  op.n_seg = in->n_seg;
  int at = 0;
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  for (int i = 0; i < in->n_seg; ++i) {
    const collie_dense_seg& s = in->seg[i];
    COLLIE_REQUIRE(s.src != nullptr && s.width >= 1 && s.ld >= s.width && s.rows >= 1,
                   "%s: segment %d: bad src / width / ld / rows", what, i);
    COLLIE_REQUIRE(s.idx != nullptr || s.rows >= m,
                   "%s: segment %d: %lld rows for %lld operand rows", what, i,
                   static_cast<long long>(s.rows), static_cast<long long>(m));
    COLLIE_REQUIRE(s.idx == nullptr || s.idx_dtype == COLLIE_I32 || s.idx_dtype == COLLIE_I64,
                   "%s: segment %d: bad index dtype", what, i);
    COLLIE_REQUIRE(s.act_ref == nullptr || s.act_ld >= s.width, "%s: segment %d: bad act_ld", what,
                   i);
    Seg& d = op.s[i];
    d.src = s.src;
    d.idx = s.idx;
    d.act = s.act_ref;
    d.ld = s.ld;
    d.act_ld = s.act_ld;
    d.rows = s.rows;
    d.idx_dtype = s.idx_dtype;
    d.begin = at;
    at += s.width;
    d.end = at;
    if (s.width % 4 || s.ld % 4 || !al16(s.src) || (s.act_ref && (s.act_ld % 4 || !al16(s.act_ref))))
      vec = false;
  }
  op.width = at;
  return 0;
}

template <bool VEC>
void launch_fwd(const FwdParams& p, hipStream_t stream) {
  const unsigned gx = static_cast<unsigned>((p.m + kFwdRows - 1) / kFwdRows);
  if (p.n_out <= 32) {
    hipLaunchKernelGGL((dense_fwd_kernel<1, VEC>), dim3(gx, 1), dim3(kFwdBlock), 0, stream, p);
  } else if (p.n_out <= 64) {
    hipLaunchKernelGGL((dense_fwd_kernel<2, VEC>), dim3(gx, 1), dim3(kFwdBlock), 0, stream, p);
  } else {
    const unsigned gy = static_cast<unsigned>((p.n_out + 127) / 128);
    hipLaunchKernelGGL((dense_fwd_kernel<4, VEC>), dim3(gx, gy), dim3(kFwdBlock), 0, stream, p);
  }
}

int64_t wgrad_chunks(int64_t m, int64_t* chunk_out) {
  int64_t nchunks = (m + 255) / 256;
  if (nchunks > 1024) nchunks = 1024;
  if (nchunks < 1) nchunks = 1;
  int64_t chunk = (m + nchunks - 1) / nchunks;
  chunk = (chunk + 31) / 32 * 32;
  nchunks = (m + chunk - 1) / chunk;
  *chunk_out = chunk;
  return nchunks;
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_dense_forward(const collie_dense_operand* x, int64_t m, const float* w,
                                    int64_t w_ld_out, int64_t w_ld_in, const float* bias,
                                    int32_t n_in, int32_t n_out, int32_t leaky, float slope,
                                    const void* users, int32_t users_dtype, const float* user_bias,
                                    int64_t num_users, const void* items, int32_t items_dtype,
                                    const float* item_bias, int64_t num_items, float* y,
                                    int64_t y_ld, int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(m >= 1 && m < (int64_t{1} << 39), "dense_forward: bad row count %lld",
                 static_cast<long long>(m));
  COLLIE_REQUIRE(w && y && n_in >= 1 && n_out >= 1 && y_ld >= n_out, "dense_forward: bad W / Y");
  COLLIE_REQUIRE((m + kFwdRows - 1) / kFwdRows < (int64_t{1} << 31), "dense_forward: too many rows");
  FwdParams p;
  memset(&p, 0, sizeof(p));
  bool vec = true;
  if (build_operand(p.x, x, m, vec, "dense_forward")) return -1;
  COLLIE_REQUIRE(p.x.width == n_in, "dense_forward: operand has %d columns, layer expects %d",
                 p.x.width, n_in);
  const bool row_bias = user_bias != nullptr || item_bias != nullptr;
  if (row_bias) {
    COLLIE_REQUIRE(user_bias && item_bias && users && items && num_users >= 1 && num_items >= 1,
                   "dense_forward: the bias-table add needs users, items and both bias tables");
    COLLIE_REQUIRE((users_dtype == COLLIE_I32 || users_dtype == COLLIE_I64) &&
                       (items_dtype == COLLIE_I32 || items_dtype == COLLIE_I64),
                   "dense_forward: bad index dtype");
  }
  p.m = m;
  p.w = w;
  p.w_ld_out = w_ld_out;
  p.w_ld_in = w_ld_in;
  p.bias = bias;
  p.n_in = n_in;
  p.n_out = n_out;
  p.leaky = leaky;
  p.slope = slope;
  p.users = users;
  p.items = items;
  p.users_dtype = users_dtype;
  p.items_dtype = items_dtype;
  p.user_bias = row_bias ? user_bias : nullptr;
  p.item_bias = item_bias;
  p.num_users = num_users;
  p.num_items = num_items;
  p.y = y;
  p.y_ld = y_ld;
  p.status = status;
  if (vec) launch_fwd<true>(p, stream);
  else launch_fwd<false>(p, stream);
  return check_launch("dense_fwd_kernel");
}

extern "C" size_t collie_dense_weight_grad_workspace_bytes(int64_t m, int32_t n_out, int32_t n_in) {
  if (m < 1 || n_out < 1 || n_in < 1) return 0;
  int64_t chunk;
  const int64_t nchunks = wgrad_chunks(m, &chunk);
  return align_up(static_cast<size_t>(nchunks) * n_out * (n_in + 1) * sizeof(float), 256);
}

extern "C" int collie_dense_weight_grad(const collie_dense_operand* dz,
                                        const collie_dense_operand* x, int64_t m, int32_t n_out,
                                        int32_t n_in, float slope, float* dw, float* db,
                                        void* workspace, size_t workspace_bytes, int32_t* status,
                                        collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(m >= 1 && n_out >= 1 && n_in >= 1, "dense_weight_grad: bad sizes");
  COLLIE_REQUIRE(dw || db, "dense_weight_grad: no output");
  const size_t need = collie_dense_weight_grad_workspace_bytes(m, n_out, n_in);
  COLLIE_REQUIRE(workspace && workspace_bytes >= need,
                 "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  WgParams p;
  memset(&p, 0, sizeof(p));
  bool vec = true;
  if (build_operand(p.dz, dz, m, vec, "dense_weight_grad(dz)")) return -1;
  if (build_operand(p.x, x, m, vec, "dense_weight_grad(x)")) return -1;
  COLLIE_REQUIRE(p.dz.width == n_out && p.x.width == n_in,
                 "dense_weight_grad: operands are [m, %d] and [m, %d], expected [m, %d] and [m, %d]",
                 p.dz.width, p.x.width, n_out, n_in);
  if (n_in % 4) vec = false;  // the ones column must start a 16-byte slot
  int64_t chunk;
  const int64_t nchunks = wgrad_chunks(m, &chunk);
  p.m = m;
  p.chunk = chunk;
  p.n_out = n_out;
  p.n_in = n_in;
  p.slope = slope;
  p.partial = static_cast<float*>(workspace);
  p.status = status;
  const dim3 grid(static_cast<unsigned>(nchunks), static_cast<unsigned>((n_in + 1 + kWgTile - 1) / kWgTile),
                  static_cast<unsigned>((n_out + kWgTile - 1) / kWgTile));
  if (vec) hipLaunchKernelGGL(dense_wgrad_kernel<true>, grid, dim3(kWgBlock), 0, stream, p);
  else hipLaunchKernelGGL(dense_wgrad_kernel<false>, grid, dim3(kWgBlock), 0, stream, p);
  if (check_launch("dense_wgrad_kernel")) return -2;
  const int64_t total = static_cast<int64_t>(n_out) * (n_in + 1);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256),
                     0, stream, p.partial, static_cast<int>(nchunks), n_out, n_in, dw, db);
  return check_launch("wgrad_reduce_kernel");
}
