// K10: the dense layers of the hybrid model's head (BASELINE configs[4]) —
// collie/model/hybrid_pretrained_matrix_factorization.py:327-464 of the reference:
//   metadata rows -> Linear + LeakyReLU stack; torch.cat((user_meta_out, U[users], I[items],
//   item_meta_out), 1) -> Linear + LeakyReLU stack -> Linear(.., 1) + user bias + item bias.
//
// Everything is exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense peak), because the parity
// bar is 1e-5 against the reference's fp32 CPU path.  Three GEMM shapes, two kernel families:
//   * dense_fwd_*:   Y[m, n] = act(sum_k X[m, k] W[n, k] + b[n])  (+ bias-table rows)
//       X is never materialised: an operand is up to four column segments, each the rows of a
//       table selected by an index array (nn.Embedding gather, metadata[ids, :]) or the previous
//       layer's activations, optionally multiplied by LeakyReLU'(saved output) — so the same kernel
//       is the data backward (dX = dZ W, W read through swapped strides) with the activation
//       backward fused into its operand load.
//   * dense_wgrad_*: dW[n, k] = sum_m dZ[m, n] X[m, k], db[n] = sum_m dZ[m, n].  The batch is cut
//       into fixed chunks; each workgroup writes the partial of its chunk and wgrad_reduce_kernel
//       adds the partials in fixed order — no floating-point atomics, deterministic.
// Each family has a generic kernel (any widths / alignment, element-wise staging; dense_fwd_kernel,
// dense_wgrad_kernel with the bias gradient as a ones column of X) and a fast kernel for the
// float4-aligned case every real layer is in (dense_fwd_fast_kernel, dense_wgrad_fast_kernel:
// register-prefetched tiles, see the comments there).
// MFMA mapping (both kernels): a lane supplies A[i = lane & 31][kk = lane >> 5] and
// B[kk = lane >> 5][j = lane & 31]; accumulator register r of a lane is D[i = (r & 3) + 8 (r >> 2)
// + 4 (lane >> 5)][j = lane & 31].
#include "common.h"

namespace collie {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// collie_dense_set_kernels(): which kernel family the dense entry points dispatch to
int g_dense_kernels = COLLIE_DENSE_AUTO;

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

// operand columns k .. k+3 of row m, element by element (generic path: any widths / alignment; the
// float4-aligned case has its own kernels below)
__device__ __forceinline__ f32x4 load_op4(const Operand& op, int64_t m, int k, float slope,
                                          int32_t* status) {
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
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
  int has_act;  // some segment of x carries an activation reference
  // optional: + addend[(m mod add_period), n] before the activation (a partial product that only
  // depends on m mod add_period, computed once: collie_dense_forward_add)
  const float* addend;
  int64_t add_ld, add_period;
};

// 8 waves x 32 operand rows; every wave multiplies its rows by the same NTILES x 32 output columns.
template <int NTILES>
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
      if (m < p.m) v = load_op4(p.x, m, k0 + c4 * 4, p.slope, p.status);
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
      if (p.addend) v += p.addend[(m % p.add_period) * p.add_ld + n];
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


// ---- fast path (every segment, ld and base float4-aligned) -----------------------------------------
// Same tiling and MFMA mapping as dense_fwd_kernel, but (i) the row numbers of the gathered
// segments are read once per workgroup into LDS, (ii) the global loads of chunk c+1 (operand rows and
// weights) are issued into registers before the MFMAs of chunk c and written to LDS after them, so
// HBM/L2 latency hides under the matrix pipe.
// Fast-path geometry: 4 waves x 32 rows per workgroup and 32-column chunks, so that TWO workgroups
// are resident per CU (registers and LDS both allow it) and run out of phase: the prologue (row ids,
// first fetch), the LDS commits and the epilogue of one hide under the MFMAs of the other.  (8 waves
// x 64-column chunks, one workgroup per CU, measured 10-25 % slower: every wave of the CU sat in the
// same non-MFMA phase at the same time.)
constexpr int kFKC = 32;
constexpr int kFLD = kFKC + 4;
constexpr int kFBlock = 256;
constexpr int kFRows = 32 * (kFBlock / kWave);  // 128

// slot e of the NT x 32 weight tile = four consecutive elements along W's contiguous direction:
// row-major W (a Linear weight): (n, kk .. kk+3); transposed W (data backward): (n .. n+3, kk),
// lanes walking kk so that the four LDS stores of a slot hit distinct banks.
// WVEC: one predicated 16-byte load (the host guarantees alignment and n_in % 4 == 0, and
// n_out % 4 == 0 for the transposed form); otherwise four predicated scalar loads.  No else-paths:
// a join of two load paths would make the compiler wait for every outstanding load right here.
template <bool WVEC>
__device__ __forceinline__ void load_w_slot(const FwdParams& p, int n0, int k0, int e, float (&w)[4],
                                            int& n, int& kk) {
  const bool rowmajor = p.w_ld_in == 1;
  n = rowmajor ? (e >> 3) : (e >> 5) * 4;
  kk = rowmajor ? (e & 7) * 4 : (e & 31);
  const int dn = rowmajor ? 0 : 1, dk = rowmajor ? 1 : 0;
  const int64_t base = static_cast<int64_t>(n0 + n) * p.w_ld_out + static_cast<int64_t>(k0 + kk) * p.w_ld_in;
  const int64_t step = rowmajor ? p.w_ld_in : p.w_ld_out;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = 0.0f;
  if (WVEC) {
    if (n0 + n + 3 * dn < p.n_out && k0 + kk + 3 * dk < p.n_in) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = v[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + n + dn * j < p.n_out && k0 + kk + dk * j < p.n_in) w[j] = p.w[base + j * step];
  }
}

template <int NTILES, bool WVEC, bool ACT>
__global__ __launch_bounds__(kFBlock, (NTILES == 4 && WVEC && !ACT ? 3 : 1)) void dense_fwd_fast_kernel(const FwdParams p) {
  constexpr int NT = 32 * NTILES;
  constexpr int WSLOTS = NT * kFKC / 4 / kFBlock;  // 16-byte weight slots per thread: 4, 2 or 1
  constexpr int WS = WSLOTS > 0 ? WSLOTS : 1;
  __shared__ __attribute__((aligned(16))) float Xs[kFRows * kFLD];
  __shared__ __attribute__((aligned(16))) float Ws[NT * kFLD];
  __shared__ int32_t rowid[COLLIE_DENSE_MAX_SEGS][kFRows];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kFRows;
  const int n0 = blockIdx.y * NT;
  const int c4 = tid & 7;    // this thread's 16-byte column slot inside a 32-column chunk
  const int r0 = tid >> 3;   // and its rows r0, r0 + 32, r0 + 64, r0 + 96

  // ---- row numbers of the gathered segments (-1: row past the end or id out of range) -----------
  for (int e = tid; e < COLLIE_DENSE_MAX_SEGS * kFRows; e += kFBlock) {
    const int si = e / kFRows, r = e - si * kFRows;
    int32_t v = -1;
    if (si < p.x.n_seg && m0 + r < p.m) {
      const Seg& s = p.x.s[si];
      if (s.idx) {
        const int64_t id = load_idx(s.idx, s.idx_dtype, m0 + r);
        if (id >= 0 && id < s.rows) v = static_cast<int32_t>(id);
        else if (p.status) atomicOr(p.status, 1);
      } else {
        v = 0;  // identity: the row is m0 + r
      }
    }
    rowid[si][r] = v;
  }
  __syncthreads();

  f32x16 acc[NTILES];
#pragma unroll
  for (int nt = 0; nt < NTILES; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;

  f32x4 xr[4], ar[ACT ? 4 : 1];
  float wr[WS][4];
  int wn[WS], wk[WS];

  // Loads of the NEXT chunk.  The segment of this thread's column slot is selected with static
  // indices (descriptors stay in SGPRs: no dependent kernarg loads in the loop) and nothing here
  // consumes a loaded value, so every load stays in flight until commit().
  auto fetch = [&](int k0) {
    const int k = k0 + c4 * 4;
    const float* src = nullptr;
    const float* act = nullptr;
    int64_t ld = 0, act_ld = 0;
    int off = 0, si = 0;
    bool gathered = false, on = false;
#pragma unroll
    for (int j = 0; j < COLLIE_DENSE_MAX_SEGS; ++j) {
      const bool hit = j < p.x.n_seg && k >= p.x.s[j].begin && k < p.x.s[j].end;
      src = hit ? p.x.s[j].src : src;
      act = hit ? p.x.s[j].act : act;
      ld = hit ? p.x.s[j].ld : ld;
      act_ld = hit ? p.x.s[j].act_ld : act_ld;
      off = hit ? k - p.x.s[j].begin : off;
      gathered = hit ? p.x.s[j].idx != nullptr : gathered;
      si = hit ? j : si;
      on = on || hit;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 32 * i;
      const int32_t id = rowid[si][r];
      const bool valid = on && id >= 0;
      const int64_t row = gathered ? static_cast<int64_t>(id) : m0 + r;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      f32x4 a = {1.0f, 1.0f, 1.0f, 1.0f};
      if (valid) {
        v = *reinterpret_cast<const f32x4*>(src + row * ld + off);
        if (ACT && act) a = *reinterpret_cast<const f32x4*>(act + (m0 + r) * act_ld + off);
      }
      xr[i] = v;
      if (ACT) ar[i] = a;
    }
#pragma unroll
    for (int j = 0; j < WS; ++j) {
      const int e = tid + kFBlock * j;
      if (e < NT * kFKC / 4) load_w_slot<WVEC>(p, n0, k0, e, wr[j], wn[j], wk[j]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = xr[i];
      if (ACT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ar[i][e] > 0.0f ? v[e] : v[e] * p.slope;
      }
      *reinterpret_cast<f32x4*>(&Xs[(r0 + 32 * i) * kFLD + c4 * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < WS; ++j) {
      const int e = tid + kFBlock * j;
      if (e < NT * kFKC / 4) {
        if (p.w_ld_in == 1) {
          *reinterpret_cast<f32x4*>(&Ws[wn[j] * kFLD + wk[j]]) =
              f32x4{wr[j][0], wr[j][1], wr[j][2], wr[j][3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) Ws[(wn[j] + q) * kFLD + wk[j]] = wr[j][q];
        }
      }
    }
  };

  fetch(0);
  for (int k0 = 0; k0 < p.n_in; k0 += kFKC) {
    __syncthreads();  // every wave is done reading the previous chunk
    commit();
    __syncthreads();
    if (k0 + kFKC < p.n_in) fetch(k0 + kFKC);  // in flight during the MFMAs below

    const float* arow = &Xs[(wave * 32 + col) * kFLD + 4 * h];
    const float* brow = &Ws[col * kFLD + 4 * h];
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(arow + 8 * q);
    // B fragments one step ahead of the MFMAs that consume them
    f32x4 b = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
    for (int step = 0; step < NTILES * 4; ++step) {
      const int nt = step >> 2, q = step & 3;
      f32x4 bn = b;
      if (step + 1 < NTILES * 4)
        bn = *reinterpret_cast<const f32x4*>(brow + ((step + 1) >> 2) * 32 * kFLD + 8 * ((step + 1) & 3));
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b[0], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b[1], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b[2], acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], b[3], acc[nt], 0, 0, 0);
      b = bn;
    }
  }

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
      if (p.addend) v += p.addend[(m % p.add_period) * p.add_ld + n];
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
  int dz_scalar;   // fast path only: dz is one column (the final Linear(.., 1)), read as scalars
};

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
        if (n < p.n_out) z = load_op4(p.dz, m, n, p.slope, p.status);
        const int k = kb + c4 * 4;
        if (k < p.n_in) x = load_op4(p.x, m, k, p.slope, p.status);
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


// Fast path of the weight gradient: a thread's 16-byte column slot is fixed for the whole kernel, so
// its segment (source, index array, activation reference) is resolved once; the rows of slab s+1
// are loaded into registers (and the row numbers of slab s+2 fetched) while the MFMAs of slab s
// run.  The bias gradient is a column sum of the staged dZ slab (no ones column: n_in = 128 / 256 /
// 384 would otherwise cost a whole extra block of column tiles).  Tiles outside [n_out) x [n_in)
// are skipped by whole waves.
struct Slot {
  const float* src;
  const void* idx;
  const float* act;
  int64_t ld, act_ld, rows;
  int idx_dtype;
  int idx_shift;  // log2 of the index element size
  int colo;       // column inside the segment
  bool on;
};

__device__ __forceinline__ Slot make_slot(const Operand& op, int k) {
  Slot t;
  t.on = false;
  t.src = nullptr; t.idx = nullptr; t.act = nullptr;
  t.ld = t.act_ld = t.rows = 0; t.idx_dtype = 0; t.idx_shift = 2; t.colo = 0;
#pragma unroll
  for (int j = 0; j < COLLIE_DENSE_MAX_SEGS; ++j) {
    if (j < op.n_seg && k >= op.s[j].begin && k < op.s[j].end) {
      const Seg& s = op.s[j];
      t.src = s.src; t.idx = s.idx; t.act = s.act; t.ld = s.ld; t.act_ld = s.act_ld;
      t.rows = s.rows; t.idx_dtype = s.idx_dtype; t.idx_shift = s.idx_dtype == COLLIE_I64 ? 3 : 2;
      t.colo = k - s.begin; t.on = true;
    }
  }
  return t;
}

// id of operand row m for a gathered slot, as the raw low 32 bits (little endian) of the index
// element: one predicated load for either index dtype and nothing else — no validation, no
// widening, no alternative value written to the same register (an if/else or a dtype branch would
// make the compiler wait for every outstanding load at the join).  Tables have < 2^31 rows, so a
// valid id is its low word; slot_load resolves and range-checks it one slab later.
__device__ __forceinline__ int32_t slot_row(const Slot& t, int64_t m, int64_t mend) {
  int32_t raw = -1;
  if (t.on && t.idx && m < mend)
    raw = *reinterpret_cast<const int32_t*>(static_cast<const char*>(t.idx) + (m << t.idx_shift));
  return raw;
}

// raw value and activation reference of one slot; nothing is consumed here, so both loads stay in
// flight until slot_value() at the LDS store
template <bool SCALAR1 = false>
__device__ __forceinline__ void slot_load(const Slot& t, int32_t raw, int64_t m, int64_t mend,
                                          f32x4& v, f32x4& a, int32_t* status) {
  v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  a = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
  if (!t.on || m >= mend) return;
  const int64_t row = t.idx ? static_cast<int64_t>(raw) : m;
  if (row < 0 || row >= t.rows) {
    if (status) atomicOr(status, 1);
    return;
  }
  if (SCALAR1) {
    v[0] = t.src[row * t.ld + t.colo];
    if (t.act) a[0] = t.act[m * t.act_ld + t.colo];
    return;
  }
  v = *reinterpret_cast<const f32x4*>(t.src + row * t.ld + t.colo);
  if (t.act) a = *reinterpret_cast<const f32x4*>(t.act + m * t.act_ld + t.colo);
}

__device__ __forceinline__ f32x4 slot_value(f32x4 v, const f32x4& a, float slope) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.0f ? v[e] : v[e] * slope;
  return v;
}

template <bool ZS>  // ZS: dz is a single column (final Linear(.., 1)), loaded as scalars
__global__ __launch_bounds__(kWgBlock) void dense_wgrad_fast_kernel(const WgParams p) {
  __shared__ __attribute__((aligned(16))) float Zs[32 * kWgLD];
  __shared__ __attribute__((aligned(16))) float Xs[32 * kWgLD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  // grid.x = column block of dW (fastest), grid.y = row chunk: the workgroups that re-read the same
  // dZ slabs for different column blocks are dispatched together, so all but the first read hit the
  // Infinity Cache instead of HBM
  const int kb = blockIdx.x * kWgTile;
  const int nb = blockIdx.z * kWgTile;
  const int64_t mbeg = static_cast<int64_t>(blockIdx.y) * p.chunk;
  int64_t mend = mbeg + p.chunk;
  if (mend > p.m) mend = p.m;
  const int c4 = tid & 31;   // 16-byte column slot
  const int r0 = tid >> 5;   // rows r0, r0 + 8, r0 + 16, r0 + 24 of a slab

  const Slot zt = make_slot(p.dz, nb + c4 * 4);
  const Slot xt = make_slot(p.x, kb + c4 * 4);
  const bool my_ntile = nb + wave * 32 < p.n_out;
  int nkt = (p.n_in - kb + 31) / 32;
  if (nkt > 4) nkt = 4;
  const bool do_bias = blockIdx.x == 0 && tid < kWgTile;

  f32x16 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[kt][i] = 0.0f;
  float bsum = 0.0f;

  int32_t zrow[4], xrow[4];   // raw row numbers of the slab being fetched next
  f32x4 zr[4], xr[4], za[4], xa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    zrow[i] = slot_row(zt, mbeg + r0 + 8 * i, mend);
    xrow[i] = slot_row(xt, mbeg + r0 + 8 * i, mend);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    slot_load<ZS>(zt, zrow[i], mbeg + r0 + 8 * i, mend, zr[i], za[i], p.status);
    slot_load(xt, xrow[i], mbeg + r0 + 8 * i, mend, xr[i], xa[i], p.status);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    zrow[i] = slot_row(zt, mbeg + 32 + r0 + 8 * i, mend);
    xrow[i] = slot_row(xt, mbeg + 32 + r0 + 8 * i, mend);
  }

  for (int64_t ms = mbeg; ms < mend; ms += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(&Zs[(r0 + 8 * i) * kWgLD + c4 * 4]) = slot_value(zr[i], za[i], p.slope);
      *reinterpret_cast<f32x4*>(&Xs[(r0 + 8 * i) * kWgLD + c4 * 4]) = slot_value(xr[i], xa[i], p.slope);
    }
    __syncthreads();
    if (ms + 32 < mend) {
      // slab s+1 into registers (row numbers were fetched one slab ago), then the numbers of s+2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        slot_load<ZS>(zt, zrow[i], ms + 32 + r0 + 8 * i, mend, zr[i], za[i], p.status);
        slot_load(xt, xrow[i], ms + 32 + r0 + 8 * i, mend, xr[i], xa[i], p.status);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        zrow[i] = slot_row(zt, ms + 64 + r0 + 8 * i, mend);
        xrow[i] = slot_row(xt, ms + 64 + r0 + 8 * i, mend);
      }
    }
    if (do_bias) {
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 32; r += 2) {
        s0 += Zs[r * kWgLD + tid];
        s1 += Zs[(r + 1) * kWgLD + tid];
      }
      bsum += s0 + s1;
    }
    if (my_ntile) {
      // operands of step t+1 are read from LDS before the MFMAs of step t
      const float* zq = &Zs[h * kWgLD + wave * 32 + col];
      const float* xq = &Xs[h * kWgLD + col];
      float a = zq[0];
      float b[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) b[kt] = xq[kt * 32];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float an = a, bn[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bn[kt] = b[kt];
        if (t + 1 < 16) {
          an = zq[2 * (t + 1) * kWgLD];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) bn[kt] = xq[2 * (t + 1) * kWgLD + kt * 32];
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          if (kt < nkt) acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kt], acc[kt], 0, 0, 0);
        a = an;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) b[kt] = bn[kt];
      }
    }
  }

  const int ldp = p.n_in + 1;
  float* out = p.partial + static_cast<int64_t>(blockIdx.y) * p.n_out * ldp;
  if (my_ntile) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = kb + kt * 32 + col;
      if (kt >= nkt || k >= p.n_in) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nb + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < p.n_out) out[static_cast<int64_t>(n) * ldp + k] = acc[kt][r];
      }
    }
  }
  if (do_bias && nb + tid < p.n_out) out[static_cast<int64_t>(nb + tid) * ldp + p.n_in] = bsum;
}

// ---- weight gradient, LDS-DMA variant: the 128 x {128, 256, 384} layers -----------------------------
// Measured on the fast kernel above (configs[4], per 128-column block of dW): its load pipeline alone
// takes 160 us (it moves 1.1 GB: every column block re-reads dZ and the activation reference), its
// MFMA + LDS loop alone 230 us, both together 290 us against an MFMA floor of 150 us -- the blocks of a
// 256- / 384-wide layer read dZ two / three times and the register-staged loads of one slab ahead do
// not cover the memory latency under that load.  Here ONE workgroup of 8 waves owns all of dW for its
// row chunk (128 x n_in: wave w holds rows 32 (w & 3) and the k-tiles of half w >> 2, NKT x 16
// accumulators), so dZ, its reference and X are read once, and no operand passes through VGPRs:
// global_load_lds_dwordx4 writes 16-row slabs straight into a ring of kWdBuf LDS images, two slabs
// ahead of the MFMAs, one barrier per slab.  LDS image of a slab: [tile][row][32 floats] for the four
// 32-column tiles of dZ, of its reference and the n_in / 32 tiles of X -- a DMA instruction fills 8
// rows of one tile (8 lanes x 16 bytes per row), and the fragment reads (row 2t + h, column = lane)
// are conflict-free without padding.  LeakyReLU' is applied where the A value is read.  Same MFMA
// order over the rows and the same chunks as the fast kernel: dW is bit-identical to it; db sums the
// A values the MFMAs consume (other order than the fast kernel's column sums, deterministic).
constexpr int kWdBlock = 512;
constexpr int kWdBuf = 3;   // the waits below are written for three
typedef __attribute__((address_space(3))) void* wd_lptr_t;

// LDS-DMA through inline asm.  After __builtin_amdgcn_global_load_lds the compiler's wait-count pass
// makes EVERY later LDS read (and every fence) wait for ALL outstanding DMA -- it cannot tell the
// images of a ring apart -- which turns a multi-slab pipeline into load-then-compute.  The asm form
// is invisible to that pass; the s_waitcnt vmcnt(N) in the loop below are the only waits, and the
// barrier is the bare instruction (no fence).  `lds` must be wave-uniform.
// `lds`: byte address inside the workgroup's LDS (lds_address() of a __shared__ array + offset).
template <int BYTES>
__device__ __forceinline__ void dma_to_lds(const void* g, uint32_t lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane(lds);
  if (BYTES == 16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l)
                 : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l)
                 : "memory", "m0");
}

// LDS byte address of a __shared__ array (taken of the array itself, so the cast folds to a constant)
template <typename T>
__device__ __forceinline__ uint32_t lds_address(T* array) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((wd_lptr_t)array));
}

// the barrier instruction alone: LDS reads are complete once their values are used, the DMA writes
// once the explicit vmcnt wait before the barrier has passed
__device__ __forceinline__ void bare_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NT x KT = 32-column tiles of dZ (n_out / 32) and of X (n_in / 32); R = operand rows per slab.
// The 8 waves split the NT x KT output tiles: wave -> (n-tile, group of KT / KH k-tiles); when there
// are fewer tiles than waves (32 x 64: two), the waves also split the rows of a slab (RG row groups)
// and their accumulators are added in fixed order at the end.
template <int NT, int KT, int R, bool ACT>
__global__ __launch_bounds__(kWdBlock, 2) void dense_wgrad_dma_kernel(const WgParams p) {
  constexpr int kWaves = kWdBlock / kWave;
  constexpr int KH = NT * KT >= kWaves ? kWaves / NT : KT;
  constexpr int NKT = KT / KH;                             // k-tiles (accumulator tiles) per wave
  constexpr int RG = kWaves / (NT * KH);
  constexpr int TS = R / 2 / RG;                           // MFMA steps per wave and slab
  constexpr int kParts = R / 8;                            // DMA instructions per tile and slab
  constexpr int kTileF = R * 32;
  constexpr int kZF = NT * kTileF;                         // floats of the dZ image (and of its reference)
  constexpr int kXOff = kZF * (ACT ? 2 : 1);
  constexpr int kSlabF = kXOff + KT * kTileF;
  constexpr int kQ = (NT * (ACT ? 2 : 1) + KT) * kParts;   // DMA instructions per slab
  constexpr int J = (kQ + kWaves - 1) / kWaves;            // ... per wave (the last ones may repeat
                                                           // instruction q - kQ: same bytes, same place)
  static_assert(NT * KH * RG == kWaves && KH * NKT == KT && TS * RG * 2 == R, "bad tiling");
  static_assert(RG == 1 || kWdBuf * kSlabF >= kWaves * (NKT * 1024 + 64), "the exchange area of the row groups");
  __shared__ __attribute__((aligned(16))) float smem[kWdBuf * kSlabF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int nt = wave % NT;
  const int kh = wave / NT % KH;
  const int rg = wave / (NT * KH);
  const int64_t mbeg = static_cast<int64_t>(blockIdx.x) * p.chunk;
  int64_t mend = mbeg + p.chunk;
  if (mend > p.m) mend = p.m;
  const int nslab = static_cast<int>((mend - mbeg) / R);

  // this wave's DMA streams: instruction q = wave + 8 j of the slab
  const char* src[J];        // per lane: address of its 16 bytes in the first slab (dense rows), or of
                             // the column inside row 0 (gathered rows)
  const char* idp[J];        // per lane: address of its row's id in the first slab
  int64_t step[J];           // bytes from one slab to the next (dense), 0 (gathered)
  int64_t ldb[J];            // row pitch in bytes (gathered), 0 (dense)
  int idstep[J];             // bytes from one slab's ids to the next; 0: the stream does not gather
  uint32_t nrows[J];
  int ldsoff[J];
  const int lrow = lane >> 3;              // row of the 8-row group
  const int lcol = (lane & 7) * 4;         // column inside the 32-column tile
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int q = wave + kWaves * j;
    if (q >= kQ) q -= kQ;
    const int ti = q / kParts;             // tile: dZ tiles, reference tiles, X tiles
    const int part = q % kParts;
    const int row = part * 8 + lrow;
    idp[j] = nullptr;
    idstep[j] = 0;
    ldb[j] = 0;
    nrows[j] = 1;
    if (ti < NT * (ACT ? 2 : 1)) {
      const bool ref = ti >= NT;
      const int tile = ref ? ti - NT : ti;
      const float* base = ref ? p.dz.s[0].act : p.dz.s[0].src;
      const int64_t ld = ref ? p.dz.s[0].act_ld : p.dz.s[0].ld;
      src[j] = reinterpret_cast<const char*>(base + (mbeg + row) * ld + tile * 32 + lcol);
      step[j] = R * ld * 4;
      ldsoff[j] = (ref ? kZF : 0) + tile * kTileF + part * 256;
    } else {
      const int tile = ti - NT * (ACT ? 2 : 1);
      const int k = tile * 32;
      // the segment of this k-tile (segment bounds are multiples of 32: checked by the host)
      const float* base = p.x.s[0].src;
      const void* idx = p.x.s[0].idx;
      int64_t ld = p.x.s[0].ld, rows = p.x.s[0].rows;
      int dt = p.x.s[0].idx_dtype, begin = 0;
#pragma unroll
      for (int si = 1; si < COLLIE_DENSE_MAX_SEGS; ++si) {
        if (si < p.x.n_seg && k >= p.x.s[si].begin) {
          base = p.x.s[si].src; idx = p.x.s[si].idx; ld = p.x.s[si].ld; rows = p.x.s[si].rows;
          dt = p.x.s[si].idx_dtype; begin = p.x.s[si].begin;
        }
      }
      ldsoff[j] = kXOff + tile * kTileF + part * 256;
      if (idx) {
        const int sh = dt == COLLIE_I64 ? 3 : 2;
        src[j] = reinterpret_cast<const char*>(base + (k - begin) + lcol);
        step[j] = 0;
        ldb[j] = ld * 4;
        nrows[j] = static_cast<uint32_t>(rows);
        idp[j] = static_cast<const char*>(idx) + ((mbeg + row) << sh);
        idstep[j] = R << sh;
      } else {
        src[j] = reinterpret_cast<const char*>(base + (mbeg + row) * ld + (k - begin) + lcol);
        step[j] = R * ld * 4;
      }
    }
  }

  // The ids of gathered rows travel by LDS-DMA as well (low word of the index element: tables have
  // < 2^31 rows, see slot_row()): a load into a VGPR inside the loop makes the compiler wait for
  // EVERY outstanding memory operation where the value is used (vmcnt counts in order), which would
  // cut the DMA depth to one slab.  Ring of two per (wave, stream): written one step ahead.
  __shared__ int32_t idring[2][kWaves][J][kWave];
  const uint32_t smem_at = lds_address(smem), idring_at = lds_address(&idring[0][0][0][0]);
  bool any_bad = false;
  auto fetch_ids = [&](int slot) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (idstep[j] != 0) {                      // uniform: the stream gathers
        dma_to_lds<4>(idp[j], idring_at + (((slot * kWaves + wave) * J + j) * kWave) * 4);
        idp[j] += idstep[j];
      }
    }
  };
  auto issue = [&](int buf, int slot) {
    // all id reads first: an asm statement is a barrier for the compiler's scheduling, and a read
    // placed between two DMA instructions would expose its LDS latency once per stream
    uint32_t r[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      r[j] = 0;
      if (idstep[j] != 0) r[j] = static_cast<uint32_t>(idring[slot][wave][j][lane]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool bad = idstep[j] != 0 && r[j] >= nrows[j];   // flagged at the end, row 0 read instead
      any_bad |= bad;
      const char* g = src[j] + static_cast<int64_t>(bad ? 0u : r[j]) * ldb[j];
      dma_to_lds<16>(g, smem_at + (buf * kSlabF + ldsoff[j]) * 4);
      src[j] += step[j];
    }
  };

  f32x16 acc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[kt][i] = 0.0f;
  float bsum = 0.0f;

  // step k of the pipeline = ids of slab k + 1, then the J data instructions of slab k -- the data
  // last, so "all but the last J operations" = everything older than the newest slab
  fetch_ids(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (1 < nslab) fetch_ids(1);
  issue(0, 0);
  if (1 < nslab) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J) : "memory");
    if (2 < nslab) fetch_ids(0);
    issue(1, 1);
  }

  int buf = 0;
  for (int s = 0; s < nslab; ++s) {
    // slab s and the ids of slab s + 2 have landed: everything but the data of slab s + 1
    if (s + 1 < nslab) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bare_barrier();      // everyone's part has; the image of slab s - 1 is free
    if (s + 2 < nslab) {
      int nb = buf + 2;
      if (nb >= kWdBuf) nb -= kWdBuf;
      // the ids of slab s + 2 are in slot s & 1; those of slab s + 3 go to the other slot
      if (s + 3 < nslab) fetch_ids((s + 1) & 1);
      issue(nb, s & 1);
    }
    const float* zq = smem + buf * kSlabF + nt * kTileF + rg * (TS * 64) + h * 32 + col;
    const float* xq = smem + buf * kSlabF + kXOff + kh * NKT * kTileF + rg * (TS * 64) + h * 32 + col;
    // operands of step t + 1 are read from LDS before the MFMAs of step t
    float a = zq[0], r = ACT ? zq[kZF] : 1.0f;
    float b[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) b[kt] = xq[kt * kTileF];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      float an = a, rn = r, bn[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) bn[kt] = b[kt];
      if (t + 1 < TS) {
        an = zq[(t + 1) * 64];
        if (ACT) rn = zq[kZF + (t + 1) * 64];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) bn[kt] = xq[kt * kTileF + (t + 1) * 64];
      }
      const float av = r > 0.0f ? a : a * p.slope;
      bsum += av;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[kt], acc[kt], 0, 0, 0);
      a = an;
      r = rn;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) b[kt] = bn[kt];
    }
    buf = buf + 1 == kWdBuf ? 0 : buf + 1;
  }

  if (RG > 1) {
    // row groups: accumulators through LDS, added in group order by the waves of group 0
    bare_barrier();                             // every wave is done with the slab images
    float* ex = smem + wave * (NKT * 1024 + 64);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) ex[kt * 1024 + i * 64 + lane] = acc[kt][i];
    ex[NKT * 1024 + lane] = bsum;
    bare_barrier();
    if (rg == 0) {
#pragma unroll
      for (int g = 1; g < RG; ++g) {
        const float* o = smem + (g * NT * KH + wave) * (NKT * 1024 + 64);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[kt][i] += o[kt * 1024 + i * 64 + lane];
        bsum += o[NKT * 1024 + lane];
      }
    }
  }

  const int ldp = p.n_in + 1;
  float* out = p.partial + static_cast<int64_t>(blockIdx.x) * p.n_out * ldp;
  if (rg == 0) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int k = (kh * NKT + kt) * 32 + col;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int n = nt * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        out[static_cast<int64_t>(n) * ldp + k] = acc[kt][rr];
      }
    }
  }
  if (any_bad && p.status) atomicOr(p.status, 1);
  // db: lanes (col, h = 0 / 1) of the waves kh = 0 hold the even- / odd-row sums of column nt * 32 + col
  const float other = __shfl_xor(bsum, 32);
  if (rg == 0 && kh == 0 && h == 0) out[static_cast<int64_t>(nt * 32 + col) * ldp + p.n_in] = bsum + other;
}

// ---- forward / data gradient, LDS-DMA variant: 128 output columns, n_in = 128 NCH ------------------
// The register-staged kernel above synchronises its workgroup twice per 32-column chunk and its loads
// are one chunk ahead; SQ counters put it at 2x its MFMA time.  Here the W fragments of a wave (its 32
// output columns x all of n_in) stay in REGISTERS for the whole kernel, a workgroup of 8 waves walks a
// run of 64-row tiles, and the operand arrives by LDS-DMA in 64 x 128 images (a ring of three, two
// ahead of the MFMAs; with an activation reference -- the data gradients -- two images of operand +
// reference, one ahead), one barrier per image.  Image layout: [k-tile][row][8 slots of 16 bytes]; a
// DMA destination is wave-linear, so the conflict-free image is an XOR swizzle on the per-lane SOURCE
// address (slot s of row r holds chunk s ^ (r & 7)), and the A-fragment reads apply the same XOR.
// Same k order and pairing as the register-staged kernel: results are bit-identical to it.
// Stores of a finished tile are issued after the NEXT barrier, before that iteration's DMA, so the
// newest J memory operations are always the DMA of the newest image (what the vmcnt waits assume).
constexpr int kDfBlock = 512;

// KS = 2 (n_in = 384: the W fragments of a 32 x 384 block do not fit one wave's registers): the two
// waves of a column tile split the K-TILES of every image instead of its rows -- each holds half of W
// and both 32-row sub-tiles -- and the second hands its sums to the first through LDS at the end of a
// tile (one more rounding order than the register-staged kernel: compared to 1e-5, not bit for bit).
// NT = 32-column output tiles (4, 2, 1): the 8 waves are NT column tiles x 8 / NT groups of 32 rows,
// so a tile has ROWS = 256 / NT operand rows (64, 128, 256).
template <int NCH, bool ACT, int KS, int KT, bool ADD, int NT>
__global__ __launch_bounds__(kDfBlock, 2) void dense_fwd_dma_kernel(const FwdParams p) {
  constexpr int NB = ACT ? 2 : 3;
  constexpr int ROWS = 32 * (8 / NT);      // operand rows per tile
  constexpr int RP = ROWS / 64;            // 64-row parts of a tile: one DMA instruction per wave each
  constexpr int kImgF4 = KT * ROWS * 8;    // one image: KT k-tiles of 32 columns
  constexpr int kSlabF4 = kImgF4 * (ACT ? 2 : 1);
  constexpr int J = KT * RP * (ACT ? 2 : 1);
  constexpr int CW = KT * 32;              // operand columns per image
  constexpr int NS = KT * 4;               // MFMA steps (of 4) per image
  static_assert(KS == 1 || (KT == 4 && NT == 4), "the K split is written for 128-column images");
  constexpr int kWaves = kDfBlock / kWave;
  constexpr int NW = NCH * NS / KS;        // W fragments (float4) per wave
  constexpr int NA = KS;                   // accumulator tiles per wave
  __shared__ f32x4 smem[NB * kSlabF4];
  __shared__ int32_t idring[2][kWaves][RP][kWave];
  __shared__ float xbuf[KS == 2 ? 4 * 2 * 16 * kWave : 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int nt = wave % NT;                // this wave's 32 output columns
  const int rh = wave / NT;                // KS = 1: its 32 rows of the tile; KS = 2: its k-tiles
  const int lrow = wave * 8 + (lane >> 3); // DMA: this lane's row of a 64-row part of the tile
  const int lslot = lane & 7;

  const int64_t ntiles = (p.m + ROWS - 1) / ROWS;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * per;
  int64_t t1 = t0 + per;
  if (t1 > ntiles) t1 = ntiles;
  if (t0 >= t1) return;
  const int nslab = static_cast<int>(t1 - t0) * NCH;

  // W fragments: W[n][k = 8 qq + 4 h + e], the k pairing of the register-staged kernel; fragment i
  // of a wave is qq = i (KS = 1), or chunk i / 8, k-tile 2 rh + (i / 4 & 1), q = i & 3 (KS = 2)
  f32x4 wf[NW];
  {
    const int n = nt * 32 + col;
    const float* wrow = p.w + static_cast<int64_t>(n) * p.w_ld_out;
    const bool wvec = p.w_ld_in == 1 && reinterpret_cast<uintptr_t>(p.w) % 16 == 0 && p.w_ld_out % 4 == 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int qq = KS == 1 ? i : (i / 8) * 16 + (2 * rh + (i / 4 & 1)) * 4 + (i & 3);  // KS = 2: KT = 4
      if (wvec) {
        wf[i] = *reinterpret_cast<const f32x4*>(wrow + qq * 8 + 4 * h);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) wf[i][e] = wrow[static_cast<int64_t>(qq * 8 + 4 * h + e) * p.w_ld_in];
      }
    }
  }
  float bias_n = p.bias ? p.bias[nt * 32 + col] : 0.0f;

  // the segment of each image's columns (segment bounds are multiples of the image width: checked by the host)
  const float* c_src[NCH];
  const float* c_act[NCH];
  const char* c_idx[NCH];
  int64_t c_ld[NCH], c_act_ld[NCH];
  uint32_t c_rows[NCH];
  int c_sh[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int si = 0;
#pragma unroll
    for (int j = 1; j < COLLIE_DENSE_MAX_SEGS; ++j)
      if (j < p.x.n_seg && c * CW >= p.x.s[j].begin) si = j;
    const Seg& sg = p.x.s[si];
    c_src[c] = sg.src + (c * CW - sg.begin);
    c_act[c] = sg.act ? sg.act + (c * CW - sg.begin) : nullptr;
    c_idx[c] = static_cast<const char*>(sg.idx);
    c_ld[c] = sg.ld;
    c_act_ld[c] = sg.act_ld;
    c_rows[c] = static_cast<uint32_t>(sg.rows);
    c_sh[c] = sg.idx_dtype == COLLIE_I64 ? 3 : 2;
  }

  // everything loaded into registers so far is made ready before the first DMA: a compiler-placed
  // s_waitcnt vmcnt(n) inside the loop would drain the DMA it does not know about
#pragma unroll
  for (int i = 0; i < NW; ++i) asm volatile("" : "+v"(wf[i]));
  asm volatile("" : "+v"(bias_n));

  const uint32_t smem_at = lds_address(smem), idring_at = lds_address(&idring[0][0][0][0]);
  bool any_bad = false;
  auto row_of = [&](int64_t tile, int part) {       // operand row of this lane, clamped into [0, m)
    int64_t m = tile * ROWS + part * 64 + lrow;
    return m < p.m ? m : p.m - 1;
  };
  auto fetch_ids = [&](int64_t tile, int c, int slot) {
    if (c_idx[c]) {
#pragma unroll
      for (int part = 0; part < RP; ++part)
        dma_to_lds<4>(c_idx[c] + (row_of(tile, part) << c_sh[c]),
                      idring_at + (((slot * kWaves + wave) * RP + part) * kWave) * 4);
    }
  };
  auto issue = [&](int64_t tile, int c, int buf, int slot) {
    // id reads of all parts first (see the weight-gradient kernel)
    uint32_t r[RP];
#pragma unroll
    for (int part = 0; part < RP; ++part) {
      r[part] = 0;
      if (c_idx[c]) r[part] = static_cast<uint32_t>(idring[slot][wave][part][lane]);
    }
    const int sw = (lslot ^ (lrow & 7)) * 4;
#pragma unroll
    for (int part = 0; part < RP; ++part) {
      const int64_t m = row_of(tile, part);
      int64_t row = m;
      if (c_idx[c]) {
        const bool bad = r[part] >= c_rows[c];          // flagged at the end, row 0 read instead
        any_bad |= bad;
        row = bad ? 0 : r[part];
      }
      const float* g = c_src[c] + row * c_ld[c] + sw;
      const uint32_t dst = smem_at + (buf * kSlabF4 + part * 512 + wave * 64) * 16;
#pragma unroll
      for (int j = 0; j < KT; ++j) dma_to_lds<16>(g + j * 32, dst + j * (ROWS * 8) * 16);
      if (ACT) {
        const float* a = c_act[c] + m * c_act_ld[c] + sw;
#pragma unroll
        for (int j = 0; j < KT; ++j) dma_to_lds<16>(a + j * 32, dst + (kImgF4 + j * (ROWS * 8)) * 16);
      }
    }
  };
  // slab s = (tile t0 + s / NCH, chunk s % NCH); its ids live in idring slot s & 1
  auto fetch_ids_slab = [&](int s) { fetch_ids(t0 + s / NCH, s % NCH, s & 1); };
  auto issue_slab = [&](int s) { issue(t0 + s / NCH, s % NCH, s % NB, s & 1); };

  fetch_ids_slab(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (1 < nslab) fetch_ids_slab(1);
  issue_slab(0);
  if (NB == 3 && 1 < nslab) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J) : "memory");
    if (2 < nslab) fetch_ids_slab(2);
    issue_slab(1);
  }

  f32x16 acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
  // store of a finished tile (KS = 2: the first wave of a pair, after adding the second's sums)
  // Addend rows of a tile are loaded into registers one tile early -- right after the previous
  // tile's stores and BEFORE that iteration's DMA issue -- and consumed after the next tile's first
  // barrier: the compiler's wait for them then only covers DMA that is a whole iteration old.  (A load
  // consumed where it is issued would drain the DMA ring once per tile: measured +200 us.)
  // ADD: the kernel instances with an addend (every load below is unconditional: a load inside a
  // branch makes the compiler wait for it at the join)
  int64_t add_row0 = ADD ? (t0 * ROWS) % p.add_period : 0;   // period >= 256: host check
  float addv[ADD ? NA : 1][16];
  auto load_addend = [&]() {
    if (ADD) {
      static_assert(!ADD || KS == 1, "addend instances keep one accumulator tile per wave");
      const float* arow = p.addend + nt * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t ar = add_row0 + rh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (ar >= p.add_period) ar -= p.add_period;
        addv[0][r] = arow[ar * p.add_ld];      // rows past m: a valid row, never stored
      }
      add_row0 += ROWS;
      if (add_row0 >= p.add_period) add_row0 -= p.add_period;
    }
  };
  auto store_tile = [&](int64_t tile) {
    const int n = nt * 32 + col;
    if (KS == 2 && rh != 0) return;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (KS == 1 ? rh : a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t m = tile * ROWS + lr;
        float v = acc[a][r];
        if (KS == 2) v += xbuf[((nt * 2 + a) * 16 + r) * kWave + lane];
        v += bias_n;
        if (ADD) v += addv[0][r];
        if (p.leaky) v = v > 0.0f ? v : v * p.slope;
        if (m < p.m) p.y[m * p.y_ld + n] = v;
      }
    }
  };
  auto hand_over = [&]() {                 // KS = 2, second wave of a pair: sums to LDS, start again
    if (KS == 2 && rh != 0) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          xbuf[((nt * 2 + a) * 16 + r) * kWave + lane] = acc[a][r];
          acc[a][r] = 0.0f;
        }
    }
  };

  int s = 0;
  for (int64_t tile = t0; tile < t1; ++tile) {
#pragma unroll
    for (int c = 0; c < NCH; ++c, ++s) {
      // image s (and the ids of the next image to issue) has landed: everything but the DMA of
      // the image after it
      if (NB == 3 && s + 1 < nslab) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bare_barrier();
      if (c == 0 && tile > t0) {
        store_tile(tile - 1);
        if (KS == 1 || rh == 0) {
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
        }
      }
      if (c == 0) load_addend();
      if (s + NB - 1 < nslab) {
        if (s + NB < nslab) fetch_ids_slab(s + NB);
        issue_slab(s + NB - 1);
      }
      const f32x4* img = smem + (s % NB) * kSlabF4;
      // A fragments two steps ahead of the MFMAs that consume them.  Step -> (accumulator / row
      // sub-tile, k-tile, q): KS = 1: rows rh, k-tile step / 4; KS = 2: rows step / 8, k-tile
      // 2 rh + (step / 4 & 1)
      auto frag = [&](int step) {
        const int a_i = KS == 1 ? 0 : step >> 3;
        const int j = KS == 1 ? step >> 2 : 2 * rh + (step >> 2 & 1);
        const int q = step & 3;
        const int arow = (KS == 1 ? rh : a_i) * 32 + col;
        const int at = j * (ROWS * 8) + arow * 8 + ((2 * q + h) ^ (arow & 7));
        f32x4 a = img[at];
        if (ACT) {
          const f32x4 r = img[kImgF4 + at];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = r[e] > 0.0f ? a[e] : a[e] * p.slope;
        }
        return a;
      };
      f32x4 a0 = frag(0), a1 = frag(1);
#pragma unroll
      for (int step = 0; step < NS; ++step) {
        const f32x4 a = a0;
        a0 = a1;
        if (step + 2 < NS) a1 = frag(step + 2);
        const f32x4 w = wf[KS == 1 ? c * NS + step : c * 8 + (step & 7)];
        const int a_i = KS == 1 ? 0 : step >> 3;
        acc[a_i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], w[0], acc[a_i], 0, 0, 0);
        acc[a_i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], w[1], acc[a_i], 0, 0, 0);
        acc[a_i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], w[2], acc[a_i], 0, 0, 0);
        acc[a_i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], w[3], acc[a_i], 0, 0, 0);
      }
      if (c == NCH - 1) hand_over();      // read by the partner after the next barrier
    }
  }
  if (KS == 2) bare_barrier();
  store_tile(t1 - 1);
  if (any_bad && p.status) atomicOr(p.status, 1);
}

// Adds the per-chunk partials in a fixed order: an output element has 16 partial sums -- sum j adds
// the chunks c = j (mod 16) in increasing order, as two interleaved chains (c = j, j + 32, ... and
// c = j + 16, j + 48, ...) -- combined by a fixed-shape butterfly (8, 4, 2, 1): deterministic, and
// 32x shorter dependent chains than one thread per element.  A workgroup owns 64 consecutive
// elements: thread (el, q) forms the sums j = 4 q .. 4 q + 3 of element el, so that every load
// instruction of a wave reads 256 contiguous bytes of one chunk (the first version gave the 16 sums
// of an element to 16 neighbouring lanes: 16-byte pieces of 16 chunks per load, 2 TB/s on partials
// that sit in the Infinity Cache); the butterfly runs over LDS in the order the shuffles had.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           int nchunks, int n_out, int n_in,
                                                           float* __restrict__ dw,
                                                           float* __restrict__ db) {
  __shared__ float sums[16][64];
  const int ldp = n_in + 1;
  const int64_t total = static_cast<int64_t>(n_out) * ldp;
  const int el = threadIdx.x & 63;
  const int q = threadIdx.x >> 6;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + el;
  // the 4 sums of this thread advance together, two rounds of their chains per step: 16 loads are
  // issued before the first add (the chains are short; what the kernel waits for is load latency)
  float s0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (e < total) {
    const float* base = partial + e;
    for (int c0 = 4 * q; c0 < nchunks; c0 += 64) {
      float x[2][2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int c = c0 + r * 32 + hlf * 16 + jj;
            x[r][hlf][jj] = c < nchunks ? base[c * total] : 0.0f;
          }
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          // (a chunk past the end adds nothing: the chains of the first version simply stopped)
          if (c0 + r * 32 + jj < nchunks) s0[jj] += x[r][0][jj];
          if (c0 + r * 32 + 16 + jj < nchunks) s1[jj] += x[r][1][jj];
        }
    }
  }
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) sums[4 * q + jj][el] = s0[jj] + s1[jj];
  __syncthreads();
  if (q != 0 || e >= total) return;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = sums[j][el];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {      // lane 0 of group_sum<16>: v[j] += v[j ^ o], j < o
#pragma unroll
    for (int j = 0; j < o; ++j) v[j] = v[j] + v[j ^ o];
  }
  const float sum = v[0];
  const int n = static_cast<int>(e / ldp);
  const int k = static_cast<int>(e - static_cast<int64_t>(n) * ldp);
  if (k < n_in) {
    if (dw) dw[static_cast<int64_t>(n) * n_in + k] = sum;
  } else if (db) {
    db[n] = sum;
  }
}

// ---- the one-column layer (the head's final Linear(.., 1)) -------------------------------------------
// With a single output (or input) column the three GEMMs are memory-bound vector operations; on the
// MFMA tiles they would pad that column to 32.  Operands here are one identity segment of width
// n <= 256, float4-aligned (checked by the host), without an activation reference.
constexpr int kNarrowBlock = 256;

// y[m] = sum_k x[m, k] w[k] + b (+ bias-table rows): LPR = n / 4 lanes per row
__global__ __launch_bounds__(kNarrowBlock) void narrow_gemv_kernel(const FwdParams p, int lpr) {
  const int rows_per_block = kNarrowBlock / lpr;
  const int sub = threadIdx.x % lpr;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * rows_per_block + threadIdx.x / lpr;
  const Seg& s = p.x.s[0];
  float acc = 0.0f;
  if (m < p.m) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(s.src + m * s.ld + sub * 4);
    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + sub * 4);
    acc = fmaf(x[3], w[3], fmaf(x[2], w[2], fmaf(x[1], w[1], x[0] * w[0])));
  }
  for (int o = lpr / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
  if (m >= p.m || sub != 0) return;
  float v = acc + (p.bias ? p.bias[0] : 0.0f);
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
  p.y[m * p.y_ld] = v;
}

// y[m, n] = g[m] * w[n] (the data gradient of that layer: x is the single column g, W' is [n, 1])
__global__ __launch_bounds__(kNarrowBlock) void narrow_outer_kernel(const FwdParams p, int lpr) {
  const int rows_per_block = kNarrowBlock / lpr;
  const int sub = threadIdx.x % lpr;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * rows_per_block + threadIdx.x / lpr;
  if (m >= p.m) return;
  const Seg& s = p.x.s[0];
  const float g = s.src[m * s.ld];
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = g * p.w[static_cast<int64_t>(sub * 4 + e) * p.w_ld_out];
  *reinterpret_cast<f32x4*>(p.y + m * p.y_ld + sub * 4) = v;
}

// partial[chunk][0][k] = sum over the chunk's rows of g[m] * x[m, k]; [n_in] = sum of g[m].  A thread
// owns one 16-byte column slot and every (256 / lpr)-th row of the chunk; the row groups are combined
// through LDS in fixed order: deterministic.
__global__ __launch_bounds__(kNarrowBlock) void narrow_wgrad_kernel(const WgParams p, int lpr) {
  __shared__ float red[kNarrowBlock * 5];
  const int groups = kNarrowBlock / lpr;
  const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  const int64_t mbeg = static_cast<int64_t>(blockIdx.x) * p.chunk;
  int64_t mend = mbeg + p.chunk;
  if (mend > p.m) mend = p.m;
  const Seg& zs = p.dz.s[0];
  const Seg& xs = p.x.s[0];
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f, ag = 0.0f;
  for (int64_t m = mbeg + grp; m < mend; m += groups) {
    const float g = zs.src[m * zs.ld];
    const f32x4 x = *reinterpret_cast<const f32x4*>(xs.src + m * xs.ld + sub * 4);
    a0 = fmaf(g, x[0], a0);
    a1 = fmaf(g, x[1], a1);
    a2 = fmaf(g, x[2], a2);
    a3 = fmaf(g, x[3], a3);
    ag += g;
  }
  float* mine = red + threadIdx.x * 5;
  mine[0] = a0; mine[1] = a1; mine[2] = a2; mine[3] = a3; mine[4] = ag;
  __syncthreads();
  if (grp != 0) return;
  for (int q = 1; q < groups; ++q) {
    const float* o = red + (q * lpr + sub) * 5;
    a0 += o[0]; a1 += o[1]; a2 += o[2]; a3 += o[3]; ag += o[4];
  }
  float* out = p.partial + static_cast<int64_t>(blockIdx.x) * (p.n_in + 1);
  out[sub * 4 + 0] = a0;
  out[sub * 4 + 1] = a1;
  out[sub * 4 + 2] = a2;
  out[sub * 4 + 3] = a3;
  if (sub == 0) out[p.n_in] = ag;
}

// one identity segment, float4-aligned rows, no activation reference
bool narrow_operand(const Operand& op, int width) {
  if (op.n_seg != 1 || op.width != width) return false;
  const Seg& s = op.s[0];
  if (s.idx || s.act) return false;
  if (width == 1) return true;
  return width % 4 == 0 && s.ld % 4 == 0 && reinterpret_cast<uintptr_t>(s.src) % 16 == 0;
}

// lanes per row for an n-column row of 16-byte slots (a power of two that divides the workgroup)
int narrow_lpr(int n) {
  const int slots = n / 4;
  return (slots >= 1 && slots <= 64 && (slots & (slots - 1)) == 0) ? slots : 0;
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

// the LDS-DMA forward takes: 128 output columns with n_in of 64 / 128 / 256 / 384, 64 output columns with
// n_in = 128, 32 with n_in = 64 (the shapes of the reference's default head), in segments that start on
// image boundaries, no per-row bias tables; an activation reference only with 128 outputs and
// n_in <= 128; an addend only with 128 outputs and n_in = 128 / 256
int fwd_dma_image_cols(const FwdParams& p) {
  if (p.n_out == 128 && (p.n_in == 128 || p.n_in == 256 || p.n_in == 384)) return 128;
  if (p.n_out == 128 && p.n_in == 64) return 64;
  if (p.n_out == 64 && p.n_in == 128 && !p.has_act && !p.addend) return 64;
  if (p.n_out == 32 && p.n_in == 64 && !p.has_act && !p.addend) return 32;
  return 0;
}

bool fwd_dma_ok(const FwdParams& p) {
  if (g_dense_kernels == COLLIE_DENSE_REGISTER_STAGED) return false;
  const int cw = fwd_dma_image_cols(p);
  if (cw == 0 || p.user_bias || p.m < 1) return false;
  if (p.has_act && p.n_in > 128) return false;
  if (p.addend && (p.has_act || (p.n_in != 128 && p.n_in != 256))) return false;
  if (p.addend && p.add_period < 256) return false;  // the kernel wraps the addend row once per tile
  for (int i = 0; i < p.x.n_seg; ++i)
    if (p.x.s[i].begin % cw || p.x.s[i].end % cw) return false;
  return true;
}

void launch_fwd(const FwdParams& p, bool vec, hipStream_t stream) {
  const unsigned gx = static_cast<unsigned>((p.m + kFwdRows - 1) / kFwdRows);
  const unsigned gy = static_cast<unsigned>((p.n_out + 127) / 128);
  if (vec && fwd_dma_ok(p)) {
    const int rows = p.n_out == 128 ? 64 : (p.n_out == 64 ? 128 : 256);
    const int64_t ntiles = (p.m + rows - 1) / rows;
    const dim3 grid(static_cast<unsigned>(ntiles < 256 ? ntiles : 256)), block(kDfBlock);
#define COLLIE_DF(NCH, KS, KT)                                                                         \
  if (p.has_act) hipLaunchKernelGGL((dense_fwd_dma_kernel<NCH, true, KS, KT, false, 4>), grid, block, 0, stream, p); \
  else hipLaunchKernelGGL((dense_fwd_dma_kernel<NCH, false, KS, KT, false, 4>), grid, block, 0, stream, p)
    if (p.n_out == 64) hipLaunchKernelGGL((dense_fwd_dma_kernel<2, false, 1, 2, false, 2>), grid, block, 0, stream, p);
    else if (p.n_out == 32) hipLaunchKernelGGL((dense_fwd_dma_kernel<2, false, 1, 1, false, 1>), grid, block, 0, stream, p);
    else if (p.addend && p.n_in == 128) hipLaunchKernelGGL((dense_fwd_dma_kernel<1, false, 1, 4, true, 4>), grid, block, 0, stream, p);
    else if (p.addend) hipLaunchKernelGGL((dense_fwd_dma_kernel<2, false, 1, 4, true, 4>), grid, block, 0, stream, p);
    else if (p.n_in == 64) { COLLIE_DF(1, 1, 2); }
    else if (p.n_in == 128) { COLLIE_DF(1, 1, 4); }
    else if (p.n_in == 256) hipLaunchKernelGGL((dense_fwd_dma_kernel<2, false, 1, 4, false, 4>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((dense_fwd_dma_kernel<3, false, 2, 4, false, 4>), grid, block, 0, stream, p);
#undef COLLIE_DF
    return;
  }
  if (vec) {
    const unsigned gx = static_cast<unsigned>((p.m + kFRows - 1) / kFRows);
    // weights load as 16-byte slots when their contiguous direction is float4-aligned
    const bool rowmajor = p.w_ld_in == 1;
    const int64_t ldw = rowmajor ? p.w_ld_out : p.w_ld_in;
    const bool w_vec = reinterpret_cast<uintptr_t>(p.w) % 16 == 0 && ldw % 4 == 0 &&
                       (rowmajor || (p.w_ld_out == 1 && p.n_out % 4 == 0));
#define COLLIE_LAUNCH_FWD(NT_, GY_)                                                                  \
  do {                                                                                               \
    if (w_vec && p.has_act)                                                                          \
      hipLaunchKernelGGL((dense_fwd_fast_kernel<NT_, true, true>), dim3(gx, GY_), dim3(kFBlock), 0, stream, p); \
    else if (w_vec)                                                                                  \
      hipLaunchKernelGGL((dense_fwd_fast_kernel<NT_, true, false>), dim3(gx, GY_), dim3(kFBlock), 0, stream, p); \
    else if (p.has_act)                                                                              \
      hipLaunchKernelGGL((dense_fwd_fast_kernel<NT_, false, true>), dim3(gx, GY_), dim3(kFBlock), 0, stream, p); \
    else                                                                                             \
      hipLaunchKernelGGL((dense_fwd_fast_kernel<NT_, false, false>), dim3(gx, GY_), dim3(kFBlock), 0, stream, p); \
  } while (0)
    if (p.n_out <= 32) COLLIE_LAUNCH_FWD(1, 1);
    else if (p.n_out <= 64) COLLIE_LAUNCH_FWD(2, 1);
    else COLLIE_LAUNCH_FWD(4, gy);
#undef COLLIE_LAUNCH_FWD
    return;
  }
  if (p.n_out <= 32)
    hipLaunchKernelGGL(dense_fwd_kernel<1>, dim3(gx, 1), dim3(kFwdBlock), 0, stream, p);
  else if (p.n_out <= 64)
    hipLaunchKernelGGL(dense_fwd_kernel<2>, dim3(gx, 1), dim3(kFwdBlock), 0, stream, p);
  else
    hipLaunchKernelGGL(dense_fwd_kernel<4>, dim3(gx, gy), dim3(kFwdBlock), 0, stream, p);
}

// the LDS-DMA weight gradient takes: dW of 128 x {128, 256, 384}, 64 x 128 or 32 x 64 (the layers of
// the reference's default head), dZ one dense segment, X in segments that start on 32-column tiles
// and carry no activation reference, whole slabs of rows
int wgrad_dma_rows(int n_out, int n_in) {
  if (n_out == 128 && (n_in == 128 || n_in == 256 || n_in == 384)) return 16;
  if ((n_out == 64 && n_in == 128) || (n_out == 32 && n_in == 64)) return 32;
  return 0;
}

bool wgrad_dma_ok(const WgParams& p) {
  if (g_dense_kernels == COLLIE_DENSE_REGISTER_STAGED) return false;
  const int rows = wgrad_dma_rows(p.n_out, p.n_in);
  if (p.dz_scalar || rows == 0) return false;
  if (p.m % rows) return false;   // chunks are multiples of 32 rows
  if (p.dz.n_seg != 1 || p.dz.s[0].idx) return false;
  for (int i = 0; i < p.x.n_seg; ++i)
    if (p.x.s[i].act || p.x.s[i].begin % 32 || p.x.s[i].end % 32) return false;
  return true;
}

// y[j, :] = act((outer[j / period, :] + bias) + inner[j mod period, :]): the first combined layer on
// the users x catalogue grid once both of its shares are known (collie_dense_outer_sum).  HBM-bound
// (one y row written per pair; outer / inner rows come from the caches).  V floats per thread.
constexpr int kOsColsX = 32, kOsRowsY = 8;
template <int V>
__global__ __launch_bounds__(kOsColsX* kOsRowsY) void dense_outer_sum_kernel(
    const float* __restrict__ outer, int64_t outer_ld, const float* __restrict__ inner,
    int64_t inner_ld, uint32_t period, const float* __restrict__ bias, int32_t n, int64_t m,
    int leaky, float slope, float* __restrict__ y, int64_t y_ld) {
  const int nv = n / V;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kOsRowsY + threadIdx.y; j < m;
       j += static_cast<int64_t>(gridDim.x) * kOsRowsY) {
    const int64_t o = j / period, i = j - o * period;
    const float* po = outer + o * outer_ld;
    const float* pi = inner + i * inner_ld;
    float* py = y + j * y_ld;
    for (int c = threadIdx.x; c < nv; c += kOsColsX) {
      float a[V], b[V], q[V];
      if (V == 4) {
        *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(po + 4 * c);
        *reinterpret_cast<float4*>(q) = *reinterpret_cast<const float4*>(pi + 4 * c);
        if (bias) *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(bias + 4 * c);
      } else {
        a[0] = po[c];
        q[0] = pi[c];
        if (bias) b[0] = bias[c];
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float v = a[e];
        if (bias) v += b[e];
        v += q[e];
        if (leaky) v = v > 0.0f ? v : v * slope;
        a[e] = v;
      }
      if (V == 4) *reinterpret_cast<float4*>(py + 4 * c) = *reinterpret_cast<float4*>(a);
      else py[c] = a[0];
    }
  }
}

// The tail of the reference's default head (combined layers 128 -> 64 -> 32 -> 1) on the users x
// catalogue grid, one launch, no activation ever written to HBM: a wave takes 32 pairs of one user
// (32 consecutive items), forms their first-layer rows from the two shares in registers, and runs
// both remaining Linear layers TRANSPOSED -- Y^T = W X^T, so the pair index stays on the lanes
// (MFMA B operand in, C columns out) from layer to layer and no LDS transpose is needed.  A lane
// half h owns the k values the previous layer's C layout hands it (layer 2: k = 64 h + s; layer 3:
// k = 32 t + (r & 3) + 8 (r >> 2) + 4 h); W2 / W3 are re-laid in LDS so that a lane's four
// consecutive k are one conflict-free 16-byte read.  MFMA-bound: 160 MFMAs per 32 pairs.
constexpr int kGtBlock = 512;
struct GridTailParams {
  const float* user_share;
  int64_t us_ld, n_users;
  const float* item_share;
  int64_t is_ld, period;
  const float *b1, *w2, *b2, *w3, *b3, *w4, *b4;
  float slope;
  const void* users;
  int32_t users_dtype;
  const float* user_bias;
  int64_t num_users, item0;
  const float* item_bias;
  int64_t num_items;
  float* y;
  int64_t y_ld;
  int32_t* status;
};

__global__ __launch_bounds__(kGtBlock) void dense_grid_tail_kernel(GridTailParams p) {
  __shared__ __align__(16) float W2s[64 * 128];
  __shared__ __align__(16) float W3s[32 * 64];
  __shared__ float consts[64 + 32 + 32];  // b2, b3, w4
  __shared__ __align__(16) float ubs[kGtBlock / 64][128];  // per wave: user_share[u] + b1 of its user
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
  for (int idx = tid; idx < 2048; idx += kGtBlock) {  // [t][s4][h][n] <- W2[32 t + n][64 h + 4 s4 ..]
    const int nn = idx & 31, hh = (idx >> 5) & 1, s4 = (idx >> 6) & 15, t = idx >> 10;
    *reinterpret_cast<f32x4*>(&W2s[idx * 4]) =
        *reinterpret_cast<const f32x4*>(&p.w2[(32 * t + nn) * 128 + 64 * hh + 4 * s4]);
  }
  for (int idx = tid; idx < 512; idx += kGtBlock) {  // [t][q][h][n] <- W3[n][32 t + 8 q + 4 h ..]
    const int nn = idx & 31, hh = (idx >> 5) & 1, q = (idx >> 6) & 3, t = idx >> 8;
    *reinterpret_cast<f32x4*>(&W3s[idx * 4]) =
        *reinterpret_cast<const f32x4*>(&p.w3[nn * 64 + 32 * t + 8 * q + 4 * hh]);
  }
  if (tid < 64) consts[tid] = p.b2 ? p.b2[tid] : 0.0f;
  else if (tid < 96) consts[tid] = p.b3 ? p.b3[tid - 64] : 0.0f;
  else if (tid < 128) consts[tid] = p.w4[tid - 96];
  __syncthreads();
  const float b4 = p.b4 ? p.b4[0] : 0.0f;
  const int64_t tiles_per_user = (p.period + 31) / 32;
  const int64_t ntiles = p.n_users * tiles_per_user;
  int64_t cur_u = -1;
  // Tiles are numbered user-fastest (tile = item_tile * n_users + user): the waves that run at the
  // same time work on the same few item tiles, so an item-share row comes from HBM once and from L2
  // for the other users.  The stride is a multiple of n_users when there are at least that many
  // waves, so that a wave keeps its user (and its LDS copy of the user's row) for the whole launch.
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kGtBlock / 64);
  const int64_t stride = p.n_users <= nwaves ? nwaves / p.n_users * p.n_users : nwaves;
  const int64_t first = static_cast<int64_t>(blockIdx.x) * (kGtBlock / 64) + wave;
  // this lane's 64 floats of the item share of a tile (rows past the range: the last valid row)
  auto share_row = [&](int64_t tile) {
    const int64_t i = (tile / p.n_users) * 32 + n;
    return p.item_share + (i < p.period ? i : p.period - 1) * p.is_ld + 64 * h;
  };
  // two register images of item-share rows: while a tile is computed from one, the NEXT tile's rows
  // are fetched into the other (the loop is unrolled by two so that the images swap roles without
  // register copies -- VALU and MFMA issue do not overlap on this chip, every VALU instruction counts)
  f32x4 qa[16], qb[16];
  {
    const float* q = share_row(first < ntiles ? first : 0);
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) qa[s4] = *reinterpret_cast<const f32x4*>(q + 4 * s4);
  }
  auto leaky = [&](float v) { return fmaxf(v, v * p.slope); };  // slope <= 1 (host check)
  auto do_tile = [&](int64_t tile, f32x4(&qreg)[16], f32x4(&qnext)[16]) {
    const int64_t it = tile / p.n_users;
    const int64_t u = tile - it * p.n_users;
    const int64_t i = it * 32 + n;
    const bool live = i < p.period;
    if (u != cur_u) {  // wave-uniform; LDS operations of one wave execute in order
      const float* pu = p.user_share + u * p.us_ld;
      ubs[wave][2 * lane] = pu[2 * lane] + (p.b1 ? p.b1[2 * lane] : 0.0f);
      ubs[wave][2 * lane + 1] = pu[2 * lane + 1] + (p.b1 ? p.b1[2 * lane + 1] : 0.0f);
      cur_u = u;
    }
    // ---- layer 2: Y2^T [64 x 32 pairs] = W2 X1^T, two 32-row tiles; the first-layer row is formed
    // four k at a time, right where the MFMAs consume it (no 64-register copy of it) -------------
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
      acc0[r] = consts[k];
      acc1[r] = consts[32 + k];
    }
    // the LDS operands one step ahead; the scheduler is fenced per step (left alone it hoists every
    // LDS read of the tile and spills)
    {
      const float* q = share_row(tile + stride < ntiles ? tile + stride : tile);
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) qnext[s4] = *reinterpret_cast<const f32x4*>(q + 4 * s4);
    }
    f32x4 un = *reinterpret_cast<const f32x4*>(&ubs[wave][64 * h]);
    f32x4 a0n = *reinterpret_cast<const f32x4*>(&W2s[((0 * 2 + h) * 32 + n) * 4]);
    f32x4 a1n = *reinterpret_cast<const f32x4*>(&W2s[(((16 + 0) * 2 + h) * 32 + n) * 4]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      const f32x4 uv = un, a0 = a0n, a1 = a1n;
      if (s4 + 1 < 16) {
        un = *reinterpret_cast<const f32x4*>(&ubs[wave][64 * h + 4 * (s4 + 1)]);
        a0n = *reinterpret_cast<const f32x4*>(&W2s[(((s4 + 1) * 2 + h) * 32 + n) * 4]);
        a1n = *reinterpret_cast<const f32x4*>(&W2s[(((16 + s4 + 1) * 2 + h) * 32 + n) * 4]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = leaky(uv[j] + qreg[s4][j]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], x, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = leaky(acc0[r]);
      acc1[r] = leaky(acc1[r]);
    }
    // ---- layer 3: Y3^T [32 x 32 pairs] = W3 Y2^T; two partial accumulators = two MFMA chains ------
    f32x16 acc3a, acc3b;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc3a[r] = consts[64 + (r & 3) + 8 * (r >> 2) + 4 * h];
      acc3b[r] = 0.0f;
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(&W3s[(((0 * 4 + qd) * 2 + h) * 32 + n) * 4]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(&W3s[(((1 * 4 + qd) * 2 + h) * 32 + n) * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc3a = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], acc0[4 * qd + j], acc3a, 0, 0, 0);
        acc3b = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], acc1[4 * qd + j], acc3b, 0, 0, 0);
      }
    }
    // ---- last Linear (32 -> 1) + the two bias tables -----------------------------------------------
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = leaky(acc3a[r] + acc3b[r]);
      part = fmaf(v, consts[96 + (r & 3) + 8 * (r >> 2) + 4 * h], part);
    }
    const float other = __shfl_xor(part, 32);
    float score = (h == 0 ? part + other : other + part) + b4;
    if (p.user_bias) {
      const int64_t uid = load_idx(p.users, p.users_dtype, u);
      const int64_t it = p.item0 + i;
      if (uid < 0 || uid >= p.num_users || (live && it >= p.num_items)) {
        if (p.status) atomicOr(p.status, 1);
      } else if (live) {
        score = (score + p.user_bias[uid]) + p.item_bias[it];
      }
    }
    if (live && h == 0) p.y[u * p.y_ld + i] = score;
  };
  for (int64_t tile = first < stride ? first : ntiles; tile < ntiles; tile += 2 * stride) {
    do_tile(tile, qa, qb);
    if (tile + stride < ntiles) do_tile(tile + stride, qb, qa);
  }
}

int64_t wgrad_chunks(int64_t m, int64_t* chunk_out, int64_t max_chunks = 512) {
  int64_t nchunks = (m + 255) / 256;
  // register-staged kernel: two resident workgroups per CU; LDS-DMA kernel: one (max_chunks 256) --
  // one round of chunks either way, and half the partial sums to write and add for the latter
  if (nchunks > max_chunks) nchunks = max_chunks;
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

extern "C" int collie_dense_set_kernels(int32_t family) {
  COLLIE_REQUIRE(family == COLLIE_DENSE_AUTO || family == COLLIE_DENSE_REGISTER_STAGED,
                 "dense_set_kernels: unknown kernel family %d", family);
  g_dense_kernels = family;
  return 0;
}

extern "C" int collie_dense_forward(const collie_dense_operand* x, int64_t m, const float* w,
                                    int64_t w_ld_out, int64_t w_ld_in, const float* bias,
                                    int32_t n_in, int32_t n_out, int32_t leaky, float slope,
                                    const void* users, int32_t users_dtype, const float* user_bias,
                                    int64_t num_users, const void* items, int32_t items_dtype,
                                    const float* item_bias, int64_t num_items, float* y,
                                    int64_t y_ld, int32_t* status, collie_stream_t stream_) {
  return collie_dense_forward_add(x, m, w, w_ld_out, w_ld_in, bias, n_in, n_out, leaky, slope, users,
                                  users_dtype, user_bias, num_users, items, items_dtype, item_bias,
                                  num_items, nullptr, 0, 0, y, y_ld, status, stream_);
}

extern "C" int collie_dense_forward_add(const collie_dense_operand* x, int64_t m, const float* w,
                                        int64_t w_ld_out, int64_t w_ld_in, const float* bias,
                                        int32_t n_in, int32_t n_out, int32_t leaky, float slope,
                                        const void* users, int32_t users_dtype,
                                        const float* user_bias, int64_t num_users, const void* items,
                                        int32_t items_dtype, const float* item_bias,
                                        int64_t num_items, const float* addend, int64_t add_ld,
                                        int64_t add_period, float* y, int64_t y_ld, int32_t* status,
                                        collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(!addend || (add_ld >= n_out && add_period >= 1),
                 "dense_forward_add: addend needs add_ld >= n_out and add_period >= 1");
  COLLIE_REQUIRE(m >= 1 && m < (int64_t{1} << 39), "dense_forward: bad row count %lld",
                 static_cast<long long>(m));
  COLLIE_REQUIRE(w && y && n_in >= 1 && n_out >= 1 && y_ld >= n_out, "dense_forward: bad W / Y");
  COLLIE_REQUIRE((m + kFRows - 1) / kFRows < (int64_t{1} << 31), "dense_forward: too many rows");
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
  p.addend = addend;
  p.add_ld = add_ld;
  p.add_period = add_period;
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
  for (int i = 0; i < p.x.n_seg; ++i)
    if (p.x.s[i].act) p.has_act = 1;
  const int lpr_in = narrow_lpr(n_in), lpr_out = narrow_lpr(n_out);
  if (!addend && n_out == 1 && lpr_in && narrow_operand(p.x, n_in) && w_ld_in == 1 &&
      reinterpret_cast<uintptr_t>(w) % 16 == 0) {
    const int64_t rows_per_block = kNarrowBlock / lpr_in;
    hipLaunchKernelGGL(narrow_gemv_kernel, dim3(static_cast<unsigned>((m + rows_per_block - 1) / rows_per_block)),
                       dim3(kNarrowBlock), 0, stream, p, lpr_in);
    return check_launch("narrow_gemv_kernel");
  }
  if (!addend && n_in == 1 && lpr_out && narrow_operand(p.x, 1) && !bias && !leaky && !p.user_bias &&
      y_ld % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0) {
    const int64_t rows_per_block = kNarrowBlock / lpr_out;
    hipLaunchKernelGGL(narrow_outer_kernel, dim3(static_cast<unsigned>((m + rows_per_block - 1) / rows_per_block)),
                       dim3(kNarrowBlock), 0, stream, p, lpr_out);
    return check_launch("narrow_outer_kernel");
  }
  launch_fwd(p, vec, stream);
  return check_launch("dense_fwd_kernel");
}

extern "C" int collie_dense_outer_sum(const float* outer, int64_t outer_ld, int64_t n_outer,
                                      const float* inner, int64_t inner_ld, int64_t period,
                                      const float* bias, int32_t n, int32_t leaky, float slope,
                                      float* y, int64_t y_ld, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(outer && inner && y && n >= 1 && n_outer >= 1 && period >= 1 &&
                     period < (int64_t{1} << 32),
                 "dense_outer_sum: bad arguments");
  COLLIE_REQUIRE(outer_ld >= n && inner_ld >= n && y_ld >= n, "dense_outer_sum: row pitch below n");
  COLLIE_REQUIRE(n_outer <= (int64_t{1} << 39) / period, "dense_outer_sum: too many rows");
  const int64_t m = n_outer * period;
  const int64_t blocks = (m + kOsRowsY - 1) / kOsRowsY;
  const dim3 grid(static_cast<unsigned>(blocks < 256 * 64 ? blocks : 256 * 64)), block(kOsColsX, kOsRowsY);
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  const bool vec = n % 4 == 0 && outer_ld % 4 == 0 && inner_ld % 4 == 0 && y_ld % 4 == 0 &&
                   al16(outer) && al16(inner) && al16(y) && (!bias || al16(bias));
  if (vec)
    hipLaunchKernelGGL(dense_outer_sum_kernel<4>, grid, block, 0, stream, outer, outer_ld, inner,
                       inner_ld, static_cast<uint32_t>(period), bias, n, m, leaky, slope, y, y_ld);
  else
    hipLaunchKernelGGL(dense_outer_sum_kernel<1>, grid, block, 0, stream, outer, outer_ld, inner,
                       inner_ld, static_cast<uint32_t>(period), bias, n, m, leaky, slope, y, y_ld);
  return check_launch("dense_outer_sum_kernel");
}

extern "C" int collie_dense_grid_tail(const float* user_share, int64_t us_ld, int64_t n_users,
                                      const float* item_share, int64_t is_ld, int64_t period,
                                      const float* b1, const float* w2, const float* b2,
                                      const float* w3, const float* b3, const float* w4,
                                      const float* b4, float slope, const void* users,
                                      int32_t users_dtype, const float* user_bias, int64_t num_users,
                                      int64_t item0, const float* item_bias, int64_t num_items,
                                      float* y, int64_t y_ld, int32_t* status,
                                      collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(user_share && item_share && w2 && w3 && w4 && y && n_users >= 1 && period >= 1 &&
                     us_ld >= 128 && is_ld >= 128 && y_ld >= period,
                 "dense_grid_tail: bad arguments");
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  COLLIE_REQUIRE(al16(user_share) && al16(item_share) && al16(w2) && al16(w3) && (!b1 || al16(b1)) &&
                     us_ld % 4 == 0 && is_ld % 4 == 0,
                 "dense_grid_tail: shares, W2, W3 and b1 must be 16-byte aligned with row pitches of "
                 "whole 16-byte units");
  const bool row_bias = user_bias != nullptr || item_bias != nullptr;
  if (row_bias) {
    COLLIE_REQUIRE(user_bias && item_bias && users && num_users >= 1 && item0 >= 0 &&
                       (users_dtype == COLLIE_I32 || users_dtype == COLLIE_I64),
                   "dense_grid_tail: the bias-table add needs user ids and both bias tables");
    COLLIE_REQUIRE(item0 + period <= num_items, "dense_grid_tail: items %lld..%lld outside the item "
                   "bias table (%lld rows)", static_cast<long long>(item0),
                   static_cast<long long>(item0 + period), static_cast<long long>(num_items));
  }
  const int64_t tiles_per_user = (period + 31) / 32;
  COLLIE_REQUIRE(n_users <= (int64_t{1} << 40) / tiles_per_user, "dense_grid_tail: too many pairs");
  // the kernel evaluates leaky_relu as max(v, slope * v)
  COLLIE_REQUIRE(slope <= 1.0f, "dense_grid_tail: negative slope %g above 1", static_cast<double>(slope));
  GridTailParams p;
  p.user_share = user_share;
  p.us_ld = us_ld;
  p.n_users = n_users;
  p.item_share = item_share;
  p.is_ld = is_ld;
  p.period = period;
  p.b1 = b1;
  p.w2 = w2;
  p.b2 = b2;
  p.w3 = w3;
  p.b3 = b3;
  p.w4 = w4;
  p.b4 = b4;
  p.slope = slope;
  p.users = users;
  p.users_dtype = users_dtype;
  p.user_bias = row_bias ? user_bias : nullptr;
  p.num_users = num_users;
  p.item0 = item0;
  p.item_bias = item_bias;
  p.num_items = num_items;
  p.y = y;
  p.y_ld = y_ld;
  p.status = status;
  const int64_t blocks = (n_users * tiles_per_user + kGtBlock / 64 - 1) / (kGtBlock / 64);
  hipLaunchKernelGGL(dense_grid_tail_kernel, dim3(static_cast<unsigned>(blocks < 256 ? blocks : 256)),
                     dim3(kGtBlock), 0, stream, p);
  return check_launch("dense_grid_tail_kernel");
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
  bool vec = true, zvec = true;
  if (build_operand(p.dz, dz, m, zvec, "dense_weight_grad(dz)")) return -1;
  if (build_operand(p.x, x, m, vec, "dense_weight_grad(x)")) return -1;
  if (!zvec) {
    if (n_out == 1 && p.dz.n_seg == 1) p.dz_scalar = 1;  // one column: scalar loads in the fast kernel
    else vec = false;
  }
  COLLIE_REQUIRE(p.dz.width == n_out && p.x.width == n_in,
                 "dense_weight_grad: operands are [m, %d] and [m, %d], expected [m, %d] and [m, %d]",
                 p.dz.width, p.x.width, n_out, n_in);
  if (n_in % 4) vec = false;  // the ones column must start a 16-byte slot
  int64_t chunk;
  int64_t nchunks = wgrad_chunks(m, &chunk);
  p.m = m;
  p.chunk = chunk;
  p.n_out = n_out;
  p.n_in = n_in;
  p.slope = slope;
  p.partial = static_cast<float*>(workspace);
  p.status = status;
  const bool dma = vec && wgrad_dma_ok(p);
  if (dma) {
    nchunks = wgrad_chunks(m, &chunk, 256);
    p.chunk = chunk;
  }
  const unsigned gz = static_cast<unsigned>((n_out + kWgTile - 1) / kWgTile);
  const int lpr_x = narrow_lpr(n_in);
  if (n_out == 1 && lpr_x && narrow_operand(p.dz, 1) && narrow_operand(p.x, n_in)) {
    hipLaunchKernelGGL(narrow_wgrad_kernel, dim3(static_cast<unsigned>(nchunks)), dim3(kNarrowBlock), 0,
                       stream, p, lpr_x);
  } else if (dma) {
    const dim3 grid(static_cast<unsigned>(nchunks)), block(kWdBlock);
    const bool act = p.dz.s[0].act != nullptr;
#define COLLIE_WD(NT, KT, R)                                                                       \
  if (act) hipLaunchKernelGGL((dense_wgrad_dma_kernel<NT, KT, R, true>), grid, block, 0, stream, p); \
  else hipLaunchKernelGGL((dense_wgrad_dma_kernel<NT, KT, R, false>), grid, block, 0, stream, p)
    if (n_out == 128 && n_in == 384) { COLLIE_WD(4, 12, 16); }
    else if (n_out == 128 && n_in == 256) { COLLIE_WD(4, 8, 16); }
    else if (n_out == 128) { COLLIE_WD(4, 4, 16); }
    else if (n_out == 64) { COLLIE_WD(2, 4, 32); }
    else { COLLIE_WD(1, 2, 32); }
#undef COLLIE_WD
  } else if (vec) {
    const dim3 grid(static_cast<unsigned>((n_in + kWgTile - 1) / kWgTile), static_cast<unsigned>(nchunks), gz);
    if (p.dz_scalar) hipLaunchKernelGGL(dense_wgrad_fast_kernel<true>, grid, dim3(kWgBlock), 0, stream, p);
    else hipLaunchKernelGGL(dense_wgrad_fast_kernel<false>, grid, dim3(kWgBlock), 0, stream, p);
  } else {
    const dim3 grid(static_cast<unsigned>(nchunks),
                    static_cast<unsigned>((n_in + 1 + kWgTile - 1) / kWgTile), gz);
    hipLaunchKernelGGL(dense_wgrad_kernel, grid, dim3(kWgBlock), 0, stream, p);
  }
  if (check_launch("dense_wgrad_kernel")) return -2;
  const int64_t total = static_cast<int64_t>(n_out) * (n_in + 1);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(static_cast<unsigned>((total + 63) / 64)), dim3(256),
                     0, stream, p.partial, static_cast<int>(nchunks), n_out, n_in, dw, db);
  return check_launch("wgrad_reduce_kernel");
}
