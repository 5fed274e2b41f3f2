// K6: full-catalog evaluation — get_preds + topk + mapk / mrr / auc of collie/metrics.py:77-282 —
// without ever materialising the [users, items] score matrix.
//
// Scoring is an exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32): items on the A rows (staged through
// a padded LDS tile shared by the four waves of a workgroup), users on the B columns (one user per
// lane column, its embedding held in registers for the whole kernel).  A lane therefore owns one
// user and receives 16 item scores per 32x32 tile, which makes the per-user epilogues private to
// a lane:
//   * top-k: a sorted (score, id) list in registers; candidates arrive in increasing id order, so
//     a strict '>' test keeps "descending score, ascending id" exact;
//   * rank counts for MRR / AUC: integer counters against the thresholds of the user's positives.
// The MFMA result is bit-for-bit an fmaf chain in k order (MI355X guide, "FP32-input MFMA"), and
// chain_score() below replays that exact chain for single (user, item) pairs, so the scores of
// the positives (thresholds) are bit-identical to the scores the GEMM produces for the same pairs.
#include <mutex>

#include "common.h"

namespace collie {

int validate_tables(const collie_mf_tables* t);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kUsersPerBlock = 32 * kWaves;  // 128
constexpr int kTileItems = 64;               // items staged in LDS per step
constexpr int kPad = 4;                      // floats of row padding: conflict-free ds_read_b128
constexpr int kCap = 16;                     // top-k capacity held in registers

// k index consumed by MFMA step t on lane half h (see header comment of eval_gemm_kernel)
__host__ __device__ __forceinline__ int kappa(int t, int h) { return 8 * (t >> 2) + 4 * h + (t & 3); }

// The exact chain the MFMA GEMM computes for one (user row, item row) pair.
__device__ __forceinline__ float chain_dot(const float* urow, const float* irow, int D, int D8) {
  float acc = 0.0f;
  for (int t = 0; t < D8 * 4; ++t) {
    const int k0 = kappa(t, 0), k1 = kappa(t, 1);
    const float a0 = k0 < D ? irow[k0] : 0.0f, b0 = k0 < D ? urow[k0] : 0.0f;
    const float a1 = k1 < D ? irow[k1] : 0.0f, b1 = k1 < D ? urow[k1] : 0.0f;
    acc = fmaf(a0, b0, acc);
    acc = fmaf(a1, b1, acc);
  }
  return acc;
}

__device__ __forceinline__ float finish_score(float dot, float ub, float ib) { return (dot + ub) + ib; }

__device__ __forceinline__ float yr(float s, int has_y, float lo, float hi) {
  return has_y ? sigmoidf_(s) * (hi - lo) + lo : s;
}

// monotone map float <-> ordered uint so that thresholds can be binary-searched
__device__ __forceinline__ uint32_t f2o(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

struct EvalParams {
  collie_mf_tables t;
  const int64_t* user_ids;
  int64_t n_users;
  int64_t item_begin, item_end;
  int64_t id_stride, id_offset;  // global item id = local row * id_stride + id_offset
  int tiles_per_chunk;
  int nchunks;
  int k;
  int D8;
  // top-k outputs (candidates)
  float* cand_scores;  // [n_users, nslots, kCap]
  int32_t* cand_ids;
  // rank-count inputs / outputs
  const int64_t* indptr;   // [n_users + 1] CSR over the batch's users
  const float* thr_lo;     // [nnz] lowest raw score whose sigmoid equals the positive's
  const float* thr_hi;     // [nnz] highest such score
  const float* best_score; // [n_users] raw score of the user's best positive
  const int64_t* best_id;  // [n_users]
  unsigned long long* counts;  // [n_users, 3]: above_best, sum_e #(sig<), sum_e #(sig<=)
  // top-k rounds beyond the first (k > kCap): only items strictly AFTER (cut_s, cut_i) in
  // (score desc, id asc) order are candidates; NULL in the first round
  const float* cut_s;    // [n_users] raw score of the last entry already emitted
  const int32_t* cut_i;  // [n_users] its global id
};

// MODE 0: top-k candidates.  MODE 1: rank counts.
template <int D8, int MODE>
__global__ __launch_bounds__(kBlock) void eval_gemm_kernel(const EvalParams p) {
  constexpr int DP = D8 * 8;           // padded depth
  constexpr int LD = DP + kPad;        // LDS row stride (floats)
  __shared__ __attribute__((aligned(16))) float tile[kTileItems * LD];
  __shared__ float tile_bias[kTileItems];

  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const int col = lane & 31;
  const int h = lane >> 5;
  const int D = p.t.dim;
  const bool vec_ok = (D % 4 == 0);

  // ---- this lane's user: embedding slices k = 8q + 4h .. +3 stay in registers ----------------
  const int64_t uslot = static_cast<int64_t>(blockIdx.x) * kUsersPerBlock + wave * 32 + col;
  const bool user_ok = uslot < p.n_users;
  int64_t uid = p.user_ids[user_ok ? uslot : p.n_users - 1];
  if (uid < 0) uid = 0;
  if (uid >= p.t.num_users) uid = p.t.num_users - 1;
  const float* urow = p.t.user_emb + uid * D;
  const float ub = p.t.user_bias[uid];
  f32x4 ufrag[D8];
#pragma unroll
  for (int q = 0; q < D8; ++q) {
    const int k0 = 8 * q + 4 * h;
    if (vec_ok && k0 + 3 < D) {
      ufrag[q] = *reinterpret_cast<const f32x4*>(urow + k0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) ufrag[q][e] = (k0 + e < D) ? urow[k0 + e] : 0.0f;
    }
  }

  // ---- per-lane epilogue state ------------------------------------------------------------------
  float top_s[kCap];
  int top_i[kCap];
  unsigned long long c_above = 0, c_lt = 0, c_le = 0;
  int64_t e_begin = 0, e_end = 0;
  float bscore = 0.0f;
  int bid = 0;
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < kCap; ++i) {
      top_s[i] = -INFINITY;
      top_i[i] = 0x7fffffff;
    }
  } else if (user_ok) {
    e_begin = p.indptr[uslot];
    e_end = p.indptr[uslot + 1];
    bscore = p.best_score[uslot];
    bid = static_cast<int>(p.best_id[uslot]);
  }
  const int k = p.k;
  const bool has_cut = MODE == 0 && p.cut_s != nullptr;  // uniform
  const float cut_s = has_cut && user_ok ? p.cut_s[uslot] : INFINITY;
  const int cut_i = has_cut && user_ok ? p.cut_i[uslot] : -1;

  const int64_t chunk_items = static_cast<int64_t>(p.tiles_per_chunk) * kTileItems;
  const int64_t chunk_begin = p.item_begin + static_cast<int64_t>(blockIdx.y) * chunk_items;
  int64_t chunk_end = chunk_begin + chunk_items;
  if (chunk_end > p.item_end) chunk_end = p.item_end;

  for (int64_t base = chunk_begin; base < chunk_end; base += kTileItems) {
    __syncthreads();
    // ---- stage 64 item rows (zero-padded to DP, rows past the shard end are zero) ------------
    for (int idx = threadIdx.x; idx < kTileItems * (DP / 4); idx += kBlock) {
      const int r = idx / (DP / 4);
      const int c4 = idx - r * (DP / 4);
      const int64_t item = base + r;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (item < chunk_end) {
        const float* irow = p.t.item_emb + item * D;
        if (vec_ok && c4 * 4 + 3 < D) {
          v = *reinterpret_cast<const f32x4*>(irow + c4 * 4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (c4 * 4 + e < D) ? irow[c4 * 4 + e] : 0.0f;
        }
      }
      *reinterpret_cast<f32x4*>(&tile[r * LD + c4 * 4]) = v;
    }
    if (threadIdx.x < kTileItems) {
      const int64_t item = base + threadIdx.x;
      tile_bias[threadIdx.x] = item < chunk_end ? p.t.item_bias[item] : 0.0f;
    }
    __syncthreads();

#pragma unroll
    for (int sub = 0; sub < kTileItems / 32; ++sub) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      const float* arow = &tile[(sub * 32 + col) * LD + 4 * h];
#pragma unroll
      for (int q = 0; q < D8; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], ufrag[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], ufrag[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], ufrag[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], ufrag[q][3], acc, 0, 0, 0);
      }
      // ---- epilogue: this lane's user x 16 items (rows (r&3) + 8(r>>2) + 4h of the tile) ------
      if (MODE == 0) {
        // scores first (rows past the shard end become -inf), one test for the whole sub-tile:
        // after warm-up almost no sub-tile holds a score that enters the list
        float smax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float s = finish_score(acc[r], ub, tile_bias[row]);
          bool ok = base + row < chunk_end;
          if (has_cut) {
            const int gid = static_cast<int>((base + row) * p.id_stride + p.id_offset);
            ok = ok && (s < cut_s || (s == cut_s && gid > cut_i));
          }
          acc[r] = ok ? s : -INFINITY;
          smax = fmaxf(smax, acc[r]);
        }
        if (smax > top_s[kCap - 1]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (acc[r] > top_s[kCap - 1]) {
              // insertion into the sorted list (items arrive in ascending id order inside a lane,
              // so the strict '>' keeps ties in id order)
              const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              float cs = acc[r];
              int ci = static_cast<int>((base + row) * p.id_stride + p.id_offset);
#pragma unroll
              for (int i = 0; i < kCap; ++i) {
                const bool better = cs > top_s[i];
                const float ts = top_s[i];
                const int ti = top_i[i];
                top_s[i] = better ? cs : ts;
                top_i[i] = better ? ci : ti;
                cs = better ? ts : cs;
                ci = better ? ti : ci;
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int64_t item = base + row;
          const float s = finish_score(acc[r], ub, tile_bias[row]);
          if (item < chunk_end && user_ok) {
            const int64_t gid = item * p.id_stride + p.id_offset;
            c_above += (s > bscore || (s == bscore && gid < bid)) ? 1ull : 0ull;
            for (int64_t e = e_begin; e < e_end; ++e) {
              c_lt += s < p.thr_lo[e] ? 1ull : 0ull;
              c_le += s <= p.thr_hi[e] ? 1ull : 0ull;
            }
          }
        }
      }
    }
  }

  if (!user_ok) return;
  if (MODE == 0) {
    const int nslots = p.nchunks * 2;
    const int64_t slot = uslot * nslots + blockIdx.y * 2 + h;
#pragma unroll
    for (int i = 0; i < kCap; ++i) {
      p.cand_scores[slot * kCap + i] = i < k ? top_s[i] : -INFINITY;
      p.cand_ids[slot * kCap + i] = i < k ? top_i[i] : 0x7fffffff;
    }
  } else {
    atomicAdd(&p.counts[uslot * 3 + 0], c_above);
    atomicAdd(&p.counts[uslot * 3 + 1], c_lt);
    atomicAdd(&p.counts[uslot * 3 + 2], c_le);
  }
}

// ---- K6, LDS-DMA variant (embedding_dim 32 / 64 / 128, float4-aligned tables) ----------------------
// Same MFMA mapping, same k order (so chain_dot() still reproduces every score bit for bit), but
//   * 8 waves = 256 users share each 64-item tile (half the L2 -> LDS traffic per flop);
//   * the tile never passes through VGPRs: global_load_lds_dwordx4 writes it straight into one of
//     two LDS buffers, so tile t+1 is in flight during all of tile t's MFMAs (one barrier per tile);
//   * an LDS-DMA destination is wave-linear, so rows cannot be padded; the bank-conflict-free image
//     is an XOR swizzle applied on the per-lane SOURCE address instead: 16-byte chunk c of tile row
//     r lives at slot r * CPR + (c ^ (r % S)), S = min(16, CPR), and the A-fragment reads apply the
//     same XOR.
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA through inline asm.  After __builtin_amdgcn_global_load_lds the compiler's wait-count pass
// makes every later LDS read wait for ALL outstanding DMA (it cannot tell the two tile buffers
// apart): it placed an s_waitcnt vmcnt(0) right after the issue of tile t+1, so the DMA never
// overlapped the MFMAs of tile t inside a workgroup.  The asm form is invisible to that pass; the
// explicit s_waitcnt vmcnt(0) at the top of the tile loop is the only wait.  `lds` is wave-uniform.
template <int BYTES>
__device__ __forceinline__ void dma_to_lds(const void* g, const void* lds) {
  const uint32_t l = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)lds)));
  if (BYTES == 16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l)
                 : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l)
                 : "memory", "m0");
}

constexpr int kBlockDma = 512;
constexpr int kWavesDma = kBlockDma / kWave;
constexpr int kUsersPerBlockDma = 32 * kWavesDma;  // 256

// ---- rank counting: a per-user search tree in LDS ---------------------------------------------------
// What the AUC needs from score s is  sum_e [s < lo_e]  and  sum_e [s <= hi_e] = sum_e [s < hn_e],
// hn_e = nextafter(hi_e, +inf), over the user's positives e.  Testing s against every threshold
// costs 2 vector instructions per test whatever the form (v_cmp + add-with-carry, or subtract +
// v_alignbit of the sign bit), and on gfx950 the fp32 MFMAs and the vector instructions of a SIMD do
// not overlap (tools/calib/mfma_valu_overlap.hip: a wave's MFMA + VALU time is the sum), so that
// counting cannot be hidden under the GEMM: with 10 positives it is 40 instructions per score.
// Instead the 2n values {lo_e} u {hn_e} of a user's first kTreePos positives are sorted once per
// workgroup (descending) into an implicit binary tree in LDS (node k has children 2k, 2k+1; unused
// nodes hold -inf).  A score walks it in 5 steps,  k <- 2k + [s < node[k]]  -- one LDS read, one
// subtract and one v_alignbit each (s < v  <=>  sign bit of s - v; exact: a negative difference
// cannot round to +0, equal operands give +0) -- and ends at k = 32 + g, g = number of values
// above s.  Entry 32 + g of the same array holds (#lo among the g largest) | (#hn among them) << 16,
// which are the two sums.  17 vector instructions per score, independent of n.
// The sigmoid-tie intervals [lo_e, hi_e] of distinct positives are disjoint or identical, which is
// all the ordering of equal values needs.  Scores are assumed finite (as the sign test always did).
constexpr int kTreePos = 15;     // positives per user in the tree: 2 * 15 values, 31 nodes
constexpr int kTreeWords = 64;   // [1, 32) nodes, [32, 64) counts

// nextafter(x, +inf) for the "<=" thresholds (x is never NaN); +inf stays +inf
__device__ __forceinline__ float next_up(float x) {
  if (x == INFINITY) return x;
  if (x == 0.0f) return __uint_as_float(1u);                    // smallest positive denormal
  const uint32_t u = __float_as_uint(x);
  return __uint_as_float(x > 0.0f ? u + 1u : u - 1u);
}

// Builds one user's tree; `tree` points at the user's column of a [kTreeWords][32] word array
// (entry k at tree[k * 32]).  Called by one lane per user.
__device__ void build_rank_tree(uint32_t* tree, const float* __restrict__ thr_lo,
                                const float* __restrict__ thr_hi, int n) {
  for (int k = 1; k < 32; ++k) tree[k * 32] = __float_as_uint(-INFINITY);
  for (int g = 0; g < 32; ++g) tree[(32 + g) * 32] = 0u;
  const int m = 2 * n;
  for (int i = 0; i < m; ++i) {
    const float vi = i < n ? thr_lo[i] : next_up(thr_hi[i - n]);
    int rank = 0;                                   // position in descending order (stable)
    for (int j = 0; j < m; ++j) {
      const float vj = j < n ? thr_lo[j] : next_up(thr_hi[j - n]);
      rank += (vj > vi || (vj == vi && j < i)) ? 1 : 0;
    }
    // in-order position rank of the perfect 31-node tree -> node number
    const int t = rank + 1, z = __builtin_ctz(t);
    tree[((1 << (4 - z)) + (t >> (z + 1))) * 32] = __float_as_uint(vi);
    tree[(32 + rank + 1) * 32] += i < n ? 1u : 0x10000u;   // delta: this value is among the g
  }                                                        // largest for every g > rank
  uint32_t run = 0;
  for (int g = 1; g < 32; ++g) {
    run += tree[(32 + g) * 32];
    tree[(32 + g) * 32] = run;
  }
}

// sum over a lane's 16 scores of (sum_e [s < lo_e]) | (sum_e [s < hn_e]) << 16.  The 16 walks go
// down the tree level by level, so that 16 LDS reads are in flight at a time (one walk at a time
// waits out the LDS latency five times per score).  MASKED: NaN scores (rows past the shard end in
// the partial last tile) walk somewhere and are dropped.
template <bool MASKED>
__device__ __forceinline__ uint32_t rank_tree_count16(const uint32_t* tree, const f32x16& acc) {
  uint32_t k[16];
  const float root = __uint_as_float(tree[32]);
#pragma unroll
  for (int rr = 0; rr < 16; ++rr)
    k[rr] = __builtin_amdgcn_alignbit(1u, __float_as_uint(acc[rr] - root), 31);
#pragma unroll
  for (int level = 1; level < 5; ++level) {
    float v[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) v[rr] = __uint_as_float(tree[k[rr] * 32]);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr)
      k[rr] = __builtin_amdgcn_alignbit(k[rr], __float_as_uint(acc[rr] - v[rr]), 31);
  }
  uint32_t c[16], total = 0;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) c[rr] = tree[k[rr] * 32];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) total += (!MASKED || acc[rr] == acc[rr]) ? c[rr] : 0u;
  return total;
}

// #{scores that precede the user's best positive (bscore, bid) in (score desc, id asc) order}
template <bool MASKED>
__device__ __forceinline__ uint32_t count_above16(const f32x16& acc, float bscore, int bid, int g0,
                                                  int st) {
  uint32_t n = 0;
  if (MASKED) {                                  // compare form: a NaN fails every test
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int gid = g0 + ((rr & 3) + 8 * (rr >> 2)) * st;
      n += ((acc[rr] > bscore) | ((acc[rr] == bscore) & (gid < bid))) ? 1u : 0u;
    }
    return n;
  }
  // s > bscore  <=>  sign bit of bscore - s; exact ties (the positive itself, or a colliding
  // score) are rare and settled by id in a branch almost no wave takes
  uint32_t bits = 0;
  bool any_eq = false;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(bscore - acc[rr]), 31);
    any_eq |= acc[rr] == bscore;
  }
  n = __popc(bits);
  if (any_eq) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int gid = g0 + ((rr & 3) + 8 * (rr >> 2)) * st;
      n += ((acc[rr] == bscore) & (gid < bid)) ? 1u : 0u;
    }
  }
  return n;
}

// MODE 2 = MODE 0 and MODE 1 in ONE pass over the catalogue (both epilogues on every sub-tile):
// `[mapk, mrr, auc]` then costs one GEMM instead of two.
template <int D8, int MODE>
__global__ __launch_bounds__(kBlockDma) void eval_gemm_dma_kernel(const EvalParams p) {
  constexpr bool TOPK = MODE != 1, RANK = MODE != 0;
  constexpr int D = D8 * 8;
  constexpr int CPR = D / 4;                     // 16-byte chunks per row
  constexpr int S = CPR < 16 ? CPR : 16;         // swizzle span
  constexpr int kTileF4 = kTileItems * CPR;      // float4 slots per tile
  constexpr int kPerWave = CPR / kWavesDma;      // DMA instructions per wave per tile
  constexpr int kBufF4 = kTileF4 + kTileItems / 4;  // tile + its 64 item biases
  static_assert(kPerWave >= 1, "embedding_dim >= 32");
  __shared__ float4 smem[2 * kBufF4];            // ONE array: both tiles and both bias vectors
  // RANK: the search trees of the workgroup's users, [wave][entry][user column]
  __shared__ uint32_t rank_trees[RANK ? kWavesDma * kTreeWords * 32 : 1];

  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int col = lane & 31;
  const int h = lane >> 5;

  const int64_t uslot = static_cast<int64_t>(blockIdx.x) * kUsersPerBlockDma + wave * 32 + col;
  const bool user_ok = uslot < p.n_users;
  int64_t uid = p.user_ids[user_ok ? uslot : p.n_users - 1];
  if (uid < 0) uid = 0;
  if (uid >= p.t.num_users) uid = p.t.num_users - 1;
  const float* urow = p.t.user_emb + uid * D;
  const float ub = p.t.user_bias[uid];
  f32x4 ufrag[D8];
#pragma unroll
  for (int q = 0; q < D8; ++q) ufrag[q] = *reinterpret_cast<const f32x4*>(urow + 8 * q + 4 * h);

  float top_s[kCap];   // TOPK: the sorted candidate list
  int top_i[kCap];
  uint32_t* tree = rank_trees + (RANK ? wave * kTreeWords * 32 + col : 0);
  unsigned long long c_above = 0, c_lt = 0, c_le = 0, c_below_all = 0;
  int64_t e_begin = 0, e_end = 0;
  float bscore = 0.0f, lo_min = INFINITY, hi_max = -INFINITY;
  int bid = 0;
  if (TOPK) {
#pragma unroll
    for (int i = 0; i < kCap; ++i) {
      top_s[i] = -INFINITY;
      top_i[i] = 0x7fffffff;
    }
  }
  if (RANK) {
    // the first kTreePos positives of the user go into its LDS tree.  The rare user with more
    // walks the rest of its CSR row per score (lo_min / hi_max bound the scores that can differ
    // from "below all" / "above all" of that remainder).
    if (user_ok) {
      e_begin = p.indptr[uslot];
      e_end = p.indptr[uslot + 1];
      bscore = p.best_score[uslot];
      bid = static_cast<int>(p.best_id[uslot]);
      for (int64_t e = e_begin + kTreePos; e < e_end; ++e) {
        lo_min = fminf(lo_min, p.thr_lo[e]);
        hi_max = fmaxf(hi_max, p.thr_hi[e]);
      }
    }
    if (h == 0) {
      int n = static_cast<int>(e_end - e_begin);
      if (n > kTreePos) n = kTreePos;
      build_rank_tree(tree, p.thr_lo + e_begin, p.thr_hi + e_begin, n);
    }
    // read by both lane halves after the first barrier of the tile loop
  }
  const bool any_long = __ballot(e_end - e_begin > kTreePos) != 0ull;
  const int k = p.k;
  float thr = -INFINITY;  // TOPK: score of the k-th best candidate so far
  const bool has_cut = MODE == 0 && p.cut_s != nullptr;  // uniform (later rounds of k > kCap)
  const float cut_s = has_cut && user_ok ? p.cut_s[uslot] : INFINITY;
  const int cut_i = has_cut && user_ok ? p.cut_i[uslot] : -1;

  const int64_t chunk_items = static_cast<int64_t>(p.tiles_per_chunk) * kTileItems;
  const int64_t chunk_begin = p.item_begin + static_cast<int64_t>(blockIdx.y) * chunk_items;
  int64_t chunk_end = chunk_begin + chunk_items;
  if (chunk_end > p.item_end) chunk_end = p.item_end;
  const int64_t last_row = p.item_end - 1;       // rows past the shard end re-read this one (masked)

  // per-lane constants of the DMA: slot i = I * 64 + lane of the tile, I = wave * kPerWave + n
  auto issue = [&](int64_t base, int buf) {
    float4* dst = smem + buf * kBufF4;
#pragma unroll
    for (int n = 0; n < kPerWave; ++n) {
      const int I = wave * kPerWave + n;
      const int i = I * kWave + lane;
      const int r = i / CPR;
      const int c = (i % CPR) ^ (r & (S - 1));
      int64_t item = base + r;
      if (item > last_row) item = last_row;
      dma_to_lds<16>(p.t.item_emb + item * D + c * 4, dst + I * kWave);
    }
    if (wave == 0) {
      int64_t item = base + lane;
      if (item > last_row) item = last_row;
      dma_to_lds<4>(p.t.item_bias + item, dst + kTileF4);
    }
  };

  // Every value loaded from global memory above is made ready HERE: the compiler would otherwise
  // keep its s_waitcnt vmcnt(n) for them inside the tile loop, and vmcnt is one in-order counter --
  // such a wait would also drain the DMA instructions it does not know about.
#pragma unroll
  for (int q = 0; q < D8; ++q) asm volatile("" : "+v"(ufrag[q]));
  float ub_ready = ub, cut_s_ready = cut_s;
  int cut_i_ready = cut_i;
  asm volatile("" : "+v"(ub_ready), "+v"(cut_s_ready), "+v"(cut_i_ready));

  int buf = 0;
  if (chunk_begin < chunk_end) issue(chunk_begin, 0);
  for (int64_t base = chunk_begin; base < chunk_end; base += kTileItems) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of tile `base` has landed
    __syncthreads();                                  // everyone's has; buffer buf ^ 1 is free
    if (base + kTileItems < chunk_end) issue(base + kTileItems, buf ^ 1);
    const float4* tile = smem + buf * kBufF4;
    const float* tile_bias = reinterpret_cast<const float*>(tile + kTileF4);
    const bool full = base + kTileItems <= chunk_end;

    // RANK keeps one sub-tile live at a time
#pragma unroll MODE == 0 ? 2 : 1
    for (int sub = 0; sub < kTileItems / 32; ++sub) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      const int r = sub * 32 + col;
      const float4* arow = tile + r * CPR;
      const int sw = r & (S - 1);
      // A fragments are fetched two steps ahead of the MFMAs that consume them (an LDS read is
      // ~2 MFMAs long; read-then-use in the same step stalls the chain on every fragment)
      float4 a0 = arow[(0 + h) ^ sw];
      float4 a1 = arow[((D8 > 1 ? 2 : 0) + h) ^ sw];
#pragma unroll
      for (int q = 0; q < D8; ++q) {
        const float4 a = a0;
        a0 = a1;
        if (q + 2 < D8) a1 = arow[(2 * (q + 2) + h) ^ sw];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, ufrag[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, ufrag[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, ufrag[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, ufrag[q][3], acc, 0, 0, 0);
      }
      // ---- epilogue: this lane's user x 16 items (rows (r&3) + 8(r>>2) + 4h of the sub-tile) ----
      float smax = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(tile_bias + sub * 32 + 8 * g + 4 * h);
        acc[4 * g + 0] = finish_score(acc[4 * g + 0], ub, b4.x);
        acc[4 * g + 1] = finish_score(acc[4 * g + 1], ub, b4.y);
        acc[4 * g + 2] = finish_score(acc[4 * g + 2], ub, b4.z);
        acc[4 * g + 3] = finish_score(acc[4 * g + 3], ub, b4.w);
      }
      if (!full) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = sub * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
          // masked rows: -inf never enters a top-k list; NaN fails every rank-count comparison
          // (and every top-k comparison in the fused mode: fmaxf drops it, `> thr` is false)
          if (base + row >= chunk_end) acc[rr] = MODE == 0 ? -INFINITY : __builtin_nanf("");
        }
      }
      if (MODE == 0 && has_cut) {
        const int g0 = static_cast<int>((base + sub * 32 + 4 * h) * p.id_stride + p.id_offset);
        const int st = static_cast<int>(p.id_stride);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int gid = g0 + ((rr & 3) + 8 * (rr >> 2)) * st;
          const bool ok = acc[rr] < cut_s || (acc[rr] == cut_s && gid > cut_i);
          acc[rr] = ok ? acc[rr] : -INFINITY;
        }
      }
      if (TOPK) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) smax = fmaxf(smax, acc[rr]);
        if (smax > thr) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            if (acc[rr] > thr) {
              const int row = sub * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
              float cs = acc[rr];
              int ci = static_cast<int>((base + row) * p.id_stride + p.id_offset);
#pragma unroll
              for (int i = 0; i < kCap; ++i) {
                const bool better = cs > top_s[i];
                const float ts = top_s[i];
                const int ti = top_i[i];
                top_s[i] = better ? cs : ts;
                top_i[i] = better ? ci : ti;
                cs = better ? ts : cs;
                ci = better ? ti : ci;
              }
              // entry threshold = the k-th best so far (k is uniform: a chain of scalar selects)
#pragma unroll
              for (int i = 0; i < kCap; ++i) thr = i == k - 1 ? top_s[i] : thr;
            }
          }
        }
      }
      if (RANK) {
        // 32-bit counters per sub-tile, folded into the 64-bit ones afterwards
        uint32_t t_above = 0, t_pair = 0;
        const int g0 = static_cast<int>((base + sub * 32 + 4 * h) * p.id_stride + p.id_offset);
        const int st = static_cast<int>(p.id_stride);   // global ids fit 31 bits (entry points)
        if (full) {
          t_above = count_above16<false>(acc, bscore, bid, g0, st);
          t_pair = rank_tree_count16<false>(tree, acc);
        } else {
          t_above = count_above16<true>(acc, bscore, bid, g0, st);
          t_pair = rank_tree_count16<true>(tree, acc);
        }
        // (without this the compiler accumulates straight into the 64-bit counters)
        asm volatile("" : "+v"(t_above), "+v"(t_pair));
        c_above += t_above;
        c_lt += t_pair & 0xffffu;
        c_le += t_pair >> 16;
        if (any_long) {
#pragma unroll 1
          for (int rr = 0; rr < 16; ++rr) {
            const float s = acc[rr];
            if (s < lo_min) {
              c_below_all += 1ull;                     // below every remaining positive
            } else if (s <= hi_max) {
              for (int64_t e = e_begin + kTreePos; e < e_end; ++e) {
                c_lt += s < p.thr_lo[e] ? 1ull : 0ull;
                c_le += s <= p.thr_hi[e] ? 1ull : 0ull;
              }
            }
          }
        }
      }
    }
    buf ^= 1;
  }

  if (!user_ok) return;
  if (TOPK) {
    const int nslots = p.nchunks * 2;
    const int64_t slot = uslot * nslots + blockIdx.y * 2 + h;
#pragma unroll
    for (int i = 0; i < kCap; ++i) {
      p.cand_scores[slot * kCap + i] = i < k ? top_s[i] : -INFINITY;
      p.cand_ids[slot * kCap + i] = i < k ? top_i[i] : 0x7fffffff;
    }
  }
  if (RANK) {
    const unsigned long long npos =
        e_end - e_begin > kTreePos ? static_cast<unsigned long long>(e_end - e_begin - kTreePos)
                                   : 0ull;
    atomicAdd(&p.counts[uslot * 3 + 0], c_above);
    atomicAdd(&p.counts[uslot * 3 + 1], c_lt + c_below_all * npos);
    atomicAdd(&p.counts[uslot * 3 + 2], c_le + c_below_all * npos);
  }
}

// ---- K6c: the MAP@k scoring pass on bf16 MFMA, split x 3 (COLLIE_EVAL_BF16X3, opt-in) --------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) carries 24
// significant bits, so  u . i ~= sum over the six products of total order <= 2 (hi hi, hi mid,
// mid hi, mid mid, hi lo, lo hi) -- each one v_mfma_f32_32x32x16_bf16 accumulating in f32 -- agrees
// with the f32 dot product to a few ulp, at 6 x 32 cycles per 16 k against 8 x 64 cycles on the f32
// instruction (2.7x fewer MFMA cycles).  The scores of this pass are NOT the reference's bits, so
// they only RANK: every slot keeps its 16 best candidates, the merge keeps 16 per user, those are
// re-scored with the exact chain (chain_dot: the bits of the f32 pass) and the k <= 12 best of the
// re-scored list are emitted in (score desc, id asc) order.  A result differs from the f32 pass only
// if an item of the true top-k ranks below 16th in the approximate order, i.e. if more than 16 - k
// items sit within ~1e-7 (relative) of the k-th score.
// Same geometry as eval_gemm_dma_kernel: 8 waves = 256 users per workgroup share a 64-item tile that
// LDS-DMA writes into one of two buffers; an item's three bf16 rows are contiguous in the split
// table ([num_items, 3, dim] bf16, collie_mf_eval_split_bf16x3), 16-byte chunk c of split s of tile
// row r lives at slot r * RC + s * CPR + (c ^ (r % S)).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// x -> (hi, mid, lo) bf16 bit patterns; the two residuals are exact in f32
__device__ __forceinline__ void split_bf16x3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = bf16_rne(x);
  const float r1 = x - __uint_as_float(hi << 16);
  mid = bf16_rne(r1);
  const float r2 = r1 - __uint_as_float(mid << 16);
  lo = bf16_rne(r2);
}

__global__ __launch_bounds__(kBlock) void split_bf16x3_kernel(const float* __restrict__ src,
                                                              int64_t rows, int dim,
                                                              uint16_t* __restrict__ out) {
  const int64_t total = rows * dim;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / dim;
    const int c = static_cast<int>(i - r * dim);
    uint32_t hi, mid, lo;
    split_bf16x3(src[i], hi, mid, lo);
    uint16_t* o = out + r * 3 * dim + c;
    o[0] = static_cast<uint16_t>(hi);
    o[dim] = static_cast<uint16_t>(mid);
    o[2 * dim] = static_cast<uint16_t>(lo);
  }
}

struct Bf16Params {
  EvalParams e;
  const uint16_t* item_split;   // [num_items, 3, dim] bf16
};

template <int D16>
__global__ __launch_bounds__(kBlockDma) void eval_gemm_bf16x3_kernel(const Bf16Params bp) {
  const EvalParams& p = bp.e;
  constexpr int D = D16 * 16;
  constexpr int CPR = D / 8;                     // 16-byte chunks per bf16 row
  constexpr int RC = 3 * CPR;                    // chunks per item (hi, mid, lo rows)
  constexpr int S = CPR < 16 ? CPR : 16;         // swizzle span
  constexpr int kTileF4 = kTileItems * RC;
  constexpr int kPerWave = RC / kWavesDma;
  constexpr int kBufF4 = kTileF4 + kTileItems / 4;
  static_assert(kPerWave * kWavesDma == RC, "embedding_dim 64 or 128");
  __shared__ float4 smem[2 * kBufF4];

  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int col = lane & 31;
  const int h = lane >> 5;

  const int64_t uslot = static_cast<int64_t>(blockIdx.x) * kUsersPerBlockDma + wave * 32 + col;
  const bool user_ok = uslot < p.n_users;
  int64_t uid = p.user_ids[user_ok ? uslot : p.n_users - 1];
  if (uid < 0) uid = 0;
  if (uid >= p.t.num_users) uid = p.t.num_users - 1;
  const float* urow = p.t.user_emb + uid * D;
  const float ub = p.t.user_bias[uid];
  // B fragments: lane (col, h) holds user element k = 16 q + 8 h + j in element j of k-block q
  bf16x8 bhi[D16], bmid[D16], blo[D16];
#pragma unroll
  for (int q = 0; q < D16; ++q) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(urow + 16 * q + 8 * h);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(urow + 16 * q + 8 * h + 4);
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t a, b, c;
      split_bf16x3(j < 4 ? x0[j & 3] : x1[j & 3], a, b, c);
      if (j & 1) {
        w[0][j >> 1] |= a << 16;
        w[1][j >> 1] |= b << 16;
        w[2][j >> 1] |= c << 16;
      } else {
        w[0][j >> 1] = a;
        w[1][j >> 1] = b;
        w[2][j >> 1] = c;
      }
    }
    const uint4 v0 = make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]);
    const uint4 v1 = make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]);
    const uint4 v2 = make_uint4(w[2][0], w[2][1], w[2][2], w[2][3]);
    bhi[q] = __builtin_bit_cast(bf16x8, v0);
    bmid[q] = __builtin_bit_cast(bf16x8, v1);
    blo[q] = __builtin_bit_cast(bf16x8, v2);
  }

  float top_s[kCap];
  int top_i[kCap];
#pragma unroll
  for (int i = 0; i < kCap; ++i) {
    top_s[i] = -INFINITY;
    top_i[i] = 0x7fffffff;
  }
  const int k = p.k;
  float thr = -INFINITY;

  const int64_t chunk_items = static_cast<int64_t>(p.tiles_per_chunk) * kTileItems;
  const int64_t chunk_begin = p.item_begin + static_cast<int64_t>(blockIdx.y) * chunk_items;
  int64_t chunk_end = chunk_begin + chunk_items;
  if (chunk_end > p.item_end) chunk_end = p.item_end;
  const int64_t last_row = p.item_end - 1;

  auto issue = [&](int64_t base, int buf) {
    float4* dst = smem + buf * kBufF4;
#pragma unroll
    for (int n = 0; n < kPerWave; ++n) {
      const int I = wave * kPerWave + n;
      const int i = I * kWave + lane;
      const int r = i / RC;
      const int c = i % RC;
      const int sp = c / CPR;
      const int cc = (c % CPR) ^ (r & (S - 1));
      int64_t item = base + r;
      if (item > last_row) item = last_row;
      dma_to_lds<16>(bp.item_split + (item * 3 + sp) * D + cc * 8, dst + I * kWave);
    }
    if (wave == 0) {
      int64_t item = base + lane;
      if (item > last_row) item = last_row;
      dma_to_lds<4>(p.t.item_bias + item, dst + kTileF4);
    }
  };

#pragma unroll
  for (int q = 0; q < D16; ++q) asm volatile("" : "+v"(bhi[q]), "+v"(bmid[q]), "+v"(blo[q]));
  float ub_ready = ub;
  asm volatile("" : "+v"(ub_ready));

  int buf = 0;
  if (chunk_begin < chunk_end) issue(chunk_begin, 0);
  for (int64_t base = chunk_begin; base < chunk_end; base += kTileItems) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (base + kTileItems < chunk_end) issue(base + kTileItems, buf ^ 1);
    const float4* tile = smem + buf * kBufF4;
    const float* tile_bias = reinterpret_cast<const float*>(tile + kTileF4);
    const bool full = base + kTileItems <= chunk_end;
#pragma unroll 1
    for (int sub = 0; sub < kTileItems / 32; ++sub) {
      // the accumulators start from the two biases (these scores only rank: the exact
      // (dot + ub) + ib order of the f32 pass is restored by rescore_kernel), 32 vector
      // instructions fewer per sub-tile than adding them afterwards
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(tile_bias + sub * 32 + 8 * g + 4 * h);
        acc[4 * g + 0] = ub + b4.x;
        acc[4 * g + 1] = ub + b4.y;
        acc[4 * g + 2] = ub + b4.z;
        acc[4 * g + 3] = ub + b4.w;
      }
      const int r = sub * 32 + col;
      const float4* arow = tile + r * RC;
      const int sw = r & (S - 1);
      // A fragments of k-block q: chunk 2 q + h of each of the item's three rows
      float4 n_hi = arow[(0 + h) ^ sw];
      float4 n_mid = arow[CPR + ((0 + h) ^ sw)];
      float4 n_lo = arow[2 * CPR + ((0 + h) ^ sw)];
#pragma unroll
      for (int q = 0; q < D16; ++q) {
        const bf16x8 ahi = __builtin_bit_cast(bf16x8, n_hi);
        const bf16x8 amid = __builtin_bit_cast(bf16x8, n_mid);
        const bf16x8 alo = __builtin_bit_cast(bf16x8, n_lo);
        if (q + 1 < D16) {
          const int cn = (2 * (q + 1) + h) ^ sw;
          n_hi = arow[cn];
          n_mid = arow[CPR + cn];
          n_lo = arow[2 * CPR + cn];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid, bmid[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid, bhi[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bmid[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi[q], acc, 0, 0, 0);
      }
      float smax = -INFINITY;
      if (!full) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int row = sub * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
          if (base + row >= chunk_end) acc[rr] = -INFINITY;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) smax = fmaxf(smax, acc[rr]);
      if (smax > thr) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          if (acc[rr] > thr) {
            const int row = sub * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            float cs = acc[rr];
            int ci = static_cast<int>((base + row) * p.id_stride + p.id_offset);
#pragma unroll
            for (int i = 0; i < kCap; ++i) {
              const bool better = cs > top_s[i];
              const float ts = top_s[i];
              const int ti = top_i[i];
              top_s[i] = better ? cs : ts;
              top_i[i] = better ? ci : ti;
              cs = better ? ts : cs;
              ci = better ? ti : ci;
            }
#pragma unroll
            for (int i = 0; i < kCap; ++i) thr = i == k - 1 ? top_s[i] : thr;
          }
        }
      }
    }
    buf ^= 1;
  }
  if (!user_ok) return;
  const int nslots = p.nchunks * 2;
  const int64_t slot = uslot * nslots + blockIdx.y * 2 + h;
#pragma unroll
  for (int i = 0; i < kCap; ++i) {
    p.cand_scores[slot * kCap + i] = i < k ? top_s[i] : -INFINITY;
    p.cand_ids[slot * kCap + i] = i < k ? top_i[i] : 0x7fffffff;
  }
}

// the candidates the approximate pass kept, re-scored with the exact chain of the f32 pass
__global__ __launch_bounds__(kBlock) void rescore_kernel(const collie_mf_tables t,
                                                         const int64_t* __restrict__ user_ids,
                                                         int64_t n_users, int kk, int D8,
                                                         int64_t id_stride, int64_t id_offset,
                                                         const int64_t* __restrict__ ids,
                                                         float* __restrict__ out_scores,
                                                         int32_t* __restrict__ out_ids) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n_users * kk) return;
  const int64_t u = i / kk;
  const int64_t gid = ids[i];
  float s = -INFINITY;
  int32_t id = 0x7fffffff;
  if (gid >= 0) {
    int64_t uid = user_ids[u];
    if (uid < 0) uid = 0;
    if (uid >= t.num_users) uid = t.num_users - 1;
    const int64_t row = (gid - id_offset) / id_stride;
    s = finish_score(chain_dot(t.user_emb + uid * t.dim, t.item_emb + row * t.dim, t.dim, D8),
                     t.user_bias[uid], t.item_bias[row]);
    id = static_cast<int32_t>(gid);
  }
  out_scores[i] = s;
  out_ids[i] = id;
}

// one wave per user: k rounds of "best remaining candidate" over the user's n candidates
__global__ __launch_bounds__(kBlock) void merge_topk_kernel(const float* __restrict__ cand_scores,
                                                            const int32_t* __restrict__ cand_ids,
                                                            int64_t n_users, int n, int k,
                                                            int has_y, float y_lo, float y_hi,
                                                            float* __restrict__ out_scores,
                                                            int64_t* __restrict__ out_ids,
                                                            int out_ld, int col0,
                                                            float* __restrict__ cut_s_out,
                                                            int32_t* __restrict__ cut_i_out) {
  const int lane = lane_id();
  const int64_t user = static_cast<int64_t>(blockIdx.x) * kWaves + threadIdx.x / kWave;
  if (user >= n_users) return;
  const float* cs = cand_scores + user * n;
  const int32_t* ci = cand_ids + user * n;
  float last_s = INFINITY;
  int last_i = -1;
  for (int round = 0; round < k; ++round) {
    // best candidate strictly after (last_s, last_i) in (score desc, id asc) order
    float bs = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < n; j += kWave) {
      const float s = cs[j];
      const int id = ci[j];
      const bool after = s < last_s || (s == last_s && id > last_i);
      const bool better = s > bs || (s == bs && id < bi);
      if (after && better && id != 0x7fffffff) {
        bs = s;
        bi = id;
      }
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, kWave);
      const int oi = __shfl_xor(bi, o, kWave);
      if (os > bs || (os == bs && oi < bi)) {
        bs = os;
        bi = oi;
      }
    }
    if (lane == 0) {
      out_scores[user * out_ld + col0 + round] = yr(bs, has_y, y_lo, y_hi);
      out_ids[user * out_ld + col0 + round] = bi == 0x7fffffff ? -1 : bi;
    }
    last_s = bs;
    last_i = bi;
  }
  if (cut_s_out && lane == 0) {  // where the next round of a k > kCap query resumes
    cut_s_out[user] = last_s;
    cut_i_out[user] = last_i;
  }
}

// scores of the positives (CSR entries) through the same chain as the GEMM + sigmoid intervals
__global__ __launch_bounds__(kBlock) void pos_scores_kernel(
    const collie_mf_tables t, const int64_t* __restrict__ user_ids, int64_t n_users,
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices, int D8, int has_y,
    float y_lo, float y_hi, int compute, int64_t id_stride, int64_t id_offset,
    float* __restrict__ pos_scores, float* __restrict__ thr_lo, float* __restrict__ thr_hi) {
  const int64_t nnz = indptr[n_users];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < nnz; e += stride) {
    // user slot of entry e: binary search in indptr
    int64_t lo = 0, hi = n_users;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) / 2;
      if (indptr[mid] <= e) lo = mid; else hi = mid;
    }
    float s;
    if (compute) {
      int64_t uid = user_ids[lo];
      if (uid < 0) uid = 0;
      if (uid >= t.num_users) uid = t.num_users - 1;
      // global item id -> row of this shard (rows of other shards contribute 0: all-reduced later)
      const int64_t rel = indices[e] - id_offset;
      const int64_t item = rel / id_stride;
      if (rel >= 0 && rel % id_stride == 0 && item < t.num_items) {
        const float dot = chain_dot(t.user_emb + uid * t.dim, t.item_emb + item * t.dim, t.dim, D8);
        s = finish_score(dot, t.user_bias[uid], t.item_bias[item]);
      } else {
        s = 0.0f;
      }
      pos_scores[e] = s;
    } else {
      s = pos_scores[e];
    }
    if (thr_lo) {
      // [lo, hi] = the raw-score interval whose sigmoid(preds) equals this positive's
      // (metrics.py:277 ranks sigmoid(preds) in fp32, which ties scores the raw values do not)
      const float target = sigmoidf_(yr(s, has_y, y_lo, y_hi));
      uint32_t a = f2o(-INFINITY), b = f2o(s);      // smallest x in [a, b] with F(x) == target
      while (a < b) {
        const uint32_t m = a + (b - a) / 2;
        if (sigmoidf_(yr(o2f(m), has_y, y_lo, y_hi)) >= target) b = m; else a = m + 1;
      }
      thr_lo[e] = o2f(a);
      a = f2o(s);
      b = f2o(INFINITY);                             // largest x in [a, b] with F(x) == target
      while (a < b) {
        const uint32_t m = a + (b - a + 1) / 2;
        if (sigmoidf_(yr(o2f(m), has_y, y_lo, y_hi)) <= target) a = m; else b = m - 1;
      }
      thr_hi[e] = o2f(a);
    }
  }
}

// best positive per user: max score, lowest item id on ties (first relevant item of the order)
__global__ __launch_bounds__(kBlock) void best_pos_kernel(const int64_t* __restrict__ indptr,
                                                          const int64_t* __restrict__ indices,
                                                          const float* __restrict__ pos_scores,
                                                          int64_t n_users,
                                                          float* __restrict__ best_score,
                                                          int64_t* __restrict__ best_id) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (u >= n_users) return;
  float bs = -INFINITY;
  int64_t bi = INT64_MAX;
  for (int64_t e = indptr[u]; e < indptr[u + 1]; ++e) {
    const float s = pos_scores[e];
    const int64_t id = indices[e];
    if (s > bs || (s == bs && id < bi)) {
      bs = s;
      bi = id;
    }
  }
  best_score[u] = bs;
  best_id[u] = bi;
}

// MAP@k, RR, AUC per user (metrics.py:178-198, :224-240, :266-282)
__global__ __launch_bounds__(kBlock) void metrics_kernel(
    const int64_t* __restrict__ topk_ids, int k, int64_t n_users, int64_t n_items,
    const int64_t* __restrict__ indptr, const int64_t* __restrict__ indices,
    const float* __restrict__ pos_scores, const unsigned long long* __restrict__ counts,
    int has_y, float y_lo, float y_hi, double* __restrict__ ap, float* __restrict__ rr,
    double* __restrict__ auc) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (u >= n_users) return;
  const int64_t b = indptr[u], e = indptr[u + 1];
  const int64_t npos = e - b;
  if (ap) {
    // AP = sum_j rel_j * cumsum(rel)_j / j  /  min(k, npos); NaN (npos == 0) -> 0
    double sum = 0.0;
    int hits = 0;
    for (int j = 0; j < k; ++j) {
      const int64_t id = topk_ids[u * k + j];
      // indices of a CSR row are sorted: binary search
      int64_t lo = b, hi = e;
      while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (indices[mid] < id) lo = mid + 1; else hi = mid;
      }
      if (lo < e && indices[lo] == id) {
        ++hits;
        sum += static_cast<double>(hits) * (1.0 / static_cast<double>(j + 1));
      }
    }
    const int64_t denom = npos < k ? npos : k;
    ap[u] = denom > 0 ? sum / static_cast<double>(denom) : 0.0;
  }
  if (rr) {
    // reciprocal rank of the best positive: 1 / (1 + #items ranked above it)
    rr[u] = npos > 0 ? 1.0f / (static_cast<float>(counts[u * 3 + 0]) + 1.0f) : 0.0f;
  }
  if (auc) {
    // Mann-Whitney with tie averaging on sigmoid(preds); the all-item counts include the
    // positives themselves, which are removed pairwise here
    double lt = static_cast<double>(counts[u * 3 + 1]);
    double le = static_cast<double>(counts[u * 3 + 2]);
    for (int64_t x = b; x < e; ++x) {
      const float sx = sigmoidf_(yr(pos_scores[x], has_y, y_lo, y_hi));
      for (int64_t y = b; y < e; ++y) {
        const float sy = sigmoidf_(yr(pos_scores[y], has_y, y_lo, y_hi));
        lt -= sy < sx ? 1.0 : 0.0;
        le -= sy <= sx ? 1.0 : 0.0;
      }
    }
    const double nneg = static_cast<double>(n_items - npos);
    auc[u] = (npos > 0 && nneg > 0) ? (lt + 0.5 * (le - lt)) / (static_cast<double>(npos) * nneg)
                                    : 0.0;
  }
}

int d8_bucket(int dim) {
  const int d8 = (dim + 7) / 8;
  int b = 1;
  while (b < d8) b *= 2;
  return b;
}

struct EvalGeom {
  int nchunks;
  int tiles_per_chunk;
};

// item chunks per user block: enough workgroups to fill 256 CUs a few times over, at most 64
EvalGeom eval_geom(int64_t n_users, int64_t n_items_shard, int users_per_block) {
  EvalGeom g;
  const int64_t ntiles = (n_items_shard + kTileItems - 1) / kTileItems;
  const int64_t user_blocks = (n_users + users_per_block - 1) / users_per_block;
  int64_t want = (2048 + user_blocks - 1) / user_blocks;
  if (want > 64) want = 64;
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  g.tiles_per_chunk = static_cast<int>((ntiles + want - 1) / want);
  if (g.tiles_per_chunk < 1) g.tiles_per_chunk = 1;
  g.nchunks = static_cast<int>((ntiles + g.tiles_per_chunk - 1) / g.tiles_per_chunk);
  if (g.nchunks < 1) g.nchunks = 1;
  return g;
}

// the LDS-DMA kernel: exact power-of-two depth 32..128, 16-byte aligned rows
bool dma_ok(const collie_mf_tables* t) {
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  return (t->dim == 32 || t->dim == 64 || t->dim == 128) && al16(t->user_emb) && al16(t->item_emb);
}

struct EvalWs {
  float* cand_scores;
  int32_t* cand_ids;
  float* cut_s;      // [n_users] resume point between the rounds of a k > kCap query
  int32_t* cut_i;
  size_t bytes;
};

// candidate lists [n_users, nslots, kCap]; sized for whichever kernel geometry needs more slots
EvalWs carve_eval_ws(void* base, int64_t n_users, int64_t n_items_shard) {
  EvalWs w;
  const int a = eval_geom(n_users, n_items_shard, kUsersPerBlock).nchunks;
  const int b = eval_geom(n_users, n_items_shard, kUsersPerBlockDma).nchunks;
  Carver c(base);
  const size_t n = static_cast<size_t>(n_users) * (a > b ? a : b) * 2 * kCap;
  w.cand_scores = c.take<float>(n);
  w.cand_ids = c.take<int32_t>(n);
  w.cut_s = c.take<float>(static_cast<size_t>(n_users));
  w.cut_i = c.take<int32_t>(static_cast<size_t>(n_users));
  w.bytes = align_up(c.off, 256);
  return w;
}

// fills p.nchunks / p.tiles_per_chunk / p.D8 and launches the scoring kernel
template <int MODE>
int launch_gemm(hipStream_t stream, EvalParams& p) {
  const int64_t n_items = p.item_end - p.item_begin;
  p.D8 = d8_bucket(p.t.dim);
  if (dma_ok(&p.t)) {
    const EvalGeom g = eval_geom(p.n_users, n_items, kUsersPerBlockDma);
    p.nchunks = g.nchunks;
    p.tiles_per_chunk = g.tiles_per_chunk;
    dim3 grid(static_cast<unsigned>((p.n_users + kUsersPerBlockDma - 1) / kUsersPerBlockDma),
              static_cast<unsigned>(g.nchunks));
    switch (p.D8) {
      case 4: hipLaunchKernelGGL((eval_gemm_dma_kernel<4, MODE>), grid, dim3(kBlockDma), 0, stream, p); break;
      case 8: hipLaunchKernelGGL((eval_gemm_dma_kernel<8, MODE>), grid, dim3(kBlockDma), 0, stream, p); break;
      default: hipLaunchKernelGGL((eval_gemm_dma_kernel<16, MODE>), grid, dim3(kBlockDma), 0, stream, p); break;
    }
    return check_launch("eval_gemm_dma_kernel");
  }
  if constexpr (MODE == 2) {
    set_error("the fused top-k + rank-count pass exists for the LDS-DMA kernel only");
    return -1;
  } else {
  const EvalGeom g = eval_geom(p.n_users, n_items, kUsersPerBlock);
  p.nchunks = g.nchunks;
  p.tiles_per_chunk = g.tiles_per_chunk;
  dim3 grid(static_cast<unsigned>((p.n_users + kUsersPerBlock - 1) / kUsersPerBlock),
            static_cast<unsigned>(g.nchunks));
  switch (p.D8) {
    case 1: hipLaunchKernelGGL((eval_gemm_kernel<1, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((eval_gemm_kernel<2, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((eval_gemm_kernel<4, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((eval_gemm_kernel<8, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((eval_gemm_kernel<16, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((eval_gemm_kernel<32, MODE>), grid, dim3(kBlock), 0, stream, p); break;
    default:
      set_error("embedding_dim > 256 is not supported by the MFMA evaluator");
      return -1;
  }
  return check_launch("eval_gemm_kernel");
  }
}

int check_range(const collie_mf_tables* t, int64_t n_users, int64_t item_begin, int64_t item_end) {
  COLLIE_REQUIRE(n_users >= 1, "eval: empty user batch");
  COLLIE_REQUIRE(item_begin >= 0 && item_end > item_begin && item_end <= t->num_items,
                 "eval: bad item range [%lld, %lld)", static_cast<long long>(item_begin),
                 static_cast<long long>(item_end));
  COLLIE_REQUIRE(t->dim <= 256, "embedding_dim > 256 is not supported by the MFMA evaluator");
  return 0;
}

}  // namespace

// thresholds (sigmoid-space tie intervals) and best positive per user from GIVEN positive scores;
// shared with the dense-preds route (preds_metrics.hip)
int launch_pos_thresholds(const int64_t* indptr, const int64_t* indices, int64_t n_users,
                          int64_t nnz, float* pos_scores, float* thr_lo, float* thr_hi,
                          float* best_score, int64_t* best_id, hipStream_t stream) {
  if (nnz > 0 && thr_lo && thr_hi) {
    int64_t blocks = (nnz + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    collie_mf_tables none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL(pos_scores_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, none, static_cast<const int64_t*>(nullptr), n_users, indptr, indices,
                       1, 0, 0.0f, 0.0f, 0, int64_t{1}, int64_t{0}, pos_scores, thr_lo, thr_hi);
    if (check_launch("pos_scores_kernel")) return -2;
  }
  if (best_score && best_id) {
    hipLaunchKernelGGL(best_pos_kernel, dim3(static_cast<unsigned>((n_users + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, indptr, indices, pos_scores, n_users, best_score,
                       best_id);
    return check_launch("best_pos_kernel");
  }
  return 0;
}

}  // namespace collie

using namespace collie;

extern "C" int32_t collie_mf_eval_topk_capacity(void) { return kCap; }

extern "C" size_t collie_mf_eval_workspace_bytes(int64_t n_users, int64_t n_items_shard) {
  if (n_users < 1 || n_items_shard < 1) return 0;
  return carve_eval_ws(nullptr, n_users, n_items_shard).bytes;
}

namespace {

// Rounds [first_round, ceil(k / kCap)) of a top-k query: a round scores the catalogue (MODE 0),
// keeping per slot the kCap best items that come strictly after the previous round's last entry,
// and merges them into columns [16 r, 16 r + 16) of the output.  Round 0 has no cut; a caller that
// already produced round 0's candidates (the fused top-k + rank-count pass) starts at its merge.
int topk_rounds(EvalParams& p, int gemm_done_rounds, int32_t k, int32_t has_y_range, float y_lo,
                float y_hi, float* topk_scores, int64_t* topk_ids, const collie::EvalWs& ws,
                hipStream_t stream) {
  const int rounds = (k + kCap - 1) / kCap;
  for (int r = 0; r < rounds; ++r) {
    const int kr = k - r * kCap < kCap ? k - r * kCap : kCap;
    p.k = kr;
    p.cut_s = r == 0 ? nullptr : ws.cut_s;
    p.cut_i = r == 0 ? nullptr : ws.cut_i;
    if (r >= gemm_done_rounds && launch_gemm<0>(stream, p)) return -1;
    const bool more = r + 1 < rounds;
    hipLaunchKernelGGL(merge_topk_kernel,
                       dim3(static_cast<unsigned>((p.n_users + kWaves - 1) / kWaves)), dim3(kBlock),
                       0, stream, ws.cand_scores, ws.cand_ids, p.n_users, p.nchunks * 2 * kCap, kr,
                       has_y_range, y_lo, y_hi, topk_scores, topk_ids, k, r * kCap,
                       more ? ws.cut_s : nullptr, more ? ws.cut_i : nullptr);
    if (check_launch("merge_topk_kernel")) return -1;
  }
  return 0;
}

// shared body of collie_mf_eval_topk / collie_mf_eval_topk_static: `ws` is already carved
int eval_topk_impl(const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users,
                   int64_t item_begin, int64_t item_end, int64_t id_stride, int64_t id_offset,
                   int32_t k, int32_t has_y_range, float y_lo, float y_hi, float* topk_scores,
                   int64_t* topk_ids, const collie::EvalWs& ws, hipStream_t stream) {
  EvalParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.user_ids = user_ids;
  p.n_users = n_users;
  p.item_begin = item_begin;
  p.item_end = item_end;
  p.id_stride = id_stride;
  p.id_offset = id_offset;
  p.k = k;
  p.cand_scores = ws.cand_scores;
  p.cand_ids = ws.cand_ids;
  return topk_rounds(p, 0, k, has_y_range, y_lo, y_hi, topk_scores, topk_ids, ws, stream);
}

int check_topk_args(const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users,
                    int64_t item_begin, int64_t item_end, int64_t id_stride, int64_t id_offset,
                    int32_t k, const float* topk_scores, const int64_t* topk_ids) {
  if (validate_tables(t) || check_range(t, n_users, item_begin, item_end)) return -1;
  COLLIE_REQUIRE(user_ids && topk_scores && topk_ids, "eval_topk: NULL pointer");
  COLLIE_REQUIRE(k >= 1, "eval_topk: k must be >= 1 (got %d)", k);
  COLLIE_REQUIRE(id_stride >= 1 && id_offset >= 0 &&
                     (item_end - 1) * id_stride + id_offset < (int64_t{1} << 31),
                 "eval_topk: bad global id mapping");
  COLLIE_REQUIRE(k <= item_end - item_begin, "eval_topk: k (%d) exceeds the number of items", k);
  return 0;
}

// ---- the allocation-free variant's scratch --------------------------------------------------------
// Candidate lists of collie_mf_eval_topk_static live in this file-scope device array instead of a
// caller workspace.  In host code `scratch` is only a stand-in for the symbol: the per-device
// address comes from hipGetSymbolAddress (each GPU holds its own copy at its own address) and is
// what every kernel argument below uses.
constexpr int64_t kStaticMaxUsers = 1024;
constexpr size_t kStaticScratchBytes = (size_t{16} << 20) + (size_t{64} << 10);  // 1024 users x 64 chunks x 2 halves
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
void* g_scratch_ptr[kMaxDevices];  // per device, filled on the first call on that device
std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device (looked up once per device)
int static_scratch(void** out) {
  int dev = -1;
  COLLIE_CHECK_HIP(hipGetDevice(&dev));
  COLLIE_REQUIRE(dev >= 0 && dev < kMaxDevices, "eval_topk_static: device index %d out of range",
                 dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* ptr = nullptr;
    COLLIE_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    COLLIE_REQUIRE(ptr != nullptr, "eval_topk_static: hipGetSymbolAddress returned NULL");
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

}  // namespace

extern "C" int collie_mf_eval_topk(const collie_mf_tables* t, const int64_t* user_ids,
                                   int64_t n_users, int64_t item_begin, int64_t item_end,
                                   int64_t id_stride, int64_t id_offset, int32_t k,
                                   int32_t has_y_range, float y_lo, float y_hi,
                                   float* topk_scores, int64_t* topk_ids, void* workspace,
                                   size_t workspace_bytes, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_topk_args(t, user_ids, n_users, item_begin, item_end, id_stride, id_offset, k,
                      topk_scores, topk_ids))
    return -1;
  EvalWs ws = carve_eval_ws(workspace, n_users, item_end - item_begin);
  COLLIE_REQUIRE(workspace && workspace_bytes >= ws.bytes,
                 "eval workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  return eval_topk_impl(t, user_ids, n_users, item_begin, item_end, id_stride, id_offset, k,
                        has_y_range, y_lo, y_hi, topk_scores, topk_ids, ws, stream);
}

extern "C" int collie_mf_eval_topk_static(const collie_mf_tables* t, const int64_t* user_ids,
                                          int64_t n_users, int64_t item_begin, int64_t item_end,
                                          int64_t id_stride, int64_t id_offset, int32_t k,
                                          int32_t has_y_range, float y_lo, float y_hi,
                                          float* topk_scores, int64_t* topk_ids,
                                          collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_topk_args(t, user_ids, n_users, item_begin, item_end, id_stride, id_offset, k,
                      topk_scores, topk_ids))
    return -1;
  COLLIE_REQUIRE(n_users <= kStaticMaxUsers,
                 "eval_topk_static: at most %lld users per call (got %lld); use collie_mf_eval_topk "
                 "with a caller workspace for larger batches",
                 static_cast<long long>(kStaticMaxUsers), static_cast<long long>(n_users));
  void* base = nullptr;
  if (static_scratch(&base)) return -1;
  EvalWs ws = carve_eval_ws(base, n_users, item_end - item_begin);
  COLLIE_REQUIRE(ws.bytes <= kStaticScratchBytes,
                 "eval_topk_static: needs %zu bytes of scratch, the static buffer holds %zu",
                 ws.bytes, kStaticScratchBytes);
  // every candidate slot the merge reads is written by the scoring kernel of THIS call, so nothing
  // of the previous call's contents has to be cleared
  return eval_topk_impl(t, user_ids, n_users, item_begin, item_end, id_stride, id_offset, k,
                        has_y_range, y_lo, y_hi, topk_scores, topk_ids, ws, stream);
}

// ---- bf16 x 3 scoring pass (opt-in) ---------------------------------------------------------------
namespace {
constexpr int kBf16MaxK = 12;   // 16 candidates kept per user: at least 4 beyond k absorb the rounding

struct Bf16Ws {
  collie::EvalWs e;
  float* ms;       // [n_users, kCap] merged approximate scores
  int64_t* mi;     // [n_users, kCap] their ids
  float* rs;       // [n_users, kCap] exact scores
  int32_t* ri;
  size_t bytes;
};

Bf16Ws carve_bf16_ws(void* base, int64_t n_users, int64_t n_items_shard) {
  Bf16Ws w;
  w.e = carve_eval_ws(base, n_users, n_items_shard);
  Carver c(base);
  c.off = w.e.bytes;
  const size_t n = static_cast<size_t>(n_users) * kCap;
  w.ms = c.take<float>(n);
  w.mi = c.take<int64_t>(n);
  w.rs = c.take<float>(n);
  w.ri = c.take<int32_t>(n);
  w.bytes = align_up(c.off, 256);
  return w;
}
}  // namespace

extern "C" int32_t collie_mf_eval_bf16x3_ok(int32_t dim, int32_t k) {
  return (dim == 64 || dim == 128) && k >= 1 && k <= kBf16MaxK ? 1 : 0;
}

extern "C" int collie_mf_eval_split_bf16x3(const float* table, int64_t rows, int32_t dim,
                                           void* out, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(table && out && rows >= 1 && dim >= 1, "split_bf16x3: bad arguments");
  int64_t blocks = (rows * dim + kBlock - 1) / kBlock;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                     stream, table, rows, dim, static_cast<uint16_t*>(out));
  return check_launch("split_bf16x3_kernel");
}

extern "C" size_t collie_mf_eval_bf16x3_workspace_bytes(int64_t n_users, int64_t n_items_shard) {
  if (n_users < 1 || n_items_shard < 1) return 0;
  return carve_bf16_ws(nullptr, n_users, n_items_shard).bytes;
}

extern "C" int collie_mf_eval_topk_bf16x3(const collie_mf_tables* t, const void* item_split,
                                          const int64_t* user_ids, int64_t n_users,
                                          int64_t item_begin, int64_t item_end, int64_t id_stride,
                                          int64_t id_offset, int32_t k, int32_t has_y_range,
                                          float y_lo, float y_hi, float* topk_scores,
                                          int64_t* topk_ids, void* workspace,
                                          size_t workspace_bytes, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (check_topk_args(t, user_ids, n_users, item_begin, item_end, id_stride, id_offset, k,
                      topk_scores, topk_ids))
    return -1;
  COLLIE_REQUIRE(item_split != nullptr, "eval_topk_bf16x3: NULL split table");
  COLLIE_REQUIRE(collie_mf_eval_bf16x3_ok(t->dim, k),
                 "eval_topk_bf16x3: embedding_dim must be 64 or 128 and k <= %d (use collie_mf_eval_topk)",
                 kBf16MaxK);
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  COLLIE_REQUIRE(al16(t->user_emb) && al16(item_split), "eval_topk_bf16x3: unaligned tables");
  Bf16Ws ws = carve_bf16_ws(workspace, n_users, item_end - item_begin);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "eval workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  const int kk = k + 6 < kCap ? k + 6 : kCap;     // candidates kept beyond k
  Bf16Params bp;
  memset(&bp, 0, sizeof(bp));
  EvalParams& p = bp.e;
  p.t = *t;
  p.user_ids = user_ids;
  p.n_users = n_users;
  p.item_begin = item_begin;
  p.item_end = item_end;
  p.id_stride = id_stride;
  p.id_offset = id_offset;
  p.k = kk;
  p.cand_scores = ws.e.cand_scores;
  p.cand_ids = ws.e.cand_ids;
  p.D8 = d8_bucket(t->dim);
  bp.item_split = static_cast<const uint16_t*>(item_split);
  const EvalGeom g = eval_geom(n_users, item_end - item_begin, kUsersPerBlockDma);
  p.nchunks = g.nchunks;
  p.tiles_per_chunk = g.tiles_per_chunk;
  dim3 grid(static_cast<unsigned>((n_users + kUsersPerBlockDma - 1) / kUsersPerBlockDma),
            static_cast<unsigned>(g.nchunks));
  if (t->dim == 128)
    hipLaunchKernelGGL((eval_gemm_bf16x3_kernel<8>), grid, dim3(kBlockDma), 0, stream, bp);
  else
    hipLaunchKernelGGL((eval_gemm_bf16x3_kernel<4>), grid, dim3(kBlockDma), 0, stream, bp);
  if (check_launch("eval_gemm_bf16x3_kernel")) return -1;
  const dim3 mgrid(static_cast<unsigned>((n_users + kWaves - 1) / kWaves));
  // 16 per slot -> kk per user in the approximate order ...
  hipLaunchKernelGGL(merge_topk_kernel, mgrid, dim3(kBlock), 0, stream, ws.e.cand_scores,
                     ws.e.cand_ids, n_users, p.nchunks * 2 * kCap, kk, 0, 0.0f, 0.0f, ws.ms, ws.mi, kk,
                     0, static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr));
  if (check_launch("merge_topk_kernel")) return -1;
  // ... exact scores for those ...
  hipLaunchKernelGGL(rescore_kernel, dim3(static_cast<unsigned>((n_users * kk + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, *t, user_ids, n_users, kk, p.D8, id_stride, id_offset,
                     ws.mi, ws.rs, ws.ri);
  if (check_launch("rescore_kernel")) return -1;
  // ... and the k best of them in the exact order
  hipLaunchKernelGGL(merge_topk_kernel, mgrid, dim3(kBlock), 0, stream, ws.rs, ws.ri, n_users, kk, k,
                     has_y_range, y_lo, y_hi, topk_scores, topk_ids, k, 0,
                     static_cast<float*>(nullptr), static_cast<int32_t*>(nullptr));
  return check_launch("merge_topk_kernel");
}

extern "C" int collie_mf_eval_merge_topk(const float* cand_scores, const int32_t* cand_ids,
                                         int64_t n_users, int32_t nslots, int32_t slot_len,
                                         int32_t k, float* topk_scores, int64_t* topk_ids,
                                         collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(cand_scores && cand_ids && topk_scores && topk_ids, "merge_topk: NULL pointer");
  COLLIE_REQUIRE(n_users >= 1 && nslots >= 1 && slot_len >= 1 && k >= 1 &&
                     static_cast<int64_t>(nslots) * slot_len < (int64_t{1} << 30) &&
                     k <= static_cast<int64_t>(nslots) * slot_len,
                 "merge_topk: bad sizes");
  hipLaunchKernelGGL(merge_topk_kernel, dim3(static_cast<unsigned>((n_users + kWaves - 1) / kWaves)),
                     dim3(kBlock), 0, stream, cand_scores, cand_ids, n_users, nslots * slot_len, k, 0, 0.0f,
                     0.0f, topk_scores, topk_ids, k, 0, static_cast<float*>(nullptr),
                     static_cast<int32_t*>(nullptr));
  return check_launch("merge_topk_kernel");
}

extern "C" int collie_mf_eval_pos_scores(const collie_mf_tables* t, const int64_t* user_ids,
                                         int64_t n_users, const int64_t* indptr,
                                         const int64_t* indices, int64_t nnz, int32_t has_y_range,
                                         float y_lo, float y_hi, int32_t compute_scores,
                                         int64_t id_stride, int64_t id_offset, float* pos_scores,
                                         float* thr_lo, float* thr_hi, float* best_score,
                                         int64_t* best_id, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t)) return -1;
  COLLIE_REQUIRE(user_ids && indptr && indices && pos_scores, "eval_pos_scores: NULL pointer");
  COLLIE_REQUIRE(n_users >= 1 && nnz >= 0 && id_stride >= 1 && id_offset >= 0,
                 "eval_pos_scores: bad sizes");
  COLLIE_REQUIRE(t->dim <= 256, "embedding_dim > 256 is not supported by the MFMA evaluator");
  if (nnz > 0) {
    int64_t blocks = (nnz + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pos_scores_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, *t, user_ids, n_users, indptr, indices, d8_bucket(t->dim),
                       has_y_range, y_lo, y_hi, compute_scores, id_stride, id_offset, pos_scores,
                       thr_lo, thr_hi);
    if (check_launch("pos_scores_kernel")) return -2;
  }
  if (best_score && best_id) {
    hipLaunchKernelGGL(best_pos_kernel, dim3(static_cast<unsigned>((n_users + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, indptr, indices, pos_scores, n_users, best_score,
                       best_id);
    return check_launch("best_pos_kernel");
  }
  return 0;
}

extern "C" int collie_mf_eval_rank_counts(const collie_mf_tables* t, const int64_t* user_ids,
                                          int64_t n_users, const int64_t* indptr,
                                          const float* thr_lo, const float* thr_hi,
                                          const float* best_score, const int64_t* best_id,
                                          int64_t item_begin, int64_t item_end, int64_t id_stride,
                                          int64_t id_offset, uint64_t* counts,
                                          collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || check_range(t, n_users, item_begin, item_end)) return -1;
  COLLIE_REQUIRE(user_ids && indptr && thr_lo && thr_hi && best_score && best_id && counts,
                 "eval_rank_counts: NULL pointer");
  COLLIE_REQUIRE(id_stride >= 1 && id_offset >= 0 &&
                     (item_end - 1) * id_stride + id_offset < (int64_t{1} << 31),
                 "eval_rank_counts: bad global id mapping");
  EvalParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.user_ids = user_ids;
  p.n_users = n_users;
  p.item_begin = item_begin;
  p.item_end = item_end;
  p.id_stride = id_stride;
  p.id_offset = id_offset;
  p.k = 1;
  p.indptr = indptr;
  p.thr_lo = thr_lo;
  p.thr_hi = thr_hi;
  p.best_score = best_score;
  p.best_id = best_id;
  p.counts = reinterpret_cast<unsigned long long*>(counts);
  return launch_gemm<1>(stream, p);
}

extern "C" int collie_mf_eval_topk_rank_counts(
    const collie_mf_tables* t, const int64_t* user_ids, int64_t n_users, int64_t item_begin,
    int64_t item_end, int64_t id_stride, int64_t id_offset, int32_t k, int32_t has_y_range,
    float y_lo, float y_hi, const int64_t* indptr, const float* thr_lo, const float* thr_hi,
    const float* best_score, const int64_t* best_id, float* topk_scores, int64_t* topk_ids,
    uint64_t* counts, void* workspace, size_t workspace_bytes, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || check_range(t, n_users, item_begin, item_end)) return -1;
  COLLIE_REQUIRE(user_ids && topk_scores && topk_ids && indptr && thr_lo && thr_hi && best_score &&
                     best_id && counts, "eval_topk_rank_counts: NULL pointer");
  COLLIE_REQUIRE(k >= 1, "eval_topk: k must be >= 1 (got %d)", k);
  COLLIE_REQUIRE(id_stride >= 1 && id_offset >= 0 &&
                     (item_end - 1) * id_stride + id_offset < (int64_t{1} << 31),
                 "eval_topk: bad global id mapping");
  COLLIE_REQUIRE(k <= item_end - item_begin, "eval_topk: k (%d) exceeds the number of items", k);
  EvalWs ws = carve_eval_ws(workspace, n_users, item_end - item_begin);
  COLLIE_REQUIRE(workspace && workspace_bytes >= ws.bytes,
                 "eval workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  EvalParams p;
  memset(&p, 0, sizeof(p));
  p.t = *t;
  p.user_ids = user_ids;
  p.n_users = n_users;
  p.item_begin = item_begin;
  p.item_end = item_end;
  p.id_stride = id_stride;
  p.id_offset = id_offset;
  p.k = k < kCap ? k : kCap;
  p.cand_scores = ws.cand_scores;
  p.cand_ids = ws.cand_ids;
  p.indptr = indptr;
  p.thr_lo = thr_lo;
  p.thr_hi = thr_hi;
  p.best_score = best_score;
  p.best_id = best_id;
  p.counts = reinterpret_cast<unsigned long long*>(counts);
  int gemm_done = 0;
  if (dma_ok(&p.t)) {
    if (launch_gemm<2>(stream, p)) return -1;     // one pass: both epilogues (top-k round 0)
    gemm_done = 1;
  } else {
    if (launch_gemm<1>(stream, p)) return -1;     // register-staged kernel: separate passes
  }
  return topk_rounds(p, gemm_done, k, has_y_range, y_lo, y_hi, topk_scores, topk_ids, ws, stream);
}

extern "C" int collie_mf_eval_metrics(const int64_t* topk_ids, int32_t k, int64_t n_users,
                                      int64_t n_items, const int64_t* indptr,
                                      const int64_t* indices, const float* pos_scores,
                                      const uint64_t* counts, int32_t has_y_range, float y_lo,
                                      float y_hi, double* ap, float* rr, double* auc,
                                      collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(indptr && indices, "eval_metrics: NULL CSR");
  COLLIE_REQUIRE(n_users >= 1, "eval_metrics: empty user batch");
  if (ap) COLLIE_REQUIRE(topk_ids && k >= 1, "eval_metrics: AP needs the top-k ids");
  if (rr || auc) COLLIE_REQUIRE(counts && pos_scores, "eval_metrics: RR / AUC need the rank counts");
  hipLaunchKernelGGL(metrics_kernel, dim3(static_cast<unsigned>((n_users + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, topk_ids, k, n_users, n_items, indptr, indices,
                     pos_scores, reinterpret_cast<const unsigned long long*>(counts), has_y_range,
                     y_lo, y_hi, ap, rr, auc);
  return check_launch("metrics_kernel");
}
