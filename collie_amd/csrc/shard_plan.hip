// Row-sharded training, requester side: which table rows does this rank's batch look up, who owns
// them, and what is each sample's row number in the compact tables the owners will send back?
//
// Round 3: the cost of a plan is proportional to the LOOKUPS of the batch, not to the size of the
// tables (the stamp-table planner of rounds 1-2 walked a cell per table row -- 110 M cells for the
// 100 M x 10 M shape on every rank, whatever the number of ranks).  The (2 + K) B ids are turned
// into route-major keys -- route = (owner, kind), owner = id mod G, kind = user | item, position
// inside a route = id div G -- and coalesced by the stable radix sort the optimizers use
// (rowsort.hip): a segment of equal keys is one distinct row, the segment index IS the compact row
// number (owner, kind, id ascending: the order in which rows are requested from / returned by the
// owners), and the entry -> segment map rewrites the batch in compact numbers.
//
// When the batch is dense in the planned key space (at most kDirectKeysPerLookup keys per lookup:
// the 11.5 M item lookups of a 1 M batch over 10 M items, say) the same numbering comes cheaper
// without a sort ("direct" route): one byte mark per looked-up key (plain stores), the marks folded
// into 64-key bit words with a running count (two streaming passes over the key space, 1 byte + 16
// bytes per 64 keys), the distinct rows written out in key order from the words, and every element
// of the batch numbered by ONE 16-byte read: words[key / 64].count + popcount(bits below the key).
// Same outputs bit for bit (tests/test_gpu_sharded.py runs both routes against plan_torch).
//
// Wire layout (one all-to-all each way, embedding rows and their bias values in ONE buffer): every
// exchange buffer is a matrix of dim-float rows; the block of one peer holds its c embedding rows
// followed by ceil(c / dim) rows that carry the c bias values (gradient direction: then as many rows
// again with the c owner-side row numbers as int32).  The requester uses the received lookup buffer
// AS its compact embedding table, so compact row numbers skip the bias rows between two peers'
// blocks: compact = segment index + (bias rows of the peers before the row's owner).
#include "rowsort.h"
#include "shard_wire.h"

namespace collie {

int validate_batch(const collie_mf_batch* b);

namespace {

constexpr int kBlock = 256;

// 16-byte accesses that leave no copy in the caches (tables far beyond them: every row the owners
// serve is read once here and its copy is read once by the forward kernel)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, const float4& v) {
  nt_f4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<nt_f4*>(p));
}

struct Routes {
  int world;
  int64_t num_users, num_items;
  int64_t key_base[2 * kMaxWorld + 1];   // first key of a route (routes are dense in key space)
  int64_t row_base[2 * kMaxWorld];       // owner-side row offset of the route's kind (0 | U_loc(owner))
};

// plan_users == false ("users are local"): the user routes are empty -- the key space is the items'
void make_routes(Routes& r, int64_t U, int64_t N, int world, bool plan_users) {
  r.world = world;
  r.num_users = U;
  r.num_items = N;
  int64_t key = 0;
  for (int o = 0; o < world; ++o) {
    const int64_t u_loc = (U - o + world - 1) / world;
    const int64_t n_loc = (N - o + world - 1) / world;
    for (int kind = 0; kind < 2; ++kind) {
      const int q = 2 * o + kind;
      r.row_base[q] = kind ? u_loc : 0;
      r.key_base[q] = key;
      key += kind ? n_loc : (plan_users ? u_loc : 0);
    }
  }
  r.key_base[2 * world] = key;   // == (U) + N
}

// ---- which route plans a batch -------------------------------------------------------------------
constexpr int64_t kDirectKeysPerLookup = 64;
constexpr int kWordsPerBlock = 256;     // 64-key words per workgroup of the direct route
constexpr int kScanBlock = 1024;

bool use_direct(int64_t nkeys, int64_t planned) {
  return planned > 0 && nkeys <= kDirectKeysPerLookup * planned;
}

struct DirectWs {
  uint8_t* marks;          // [nwords * 64] one byte per key: looked up in this batch
  uint4* words;            // [nwords] {bits 0-31, bits 32-63, distinct keys before the word, -}
  uint32_t* blk;           // [nblocks + 1] distinct keys of a workgroup's words, then their prefix
  int64_t* shift;          // [world]
  int64_t nwords, nblocks;
  size_t bytes;
};

DirectWs carve_direct_ws(void* base, int64_t nkeys, int world) {
  Carver c(base);
  DirectWs w;
  w.nwords = (nkeys + 63) / 64;
  w.nblocks = (w.nwords + kWordsPerBlock - 1) / kWordsPerBlock;
  w.marks = c.take<uint8_t>(w.nblocks * kWordsPerBlock * 64);
  w.words = c.take<uint4>(w.nblocks * kWordsPerBlock);
  w.blk = c.take<uint32_t>(w.nblocks + 1);
  w.shift = c.take<int64_t>(world);
  w.bytes = align_up(c.off, 256);
  return w;
}

struct PlanWs {
  SortWs sort;
  uint32_t* seg_of;        // [m] compact row of every sorted entry
  int64_t* shift;          // [world] compact row = segment index + shift[owner]
  size_t bytes;
};

PlanWs carve_plan_ws(void* base, int64_t m, int world) {
  Carver c(base);
  PlanWs w;
  carve_sort_ws(c, w.sort, m, 1);
  w.seg_of = c.take<uint32_t>(m);
  w.shift = c.take<int64_t>(world);
  w.bytes = align_up(c.off, 256);
  return w;
}

// element e of the batch: [0,B) users, [B,2B) positives, [2B, 2B+BK) negatives (b major)
__device__ __forceinline__ int64_t batch_id(const collie_mf_batch& b, int64_t e, int& kind) {
  const int64_t B = b.batch;
  if (e < B) {
    kind = 0;
    return load_idx(b.users, b.users_dtype, e);
  }
  kind = 1;
  if (e < 2 * B) return load_idx(b.pos_items, b.pos_dtype, e - B);
  const int64_t j = e - 2 * B;
  const int64_t row = j / b.num_neg, k = j - row * b.num_neg;
  return load_idx(b.neg_items, b.neg_dtype, row * b.neg_stride_b + k * b.neg_stride_k);
}

// local_rank >= 0 ("users are local"): every user of the batch is owned by this rank -- its row is
// read and updated in place, so only the item ids are planned (entries [B, (2 + K) B) of the batch,
// stored from 0) and compact_users is simply the local row id / world; a user owned by another rank
// is reported like an id outside the tables.
// DIRECT: instead of a (key, element) entry for the sort, the key's byte in `marks` is set.
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void plan_keys_kernel(const collie_mf_batch b, const Routes r,
                                                           int local_rank,
                                                           uint32_t* __restrict__ ekeys,
                                                           uint32_t* __restrict__ ecodes,
                                                           uint8_t* __restrict__ marks,
                                                           int32_t* __restrict__ cu,
                                                           int64_t* __restrict__ bad_flag) {
  const int64_t total = b.batch * (2 + b.num_neg);
  const int64_t skip = local_rank >= 0 ? b.batch : 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * kBlock) {
    int kind;
    int64_t id = batch_id(b, e, kind);
    const int64_t rows = kind ? r.num_items : r.num_users;
    if (id < 0 || id >= rows) {   // reported; the entry then stands for the nearest valid row
      *bad_flag = 1;
      id = id < 0 ? 0 : rows - 1;
    }
    // (ids are below 2^31 here: 32-bit division)
    const uint32_t uid = static_cast<uint32_t>(id), uworld = static_cast<uint32_t>(r.world);
    const int owner = static_cast<int>(uid % uworld);
    const int64_t local = uid / uworld;
    if (e < skip) {
      if (owner != local_rank) *bad_flag = 1;
      cu[e] = owner == local_rank ? static_cast<int32_t>(local) : 0;
      continue;
    }
    const int64_t key = r.key_base[2 * owner + kind] + local;
    if (DIRECT) {
      marks[key] = 1;   // (plain store; reading first -- "already marked?" -- measured 13 % slower,
                        // a bitmap with atomicOr 30-70 % slower)
    } else {
      ekeys[e - skip] = static_cast<uint32_t>(key);
      ecodes[e - skip] = static_cast<uint32_t>(e);
    }
  }
}

__device__ __forceinline__ int route_of(const Routes& r, uint32_t key) {
  int q = 0;
  while (q + 1 < 2 * r.world && static_cast<int64_t>(key) >= r.key_base[q + 1]) ++q;
  return q;
}

// one thread per route boundary: where the route starts among the segments (lower bound of its
// first key in the sorted keys); thread 0 then turns the boundaries into rows per route and into the
// shift of every owner's compact numbers (the bias rows of the blocks before it, see above).
__global__ __launch_bounds__(64) void plan_layout_kernel(const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ seg_of,
                                                         const uint32_t* __restrict__ nseg, int64_t m,
                                                         const Routes r, int dim,
                                                         int64_t* __restrict__ route_counts,
                                                         int64_t* __restrict__ shift) {
  __shared__ int64_t first[2 * kMaxWorld + 1];
  const int nroutes = 2 * r.world;
  if (static_cast<int>(threadIdx.x) <= nroutes) {
    const int64_t want = r.key_base[threadIdx.x];
    int64_t lo = 0, hi = m;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (static_cast<int64_t>(keys[mid]) < want) lo = mid + 1;
      else hi = mid;
    }
    first[threadIdx.x] = lo < m ? static_cast<int64_t>(seg_of[lo]) : static_cast<int64_t>(nseg[0]);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int64_t pad = 0;
  for (int o = 0; o < r.world; ++o) {
    shift[o] = pad;
    const int64_t rows = first[2 * o + 2] - first[2 * o];
    pad += (rows + dim - 1) / dim;
  }
  for (int q = 0; q < nroutes; ++q) route_counts[q] = first[q + 1] - first[q];
  route_counts[nroutes] = static_cast<int64_t>(nseg[0]);
}

// one thread per sorted entry: the entry's element of the batch gets its compact row number; a
// segment's first entry also writes the row to request.
__global__ __launch_bounds__(kBlock) void plan_emit_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
    const uint32_t* __restrict__ seg_of, int64_t m, int64_t B, const Routes r,
    const int64_t* __restrict__ shift, int32_t* __restrict__ cu, int32_t* __restrict__ cp,
    int64_t* __restrict__ cn, int64_t* __restrict__ send_rows) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j >= m) return;
  const uint32_t key = keys[j];
  const uint32_t seg = seg_of[j];
  const int64_t e = vals[j];
  const int q = route_of(r, key);
  const int64_t compact = static_cast<int64_t>(seg) + shift[q >> 1];
  if (e < B) cu[e] = static_cast<int32_t>(compact);
  else if (e < 2 * B) cp[e - B] = static_cast<int32_t>(compact);
  else cn[e - 2 * B] = compact;
  if (j == 0 || keys[j - 1] != key)
    send_rows[seg] = r.row_base[q] + (static_cast<int64_t>(key) - r.key_base[q]);
}

// ---- direct route --------------------------------------------------------------------------------
// one thread per 64-key word: the 64 mark bytes become a bit mask; the word keeps its own count
// until plan_rows_kernel replaces it by the running count; the workgroup's total goes to blk.
__global__ __launch_bounds__(kWordsPerBlock) void plan_bits_kernel(const uint8_t* __restrict__ marks,
                                                                 uint4* __restrict__ words,
                                                                 uint32_t* __restrict__ blk) {
  __shared__ uint32_t wsum[kWordsPerBlock / 64];
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kWordsPerBlock + threadIdx.x;
  const uint4* src = reinterpret_cast<const uint4*>(marks + w * 64);
  uint32_t half[2] = {0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint4 v = src[q];
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int byte = 0; byte < 4; ++byte)
        if ((x[i] >> (8 * byte)) & 0xffu) half[q >> 1] |= 1u << ((q & 1) * 16 + i * 4 + byte);
  }
  const uint32_t cnt = __popc(half[0]) + __popc(half[1]);
  words[w] = make_uint4(half[0], half[1], cnt, 0u);
  uint32_t s = cnt;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x % 64 == 0) wsum[threadIdx.x / 64] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int i = 0; i < kWordsPerBlock / 64; ++i) t += wsum[i];
    blk[blockIdx.x] = t;
  }
}

// one workgroup: blk[i] <- distinct keys before workgroup i's words, blk[n] <- all distinct keys
__global__ __launch_bounds__(kScanBlock) void plan_scan_kernel(uint32_t* __restrict__ blk, int64_t n) {
  __shared__ uint32_t part[kScanBlock / 64];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid % 64, wv = tid / 64;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += kScanBlock) {
    const int64_t i = base + tid;
    const uint32_t v = i < n ? blk[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) part[wv] = inc;
    __syncthreads();
    uint32_t before = carry_s;
    for (int j = 0; j < wv; ++j) before += part[j];
    if (i < n) blk[i] = before + inc - v;
    __syncthreads();
    if (tid == kScanBlock - 1) carry_s = before + inc;
    __syncthreads();
  }
  if (tid == 0) blk[n] = carry_s;
}

// one workgroup per kWordsPerBlock words: running counts into the words, and the workgroup's distinct
// keys -- gathered in key order in LDS first, so that the rows to request leave as coalesced stores
__global__ __launch_bounds__(kWordsPerBlock) void plan_rows_kernel(uint4* __restrict__ words,
                                                                 const uint32_t* __restrict__ blk,
                                                                 const Routes r,
                                                                 int64_t* __restrict__ send_rows) {
  __shared__ uint16_t lkey[kWordsPerBlock * 64];
  __shared__ uint32_t wsum[kWordsPerBlock / 64];
  const int tid = threadIdx.x, lane = tid % 64, wv = tid / 64;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * kWordsPerBlock + tid;
  uint4 word = words[w];
  const uint32_t cnt = word.z;
  uint32_t inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t local = inc - cnt;
  uint32_t total = 0;
  for (int j = 0; j < kWordsPerBlock / 64; ++j) {
    if (j < wv) local += wsum[j];
    total += wsum[j];
  }
  const uint32_t before = blk[blockIdx.x];
  word.z = before + local;
  words[w] = word;
  uint64_t bits = (static_cast<uint64_t>(word.y) << 32) | word.x;
  uint32_t at = local;
  while (bits) {
    const int bit = __ffsll(static_cast<unsigned long long>(bits)) - 1;
    bits &= bits - 1;
    lkey[at++] = static_cast<uint16_t>(tid * 64 + bit);
  }
  __syncthreads();
  const int64_t key0 = static_cast<int64_t>(blockIdx.x) * kWordsPerBlock * 64;
  for (uint32_t j = tid; j < total; j += kWordsPerBlock) {
    const int64_t key = key0 + lkey[j];
    const int q = route_of(r, static_cast<uint32_t>(key));
    send_rows[static_cast<int64_t>(before) + j] = r.row_base[q] + (key - r.key_base[q]);
  }
}

__device__ __forceinline__ int64_t rank_of_key(const uint4* __restrict__ words, int64_t key) {
  const uint4 w = words[key >> 6];
  const uint64_t bits = (static_cast<uint64_t>(w.y) << 32) | w.x;
  const uint64_t below = bits & ((uint64_t{1} << (key & 63)) - 1);
  return static_cast<int64_t>(w.z) + __popcll(static_cast<unsigned long long>(below));
}

// plan_layout_kernel of the direct route: a route starts at the rank of its first key
__global__ __launch_bounds__(64) void plan_direct_layout_kernel(const uint4* __restrict__ words,
                                                                const uint32_t* __restrict__ blk,
                                                                int64_t nblocks, const Routes r, int dim,
                                                                int64_t* __restrict__ route_counts,
                                                                int64_t* __restrict__ shift) {
  __shared__ int64_t first[2 * kMaxWorld + 1];
  const int nroutes = 2 * r.world;
  const int64_t nkeys = r.key_base[nroutes];
  const int64_t total = blk[nblocks];
  if (static_cast<int>(threadIdx.x) <= nroutes) {
    const int64_t k = r.key_base[threadIdx.x];
    first[threadIdx.x] = k < nkeys ? rank_of_key(words, k) : total;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int64_t pad = 0;
  for (int o = 0; o < r.world; ++o) {
    shift[o] = pad;
    const int64_t rows = first[2 * o + 2] - first[2 * o];
    pad += (rows + dim - 1) / dim;
  }
  for (int q = 0; q < nroutes; ++q) route_counts[q] = first[q + 1] - first[q];
  route_counts[nroutes] = total;
}

// one thread per planned element of the batch, in batch order (coalesced stores)
__global__ __launch_bounds__(kBlock) void plan_direct_emit_kernel(
    const collie_mf_batch b, const Routes r, int local_rank, const uint4* __restrict__ words,
    const int64_t* __restrict__ shift, int32_t* __restrict__ cu, int32_t* __restrict__ cp,
    int64_t* __restrict__ cn) {
  const int64_t B = b.batch;
  const int64_t total = B * (2 + b.num_neg);
  const int64_t skip = local_rank >= 0 ? B : 0;
  for (int64_t e = skip + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * kBlock) {
    int kind;
    int64_t id = batch_id(b, e, kind);
    const int64_t rows = kind ? r.num_items : r.num_users;
    if (id < 0 || id >= rows) id = id < 0 ? 0 : rows - 1;   // as plan_keys_kernel (reported there)
    const uint32_t uid = static_cast<uint32_t>(id), uworld = static_cast<uint32_t>(r.world);
    const int owner = static_cast<int>(uid % uworld);
    const int64_t key = r.key_base[2 * owner + kind] + uid / uworld;
    const int64_t compact = rank_of_key(words, key) + shift[owner];
    if (e < B) cu[e] = static_cast<int32_t>(compact);
    else if (e < 2 * B) cp[e - B] = static_cast<int32_t>(compact);
    else cn[e - 2 * B] = compact;
  }
}

// owner side of the lookup exchange: the rows the peers requested, written as the peers' blocks of the
// lookup buffer (embedding rows, then the bias values).  One 16-lane group per requested row (16-byte
// pieces when the rows allow it).  Item rows (rows at or behind item_base) are stamped "looked up in
// this step": the owner-side update finds the looked-up rows that received no gradient through these
// stamps (SparseAdam moves them too).
template <int VEC, bool STREAM>
__global__ __launch_bounds__(kBlock) void serve_rows_kernel(
    const float* __restrict__ emb, const float* __restrict__ bias, int64_t num_rows, int dim,
    const int64_t* __restrict__ rows, const WireBlocks w, float* __restrict__ out,
    uint8_t* __restrict__ touch, int64_t item_base, int tag, int32_t* __restrict__ status) {
  constexpr int LPR = 16;
  constexpr int RIF = 4;   // rows in flight per group: all their loads are issued before the first store
  const int lig = threadIdx.x % LPR;
  const int64_t m = w.first[w.world];
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * (kBlock / LPR);
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * (kBlock / LPR) + threadIdx.x / LPR;
  for (int64_t j0 = g0 * RIF; j0 < m; j0 += ngroups * RIF) {
    int64_t r[RIF];
    bool ok[RIF];
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      r[q] = j0 + q < m ? rows[j0 + q] : -1;
      ok[q] = r[q] >= 0 && r[q] < num_rows;
      if (j0 + q < m && !ok[q] && lig == 0 && status) atomicOr(status, 1);
    }
    if (VEC == 4 && dim <= 2 * 4 * LPR) {      // up to 128 floats: two 16-byte pieces per lane and row
      float4 v[RIF][2];
      float bv[RIF];
#pragma unroll
      for (int q = 0; q < RIF; ++q) {
        const float4* src = reinterpret_cast<const float4*>(emb + (ok[q] ? r[q] : 0) * dim);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v[q][h] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[q] && lig + h * LPR < dim / 4)
            v[q][h] = STREAM ? nt_load4(src + lig + h * LPR) : src[lig + h * LPR];
        }
        bv[q] = ok[q] && lig == 0 ? bias[r[q]] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < RIF; ++q) {
        const int64_t j = j0 + q;
        if (j >= m) break;
        const int p = block_of(w, j);
        const int64_t jj = j - w.first[p];
        float4* dst = reinterpret_cast<float4*>(out + (w.row0[p] + jj) * dim);
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (lig + h * LPR < dim / 4) {
            if (STREAM) nt_store4(dst + lig + h * LPR, v[q][h]);
            else dst[lig + h * LPR] = v[q][h];
          }
        if (lig == 0) out[(w.row0[p] + (w.first[p + 1] - w.first[p])) * dim + jj] = bv[q];
      }
    } else {
      for (int q = 0; q < RIF; ++q) {
        const int64_t j = j0 + q;
        if (j >= m) break;
        const int p = block_of(w, j);
        const int64_t jj = j - w.first[p];
        const float* src = emb + (ok[q] ? r[q] : 0) * dim;
        float* dst = out + (w.row0[p] + jj) * dim;
        if (VEC == 4) {
          for (int c = lig; c < dim / 4; c += LPR) {
            float4 x = {0.f, 0.f, 0.f, 0.f};
            if (ok[q]) x = reinterpret_cast<const float4*>(src)[c];
            reinterpret_cast<float4*>(dst)[c] = x;
          }
        } else {
          for (int c = lig; c < dim; c += LPR) dst[c] = ok[q] ? src[c] : 0.0f;
        }
        if (lig == 0)
          out[(w.row0[p] + (w.first[p + 1] - w.first[p])) * dim + jj] = ok[q] ? bias[r[q]] : 0.0f;
      }
    }
    if (touch && lig == 0) {
#pragma unroll
      for (int q = 0; q < RIF; ++q)
        if (ok[q] && r[q] >= item_base) {
          uint8_t* cell = touch + (r[q] - item_base);
          if (*cell != static_cast<uint8_t>(tag)) *cell = static_cast<uint8_t>(tag);
        }
    }
  }
}

// requester side: the bias values of the received lookup buffer as an array indexed like its rows
// (compact row numbers), which is what the fused forward kernel reads.
__global__ __launch_bounds__(kBlock) void unpack_bias_kernel(const float* __restrict__ buf, int dim,
                                                             const WireBlocks w,
                                                             float* __restrict__ bias_out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (j >= w.first[w.world]) return;
  const int p = block_of(w, j);
  const int64_t jj = j - w.first[p];
  bias_out[w.row0[p] + jj] = buf[(w.row0[p] + (w.first[p + 1] - w.first[p])) * dim + jj];
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int64_t collie_shard_wire_rows(const int64_t* counts, int32_t world, int32_t dim,
                                          int32_t gradient) {
  WireBlocks w;
  if (dim < 1 || make_blocks(w, counts, world, dim, gradient ? 2 : 1)) return -1;
  return w.row0[world];
}

extern "C" int collie_shard_serve_rows(const float* emb, const float* bias, int64_t num_rows,
                                       int32_t dim, const int64_t* rows, const int64_t* peer_counts,
                                       int32_t world, float* out, int64_t out_rows,
                                       uint8_t* item_touch, int64_t item_base, int32_t step_id,
                                       int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(emb && bias && num_rows >= 1 && dim >= 1, "serve_rows: bad table");
  WireBlocks w;
  if (make_blocks(w, peer_counts, world, dim, 1)) return -1;
  int tag = 0;
  if (item_touch) {
    COLLIE_REQUIRE(item_base >= 0 && item_base <= num_rows && step_id >= 1,
                   "serve_rows: bad item_base / step_id");
    tag = touch_tag(step_id);
    // the 8-bit tags start a new cycle: forget the previous cycle's stamps (train_ws.h)
    if (step_id % kTouchCycle == 0 && num_rows > item_base)
      COLLIE_CHECK_HIP(hipMemsetAsync(item_touch, 0, static_cast<size_t>(num_rows - item_base), stream));
  }
  const int64_t m = w.first[world];
  if (m == 0) return 0;
  COLLIE_REQUIRE(rows && out, "serve_rows: NULL pointer");
  COLLIE_REQUIRE(out_rows >= w.row0[world], "serve_rows: out holds %lld rows, need %lld",
                 static_cast<long long>(out_rows), static_cast<long long>(w.row0[world]));
  int64_t blocks = (m + 4 * (kBlock / 16) - 1) / (4 * (kBlock / 16));   // four rows per 16-lane group
  if (blocks > 65536) blocks = 65536;
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  const bool stream_rows = num_rows * dim * int64_t{12} > (int64_t{1} << 30);   // (as mf_update.hip)
  if (dim % 4 == 0 && al16(emb) && al16(out) && stream_rows)
    hipLaunchKernelGGL((serve_rows_kernel<4, true>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock),
                       0, stream, emb, bias, num_rows, dim, rows, w, out, item_touch, item_base, tag,
                       status);
  else if (dim % 4 == 0 && al16(emb) && al16(out))
    hipLaunchKernelGGL((serve_rows_kernel<4, false>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock),
                       0, stream, emb, bias, num_rows, dim, rows, w, out, item_touch, item_base, tag,
                       status);
  else
    hipLaunchKernelGGL((serve_rows_kernel<1, false>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock),
                       0, stream, emb, bias, num_rows, dim, rows, w, out, item_touch, item_base, tag,
                       status);
  return check_launch("serve_rows_kernel");
}

extern "C" int collie_shard_unpack_bias(const float* buf, int64_t buf_rows, int32_t dim,
                                        const int64_t* peer_counts, int32_t world, float* bias_out,
                                        collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(dim >= 1, "unpack_bias: bad dim");
  WireBlocks w;
  if (make_blocks(w, peer_counts, world, dim, 1)) return -1;
  const int64_t m = w.first[world];
  if (m == 0) return 0;
  COLLIE_REQUIRE(buf && bias_out, "unpack_bias: NULL pointer");
  COLLIE_REQUIRE(buf_rows >= w.row0[world], "unpack_bias: buffer holds %lld rows, need %lld",
                 static_cast<long long>(buf_rows), static_cast<long long>(w.row0[world]));
  hipLaunchKernelGGL(unpack_bias_kernel, dim3(static_cast<unsigned>((m + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, buf, dim, w, bias_out);
  return check_launch("unpack_bias_kernel");
}

extern "C" size_t collie_shard_plan_workspace_bytes(int64_t batch, int32_t num_neg, int32_t world,
                                                    int64_t num_users, int64_t num_items,
                                                    int32_t local_rank) {
  if (batch < 1 || num_neg < 1 || world < 1 || world > kMaxWorld || num_users < 1 || num_items < 1 ||
      num_users + num_items >= (int64_t{1} << 31))
    return 0;
  const int64_t lookups = batch * (2 + static_cast<int64_t>(num_neg));
  const int64_t planned = local_rank >= 0 ? lookups - batch : lookups;
  const int64_t nkeys = (local_rank >= 0 ? 0 : num_users) + num_items;
  if (use_direct(nkeys, planned)) return carve_direct_ws(nullptr, nkeys, world).bytes;
  return carve_plan_ws(nullptr, lookups, world).bytes;
}

extern "C" int collie_shard_plan(const collie_mf_batch* b, int64_t num_users, int64_t num_items,
                                 int32_t world, int32_t dim, int32_t local_rank, void* workspace,
                                 size_t workspace_bytes,
                                 int32_t* compact_users, int32_t* compact_pos, int64_t* compact_neg,
                                 int64_t* send_rows, int64_t send_capacity, int64_t* route_counts,
                                 collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_batch(b)) return -1;
  COLLIE_REQUIRE(world >= 1 && world <= kMaxWorld, "shard_plan: world must be in [1, %d]", kMaxWorld);
  COLLIE_REQUIRE(dim >= 1, "shard_plan: bad dim");
  COLLIE_REQUIRE(local_rank >= -1 && local_rank < world, "shard_plan: bad local_rank");
  COLLIE_REQUIRE(num_users >= 1 && num_items >= 1 && num_users + num_items < (int64_t{1} << 31),
                 "shard_plan: bad table sizes");
  COLLIE_REQUIRE(compact_users && compact_pos && compact_neg && send_rows && route_counts,
                 "shard_plan: NULL output");
  const int64_t lookups = b->batch * (2 + b->num_neg);
  const int64_t max_rows = lookups < num_users + num_items ? lookups : num_users + num_items;
  COLLIE_REQUIRE(send_capacity >= max_rows, "shard_plan: send_rows holds %lld rows, need %lld",
                 static_cast<long long>(send_capacity), static_cast<long long>(max_rows));
  Routes r;
  make_routes(r, num_users, num_items, world, local_rank < 0);
  const int nroutes = 2 * world;
  // users local: the first B entries of the batch are not planned (only the item ids are)
  const int64_t planned = local_rank >= 0 ? lookups - b->batch : lookups;
  const int64_t nkeys = r.key_base[nroutes];
  COLLIE_CHECK_HIP(hipMemsetAsync(route_counts, 0, sizeof(int64_t) * (nroutes + 2), stream));
  int64_t grid = (lookups + kBlock - 1) / kBlock;
  if (grid > 8192) grid = 8192;
  if (use_direct(nkeys, planned)) {
    DirectWs ws = carve_direct_ws(workspace, nkeys, world);
    COLLIE_REQUIRE(workspace && workspace_bytes >= ws.bytes,
                   "shard_plan: workspace too small: need %zu bytes, got %zu", ws.bytes,
                   workspace_bytes);
    COLLIE_CHECK_HIP(hipMemsetAsync(ws.marks, 0, static_cast<size_t>(ws.nblocks) * kWordsPerBlock * 64,
                                    stream));
    hipLaunchKernelGGL(plan_keys_kernel<true>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       stream, *b, r, local_rank, nullptr, nullptr, ws.marks, compact_users,
                       route_counts + nroutes + 1);
    if (check_launch("plan_keys_kernel")) return -2;
    hipLaunchKernelGGL(plan_bits_kernel, dim3(static_cast<unsigned>(ws.nblocks)),
                       dim3(kWordsPerBlock), 0, stream, ws.marks, ws.words, ws.blk);
    if (check_launch("plan_bits_kernel")) return -2;
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(kScanBlock), 0, stream, ws.blk, ws.nblocks);
    if (check_launch("plan_scan_kernel")) return -2;
    hipLaunchKernelGGL(plan_rows_kernel, dim3(static_cast<unsigned>(ws.nblocks)),
                       dim3(kWordsPerBlock), 0, stream, ws.words, ws.blk, r, send_rows);
    if (check_launch("plan_rows_kernel")) return -2;
    hipLaunchKernelGGL(plan_direct_layout_kernel, dim3(1), dim3(64), 0, stream, ws.words, ws.blk,
                       ws.nblocks, r, dim, route_counts, ws.shift);
    if (check_launch("plan_direct_layout_kernel")) return -2;
    hipLaunchKernelGGL(plan_direct_emit_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                       stream, *b, r, local_rank, ws.words, ws.shift, compact_users, compact_pos,
                       compact_neg);
    return check_launch("plan_direct_emit_kernel");
  }
  PlanWs ws = carve_plan_ws(workspace, lookups, world);
  COLLIE_REQUIRE(workspace && workspace_bytes >= ws.bytes,
                 "shard_plan: workspace too small: need %zu bytes, got %zu", ws.bytes,
                 workspace_bytes);
  hipLaunchKernelGGL(plan_keys_kernel<false>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                     stream, *b, r, local_rank, ws.sort.ekeys, ws.sort.ecodes, nullptr, compact_users,
                     route_counts + nroutes + 1);
  if (check_launch("plan_keys_kernel")) return -2;
  ws.sort.m = planned;
  ws.sort.ntiles = (planned + kSortTile - 1) / kSortTile;
  if (sort_and_segment(ws.sort, nkeys, 0xffffffffu, stream, ws.seg_of)) return -1;
  hipLaunchKernelGGL(plan_layout_kernel, dim3(1), dim3(64), 0, stream, sorted_keys(ws.sort, nkeys),
                     ws.seg_of, ws.sort.nseg, planned, r, dim, route_counts, ws.shift);
  if (check_launch("plan_layout_kernel")) return -2;
  hipLaunchKernelGGL(plan_emit_kernel, dim3(static_cast<unsigned>((planned + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, sorted_keys(ws.sort, nkeys), sorted_vals(ws.sort, nkeys),
                     ws.seg_of, planned, b->batch, r, ws.shift, compact_users, compact_pos, compact_neg,
                     send_rows);
  return check_launch("plan_emit_kernel");
}
