// Row-sharded training, requester side: which table rows does this rank's batch look up, who owns
// them, and what is each sample's row number in the compact tables the owners will send back?
//
// The set of looked-up ids is built without sorting the (2+K)*B ids: every id marks a stamp table
// (plain stores of the step id: all writers of a cell write the same value), and the compact
// numbering is a prefix sum over that table.  The table is laid out ROUTE-major -- route = (owner,
// kind), owner = id mod G, kind = user|item, position inside a route = id div G -- so the prefix
// sum walks memory linearly and the resulting order (owner, kind, id ascending) is exactly the order
// in which the rows are requested from / returned by the owners.
#include "common.h"

namespace collie {

int validate_batch(const collie_mf_batch* b);

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 8;
constexpr int kTile = kBlock * kPerThread;  // table cells per workgroup
constexpr int kMaxWorld = 16;   // GPUs of one node; keeps Routes well inside the kernarg limit

struct Routes {
  int world;
  int64_t num_users, num_items;
  int64_t cell_base[2 * kMaxWorld + 1];  // first table cell of a route (tile aligned)
  int64_t tile_base[2 * kMaxWorld + 1];  // first tile of a route
  int64_t len[2 * kMaxWorld];            // rows in the route
  int64_t row_base[2 * kMaxWorld];       // owner-side row offset of the route's kind (0 | U_loc(owner))
};

struct PlanWs {
  int32_t* stamp;       // [cells]  step id of the last step that looked the row up
  int32_t* slot;        // [cells]  compact row number of the row in that step
  uint32_t* tile_count; // [tiles]
  uint32_t* tile_base_; // [tiles]  exclusive prefix
  size_t bytes;
};

void make_routes(Routes& r, int64_t U, int64_t N, int world) {
  r.world = world;
  r.num_users = U;
  r.num_items = N;
  int64_t cell = 0, tile = 0;
  for (int o = 0; o < world; ++o) {
    const int64_t u_loc = (U - o + world - 1) / world;
    const int64_t n_loc = (N - o + world - 1) / world;
    for (int kind = 0; kind < 2; ++kind) {
      const int q = 2 * o + kind;
      r.len[q] = kind ? n_loc : u_loc;
      r.row_base[q] = kind ? u_loc : 0;
      r.cell_base[q] = cell;
      r.tile_base[q] = tile;
      const int64_t tiles = (r.len[q] + kTile - 1) / kTile;
      tile += tiles;
      cell += tiles * kTile;
    }
  }
  r.cell_base[2 * world] = cell;
  r.tile_base[2 * world] = tile;
}

PlanWs carve_plan_ws(void* base, const Routes& r) {
  Carver c(base);
  PlanWs w;
  const size_t cells = static_cast<size_t>(r.cell_base[2 * r.world]);
  const size_t tiles = static_cast<size_t>(r.tile_base[2 * r.world]);
  w.stamp = c.take<int32_t>(cells);
  w.slot = c.take<int32_t>(cells);
  w.tile_count = c.take<uint32_t>(tiles);
  w.tile_base_ = c.take<uint32_t>(tiles + 1);
  w.bytes = align_up(c.off, 256);
  return w;
}

__device__ __forceinline__ int64_t cell_of(const Routes& r, int kind, int64_t id) {
  const int64_t owner = id % r.world;
  return r.cell_base[2 * owner + kind] + id / r.world;
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

__global__ __launch_bounds__(kBlock) void plan_mark_kernel(const collie_mf_batch b, const Routes r,
                                                           int32_t* __restrict__ stamp, int step,
                                                           int64_t* __restrict__ bad_flag) {
  const int64_t total = b.batch * (2 + b.num_neg);
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * kBlock) {
    int kind;
    const int64_t id = batch_id(b, e, kind);
    if (id < 0 || id >= (kind ? r.num_items : r.num_users)) {
      *bad_flag = 1;
      continue;
    }
    // read before write: after the first few thousand ids almost every cell is already stamped, and
    // a (cached) load is far cheaper than a store to a line every CU is writing
    int32_t* cell = stamp + cell_of(r, kind, id);
    if (*cell != step) *cell = step;
  }
}

__global__ __launch_bounds__(kBlock) void plan_count_kernel(const int32_t* __restrict__ stamp,
                                                            int step,
                                                            uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t wave_tot[kBlock / kWave];
  const int64_t cell0 = static_cast<int64_t>(blockIdx.x) * kTile + threadIdx.x * kPerThread;
  const int4 a = *reinterpret_cast<const int4*>(stamp + cell0);
  const int4 c = *reinterpret_cast<const int4*>(stamp + cell0 + 4);
  uint32_t n = (a.x == step) + (a.y == step) + (a.z == step) + (a.w == step) + (c.x == step) +
               (c.y == step) + (c.z == step) + (c.w == step);
  for (int o = kWave / 2; o > 0; o >>= 1) n += __shfl_xor(n, o, kWave);
  if (lane_id() == 0) wave_tot[threadIdx.x / kWave] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += wave_tot[w];
    tile_count[blockIdx.x] = t;
  }
}

// one workgroup: exclusive prefix over the tiles, route counts and the total
__global__ __launch_bounds__(1024) void plan_scan_kernel(const uint32_t* __restrict__ tile_count,
                                                         uint32_t* __restrict__ tile_base,
                                                         const Routes r,
                                                         int64_t* __restrict__ route_counts) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  const int64_t tiles = r.tile_base[2 * r.world];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < tiles; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < tiles ? tile_count[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
      const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < tiles) tile_base[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_base[tiles] = carry;
  __syncthreads();
  const int nroutes = 2 * r.world;
  if (threadIdx.x < nroutes) {
    const uint32_t lo = tile_base[r.tile_base[threadIdx.x]];
    const uint32_t hi = tile_base[r.tile_base[threadIdx.x + 1]];
    route_counts[threadIdx.x] = static_cast<int64_t>(hi - lo);
  }
  if (threadIdx.x == 0) route_counts[nroutes] = carry;
}

__global__ __launch_bounds__(kBlock) void plan_number_kernel(
    const int32_t* __restrict__ stamp, int step, const uint32_t* __restrict__ tile_base,
    const Routes r, int32_t* __restrict__ slot, int64_t* __restrict__ send_rows) {
  __shared__ uint32_t wave_tot[kBlock / kWave];
  const int64_t tile = blockIdx.x;
  int route = 0;
  while (route + 1 < 2 * r.world && tile >= r.tile_base[route + 1]) ++route;
  const int64_t cell0 = tile * kTile + threadIdx.x * kPerThread;
  const int64_t local0 = cell0 - r.cell_base[route];  // row inside the route
  int32_t s[kPerThread];
  *reinterpret_cast<int4*>(s) = *reinterpret_cast<const int4*>(stamp + cell0);
  *reinterpret_cast<int4*>(s + 4) = *reinterpret_cast<const int4*>(stamp + cell0 + 4);
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) mine += (s[i] == step);
  // exclusive prefix of `mine` over the workgroup
  uint32_t incl = mine;
  const int lane = lane_id();
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += up;
  }
  if (lane == kWave - 1) wave_tot[threadIdx.x / kWave] = incl;
  __syncthreads();
  uint32_t before = tile_base[tile];
  for (int w = 0; w < static_cast<int>(threadIdx.x / kWave); ++w) before += wave_tot[w];
  uint32_t pos = before + incl - mine;
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) {
    if (s[i] == step) {
      slot[cell0 + i] = static_cast<int32_t>(pos);
      send_rows[pos] = r.row_base[route] + local0 + i;
      ++pos;
    }
  }
}

__global__ __launch_bounds__(kBlock) void plan_translate_kernel(
    const collie_mf_batch b, const Routes r, const int32_t* __restrict__ slot,
    int32_t* __restrict__ cu, int32_t* __restrict__ cp, int64_t* __restrict__ cn) {
  const int64_t B = b.batch;
  const int64_t total = B * (2 + b.num_neg);
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < total;
       e += static_cast<int64_t>(gridDim.x) * kBlock) {
    int kind;
    int64_t id = batch_id(b, e, kind);
    const int64_t rows = kind ? r.num_items : r.num_users;
    int32_t out = 0;
    if (id >= 0 && id < rows) out = slot[cell_of(r, kind, id)];
    if (e < B) cu[e] = out;
    else if (e < 2 * B) cp[e - B] = out;
    else cn[e - 2 * B] = out;
  }
}

// owner side of the lookup exchange: out_emb[j, :] = emb[rows[j], :], out_bias[j] = bias[rows[j]].
// One 16-lane group per requested row (16-byte pieces when the rows allow it).
template <int VEC>
__global__ __launch_bounds__(kBlock) void serve_rows_kernel(
    const float* __restrict__ emb, const float* __restrict__ bias, int64_t num_rows, int dim,
    const int64_t* __restrict__ rows, int64_t m, float* __restrict__ out_emb,
    float* __restrict__ out_bias, int32_t* __restrict__ status) {
  constexpr int LPR = 16;
  const int lig = threadIdx.x % LPR;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * (kBlock / LPR);
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * (kBlock / LPR) + threadIdx.x / LPR; j < m;
       j += ngroups) {
    const int64_t r = rows[j];
    const bool ok = r >= 0 && r < num_rows;
    if (!ok && lig == 0 && status) atomicOr(status, 1);
    const float* src = emb + (ok ? r : 0) * dim;
    float* dst = out_emb + j * dim;
    if (VEC == 4) {
      for (int c = lig; c < dim / 4; c += LPR) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = reinterpret_cast<const float4*>(src)[c];
        reinterpret_cast<float4*>(dst)[c] = v;
      }
    } else {
      for (int c = lig; c < dim; c += LPR) dst[c] = ok ? src[c] : 0.0f;
    }
    if (lig == 0) out_bias[j] = ok ? bias[r] : 0.0f;
  }
}

}  // namespace
}  // namespace collie

using namespace collie;

extern "C" int collie_shard_serve_rows(const float* emb, const float* bias, int64_t num_rows,
                                       int32_t dim, const int64_t* rows, int64_t m, float* out_emb,
                                       float* out_bias, int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(emb && bias && num_rows >= 1 && dim >= 1, "serve_rows: bad table");
  COLLIE_REQUIRE(m >= 0, "serve_rows: bad row count");
  if (m == 0) return 0;
  COLLIE_REQUIRE(rows && out_emb && out_bias, "serve_rows: NULL pointer");
  int64_t blocks = (m + kBlock / 16 - 1) / (kBlock / 16);
  if (blocks > 65536) blocks = 65536;
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  if (dim % 4 == 0 && al16(emb) && al16(out_emb))
    hipLaunchKernelGGL(serve_rows_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, emb, bias, num_rows, dim, rows, m, out_emb, out_bias, status);
  else
    hipLaunchKernelGGL(serve_rows_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                       stream, emb, bias, num_rows, dim, rows, m, out_emb, out_bias, status);
  return check_launch("serve_rows_kernel");
}

extern "C" size_t collie_shard_plan_workspace_bytes(int64_t num_users, int64_t num_items,
                                                    int32_t world) {
  if (num_users < 1 || num_items < 1 || world < 1 || world > kMaxWorld) return 0;
  Routes r;
  make_routes(r, num_users, num_items, world);
  return carve_plan_ws(nullptr, r).bytes;
}

extern "C" int collie_shard_plan(const collie_mf_batch* b, int64_t num_users, int64_t num_items,
                                 int32_t world, int32_t step_id, void* workspace,
                                 size_t workspace_bytes, int32_t* compact_users,
                                 int32_t* compact_pos, int64_t* compact_neg, int64_t* send_rows,
                                 int64_t send_capacity, int64_t* route_counts,
                                 collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_batch(b)) return -1;
  COLLIE_REQUIRE(world >= 1 && world <= kMaxWorld, "shard_plan: world must be in [1, %d]", kMaxWorld);
  COLLIE_REQUIRE(num_users >= 1 && num_items >= 1 && num_users + num_items < (int64_t{1} << 31),
                 "shard_plan: bad table sizes");
  COLLIE_REQUIRE(step_id >= 1, "shard_plan: step_id must be >= 1");
  COLLIE_REQUIRE(compact_users && compact_pos && compact_neg && send_rows && route_counts,
                 "shard_plan: NULL output");
  const int64_t lookups = b->batch * (2 + b->num_neg);
  const int64_t max_rows = lookups < num_users + num_items ? lookups : num_users + num_items;
  COLLIE_REQUIRE(send_capacity >= max_rows, "shard_plan: send_rows holds %lld rows, need %lld",
                 static_cast<long long>(send_capacity), static_cast<long long>(max_rows));
  Routes r;
  make_routes(r, num_users, num_items, world);
  PlanWs ws = carve_plan_ws(workspace, r);
  COLLIE_REQUIRE(workspace && workspace_bytes >= ws.bytes,
                 "shard_plan: workspace too small: need %zu bytes, got %zu", ws.bytes,
                 workspace_bytes);
  const int64_t tiles = r.tile_base[2 * world];
  COLLIE_CHECK_HIP(hipMemsetAsync(route_counts, 0, sizeof(int64_t) * (2 * world + 2), stream));
  int64_t grid = (lookups + kBlock - 1) / kBlock;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(plan_mark_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream,
                     *b, r, ws.stamp, step_id, route_counts + 2 * world + 1);
  if (check_launch("plan_mark_kernel")) return -2;
  hipLaunchKernelGGL(plan_count_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kBlock), 0, stream,
                     ws.stamp, step_id, ws.tile_count);
  if (check_launch("plan_count_kernel")) return -2;
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, stream, ws.tile_count, ws.tile_base_,
                     r, route_counts);
  if (check_launch("plan_scan_kernel")) return -2;
  hipLaunchKernelGGL(plan_number_kernel, dim3(static_cast<unsigned>(tiles)), dim3(kBlock), 0,
                     stream, ws.stamp, step_id, ws.tile_base_, r, ws.slot, send_rows);
  if (check_launch("plan_number_kernel")) return -2;
  hipLaunchKernelGGL(plan_translate_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0,
                     stream, *b, r, ws.slot, compact_users, compact_pos, compact_neg);
  return check_launch("plan_translate_kernel");
}
