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
#include <vector>

#include "rowsort.h"
#include "shard_wire.h"

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
// neg_lr / bc2_sqrt are passed apart from the rest of Hyper: they are the two step-dependent
// scalars, which a replayed step reads from device memory (the row kernels keep reading everything
// else through the table descriptor: selecting whole Hyper structs per lane made `algo` a per-lane
// value and the switch below divergent -- emb_rows_kernel 2.2 -> 2.9 ms on the 100 M x 10 M shape).
__device__ __forceinline__ float apply_update2(const Hyper& h, float neg_lr, float bc2_sqrt, float p,
                                               float g, float& s1, float& s2) {
  switch (h.algo) {
    case COLLIE_OPT_SGD: {
      if (h.dense && h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      return p + neg_lr * g;
    }
    case COLLIE_OPT_ADAGRAD: {
      if (h.dense && h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      s1 = s1 + g * g;
      const float std_ = sqrtf(s1) + h.eps;
      return p + neg_lr * (g / std_);
    }
    case COLLIE_OPT_SPARSE_ADAM: {
      const float um = (g - s1) * h.c1;
      const float uv = (g * g - s2) * h.c2;
      s1 = s1 + um;
      s2 = s2 + uv;
      const float denom = sqrtf(s2) + h.eps;
      return p + neg_lr * (s1 / denom);
    }
    default: {  // COLLIE_OPT_ADAM (dense)
      if (h.weight_decay != 0.0f) g = g + h.weight_decay * p;
      s1 = s1 + h.c1 * (g - s1);                 // exp_avg.lerp_(grad, 1 - beta1)
      s2 = s2 * h.beta2;                         // exp_avg_sq.mul_(beta2)
      s2 = s2 + h.c2 * (g * g);                  //            .addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(s2) / bc2_sqrt + h.eps;
      return p + neg_lr * (s1 / denom);
    }
  }
}

// captured-step replay: the step-dependent scalars come from device memory (train_ws.h)
__device__ __forceinline__ Hyper dyn_hyper(Hyper h, const StepDyn* dyn, int idx) {
  if (dyn) {
    h.neg_lr = dyn->neg_lr[idx];
    h.bc2_sqrt = dyn->bc2_sqrt[idx];
  }
  return h;
}

__device__ __forceinline__ float apply_update(const Hyper& h, float p, float g, float& s1,
                                              float& s2) {
  return apply_update2(h, h.neg_lr, h.bc2_sqrt, p, g, s1, s2);
}

struct TableDesc {
  float* table;        // [rows, D] (D = 1 for a bias table); NULL = not owned by this call
  float* s1;
  float* s2;
  int32_t* row_stamp;  // dense algorithms: row_stamp[row] = step for rows that had a segment
  int32_t* seg_stamp;  // sparse item side: seg_stamp[row] = step for rows that had a segment
  const float* stage;  // [*, D] staged rows (embedding) | gcoef [*] (bias) | NULL = zero gradient
  Hyper h;
};

struct UpdParams {
  TableDesc d[2];      // [0]: keys < split, [1]: keys >= split (row = key - split)
  uint32_t split;
  const uint32_t* keys;
  const uint32_t* vals;
  const uint32_t* seg_start;
  const uint32_t* nseg;
  const uint32_t* long_base;    // per segment: first piece of a long segment, or kNoPiece
  const uint32_t* piece_start;
  const uint32_t* piece_end;
  float* partial;               // [pieces, D] piece sums
  const float* grad_scale;
  int64_t* rows_out;   // export mode
  float* vals_out;
  int32_t* count_out;
  int export_side;
  int D;
  int step;
  const StepDyn* dyn;  // non-NULL: step and the step-dependent Hyper fields are read from here
  int dyn_base;        // 2 * is_bias: index of d[0]'s table in StepDyn
  bool stream_rows;    // host side only: pick the STREAM row kernels (beyond_caches())
};

// 16-byte accesses that leave no copy in the caches: a swept row is read and written once per step
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
  nt_f4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<nt_f4*>(p));
}

template <int VEC>
struct RVec {
  float v[VEC];
};
template <int VEC, bool STREAM = false>
__device__ __forceinline__ RVec<VEC> rload(const float* p) {
  RVec<VEC> r;
  if (VEC == 4) {
    const float4 x = STREAM ? nt_load4(p) : *reinterpret_cast<const float4*>(p);
    r.v[0] = x.x; r.v[1 % VEC] = x.y; r.v[2 % VEC] = x.z; r.v[3 % VEC] = x.w;
  } else if (VEC == 2) {
    const float2 x = *reinterpret_cast<const float2*>(p);
    r.v[0] = x.x; r.v[1 % VEC] = x.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int VEC, bool STREAM = false>
__device__ __forceinline__ void rstore(float* p, const RVec<VEC>& r) {
  if (VEC == 4) {
    const float4 x = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
    if (STREAM) nt_store4(p, x);
    else *reinterpret_cast<float4*>(p) = x;
  } else if (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1 % VEC]);
  } else {
    *p = r.v[0];
  }
}

// Piece sums of the long segments (embedding rows): LPR lanes own one piece and add its staged rows
// in entry order, four loads in flight; pieces of tables this call does not own are skipped.
template <bool EXPORT, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void piece_rows_kernel(const UpdParams p) {
  constexpr int GPW = kWave / LPR;
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const uint32_t npieces = p.nseg[2];
  const int D = p.D;
  const int nchunks = (D + VEC - 1) / VEC;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t pc0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * GPW;
       pc0 < npieces; pc0 += nwaves * GPW) {
    const int64_t pc = pc0 + grp;
    if (pc >= npieces) continue;
    const uint32_t start = p.piece_start[pc], end = p.piece_end[pc];
    const int side = p.keys[start] >= p.split ? 1 : 0;
    const TableDesc& t = p.d[side];
    if (EXPORT ? side != p.export_side : t.table == nullptr) continue;
    const float* stage = t.stage;
    for (int chunk = lig; chunk < nchunks; chunk += LPR) {
      RVec<VEC> acc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] = 0.0f;
      for (uint32_t e0 = start; e0 < end; e0 += 4) {
        RVec<VEC> x[4];
        float sg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t c = e0 + q < end ? p.vals[e0 + q] : kCodeNone;
          const bool live = c != kCodeNone && stage != nullptr;
          sg[q] = live ? ((c & kCodeNeg) ? -1.0f : 1.0f) : 0.0f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) x[q].v[i] = 0.0f;
          if (live) x[q] = rload<VEC>(stage + static_cast<int64_t>(c & kCodeMask) * D + chunk * VEC);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (sg[q] != 0.0f) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.v[i] += sg[q] * x[q].v[i];
          }
      }
      rstore<VEC>(p.partial + pc * D + chunk * VEC, acc);
    }
  }
}

// Piece sums for a bias table: one wave per piece, lanes stride the entries, fixed butterfly.
template <bool EXPORT>
__global__ __launch_bounds__(kBlock) void piece_bias_kernel(const UpdParams p) {
  const int lane = lane_id();
  const uint32_t npieces = p.nseg[2];
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t pc = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
       pc < npieces; pc += nwaves) {
    const uint32_t start = p.piece_start[pc], end = p.piece_end[pc];
    const int side = p.keys[start] >= p.split ? 1 : 0;
    const TableDesc& t = p.d[side];
    if (EXPORT ? side != p.export_side : t.table == nullptr) continue;
    float acc = 0.0f;
    if (t.stage) {
      for (uint32_t e = start + lane; e < end; e += kWave) {
        const uint32_t c = p.vals[e];
        if (c != kCodeNone) {
          const float gv = t.stage[c & kCodeMask];
          acc += (c & kCodeNeg) ? -gv : gv;
        }
      }
    }
    acc = group_sum<kWave>(acc);
    if (lane == 0) p.partial[pc] = acc;
  }
}

// EXPORT = true writes the coalesced gradient of one side instead of updating the tables.
// LPR lanes own one segment (VEC floats per lane per chunk), so 64 / LPR segments are reduced and
// updated concurrently by a wave: independent rows keep 64 / LPR times more loads in flight than
// one-wave-per-row.  The sum over a segment is still formed entry by entry in batch order.
// STREAM: the rows of the table and of its optimizer state, and the staged gradient rows, are read
// and written non-temporally (tables far beyond the caches, beyond_caches()).  Compare variants
// inside ONE GPU job only: the same binary differs by up to 5 % between two boxes of the pool.
template <bool EXPORT, int VEC, int LPR, bool STREAM = false>
__global__ __launch_bounds__(kBlock) void emb_rows_kernel(const UpdParams p) {
  constexpr int GPW = kWave / LPR;  // segments per wave
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const uint32_t nseg = p.nseg[0];
  const uint32_t first1 = p.nseg[1];
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  // the two step-dependent scalars of each side (uniform); everything else of Hyper through t.h
  const float nl0 = p.dyn ? p.dyn->neg_lr[p.dyn_base] : p.d[0].h.neg_lr;
  const float nl1 = p.dyn ? p.dyn->neg_lr[p.dyn_base + 1] : p.d[1].h.neg_lr;
  const float bc0 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base] : p.d[0].h.bc2_sqrt;
  const float bc1 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base + 1] : p.d[1].h.bc2_sqrt;
  const int step = p.dyn ? p.dyn->step : p.step;
  const int D = p.D;
  const int nchunks = (D + VEC - 1) / VEC;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  if (EXPORT && blockIdx.x == 0 && threadIdx.x == 0)
    *p.count_out = static_cast<int32_t>(p.export_side ? nseg - first1 : first1);
  for (int64_t seg0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * GPW;
       seg0 < nseg; seg0 += nwaves * GPW) {
    const int64_t seg = seg0 + grp;
    bool active = seg < nseg;
    uint32_t start = 0, end = 0, key = 0;
    if (active) {
      start = p.seg_start[seg];
      end = p.seg_start[seg + 1];
      key = p.keys[start];
    }
    const int side = key >= p.split ? 1 : 0;
    const TableDesc& t = p.d[side];
    if (EXPORT ? side != p.export_side : t.table == nullptr) active = false;
    if (!active) end = start;
    // a long segment was summed piecewise (piece_rows_kernel): combine its pieces in order
    const uint32_t lbase = active ? p.long_base[seg] : kNoPiece;
    const uint32_t npc = lbase != kNoPiece ? (end - start + kPieceLen - 1) / kPieceLen : 0u;
    if (lbase != kNoPiece) end = start;
    const int64_t row = side ? key - p.split : key;
    const int64_t oseg = EXPORT ? (side ? seg - first1 : seg) : 0;
    if (EXPORT && active && lig == 0) p.rows_out[oseg] = row;
    const float* stage = active ? t.stage : nullptr;
    for (int c0 = 0; c0 < nchunks; c0 += LPR) {
      const int chunk = c0 + lig;
      const bool on = active && chunk < nchunks;
      const int64_t off = row * D + chunk * VEC;
      RVec<VEC> pv, s1, s2, acc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) pv.v[i] = s1.v[i] = s2.v[i] = acc.v[i] = 0.0f;
      if (!EXPORT && on) {
        pv = rload<VEC, STREAM>(t.table + off);
        if (t.s1) s1 = rload<VEC, STREAM>(t.s1 + off);
        if (t.s2) s2 = rload<VEC, STREAM>(t.s2 + off);
      }
      uint32_t e0 = start;
      while (__any(e0 < end)) {
        const uint32_t code = (e0 + lig < end) ? p.vals[e0 + lig] : kCodeNone;
        const int n_here = e0 < end ? static_cast<int>(min(end - e0, static_cast<uint32_t>(LPR))) : 0;
        for (int j0 = 0; j0 < LPR; j0 += 4) {
          if (!__any(j0 < n_here)) break;
          // up to four staged rows in flight per segment, added in batch order
          RVec<VEC> x[4];
          float sg[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t c = __shfl(code, (j0 + q) % LPR, LPR);
            const bool live = (j0 + q < LPR) && c != kCodeNone;
            sg[q] = live ? ((c & kCodeNeg) ? -1.0f : 1.0f) : 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[q].v[i] = 0.0f;
            if (live && on && stage)
              x[q] = rload<VEC, STREAM>(stage + static_cast<int64_t>(c & kCodeMask) * D + chunk * VEC);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (sg[q] != 0.0f) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc.v[i] += sg[q] * x[q].v[i];
            }
        }
        e0 += LPR;
      }
      if (on) {
        for (uint32_t i0 = 0; i0 < npc; i0 += 4) {
          RVec<VEC> x[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) x[q].v[i] = 0.0f;
            if (i0 + q < npc)
              x[q] = rload<VEC>(p.partial + static_cast<int64_t>(lbase + i0 + q) * D + chunk * VEC);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (i0 + q < npc) {
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc.v[i] += x[q].v[i];
            }
        }
      }
      if (on) {
        RVec<VEC> g;
#pragma unroll
        for (int i = 0; i < VEC; ++i) g.v[i] = acc.v[i] * scale;
        if (EXPORT) {
          rstore<VEC>(p.vals_out + oseg * D + chunk * VEC, g);
        } else {
          RVec<VEC> np;
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            np.v[i] = apply_update2(t.h, side ? nl1 : nl0, side ? bc1 : bc0, pv.v[i], g.v[i], s1.v[i],
                                    s2.v[i]);
          rstore<VEC, STREAM>(t.table + off, np);
          if (t.s1) rstore<VEC, STREAM>(t.s1 + off, s1);
          if (t.s2) rstore<VEC, STREAM>(t.s2 + off, s2);
        }
      }
    }
    if (!EXPORT && active && lig == 0) {
      if (t.row_stamp) t.row_stamp[row] = step;
      if (t.seg_stamp) t.seg_stamp[row] = step;
    }
  }
}

// bias tables.  The sum of a segment has a fixed shape: 16 lanes stride its entries, then a butterfly
// (offsets 8, 4, 2, 1).  Nearly every segment of a batch with uniform ids has one or two entries, so
// a wave first takes 64 segments, ONE PER LANE: a lane whose segment has at most four entries forms
// the same sum by itself -- for entries e0..e3 the butterfly gives (e0 + e2) + (e1 + e3), absent
// entries being +0 -- and only the longer segments (and those summed piecewise) are then reduced
// four at a time by 16-lane groups as before.  The read-modify-write of the bias, its moments and
// its stamps runs on all 64 lanes at once (16 times the rows in flight of one group per segment).
// d loss / d user_bias is identically zero for every pairwise loss (the user bias cancels in
// s+ - s-); the reference's value is only the fp32 residue of (+g) + (-g) in coalesce order, so
// the user-bias gradient is produced as an exact 0 (stage == NULL; the row is still "touched").
template <bool EXPORT>
__global__ __launch_bounds__(kBlock) void bias_rows_kernel(const UpdParams p) {
  constexpr int LPS = 16;  // lanes per long segment
  constexpr int GPW = kWave / LPS;
  const int lane = lane_id();
  const int grp = lane / LPS;
  const int lig = lane % LPS;
  const uint32_t nseg = p.nseg[0];
  const uint32_t first1 = p.nseg[1];
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const float nl0 = p.dyn ? p.dyn->neg_lr[p.dyn_base] : p.d[0].h.neg_lr;
  const float nl1 = p.dyn ? p.dyn->neg_lr[p.dyn_base + 1] : p.d[1].h.neg_lr;
  const float bc0 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base] : p.d[0].h.bc2_sqrt;
  const float bc1 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base + 1] : p.d[1].h.bc2_sqrt;
  const int step = p.dyn ? p.dyn->step : p.step;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  if (EXPORT && blockIdx.x == 0 && threadIdx.x == 0)
    *p.count_out = static_cast<int32_t>(p.export_side ? nseg - first1 : first1);
  for (int64_t seg0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * kWave;
       seg0 < nseg; seg0 += nwaves * kWave) {
    const int64_t seg = seg0 + lane;
    bool active = seg < nseg;
    uint32_t start = 0, end = 0, key = 0;
    if (active) {
      start = p.seg_start[seg];
      end = p.seg_start[seg + 1];
      key = p.keys[start];
    }
    const int side = key >= p.split ? 1 : 0;
    const TableDesc& t = p.d[side];
    if (EXPORT ? side != p.export_side : t.table == nullptr) active = false;
    const int64_t row = side ? key - p.split : key;
    const uint32_t lbase = active ? p.long_base[seg] : kNoPiece;
    const bool summed = active && t.stage != nullptr;           // (no stage: the gradient is 0)
    const bool quick = summed && lbase == kNoPiece && end - start <= 4;
    float acc = 0.0f;
    if (quick) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = 0.0f;
        if (start + i < end) {
          const uint32_t c = p.vals[start + i];
          if (c != kCodeNone) {
            const float gv = t.stage[c & kCodeMask];
            e[i] += (c & kCodeNeg) ? -gv : gv;
          }
        }
      }
      acc = (e[0] + e[2]) + (e[1] + e[3]);
    }
    // the long segments of these 64, four per round, in lane order
    uint64_t todo = __ballot(summed && !quick);
    while (todo) {
      const uint64_t round = todo;
      int src = 0;
      bool have = false;
#pragma unroll
      for (int i = 0; i < GPW; ++i) {
        if (todo) {
          const int b = __ffsll(static_cast<unsigned long long>(todo)) - 1;
          todo &= todo - 1;
          if (i == grp) {
            src = b;
            have = true;
          }
        }
      }
      const uint32_t s_start = __shfl(start, src, kWave);
      const uint32_t s_end = __shfl(end, src, kWave);
      const uint32_t s_lbase = __shfl(lbase, src, kWave);
      const int s_side = __shfl(side, src, kWave);
      const float* stage = p.d[s_side].stage;
      float a = 0.0f;
      if (have) {
        if (s_lbase != kNoPiece) {
          const uint32_t npc = (s_end - s_start + kPieceLen - 1) / kPieceLen;
          for (uint32_t i = lig; i < npc; i += LPS) a += p.partial[s_lbase + i];
        } else {
          for (uint32_t e = s_start + lig; e < s_end; e += LPS) {
            const uint32_t c = p.vals[e];
            if (c != kCodeNone) {
              const float gv = stage[c & kCodeMask];
              a += (c & kCodeNeg) ? -gv : gv;
            }
          }
        }
      }
      a = group_sum<LPS>(a);
      // back to the lane that owns the segment: it was the nth long one of this round
      const int nth = __popcll(static_cast<unsigned long long>(round & ((uint64_t{1} << lane) - 1)));
      const bool mine = ((round >> lane) & 1) && nth < GPW;
      const float got = __shfl(a, (nth < GPW ? nth : 0) * LPS, kWave);
      if (mine) acc = got;
    }
    const float g = acc * scale;
    if (active) {
      if (EXPORT) {
        const int64_t oseg = side ? seg - first1 : seg;
        p.rows_out[oseg] = row;
        p.vals_out[oseg] = g;
      } else {
        float s1 = t.s1 ? t.s1[row] : 0.0f;
        float s2 = t.s2 ? t.s2[row] : 0.0f;
        t.table[row] = apply_update2(t.h, side ? nl1 : nl0, side ? bc1 : bc0, t.table[row], g, s1, s2);
        if (t.s1) t.s1[row] = s1;
        if (t.s2) t.s2[row] = s2;
        if (t.row_stamp) t.row_stamp[row] = step;
        if (t.seg_stamp) t.seg_stamp[row] = step;
      }
    }
  }
}

// Batches whose segments are long on average (tables much smaller than the batch: the ML-20M shape has
// 18 entries per row at B = 1 M): 16 lanes per segment from the start, four segments per wave.
template <bool EXPORT>
__global__ __launch_bounds__(kBlock) void bias_rows_group_kernel(const UpdParams p) {
  constexpr int LPS = 16;  // lanes per segment
  constexpr int GPW = kWave / LPS;
  const int lane = lane_id();
  const int grp = lane / LPS;
  const int lig = lane % LPS;
  const uint32_t nseg = p.nseg[0];
  const uint32_t first1 = p.nseg[1];
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const float nl0 = p.dyn ? p.dyn->neg_lr[p.dyn_base] : p.d[0].h.neg_lr;
  const float nl1 = p.dyn ? p.dyn->neg_lr[p.dyn_base + 1] : p.d[1].h.neg_lr;
  const float bc0 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base] : p.d[0].h.bc2_sqrt;
  const float bc1 = p.dyn ? p.dyn->bc2_sqrt[p.dyn_base + 1] : p.d[1].h.bc2_sqrt;
  const int step = p.dyn ? p.dyn->step : p.step;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  if (EXPORT && blockIdx.x == 0 && threadIdx.x == 0)
    *p.count_out = static_cast<int32_t>(p.export_side ? nseg - first1 : first1);
  for (int64_t seg0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * GPW;
       seg0 < nseg; seg0 += nwaves * GPW) {
    const int64_t seg = seg0 + grp;
    bool active = seg < nseg;
    uint32_t start = 0, end = 0, key = 0;
    if (active) {
      start = p.seg_start[seg];
      end = p.seg_start[seg + 1];
      key = p.keys[start];
    }
    const int side = key >= p.split ? 1 : 0;
    const TableDesc& t = p.d[side];
    if (EXPORT ? side != p.export_side : t.table == nullptr) active = false;
    const int64_t row = side ? key - p.split : key;
    float acc = 0.0f;
    const uint32_t lbase = active ? p.long_base[seg] : kNoPiece;
    if (lbase != kNoPiece) {
      if (t.stage) {
        const uint32_t npc = (end - start + kPieceLen - 1) / kPieceLen;
        for (uint32_t i = lig; i < npc; i += LPS) acc += p.partial[lbase + i];
      }
    } else if (active && t.stage) {
      for (uint32_t e = start + lig; e < end; e += LPS) {
        const uint32_t c = p.vals[e];
        if (c != kCodeNone) {
          const float gv = t.stage[c & kCodeMask];
          acc += (c & kCodeNeg) ? -gv : gv;
        }
      }
    }
    acc = group_sum<LPS>(acc);
    const float g = acc * scale;
    if (active && lig == 0) {
      if (EXPORT) {
        const int64_t oseg = side ? seg - first1 : seg;
        p.rows_out[oseg] = row;
        p.vals_out[oseg] = g;
      } else {
        float s1 = t.s1 ? t.s1[row] : 0.0f;
        float s2 = t.s2 ? t.s2[row] : 0.0f;
        t.table[row] = apply_update2(t.h, side ? nl1 : nl0, side ? bc1 : bc0, t.table[row], g, s1, s2);
        if (t.s1) t.s1[row] = s1;
        if (t.s2) t.s2[row] = s2;
        if (t.row_stamp) t.row_stamp[row] = step;
        if (t.seg_stamp) t.seg_stamp[row] = step;
      }
    }
  }
}

// ---- small batches: coalesce and update in ONE launch, without a sort --------------------------------
// At most kSmallMax entries (the reference's default batch of 1,024 samples): every launch costs more
// than its work, so the sort, the embedding-row update and the bias-row update become one kernel.
// LPR lanes own one ENTRY i of the unsorted (key, code) list the forward kernel wrote.  The group
// scans all keys (staged in LDS, four per lane and iteration) for equal ones: an equal key at j < i
// means i is not the row's first entry and the group has nothing to do; equal keys at j > i are the
// rest of the row's segment, met in increasing j -- the batch order a stable sort would give them --
// and added as they are met.  Nearly every row of a small batch is looked up once, so a scan
// iteration is a 16-byte LDS read, four compares and one wave-wide ballot that says "nothing here".
// The sums are formed entry by entry in batch order, exactly as emb_rows_kernel forms them.
struct SmallParams {
  TableDesc e[2];      // embedding tables (user, item); table == NULL: not updated by this launch
  TableDesc z[2];      // bias tables
  const uint32_t* ekeys;
  const uint32_t* ecodes;
  const float* grad_scale;
  const StepDyn* dyn;
  uint32_t split;
  int m;
  int D;
  int step;
};

template <int LPR, int THREADS>
__global__ __launch_bounds__(THREADS) void small_rows_kernel(const SmallParams p) {
  constexpr int GPW = kWave / LPR;
  constexpr int U = 4;               // scan iterations whose LDS reads are issued together
  constexpr int kKeys = kSmallRowsMax + 4 * U * kWave;
  // (padded so that the unrolled scan may read past m: padding never equals a key)
  __shared__ __attribute__((aligned(16))) uint32_t keys[kKeys];
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int m = p.m;
  const int m_pad = (m + 4 * U * kWave + 3) & ~3;      // what the scan can touch
  for (int i = tid; i < m_pad; i += THREADS) keys[i] = i < m ? p.ekeys[i] : 0xffffffffu;
  const int i = (blockIdx.x * (THREADS / kWave) + tid / kWave) * GPW + grp;
  const bool active = i < m;
  const uint32_t key = active ? p.ekeys[i] : 0xfffffffeu;   // matches nothing (keys < 2^32 - 2)
  const int side = key >= p.split ? 1 : 0;
  const int64_t row = side ? key - p.split : key;
  const TableDesc& et = p.e[side];
  const TableDesc& zt = p.z[side];
  const int D = p.D;
  const int chunk = lig;                               // D / 4 <= LPR (host)
  const bool on = active && chunk * 4 < D;
  const float* estage = et.table ? et.stage : nullptr;
  const float* zstage = zt.table ? zt.stage : nullptr;

  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float zacc = 0.0f;
  // up to four entries at a time: their codes, then their staged rows, are loaded together (a hot
  // row's segment would otherwise be a chain of two dependent loads per entry); added in order
  auto add_entries = [&](const int (&js)[4], int n) {
    uint32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = q < n ? p.ecodes[js[q]] : kCodeNone;
    float4 x[4];
    float gv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      x[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      gv[q] = 0.0f;
      if (c[q] != kCodeNone) {
        const int64_t src = c[q] & kCodeMask;
        if (estage && on) x[q] = *reinterpret_cast<const float4*>(estage + src * D + chunk * 4);
        if (zstage && lig == 0) gv[q] = zstage[src];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c[q] == kCodeNone) continue;
      const float sg = (c[q] & kCodeNeg) ? -1.0f : 1.0f;
      acc.x += sg * x[q].x;
      acc.y += sg * x[q].y;
      acc.z += sg * x[q].z;
      acc.w += sg * x[q].w;
      zacc += (c[q] & kCodeNeg) ? -gv[q] : gv[q];
    }
  };
  int pend[4] = {0, 0, 0, 0};
  int npend = 0;
  // speculative (wasted only for the few entries that are not the first of their row): the entry's
  // own gradient row and the row of the tables it will update are in flight during the scan
  if (active) {
    pend[0] = i;
    add_entries(pend, 1);
  }
  const int64_t off = row * D + chunk * 4;
  float4 pv = make_float4(0, 0, 0, 0), s1 = pv, s2 = pv;
  if (et.table && on) {
    pv = *reinterpret_cast<const float4*>(et.table + off);
    if (et.s1) s1 = *reinterpret_cast<const float4*>(et.s1 + off);
    if (et.s2) s2 = *reinterpret_cast<const float4*>(et.s2 + off);
  }
  float zp = 0.0f, zs1 = 0.0f, zs2 = 0.0f;
  if (zt.table && active && lig == 0) {
    zp = zt.table[row];
    if (zt.s1) zs1 = zt.s1[row];
    if (zt.s2) zs2 = zt.s2[row];
  }
  __syncthreads();

  bool first = active;
  const uint4* keys4 = reinterpret_cast<const uint4*>(keys);
  const int iters = (m + 4 * LPR - 1) / (4 * LPR);
  const unsigned long long gmask = (LPR == kWave ? ~0ull : ((1ull << LPR) - 1ull)) << (grp * LPR);
  for (int t0 = 0; t0 < iters; t0 += U) {
    uint4 k4[U];
    uint32_t eq[U], any_eq = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) k4[u] = keys4[(t0 + u) * LPR + lig];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j0 = 4 * ((t0 + u) * LPR + lig);
      eq[u] = (k4[u].x == key && j0 != i ? 1u : 0u) | (k4[u].y == key && j0 + 1 != i ? 2u : 0u) |
              (k4[u].z == key && j0 + 2 != i ? 4u : 0u) | (k4[u].w == key && j0 + 3 != i ? 8u : 0u);
      any_eq |= eq[u];
    }
    if (__ballot(any_eq != 0u) == 0ull) continue;      // wave-uniform: the common case
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u;
      const int j0 = 4 * (t * LPR + lig);
      // an equal key before i anywhere in the group's window: i is not the first entry of its row
      uint32_t before = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) before |= ((eq[u] >> c) & 1u) && j0 + c < i ? 1u : 0u;
      if (__ballot(before != 0u) & gmask) first = false;
      // equal keys after i, in increasing j = (lane, component) order
      unsigned long long later[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        later[c] = (__ballot(((eq[u] >> c) & 1u) && j0 + c > i) & gmask) >> (grp * LPR);
      unsigned long long any_later = later[0] | later[1] | later[2] | later[3];
      if (!first) any_later = 0ull;
      while (any_later) {
        const int l = __builtin_ctzll(any_later);
        any_later &= any_later - 1ull;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if ((later[c] >> l) & 1ull) {
            const int j = 4 * (t * LPR + l) + c;
            // (static indexing: the list stays in registers)
            pend[0] = npend == 0 ? j : pend[0];
            pend[1] = npend == 1 ? j : pend[1];
            pend[2] = npend == 2 ? j : pend[2];
            pend[3] = npend == 3 ? j : pend[3];
            if (++npend == 4) {
              add_entries(pend, 4);
              npend = 0;
            }
          }
      }
    }
  }
  if (!first) return;
  if (npend > 0) add_entries(pend, npend);

  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const int step = p.dyn ? p.dyn->step : p.step;
  if (et.table && on) {
    const float nl = p.dyn ? p.dyn->neg_lr[side] : et.h.neg_lr;
    const float bc = p.dyn ? p.dyn->bc2_sqrt[side] : et.h.bc2_sqrt;
    pv.x = apply_update2(et.h, nl, bc, pv.x, acc.x * scale, s1.x, s2.x);
    pv.y = apply_update2(et.h, nl, bc, pv.y, acc.y * scale, s1.y, s2.y);
    pv.z = apply_update2(et.h, nl, bc, pv.z, acc.z * scale, s1.z, s2.z);
    pv.w = apply_update2(et.h, nl, bc, pv.w, acc.w * scale, s1.w, s2.w);
    *reinterpret_cast<float4*>(et.table + off) = pv;
    if (et.s1) *reinterpret_cast<float4*>(et.s1 + off) = s1;
    if (et.s2) *reinterpret_cast<float4*>(et.s2 + off) = s2;
  }
  if (lig == 0) {
    if (zt.table) {
      const float nl = p.dyn ? p.dyn->neg_lr[2 + side] : zt.h.neg_lr;
      const float bc = p.dyn ? p.dyn->bc2_sqrt[2 + side] : zt.h.bc2_sqrt;
      zt.table[row] = apply_update2(zt.h, nl, bc, zp, zacc * scale, zs1, zs2);
      if (zt.s1) zt.s1[row] = zs1;
      if (zt.s2) zt.s2[row] = zs2;
      if (zt.seg_stamp) zt.seg_stamp[row] = step;
    }
    if (et.table && et.seg_stamp) et.seg_stamp[row] = step;
  }
}

// K5: dense optimizers also move every row that had NO segment this step (gradient 0, plus
// weight decay): Adam decays m, v and keeps stepping along m.
__global__ __launch_bounds__(kBlock) void dense_sweep_kernel(float* __restrict__ table,
                                                             float* __restrict__ s1p,
                                                             float* __restrict__ s2p,
                                                             const int32_t* __restrict__ stamp,
                                                             int64_t rows, int D, int step_,
                                                             const Hyper h_, const StepDyn* dyn,
                                                             int dyn_idx) {
  const Hyper h = dyn_hyper(h_, dyn, dyn_idx);
  const int step = dyn ? dyn->step : step_;
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

// SparseAdam "touched" semantics for the items that were looked up (stamped by the forward
// kernel) but carry no gradient entry: their moments decay and the parameter moves along m.
__global__ __launch_bounds__(kBlock) void touched_sweep_kernel(float* __restrict__ table,
                                                               float* __restrict__ s1p,
                                                               float* __restrict__ s2p,
                                                               const uint8_t* __restrict__ touch,
                                                               const int32_t* __restrict__ seg,
                                                               int64_t rows, int D, int step_,
                                                               const Hyper h_, const StepDyn* dyn,
                                                               int dyn_idx) {
  const Hyper h = dyn_hyper(h_, dyn, dyn_idx);
  const int step = dyn ? dyn->step : step_;
  const int64_t total = rows * D;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += stride) {
    const int64_t row = i / D;
    if (touch[row] != touch_tag(step) || seg[row] == step) continue;
    float s1 = s1p[i];
    float s2 = s2p[i];
    table[i] = apply_update(h, table[i], 0.0f, s1, s2);
    s1p[i] = s1;
    s2p[i] = s2;
  }
}

// Row-granular variant of the sweep for float4-aligned rows: a wave scans 64 consecutive stamps
// with one coalesced load, compacts the rows that need the zero-gradient update into a wave-private
// LDS list, and then LPR lanes own one such row (64 / LPR rows in flight per wave, every lane with
// up to 4 chunks x (p, m, v) 16-byte loads outstanding before the first use).
template <int LPR, bool STREAM>
__global__ __launch_bounds__(kBlock) void touched_rows_kernel(float* __restrict__ table,
                                                              float* __restrict__ s1p,
                                                              float* __restrict__ s2p,
                                                              const uint8_t* __restrict__ touch,
                                                              const int32_t* __restrict__ seg,
                                                              int64_t rows, int D, int step_,
                                                              const Hyper h_, const StepDyn* dyn,
                                                              int dyn_idx, int span) {
  // span = stamps a wave scans at a time: 64 (one coalesced line per wave) for big tables; 16 for
  // small ones, where the grid has a wave per 16 rows anyway and a wave that took 64 would leave
  // three idle and walk its rows in four dependent rounds instead of one
  const Hyper h = dyn_hyper(h_, dyn, dyn_idx);
  const int step = dyn ? dyn->step : step_;
  constexpr int GPW = kWave / LPR;
  __shared__ int list[kWavesPerBlock][kWave];
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const int nchunks = D / 4;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave) * span;
       base < rows; base += nwaves * span) {
    const int64_t r = base + lane;
    const bool need = lane < span && r < rows && touch[r] == touch_tag(step) && seg[r] != step;
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) continue;
    const int n = __popcll(mask);
    if (need) list[wave][__popcll(mask & ((1ull << lane) - 1ull))] = lane;
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < n; i0 += GPW) {
      const int i = i0 + grp;
      const bool on_row = i < n;
      const int64_t row = base + list[wave][on_row ? i : 0];
      for (int c0 = 0; c0 < nchunks; c0 += 4 * LPR) {
        float4 pv[4], m[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + q * LPR + lig;
          if (on_row && c < nchunks) {
            const int64_t off = row * D + c * 4;
            pv[q] = STREAM ? nt_load4(table + off) : *reinterpret_cast<const float4*>(table + off);
            m[q] = STREAM ? nt_load4(s1p + off) : *reinterpret_cast<const float4*>(s1p + off);
            v[q] = STREAM ? nt_load4(s2p + off) : *reinterpret_cast<const float4*>(s2p + off);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + q * LPR + lig;
          if (on_row && c < nchunks) {
            const int64_t off = row * D + c * 4;
            float4 np;
            np.x = apply_update(h, pv[q].x, 0.0f, m[q].x, v[q].x);
            np.y = apply_update(h, pv[q].y, 0.0f, m[q].y, v[q].y);
            np.z = apply_update(h, pv[q].z, 0.0f, m[q].z, v[q].z);
            np.w = apply_update(h, pv[q].w, 0.0f, m[q].w, v[q].w);
            if (STREAM) {
              nt_store4(table + off, np);
              nt_store4(s1p + off, m[q]);
              nt_store4(s2p + off, v[q]);
            } else {
              *reinterpret_cast<float4*>(table + off) = np;
              *reinterpret_cast<float4*>(s1p + off) = m[q];
              *reinterpret_cast<float4*>(s2p + off) = v[q];
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(kBlock) void export_touched_kernel(const uint8_t* __restrict__ touch,
                                                                int64_t rows, int step,
                                                                uint8_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < rows) out[i] = touch[i] == touch_tag(step) ? 1 : 0;
}

// entries of a stand-alone sparse gradient: key = row id (clamped), code = position in vals
__global__ __launch_bounds__(kBlock) void coo_entries_kernel(const void* __restrict__ rows,
                                                             int dtype, int64_t m,
                                                             int64_t num_rows,
                                                             uint32_t* __restrict__ ekeys,
                                                             uint32_t* __restrict__ ecodes,
                                                             int32_t* __restrict__ status) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= m) return;
  int64_t r = load_idx(rows, dtype, i);
  uint32_t code = static_cast<uint32_t>(i);
  if (r < 0 || r >= num_rows) {
    if (status) atomicOr(status, 1);
    r = 0;
    code = kCodeNone;
  }
  ekeys[i] = static_cast<uint32_t>(r);
  ecodes[i] = code;
}


// ---- row-sharded step: the requester's gradient rows as the owners' blocks of the wire buffer --------
// (shard_wire.h).  The batch was run on COMPACT tables (the rows the owners served), user and item
// rows in one key space: an entry's code says which staging buffer its gradient row lives in.
struct WireExport {
  const uint32_t* keys;
  const uint32_t* vals;
  const uint32_t* seg_start;
  const uint32_t* nseg;
  const uint32_t* long_base;
  const uint32_t* piece_start;
  const uint32_t* piece_end;
  float* partial;        // [max_pieces, D] piece sums, then [max_pieces] bias piece sums
  int64_t max_pieces;
  const float* stage_u;
  const float* stage_i;
  const float* gcoef;
  const float* grad_scale;
  const int64_t* send_rows;   // dense request index -> row of the owner's combined table
  const GradLayout* lay;
  float* wire;
  uint32_t key_sub;   // users local: keys of item rows are num_users + compact row, user segments stay
  int all_items;      // users local: every exported entry is an item entry (codes carry no kind)
  int world;
  int D;
};

// entries [B, 3B) of the forward kernel's list are item rows: key - split is their compact row
__global__ __launch_bounds__(kBlock) void wire_rekey_kernel(uint32_t* __restrict__ ekeys,
                                                            uint32_t* __restrict__ ecodes, int64_t B,
                                                            uint32_t split) {
  const int64_t e = B + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= 3 * B) return;
  ekeys[e] -= split;
  const uint32_t c = ecodes[e];
  if (c != kCodeNone) ecodes[e] = c | kCodeItem;
}

// where each peer's gradient segments start, how many there are, and where its block of the wire
// buffer starts: thread p finds the first sorted entry at or behind peer p's first compact row
__global__ __launch_bounds__(64) void wire_layout_kernel(const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ seg_of,
                                                         const uint32_t* __restrict__ nseg, int64_t m,
                                                         const WireBlocks look, int dim,
                                                         uint32_t key_sub,
                                                         GradLayout* __restrict__ lay,
                                                         int64_t* __restrict__ peer_counts) {
  __shared__ int64_t gfirst[kMaxWorld + 1];
  const int world = look.world;
  if (static_cast<int>(threadIdx.x) <= world) {
    const int p = threadIdx.x;
    int64_t at = static_cast<int64_t>(nseg[0]);
    if (p < world) {
      const int64_t want = look.row0[p] + static_cast<int64_t>(key_sub);
      int64_t lo = 0, hi = m;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (static_cast<int64_t>(keys[mid]) < want) lo = mid + 1;
        else hi = mid;
      }
      if (lo < m) at = static_cast<int64_t>(seg_of[lo]);
    }
    gfirst[p] = at;
    lay->gfirst[p] = at;
    lay->row0[p] = look.row0[p];
    lay->dense0[p] = look.first[p];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int64_t w = 0;
  for (int p = 0; p < world; ++p) {
    const int64_t c = gfirst[p + 1] - gfirst[p];
    lay->wire0[p] = w;
    peer_counts[p] = c;
    w += c + 2 * ((c + dim - 1) / dim);
  }
  lay->wire0[world] = w;
  peer_counts[world] = w;
}

template <int VEC>
__device__ __forceinline__ RVec<VEC> wire_entry(const WireExport& p, uint32_t c, int chunk,
                                                float& sg) {
  RVec<VEC> x;
#pragma unroll
  for (int i = 0; i < VEC; ++i) x.v[i] = 0.0f;
  sg = 0.0f;
  if (c != kCodeNone) {
    sg = (c & kCodeNeg) ? -1.0f : 1.0f;
    const float* stage = (p.all_items || (c & kCodeItem)) ? p.stage_i : p.stage_u;
    x = rload<VEC>(stage + static_cast<int64_t>(c & kCodeSample) * p.D + chunk * VEC);
  }
  return x;
}

// d loss / d bias of an entry: items only (the user bias cancels in s+ - s-, see bias_rows_kernel)
__device__ __forceinline__ float wire_entry_bias(const WireExport& p, uint32_t c) {
  if (c == kCodeNone || !(p.all_items || (c & kCodeItem))) return 0.0f;
  const float gv = p.gcoef[c & kCodeSample];
  return (c & kCodeNeg) ? -gv : gv;
}

template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void wire_piece_kernel(const WireExport p) {
  constexpr int GPW = kWave / LPR;
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const uint32_t npieces = p.nseg[2];
  const int D = p.D;
  const int nchunks = (D + VEC - 1) / VEC;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t pc0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * GPW;
       pc0 < npieces; pc0 += nwaves * GPW) {
    const int64_t pc = pc0 + grp;
    if (pc >= npieces) continue;
    const uint32_t start = p.piece_start[pc], end = p.piece_end[pc];
    if (p.keys[start] < p.key_sub) continue;     // a user row (users local): updated in place
    for (int chunk = lig; chunk < nchunks; chunk += LPR) {
      RVec<VEC> acc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] = 0.0f;
      for (uint32_t e0 = start; e0 < end; e0 += 4) {
        RVec<VEC> x[4];
        float sg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          x[q] = wire_entry<VEC>(p, e0 + q < end ? p.vals[e0 + q] : kCodeNone, chunk, sg[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (sg[q] != 0.0f) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.v[i] += sg[q] * x[q].v[i];
          }
      }
      rstore<VEC>(p.partial + pc * D + chunk * VEC, acc);
    }
    if (lig == 0) {
      float b = 0.0f;
      for (uint32_t e = start; e < end; ++e) b += wire_entry_bias(p, p.vals[e]);
      p.partial[p.max_pieces * D + pc] = b;
    }
  }
}

// LPR lanes own one segment, every lane CPL chunks of VEC floats of its row (chunks lig, lig + LPR,
// ...): 64 / LPR segments per wave with up to 4 x CPL loads in flight per lane -- nearly every
// segment is a single entry, so the kernel lives on independent rows in flight, not on long sums.
template <int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void wire_rows_kernel(const WireExport p) {
  constexpr int GPW = kWave / LPR;
  constexpr int CPL = 4;
  __shared__ GradLayout lay;
  for (int i = threadIdx.x; i < static_cast<int>(sizeof(GradLayout) / sizeof(int64_t)); i += kBlock)
    reinterpret_cast<int64_t*>(&lay)[i] = reinterpret_cast<const int64_t*>(p.lay)[i];
  __syncthreads();
  const int lane = lane_id();
  const int grp = lane / LPR;
  const int lig = lane % LPR;
  const uint32_t nseg = p.nseg[0];
  const float scale = p.grad_scale ? *p.grad_scale : 1.0f;
  const int D = p.D;
  const int nchunks = (D + VEC - 1) / VEC;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  // (users local: the segments before gfirst[0] are user rows, which are not exported)
  for (int64_t seg0 = lay.gfirst[0] +
                      (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave) * GPW;
       seg0 < nseg; seg0 += nwaves * GPW) {
    const int64_t seg = seg0 + grp;
    const bool active = seg < nseg;
    uint32_t start = 0, end = 0, key = 0;
    uint32_t lbase = kNoPiece;
    if (active) {
      start = p.seg_start[seg];
      end = p.seg_start[seg + 1];
      lbase = p.long_base[seg];
      key = p.keys[start] - p.key_sub;
    }
    const uint32_t npc = lbase != kNoPiece ? (end - start + kPieceLen - 1) / kPieceLen : 0u;
    if (lbase != kNoPiece) end = start;
    // the peer that owns the row, and the row's place in that peer's block
    int peer = 0;
    while (peer + 1 < p.world && static_cast<int64_t>(key) >= lay.row0[peer + 1]) ++peer;
    const int64_t cnt = lay.gfirst[peer + 1] - lay.gfirst[peer];
    const int64_t jj = seg - lay.gfirst[peer];
    const int64_t w0 = lay.wire0[peer];
    const int64_t extra = (cnt + D - 1) / D;
    int64_t owner_row = 0;
    if (active && lig == 0)
      owner_row = p.send_rows[lay.dense0[peer] + (static_cast<int64_t>(key) - lay.row0[peer])];
    float bacc = 0.0f;
    for (int c0 = 0; c0 < nchunks; c0 += LPR * CPL) {
      RVec<VEC> acc[CPL];
#pragma unroll
      for (int u = 0; u < CPL; ++u)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[u].v[i] = 0.0f;
      uint32_t e0 = start;
      while (__any(e0 < end)) {
        // four entries at a time, added in order
        uint32_t cc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cc[q] = e0 + q < end ? p.vals[e0 + q] : kCodeNone;
        RVec<VEC> x[4][CPL];
        float sg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int u = 0; u < CPL; ++u) {
            const int chunk = c0 + u * LPR + lig;
            x[q][u] = wire_entry<VEC>(p, chunk < nchunks ? cc[q] : kCodeNone, chunk, sg[q]);
          }
          sg[q] = cc[q] == kCodeNone ? 0.0f : ((cc[q] & kCodeNeg) ? -1.0f : 1.0f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (sg[q] != 0.0f) {
#pragma unroll
            for (int u = 0; u < CPL; ++u)
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[u].v[i] += sg[q] * x[q][u].v[i];
          }
          if (c0 == 0 && lig == 0) bacc += wire_entry_bias(p, cc[q]);
        }
        e0 += 4;
      }
#pragma unroll
      for (int u = 0; u < CPL; ++u) {
        const int chunk = c0 + u * LPR + lig;
        if (!active || chunk >= nchunks) continue;
        for (uint32_t i = 0; i < npc; ++i) {     // a long segment: its piece sums, in piece order
          const RVec<VEC> x = rload<VEC>(p.partial + static_cast<int64_t>(lbase + i) * D + chunk * VEC);
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[u].v[k] += x.v[k];
        }
        RVec<VEC> g;
#pragma unroll
        for (int k = 0; k < VEC; ++k) g.v[k] = acc[u].v[k] * scale;
        rstore<VEC>(p.wire + (w0 + jj) * D + chunk * VEC, g);
      }
    }
    if (active && lig == 0) {
      for (uint32_t i = 0; i < npc; ++i) bacc += p.partial[p.max_pieces * D + lbase + i];
      p.wire[(w0 + cnt) * D + jj] = bacc * scale;
      reinterpret_cast<int32_t*>(p.wire)[(w0 + cnt + extra) * D + jj] = static_cast<int32_t>(owner_row);
    }
  }
}

// owner side: the received blocks as (row key, code = wire row of the gradient) entries, and the
// bias gradients as an array indexed like the wire rows
__global__ __launch_bounds__(kBlock) void wire_entries_kernel(const float* __restrict__ wire, int dim,
                                                              const WireBlocks w, int64_t num_rows,
                                                              uint32_t* __restrict__ ekeys,
                                                              uint32_t* __restrict__ ecodes,
                                                              float* __restrict__ gbias,
                                                              int32_t* __restrict__ status) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= w.first[w.world]) return;
  const int p = block_of(w, e);
  const int64_t jj = e - w.first[p];
  const int64_t cnt = w.first[p + 1] - w.first[p];
  const int64_t extra = (cnt + dim - 1) / dim;
  const int64_t w0 = w.row0[p];
  int64_t r = reinterpret_cast<const int32_t*>(wire)[(w0 + cnt + extra) * dim + jj];
  uint32_t code = static_cast<uint32_t>(w0 + jj);
  if (r < 0 || r >= num_rows) {
    if (status) atomicOr(status, 1);
    r = 0;
    code = kCodeNone;
  }
  ekeys[e] = static_cast<uint32_t>(r);
  ecodes[e] = code;
  gbias[w0 + jj] = wire[(w0 + cnt) * dim + jj];
}

}  // namespace

int validate_tables(const collie_mf_tables* t);
int validate_batch(const collie_mf_batch* b);

namespace {

// the two Hyper fields that depend on the step count (also what a step schedule holds)
void step_scalars(const collie_optim_cfg* cfg, float& neg_lr, float& bc2_sqrt) {
  const double lr = cfg->lr;
  bc2_sqrt = 1.0f;
  switch (cfg->algo) {
    case COLLIE_OPT_ADAGRAD:
      neg_lr = static_cast<float>(
          -(lr / (1.0 + (cfg->step - 1) * static_cast<double>(cfg->lr_decay))));
      break;
    case COLLIE_OPT_SPARSE_ADAM: {
      const double bc1 = 1.0 - pow(static_cast<double>(cfg->beta1), cfg->step);
      const double bc2 = 1.0 - pow(static_cast<double>(cfg->beta2), cfg->step);
      neg_lr = static_cast<float>(-(lr * sqrt(bc2) / bc1));
      break;
    }
    case COLLIE_OPT_ADAM: {
      const double bc1 = 1.0 - pow(static_cast<double>(cfg->beta1), cfg->step);
      const double bc2 = 1.0 - pow(static_cast<double>(cfg->beta2), cfg->step);
      neg_lr = static_cast<float>(-(lr / bc1));
      bc2_sqrt = static_cast<float>(sqrt(bc2));
      break;
    }
    default:  // COLLIE_OPT_SGD
      neg_lr = static_cast<float>(-lr);
  }
}

int make_hyper(Hyper& h, const collie_optim_cfg* cfg, int dense, bool need_s1, float* s1,
               float* s2) {
  COLLIE_REQUIRE(cfg->step >= 1, "optimizer step must be >= 1 (got %d)", cfg->step);
  h.algo = cfg->algo;
  h.dense = dense ? 1 : 0;
  h.eps = cfg->eps;
  h.weight_decay = cfg->weight_decay;
  h.beta2 = cfg->beta2;
  h.c1 = static_cast<float>(1.0 - static_cast<double>(cfg->beta1));
  h.c2 = static_cast<float>(1.0 - static_cast<double>(cfg->beta2));
  (void)need_s1;
  switch (cfg->algo) {
    case COLLIE_OPT_SGD:
      break;
    case COLLIE_OPT_ADAGRAD:
      COLLIE_REQUIRE(s1 != nullptr, "Adagrad needs state1 = state_sum");
      break;
    case COLLIE_OPT_SPARSE_ADAM:
      COLLIE_REQUIRE(!dense,
                     "SparseAdam does not support dense gradients, please consider Adam instead");
      COLLIE_REQUIRE(s1 && s2, "SparseAdam needs state1 = exp_avg, state2 = exp_avg_sq");
      break;
    case COLLIE_OPT_ADAM:
      COLLIE_REQUIRE(dense,
                     "Adam does not support sparse gradients, please consider SparseAdam instead");
      COLLIE_REQUIRE(s1 && s2, "Adam needs state1 = exp_avg, state2 = exp_avg_sq");
      break;
    default:
      COLLIE_REQUIRE(false, "%d is not a valid optimizer!", cfg->algo);
  }
  step_scalars(cfg, h.neg_lr, h.bc2_sqrt);
  if (!dense)
    COLLIE_REQUIRE(cfg->weight_decay == 0.0f,
                   "weight_decay option is not compatible with sparse gradients");
  return 0;
}

unsigned grid_for(int64_t ids) {
  int64_t blocks = (ids + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > kUpdateBlocks) blocks = kUpdateBlocks;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

unsigned sweep_grid(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

// widest vector every row start of every buffer the kernel touches is aligned for
int rows_vec(const UpdParams& p) {
  auto ok = [&](int v) {
    if (p.D % v) return false;
    const size_t a = sizeof(float) * v;
    auto al = [&](const void* q) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % a == 0; };
    for (int i = 0; i < 2; ++i)
      if (!(al(p.d[i].table) && al(p.d[i].s1) && al(p.d[i].s2) && al(p.d[i].stage))) return false;
    return al(p.vals_out);
  };
  return ok(4) ? 4 : (ok(2) ? 2 : 1);
}

template <bool EXPORT, int VEC>
void launch_piece_lpr(int lpr, unsigned grid, hipStream_t stream, const UpdParams& p) {
  switch (lpr) {
    case 1: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 1>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 2>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 4>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 8>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 16>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 32>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((piece_rows_kernel<EXPORT, VEC, 64>), dim3(grid), dim3(kBlock), 0, stream, p); break;
  }
}

// parameter + optimizer state of `rows` rows far beyond the 256 MB Infinity Cache: the update streams
// them (non-temporal loads / stores), a cached copy would only evict what the next kernel reads
bool beyond_caches(int64_t rows, int dim) {
  return rows * dim * int64_t{12} > (int64_t{1} << 30);
}

template <bool EXPORT, int VEC>
void launch_emb_lpr(int lpr, unsigned grid, hipStream_t stream, const UpdParams& p) {
  if (!EXPORT && VEC == 4 && p.stream_rows && (lpr == 8 || lpr == 16 || lpr == 32)) {
    if (lpr == 8) hipLaunchKernelGGL((emb_rows_kernel<false, 4, 8, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    else if (lpr == 16) hipLaunchKernelGGL((emb_rows_kernel<false, 4, 16, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    else hipLaunchKernelGGL((emb_rows_kernel<false, 4, 32, true>), dim3(grid), dim3(kBlock), 0, stream, p);
    return;
  }
  switch (lpr) {
    case 1: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 1>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 2>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 4>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 8>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 16>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    case 32: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 32>), dim3(grid), dim3(kBlock), 0, stream, p); break;
    default: hipLaunchKernelGGL((emb_rows_kernel<EXPORT, VEC, 64>), dim3(grid), dim3(kBlock), 0, stream, p); break;
  }
}

// segs: upper bound on the number of segments
template <bool EXPORT>
int launch_emb(const UpdParams& p, int64_t segs, hipStream_t stream) {
  const int vec = rows_vec(p);
  const int nchunks = (p.D + vec - 1) / vec;
  int lpr = 1;
  while (lpr < nchunks && lpr < kWave) lpr *= 2;
  const int gpw = kWave / lpr;
  int64_t blocks = (segs + gpw * kWavesPerBlock - 1) / (gpw * kWavesPerBlock);
  if (blocks > 4 * kUpdateBlocks) blocks = 4 * kUpdateBlocks;
  if (blocks < 1) blocks = 1;
  const unsigned grid = static_cast<unsigned>(blocks);
  // long segments first: at most max_pieces_for(3B) pieces, usually none (the grid-stride loop
  // reads the piece count on the device)
  int64_t pblocks = (max_pieces_for(segs) + gpw * kWavesPerBlock - 1) / (gpw * kWavesPerBlock);
  if (pblocks > 2048) pblocks = 2048;
  const unsigned pgrid = static_cast<unsigned>(pblocks);
  if (small_coalesce(segs)) {
    // no pieces exist (rowsort.hip): the row kernel sums every segment itself
  } else if (vec == 4) {
    launch_piece_lpr<EXPORT, 4>(lpr, pgrid, stream, p);
  } else if (vec == 2) {
    launch_piece_lpr<EXPORT, 2>(lpr, pgrid, stream, p);
  } else {
    launch_piece_lpr<EXPORT, 1>(lpr, pgrid, stream, p);
  }
  if (vec == 4) launch_emb_lpr<EXPORT, 4>(lpr, grid, stream, p);
  else if (vec == 2) launch_emb_lpr<EXPORT, 2>(lpr, grid, stream, p);
  else launch_emb_lpr<EXPORT, 1>(lpr, grid, stream, p);
  return check_launch("emb_rows_kernel");
}

unsigned bias_grid(int64_t segs, int segs_per_wave) {
  int64_t blocks = (segs + segs_per_wave * kWavesPerBlock - 1) / (segs_per_wave * kWavesPerBlock);
  if (blocks > 4 * kUpdateBlocks) blocks = 4 * kUpdateBlocks;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

// entries: upper bound of the segments; nkeys: size of the key space they fall into.  Sparse batches
// (at most two entries per key on average) take the lane-per-segment kernel.
template <bool EXPORT>
int launch_bias_rows(const UpdParams& p, int64_t entries, int64_t nkeys, hipStream_t stream) {
  if (2 * nkeys >= entries)
    hipLaunchKernelGGL(bias_rows_kernel<EXPORT>, dim3(bias_grid(entries, kWave)), dim3(kBlock), 0,
                       stream, p);
  else
    hipLaunchKernelGGL(bias_rows_group_kernel<EXPORT>, dim3(bias_grid(entries, 4)), dim3(kBlock), 0,
                       stream, p);
  return check_launch("bias_rows_kernel");
}

unsigned piece_bias_grid(int64_t entries) {
  int64_t blocks = (max_pieces_for(entries) + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > 2048) blocks = 2048;
  return static_cast<unsigned>(blocks);
}

void fill_sorted(UpdParams& p, const SortWs& s, int64_t nkeys, uint32_t split) {
  p.keys = sorted_keys(s, nkeys);
  p.vals = sorted_vals(s, nkeys);
  p.seg_start = s.seg_start;
  p.nseg = s.nseg;
  p.long_base = s.long_base;
  p.piece_start = s.piece_start;
  p.piece_end = s.piece_end;
  p.partial = s.partial;
  p.split = split;
}

}  // namespace
}  // namespace collie

using namespace collie;

namespace collie {
namespace {

// the TableDesc pair of one kind of table (embeddings: is_bias = 0; biases: 1, or 2 after an
// explicit loss) for the sides in table_mask, and which of them need a sweep afterwards
struct KindPlan {
  TableDesc d[2];
  bool sweep[2], touched[2];
  int D;
};

int plan_kind(KindPlan& k, const collie_mf_tables* t, const TrainWs& ws, int table_mask, int is_bias,
              int dense, bool from_device, const collie_optim_cfg* cfg, float* const* state1,
              float* const* state2, int32_t* const* row_stamp) {
  memset(&k, 0, sizeof(k));
  k.D = is_bias ? 1 : t->dim;
  for (int side = 0; side < 2; ++side) {
    if (!(table_mask & (1 << side))) continue;
    TableDesc& d = k.d[side];
    if (is_bias) {
      d.table = side == COLLIE_SIDE_USER ? t->user_bias : t->item_bias;
      // pairwise losses: d loss / d user_bias == 0 exactly; explicit losses (is_bias == 2): g
      d.stage = (side == COLLIE_SIDE_USER && is_bias != 2) ? nullptr : ws.gcoef;
    } else {
      d.table = side == COLLIE_SIDE_USER ? t->user_emb : t->item_emb;
      d.stage = side == COLLIE_SIDE_USER ? ws.stage_u : ws.stage_i;
    }
    d.s1 = cfg[side].algo == COLLIE_OPT_SGD ? nullptr : state1[side];
    d.s2 = (cfg[side].algo == COLLIE_OPT_SPARSE_ADAM || cfg[side].algo == COLLIE_OPT_ADAM)
               ? state2[side]
               : nullptr;
    collie_optim_cfg c = cfg[side];
    if (from_device) c.step = 1;
    if (make_hyper(d.h, &c, dense, true, d.s1, d.s2)) return -1;
    // dense algorithms whose zero-gradient update is not the identity sweep the other rows
    k.sweep[side] = dense && (cfg[side].algo == COLLIE_OPT_ADAM || cfg[side].weight_decay != 0.0f);
    if (k.sweep[side]) {
      COLLIE_REQUIRE(row_stamp[side] != nullptr,
                     "dense Adam / weight decay need row_stamp = int32[rows] scratch");
      d.row_stamp = row_stamp[side];
    }
    // SparseAdam moves looked-up rows even with a zero gradient: the K-1 other negatives
    k.touched[side] =
        !dense && side == COLLIE_SIDE_ITEM && cfg[side].algo == COLLIE_OPT_SPARSE_ADAM;
    if (k.touched[side]) d.seg_stamp = ws.seg_item;
  }
  return 0;
}

// rows of 16-byte chunks: 4, 8 or 16 lanes per row (at most 4 chunks per lane and pass)
template <bool STREAM>
void launch_touched_rows_t(int nchunks, unsigned grid, hipStream_t stream, float* table, float* s1,
                           float* s2, const uint8_t* touch, const int32_t* seg, int64_t rows, int D,
                           int step, const Hyper& h, const StepDyn* dyn, int dyn_idx, int span) {
  if (nchunks <= 16)
    hipLaunchKernelGGL((touched_rows_kernel<4, STREAM>), dim3(grid), dim3(kBlock), 0, stream, table,
                       s1, s2, touch, seg, rows, D, step, h, dyn, dyn_idx, span);
  else if (nchunks <= 32)
    hipLaunchKernelGGL((touched_rows_kernel<8, STREAM>), dim3(grid), dim3(kBlock), 0, stream, table,
                       s1, s2, touch, seg, rows, D, step, h, dyn, dyn_idx, span);
  else
    hipLaunchKernelGGL((touched_rows_kernel<16, STREAM>), dim3(grid), dim3(kBlock), 0, stream, table,
                       s1, s2, touch, seg, rows, D, step, h, dyn, dyn_idx, span);
}

void launch_touched_rows(int nchunks, bool stream_rows, unsigned grid, hipStream_t stream,
                         float* table, float* s1, float* s2, const uint8_t* touch,
                         const int32_t* seg, int64_t rows, int D, int step, const Hyper& h,
                         const StepDyn* dyn, int dyn_idx, int span) {
  if (stream_rows)
    launch_touched_rows_t<true>(nchunks, grid, stream, table, s1, s2, touch, seg, rows, D, step, h,
                                dyn, dyn_idx, span);
  else
    launch_touched_rows_t<false>(nchunks, grid, stream, table, s1, s2, touch, seg, rows, D, step, h,
                                 dyn, dyn_idx, span);
}

// rows without a gradient entry (K5 / K5b) of one kind, after its row kernel
int sweep_kind(const KindPlan& k, const collie_mf_tables* t, const TrainWs& ws, int step,
               const StepDyn* dyn, int dyn_base, hipStream_t stream) {
  for (int side = 0; side < 2; ++side) {
    const TableDesc& d = k.d[side];
    const int64_t rows = side == COLLIE_SIDE_USER ? t->num_users : t->num_items;
    if (k.sweep[side]) {
      hipLaunchKernelGGL(dense_sweep_kernel, dim3(sweep_grid(rows * k.D)), dim3(kBlock), 0, stream,
                         d.table, d.s1, d.s2, d.row_stamp, rows, k.D, step, d.h, dyn,
                         dyn_base + side);
      if (check_launch("dense_sweep_kernel")) return -2;
    }
    if (k.touched[side]) {
      auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
      if (k.D % 4 == 0 && al16(d.table) && al16(d.s1) && al16(d.s2)) {
        // rows of 16-byte chunks: 4, 8 or 16 lanes per row (at most 4 chunks per lane and pass)
        const int nchunks = k.D / 4;
        const unsigned grid = sweep_grid(rows * 4);
        const int span = rows * 4 <= 4096 * static_cast<int64_t>(kBlock) ? 16 : kWave;
        launch_touched_rows(nchunks, beyond_caches(rows, k.D), grid, stream, d.table, d.s1, d.s2,
                            ws.touch_item, ws.seg_item, rows, k.D, step, d.h, dyn, dyn_base + side,
                            span);
      } else {
        hipLaunchKernelGGL(touched_sweep_kernel, dim3(sweep_grid(rows * k.D)), dim3(kBlock), 0,
                           stream, d.table, d.s1, d.s2, ws.touch_item, ws.seg_item, rows, k.D,
                           step, d.h, dyn, dyn_base + side);
      }
      if (check_launch("touched_sweep_kernel")) return -2;
    }
  }
  return 0;
}

// The sort-free kernel serves sparse algorithms on coalesces of at most kSmallMax entries whose
// embedding rows are float4 chunks, at most 64 of them (dim <= 256).
bool small_rows_ok(const collie_mf_tables* t, const collie_mf_batch* b, int dense) {
  return !dense && 3 * b->batch <= kSmallRowsMax && t->dim % 4 == 0 && t->dim <= 4 * kWave;
}

bool small_descs_ok(const KindPlan* emb) {
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  if (!emb) return true;
  for (int side = 0; side < 2; ++side) {
    const TableDesc& d = emb->d[side];
    if (!(al16(d.table) && al16(d.s1) && al16(d.s2) && al16(d.stage))) return false;
  }
  return true;
}

int launch_small_rows(const collie_mf_tables* t, const collie_mf_batch* b, const TrainWs& ws,
                      const KindPlan* emb, const KindPlan* bias, int step, const StepDyn* dyn,
                      const float* grad_scale, hipStream_t stream) {
  SmallParams q;
  memset(&q, 0, sizeof(q));
  for (int side = 0; side < 2; ++side) {
    if (emb) q.e[side] = emb->d[side];
    if (bias) q.z[side] = bias->d[side];
  }
  q.ekeys = ws.sort.ekeys;
  q.ecodes = ws.sort.ecodes;
  q.grad_scale = grad_scale;
  q.dyn = dyn;
  q.split = static_cast<uint32_t>(t->num_users);
  q.m = static_cast<int>(3 * b->batch);
  q.D = t->dim;
  q.step = step;
  const int nchunks = emb ? t->dim / 4 : 1;
  int lpr = 16;                      // at least 16 lanes per entry: the scan is split among them
  while (lpr < nchunks) lpr *= 2;
  // every workgroup stages all m keys: beyond a few thousand entries 1,024-thread workgroups (four
  // times fewer of them) keep that traffic down
  const bool big = q.m > kSmallMax;
  const int threads = big ? 1024 : kBlock;
  const int per_block = (threads / kWave) * (kWave / lpr);
  const dim3 grid(static_cast<unsigned>((q.m + per_block - 1) / per_block));
  if (big) {
    switch (lpr) {
      case 16: hipLaunchKernelGGL((small_rows_kernel<16, 1024>), grid, dim3(1024), 0, stream, q); break;
      case 32: hipLaunchKernelGGL((small_rows_kernel<32, 1024>), grid, dim3(1024), 0, stream, q); break;
      default: hipLaunchKernelGGL((small_rows_kernel<64, 1024>), grid, dim3(1024), 0, stream, q); break;
    }
  } else {
    switch (lpr) {
      case 16: hipLaunchKernelGGL((small_rows_kernel<16, kBlock>), grid, dim3(kBlock), 0, stream, q); break;
      case 32: hipLaunchKernelGGL((small_rows_kernel<32, kBlock>), grid, dim3(kBlock), 0, stream, q); break;
      default: hipLaunchKernelGGL((small_rows_kernel<64, kBlock>), grid, dim3(kBlock), 0, stream, q); break;
    }
  }
  return check_launch("small_rows_kernel");
}

}  // namespace
}  // namespace collie

extern "C" int collie_mf_train_update_needs_segments(const collie_mf_tables* t,
                                                     const collie_mf_batch* b, int32_t dense) {
  if (validate_tables(t) || validate_batch(b)) return -1;
  return small_rows_ok(t, b, dense) ? 0 : 1;
}

extern "C" int collie_mf_train_update(const collie_mf_tables* t, const collie_mf_batch* b,
                                      void* workspace, size_t workspace_bytes, int32_t table_mask,
                                      int32_t is_bias, int32_t dense, int32_t step_id,
                                      const collie_optim_cfg* cfg, float* const* state1,
                                      float* const* state2, int32_t* const* row_stamp,
                                      const float* grad_scale, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(cfg && state1 && state2 && row_stamp, "update: NULL argument array");
  COLLIE_REQUIRE((table_mask & 3) != 0 && (table_mask & ~3) == 0, "bad table_mask %d", table_mask);
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  // a replayed step: the step id and the step-dependent scalars of cfg come from the schedule in
  // the workspace (cfg[].step is ignored; every other field of cfg is read now, as usual)
  const bool from_device = step_id == COLLIE_STEP_DEVICE;
  COLLIE_REQUIRE(step_id >= 1 || from_device, "step_id must be >= 1 (got %d)", step_id);
  const StepDyn* dyn = from_device ? ws.dyn_cur : nullptr;
  const int dyn_base = is_bias ? 2 : 0;
  KindPlan k;
  if (plan_kind(k, t, ws, table_mask, is_bias, dense, from_device, cfg, state1, state2, row_stamp))
    return -1;
  if (small_rows_ok(t, b, dense)) {
    if (small_descs_ok(is_bias ? nullptr : &k)) {
      if (launch_small_rows(t, b, ws, is_bias ? nullptr : &k, is_bias ? &k : nullptr, step_id, dyn,
                            grad_scale, stream))
        return -2;
      return sweep_kind(k, t, ws, step_id, dyn, dyn_base, stream);
    }
    // collie_mf_train_update_needs_segments answered "no" for this batch (it cannot see the state
    // pointers), but a table / state / staging row is not 16-byte aligned, so the sorted path below
    // runs: coalesce here instead of reading segments nobody built for this step
    if (sort_and_segment(ws.sort, t->num_users + t->num_items, static_cast<uint32_t>(t->num_users),
                         stream))
      return -1;
  }
  UpdParams p;
  memset(&p, 0, sizeof(p));
  fill_sorted(p, ws.sort, t->num_users + t->num_items, static_cast<uint32_t>(t->num_users));
  p.grad_scale = grad_scale;
  p.step = step_id;
  p.dyn = dyn;
  p.dyn_base = dyn_base;
  p.D = k.D;
  p.d[0] = k.d[0];
  p.d[1] = k.d[1];
  p.stream_rows = !is_bias && beyond_caches(t->num_users + t->num_items, t->dim);
  if (is_bias) {
    if (!small_coalesce(3 * b->batch))
      hipLaunchKernelGGL(piece_bias_kernel<false>, dim3(piece_bias_grid(3 * b->batch)), dim3(kBlock),
                         0, stream, p);
    if (launch_bias_rows<false>(p, 3 * b->batch, t->num_users + t->num_items, stream)) return -2;
  } else if (launch_emb<false>(p, 3 * b->batch, stream)) {
    return -2;
  }
  return sweep_kind(k, t, ws, step_id, dyn, dyn_base, stream);
}

// Both kinds of table in one call: cfg / state1 / state2 / row_stamp are indexed [user_emb,
// item_emb, user_bias, item_bias]; cfg[i].algo < 0: table i is not updated.
extern "C" int collie_mf_train_update_all(const collie_mf_tables* t, const collie_mf_batch* b,
                                          void* workspace, size_t workspace_bytes,
                                          int32_t bias_mode, int32_t dense, int32_t step_id,
                                          const collie_optim_cfg* cfg, float* const* state1,
                                          float* const* state2, int32_t* const* row_stamp,
                                          const float* grad_scale, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(cfg && state1 && state2 && row_stamp, "update: NULL argument array");
  COLLIE_REQUIRE(bias_mode == 1 || bias_mode == 2, "bias_mode must be 1 or 2 (got %d)", bias_mode);
  int mask[2] = {0, 0};
  for (int i = 0; i < 4; ++i)
    if (cfg[i].algo >= 0) mask[i / 2] |= 1 << (i % 2);
  COLLIE_REQUIRE(mask[0] || mask[1], "update_all: no table to update");
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  const bool from_device = step_id == COLLIE_STEP_DEVICE;
  COLLIE_REQUIRE(step_id >= 1 || from_device, "step_id must be >= 1 (got %d)", step_id);
  const StepDyn* dyn = from_device ? ws.dyn_cur : nullptr;
  KindPlan ke, kz;
  if (mask[0] && plan_kind(ke, t, ws, mask[0], 0, dense, from_device, cfg, state1, state2, row_stamp))
    return -1;
  if (mask[1] && plan_kind(kz, t, ws, mask[1], bias_mode, dense, from_device, cfg + 2, state1 + 2,
                           state2 + 2, row_stamp + 2))
    return -1;
  if (small_rows_ok(t, b, dense) && small_descs_ok(mask[0] ? &ke : nullptr)) {
    if (launch_small_rows(t, b, ws, mask[0] ? &ke : nullptr, mask[1] ? &kz : nullptr, step_id, dyn,
                          grad_scale, stream))
      return -2;
    if (mask[0] && sweep_kind(ke, t, ws, step_id, dyn, 0, stream)) return -2;
    if (mask[1] && sweep_kind(kz, t, ws, step_id, dyn, 2, stream)) return -2;
    return 0;
  }
  // general path: the two ordinary calls (they need collie_mf_train_build_segments first; for a
  // small batch with unaligned rows collie_mf_train_update builds the segments itself)
  if (mask[0] && collie_mf_train_update(t, b, workspace, workspace_bytes, mask[0], 0, dense, step_id,
                                        cfg, state1, state2, row_stamp, grad_scale, stream_))
    return -2;
  if (mask[1] && collie_mf_train_update(t, b, workspace, workspace_bytes, mask[1], bias_mode, dense,
                                        step_id, cfg + 2, state1 + 2, state2 + 2, row_stamp + 2,
                                        grad_scale, stream_))
    return -2;
  return 0;
}

// The whole loop body of the trainers (trainer.py:458-470) for one batch in ONE call: forward,
// coalesce (where the update needs it), every table update.  For launch-bound batch sizes the host
// side of a step is then one FFI call.
extern "C" int collie_mf_train_step(const collie_mf_tables* t, const collie_mf_batch* b,
                                    const collie_loss_cfg* loss, void* workspace,
                                    size_t workspace_bytes, int32_t step_id, int32_t dense,
                                    float* loss_out, int32_t* status, const collie_optim_cfg* cfg,
                                    float* const* state1, float* const* state2,
                                    int32_t* const* row_stamp, collie_stream_t stream) {
  int rc = collie_mf_train_forward(t, b, loss, workspace, workspace_bytes, 1, step_id, loss_out,
                                   nullptr, nullptr, status, stream);
  if (rc) return rc;
  if (!small_rows_ok(t, b, dense)) {
    rc = collie_mf_train_build_segments(t, b, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return collie_mf_train_update_all(t, b, workspace, workspace_bytes, 1, dense, step_id, cfg, state1,
                                    state2, row_stamp, nullptr, stream);
}

// ---- captured-step replay --------------------------------------------------------------------------
namespace collie {
namespace {

__global__ void dyn_reset_kernel(DynCtl* ctl, StepDyn* cur, const StepDyn* sched, int n) {
  ctl->cursor = 0;
  ctl->n = n;
  ctl->overrun = 0;
  *cur = sched[0];
}

// last node of a captured step: move `cur` to the next schedule entry and, when that step starts a
// new cycle of stamp tags, clear the stamps (what the forward entry point does from the host for a
// step id it is given by value).  One workgroup: the clear runs once in 255 steps.
__global__ __launch_bounds__(1024) void dyn_advance_kernel(DynCtl* ctl, StepDyn* cur,
                                                           const StepDyn* sched,
                                                           uint8_t* touch_item, int64_t num_items,
                                                           const float* partials, int nblocks,
                                                           float inv_b, float* loss_out,
                                                           int32_t* status) {
  // the loss the forward deferred (COLLIE_DEFER_LOSS): one launch less in the replayed step
  if (loss_out) finalize_loss_block(partials, nblocks, inv_b, loss_out, status);
  __shared__ StepDyn next;
  __shared__ int next_at;
  if (threadIdx.x == 0) {
    int c = ctl->cursor + 1;
    if (c >= ctl->n) {
      c = ctl->n - 1;
      ctl->overrun = 1;
    }
    next_at = c;
    next = sched[c];
  }
  __syncthreads();
  if (next.tag == touch_tag(0)) {
    const int64_t words = num_items / 16;
    uint4* w = reinterpret_cast<uint4*>(touch_item);       // Carver blocks are 256-byte aligned
    for (int64_t i = threadIdx.x; i < words; i += blockDim.x) w[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = words * 16 + threadIdx.x; i < num_items; i += blockDim.x) touch_item[i] = 0;
  }
  if (threadIdx.x == 0) {
    ctl->cursor = next_at;
    *cur = next;
  }
}

}  // namespace
}  // namespace collie

extern "C" int collie_mf_train_schedule_steps(const collie_mf_tables* t, const collie_mf_batch* b,
                                              void* workspace, size_t workspace_bytes,
                                              int32_t first_step_id, int32_t n_steps,
                                              const collie_optim_cfg* cfg, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(cfg != nullptr, "schedule_steps: cfg is NULL");
  COLLIE_REQUIRE(first_step_id >= 1, "first_step_id must be >= 1 (got %d)", first_step_id);
  COLLIE_REQUIRE(n_steps >= 1 && n_steps <= kSchedLen, "n_steps must be in [1, %d] (got %d)",
                 kSchedLen, n_steps);
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  std::vector<StepDyn> host(static_cast<size_t>(n_steps));
  for (int s = 0; s < n_steps; ++s) {
    StepDyn& e = host[static_cast<size_t>(s)];
    memset(&e, 0, sizeof(e));
    e.step = first_step_id + s;
    e.tag = touch_tag(e.step);
    for (int i = 0; i < 4; ++i) {
      e.bc2_sqrt[i] = 1.0f;
      if (cfg[i].algo < 0) continue;                // table not updated by the captured step
      COLLIE_REQUIRE(cfg[i].algo <= COLLIE_OPT_ADAM, "%d is not a valid optimizer!", cfg[i].algo);
      COLLIE_REQUIRE(cfg[i].step >= 1, "optimizer step must be >= 1 (got %d)", cfg[i].step);
      collie_optim_cfg c = cfg[i];
      c.step = cfg[i].step + s;
      step_scalars(&c, e.neg_lr[i], e.bc2_sqrt[i]);
    }
  }
  // blocks the host until the copy is done (`host` is pageable and dies with this call)
  COLLIE_CHECK_HIP(hipMemcpyWithStream(ws.dyn_sched, host.data(), host.size() * sizeof(StepDyn),
                                       hipMemcpyHostToDevice, stream));
  if (first_step_id % kTouchCycle == 0)
    COLLIE_CHECK_HIP(hipMemsetAsync(ws.touch_item, 0, static_cast<size_t>(t->num_items), stream));
  hipLaunchKernelGGL(dyn_reset_kernel, dim3(1), dim3(1), 0, stream, ws.dyn_ctl, ws.dyn_cur,
                     ws.dyn_sched, n_steps);
  return check_launch("dyn_reset_kernel");
}

extern "C" int collie_mf_train_advance_step(const collie_mf_tables* t, const collie_mf_batch* b,
                                            void* workspace, size_t workspace_bytes,
                                            float* loss_out, int32_t* status,
                                            collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  int64_t nblocks = (b->batch + 3) / 4;   // the forward's grid: one wave per sample, 4 per block
  if (nblocks > kMaxLossBlocks) nblocks = kMaxLossBlocks;
  hipLaunchKernelGGL(dyn_advance_kernel, dim3(1), dim3(1024), 0, stream, ws.dyn_ctl, ws.dyn_cur,
                     ws.dyn_sched, ws.touch_item, t->num_items, ws.loss_partials,
                     static_cast<int>(nblocks), 1.0f / static_cast<float>(b->batch), loss_out,
                     status);
  return check_launch("dyn_advance_kernel");
}

extern "C" int collie_mf_train_export_grad(const collie_mf_tables* t, const collie_mf_batch* b,
                                           void* workspace, size_t workspace_bytes, int32_t side,
                                           int32_t is_bias, int32_t step_id, int64_t* rows_out,
                                           float* vals_out, int32_t* count_out,
                                           uint8_t* touched_out, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  COLLIE_REQUIRE(side == COLLIE_SIDE_USER || side == COLLIE_SIDE_ITEM, "bad side %d", side);
  COLLIE_REQUIRE(rows_out && vals_out && count_out, "export: NULL output");
  TrainWs ws = carve_train_ws(workspace, b->batch, b->num_neg, t->dim, t->num_users, t->num_items);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, workspace_bytes);
  UpdParams p;
  memset(&p, 0, sizeof(p));
  fill_sorted(p, ws.sort, t->num_users + t->num_items, static_cast<uint32_t>(t->num_users));
  p.D = is_bias ? 1 : t->dim;
  p.export_side = side;
  p.rows_out = rows_out;
  p.vals_out = vals_out;
  p.count_out = count_out;
  if (is_bias) {
    p.d[0].stage = nullptr;
    p.d[1].stage = ws.gcoef;
  } else {
    p.d[0].stage = ws.stage_u;
    p.d[1].stage = ws.stage_i;
  }
  if (is_bias) {
    if (!small_coalesce(3 * b->batch))
      hipLaunchKernelGGL(piece_bias_kernel<true>, dim3(piece_bias_grid(3 * b->batch)), dim3(kBlock),
                       0, stream, p);
    if (launch_bias_rows<true>(p, 3 * b->batch, t->num_users + t->num_items, stream)) return -2;
  } else if (launch_emb<true>(p, 3 * b->batch, stream)) {
    return -2;
  }
  if (touched_out && side == COLLIE_SIDE_ITEM) {
    hipLaunchKernelGGL(export_touched_kernel,
                       dim3(static_cast<unsigned>((t->num_items + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, ws.touch_item, t->num_items, step_id, touched_out);
    return check_launch("export_touched_kernel");
  }
  return 0;
}

extern "C" size_t collie_sparse_rows_workspace_bytes(int64_t m, int32_t dim) {
  if (m < 1 || dim < 1) return 0;
  Carver c(nullptr);
  SortWs s;
  carve_sort_ws(c, s, m, dim);
  return align_up(c.off, 256);
}

// ---- row-sharded step ----------------------------------------------------------------------------------
namespace collie {
namespace {

struct ExportWs {
  uint32_t* seg_of;     // [3B]
  GradLayout* lay;
  size_t bytes;
};

ExportWs carve_export_ws(void* base, int64_t B) {
  Carver c(base);
  ExportWs w;
  w.seg_of = c.take<uint32_t>(3 * B);
  w.lay = c.take<GradLayout>(1);
  w.bytes = align_up(c.off, 256);
  return w;
}

template <int VEC>
void launch_wire(int lpr, unsigned pgrid, unsigned grid, bool pieces, hipStream_t stream,
                 const WireExport& p) {
#define COLLIE_WIRE_CASE(L)                                                                         \
  case L:                                                                                           \
    if (pieces)                                                                                     \
      hipLaunchKernelGGL((wire_piece_kernel<VEC, L>), dim3(pgrid), dim3(kBlock), 0, stream, p);     \
    hipLaunchKernelGGL((wire_rows_kernel<VEC, L>), dim3(grid), dim3(kBlock), 0, stream, p);         \
    break;
  switch (lpr) {
    COLLIE_WIRE_CASE(1)
    COLLIE_WIRE_CASE(2)
    COLLIE_WIRE_CASE(4)
    COLLIE_WIRE_CASE(8)
    COLLIE_WIRE_CASE(16)
    COLLIE_WIRE_CASE(32)
    default:
      if (pieces)
        hipLaunchKernelGGL((wire_piece_kernel<VEC, 64>), dim3(pgrid), dim3(kBlock), 0, stream, p);
      hipLaunchKernelGGL((wire_rows_kernel<VEC, 64>), dim3(grid), dim3(kBlock), 0, stream, p);
  }
#undef COLLIE_WIRE_CASE
}

struct OwnerWs {
  SortWs sort;
  float* gbias;       // [wire rows]
  size_t bytes;
};

OwnerWs carve_owner_ws(void* base, int64_t entries, int64_t wire_rows, int D) {
  Carver c(base);
  OwnerWs w;
  carve_sort_ws(c, w.sort, entries, D);
  w.gbias = c.take<float>(wire_rows);
  w.bytes = align_up(c.off, 256);
  return w;
}

}  // namespace
}  // namespace collie

extern "C" size_t collie_shard_export_workspace_bytes(int64_t batch) {
  if (batch < 1) return 0;
  return carve_export_ws(nullptr, batch).bytes;
}

extern "C" int collie_shard_export_grads(const collie_mf_tables* t, const collie_mf_batch* b,
                                         void* train_workspace, size_t train_workspace_bytes,
                                         int32_t world, int32_t users_local,
                                         const int64_t* lookup_counts, const int64_t* send_rows,
                                         const float* grad_scale, float* wire, int64_t wire_rows,
                                         int64_t* peer_counts, void* workspace,
                                         size_t workspace_bytes, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (validate_tables(t) || validate_batch(b)) return -1;
  if (!users_local)
    COLLIE_REQUIRE(t->num_users == t->num_items && t->user_emb == t->item_emb,
                   "export_grads: the tables must be the compact table under both names");
  COLLIE_REQUIRE(send_rows && wire && peer_counts, "export_grads: NULL pointer");
  const int64_t B = b->batch;
  const int D = t->dim;
  COLLIE_REQUIRE(B < (int64_t{1} << 29), "export_grads: batch too large");
  WireBlocks look;
  if (make_blocks(look, lookup_counts, world, D, 1)) return -1;
  COLLIE_REQUIRE(look.row0[world] <= t->num_items,
                 "export_grads: the lookup blocks (%lld rows) exceed the compact table (%lld rows)",
                 static_cast<long long>(look.row0[world]), static_cast<long long>(t->num_items));
  // at most 3B gradient rows; every peer's block has at most 2 (ceil(c / D)) more rows
  const int64_t need_rows = 3 * B + 2 * ((3 * B + D - 1) / D + world);
  COLLIE_REQUIRE(wire_rows >= need_rows, "export_grads: wire holds %lld rows, need %lld",
                 static_cast<long long>(wire_rows), static_cast<long long>(need_rows));
  TrainWs ws = carve_train_ws(train_workspace, B, b->num_neg, D, t->num_users, t->num_items);
  COLLIE_REQUIRE(train_workspace != nullptr && train_workspace_bytes >= ws.bytes,
                 "workspace too small: need %zu bytes, got %zu", ws.bytes, train_workspace_bytes);
  ExportWs xw = carve_export_ws(workspace, B);
  COLLIE_REQUIRE(workspace != nullptr && workspace_bytes >= xw.bytes,
                 "export workspace too small: need %zu bytes, got %zu", xw.bytes, workspace_bytes);
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  const int64_t m = 3 * B;
  int64_t nkeys = t->num_users;
  uint32_t split = 0xffffffffu, key_sub = 0;
  if (users_local) {
    // the forward kernel's own key space: user rows of the shard, then the compact item rows; the
    // segments of the sort below also serve collie_mf_train_update on the user side
    nkeys = t->num_users + t->num_items;
    split = key_sub = static_cast<uint32_t>(t->num_users);
  } else {
    hipLaunchKernelGGL(wire_rekey_kernel, dim3(static_cast<unsigned>((2 * B + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, ws.sort.ekeys, ws.sort.ecodes, B,
                       static_cast<uint32_t>(t->num_users));
    if (check_launch("wire_rekey_kernel")) return -2;
  }
  if (sort_and_segment(ws.sort, nkeys, split, stream, xw.seg_of)) return -1;
  hipLaunchKernelGGL(wire_layout_kernel, dim3(1), dim3(64), 0, stream, sorted_keys(ws.sort, nkeys),
                     xw.seg_of, ws.sort.nseg, m, look, D, key_sub, xw.lay, peer_counts);
  if (check_launch("wire_layout_kernel")) return -2;
  WireExport p;
  memset(&p, 0, sizeof(p));
  p.keys = sorted_keys(ws.sort, nkeys);
  p.vals = sorted_vals(ws.sort, nkeys);
  p.seg_start = ws.sort.seg_start;
  p.nseg = ws.sort.nseg;
  p.long_base = ws.sort.long_base;
  p.piece_start = ws.sort.piece_start;
  p.piece_end = ws.sort.piece_end;
  p.partial = ws.sort.partial;
  p.max_pieces = ws.sort.max_pieces;
  p.stage_u = ws.stage_u;
  p.stage_i = ws.stage_i;
  p.gcoef = ws.gcoef;
  p.grad_scale = grad_scale;
  p.send_rows = send_rows;
  p.lay = xw.lay;
  p.wire = wire;
  p.key_sub = key_sub;
  p.all_items = users_local ? 1 : 0;
  p.world = world;
  p.D = D;
  const int vec = (D % 4 == 0 && al16(wire) && al16(ws.stage_u) && al16(ws.stage_i)) ? 4 : 1;
  const int nchunks = (D + vec - 1) / vec;
  // the row kernel gives every lane 4 chunks of its row, the piece kernel one chunk per pass
  int lpr = 1;
  while (lpr * 4 < nchunks && lpr < kWave) lpr *= 2;
  const int gpw = kWave / lpr;
  int64_t blocks = (m + gpw * kWavesPerBlock - 1) / (gpw * kWavesPerBlock);
  if (blocks > 4 * kUpdateBlocks) blocks = 4 * kUpdateBlocks;
  int64_t pblocks = (max_pieces_for(m) + gpw * kWavesPerBlock - 1) / (gpw * kWavesPerBlock);
  if (pblocks > 2048) pblocks = 2048;
  const bool pieces = !small_coalesce(m);
  if (vec == 4)
    launch_wire<4>(lpr, static_cast<unsigned>(pblocks), static_cast<unsigned>(blocks), pieces, stream, p);
  else
    launch_wire<1>(lpr, static_cast<unsigned>(pblocks), static_cast<unsigned>(blocks), pieces, stream, p);
  return check_launch("wire_rows_kernel");
}

extern "C" size_t collie_shard_owner_workspace_bytes(int64_t entries, int64_t wire_rows, int32_t dim) {
  if (entries < 0 || wire_rows < 0 || dim < 1) return 0;
  return carve_owner_ws(nullptr, entries < 1 ? 1 : entries, wire_rows < 1 ? 1 : wire_rows, dim).bytes;
}

extern "C" int collie_shard_owner_update(float* emb, float* bias, int64_t num_rows, int32_t dim,
                                         int64_t item_base, const float* wire, int64_t wire_rows,
                                         const int64_t* peer_counts, int32_t world,
                                         const uint8_t* item_touch, int32_t* row_seg,
                                         const collie_optim_cfg* cfg, const collie_optim_cfg* bias_cfg,
                                         float* state1, float* state2, float* bias_state1,
                                         float* bias_state2, void* workspace, size_t workspace_bytes,
                                         int32_t* status, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(emb && bias && cfg && bias_cfg && item_touch && row_seg,
                 "owner_update: NULL pointer");
  COLLIE_REQUIRE(num_rows >= 1 && num_rows < (int64_t{1} << 31) && dim >= 1 && item_base >= 0 &&
                     item_base <= num_rows,
                 "owner_update: bad table shape");
  COLLIE_REQUIRE(cfg->algo != COLLIE_OPT_ADAM && bias_cfg->algo != COLLIE_OPT_ADAM,
                 "Adam does not support sparse gradients, please consider SparseAdam instead");
  COLLIE_REQUIRE(cfg->step == bias_cfg->step, "owner_update: the two optimizers must be at the same step");
  WireBlocks w;
  if (make_blocks(w, peer_counts, world, dim, 2)) return -1;
  const int64_t m = w.first[world];
  COLLIE_REQUIRE(m < (int64_t{1} << 30), "owner_update: too many entries");
  COLLIE_REQUIRE(m == 0 || (wire != nullptr && wire_rows >= w.row0[world]),
                 "owner_update: wire holds %lld rows, need %lld", static_cast<long long>(wire_rows),
                 static_cast<long long>(w.row0[world]));
  OwnerWs ow = carve_owner_ws(workspace, m < 1 ? 1 : m, w.row0[world] < 1 ? 1 : w.row0[world], dim);
  COLLIE_REQUIRE(workspace && workspace_bytes >= ow.bytes,
                 "owner workspace too small: need %zu bytes, got %zu", ow.bytes, workspace_bytes);
  const int step = cfg->step;
  UpdParams p;
  memset(&p, 0, sizeof(p));
  p.D = dim;
  p.step = step;
  p.stream_rows = beyond_caches(num_rows, dim);
  p.d[0].table = emb;
  p.d[0].stage = wire;
  p.d[0].s1 = cfg->algo == COLLIE_OPT_SGD ? nullptr : state1;
  p.d[0].s2 = cfg->algo == COLLIE_OPT_SPARSE_ADAM ? state2 : nullptr;
  p.d[0].seg_stamp = row_seg;
  if (make_hyper(p.d[0].h, cfg, 0, true, p.d[0].s1, p.d[0].s2)) return -1;
  UpdParams q;
  memset(&q, 0, sizeof(q));
  q.D = 1;
  q.step = step;
  q.d[0].table = bias;
  q.d[0].stage = ow.gbias;
  q.d[0].s1 = bias_cfg->algo == COLLIE_OPT_SGD ? nullptr : bias_state1;
  q.d[0].s2 = bias_cfg->algo == COLLIE_OPT_SPARSE_ADAM ? bias_state2 : nullptr;
  q.d[0].seg_stamp = row_seg;
  if (make_hyper(q.d[0].h, bias_cfg, 0, true, q.d[0].s1, q.d[0].s2)) return -1;
  if (m > 0) {
    hipLaunchKernelGGL(wire_entries_kernel, dim3(static_cast<unsigned>((m + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, stream, wire, dim, w, num_rows, ow.sort.ekeys,
                       ow.sort.ecodes, ow.gbias, status);
    if (check_launch("wire_entries_kernel")) return -2;
    if (sort_and_segment(ow.sort, num_rows, 0xffffffffu, stream)) return -1;
    fill_sorted(p, ow.sort, num_rows, 0xffffffffu);
    fill_sorted(q, ow.sort, num_rows, 0xffffffffu);
    if (launch_emb<false>(p, m, stream)) return -2;
    if (!small_coalesce(m))
      hipLaunchKernelGGL(piece_bias_kernel<false>, dim3(piece_bias_grid(m)), dim3(kBlock), 0, stream, q);
    if (launch_bias_rows<false>(q, m, num_rows, stream)) return -2;
  }
  // looked-up item rows without a gradient entry: SparseAdam decays their moments (K5b)
  const int64_t items = num_rows - item_base;
  if (items == 0) return 0;
  auto al16 = [](const void* x) { return reinterpret_cast<uintptr_t>(x) % 16 == 0; };
  if (cfg->algo == COLLIE_OPT_SPARSE_ADAM) {
    float* tb = emb + item_base * dim;
    float* s1 = state1 + item_base * dim;
    float* s2 = state2 + item_base * dim;
    const int32_t* seg = row_seg + item_base;
    if (dim % 4 == 0 && al16(tb) && al16(s1) && al16(s2)) {
      const int nchunks = dim / 4;
      const unsigned grid = sweep_grid(items * 4);
      const int span = items * 4 <= 4096 * static_cast<int64_t>(kBlock) ? 16 : kWave;
      launch_touched_rows(nchunks, beyond_caches(num_rows, dim), grid, stream, tb, s1, s2, item_touch,
                          seg, items, dim, step, p.d[0].h, nullptr, 0, span);
    } else {
      hipLaunchKernelGGL(touched_sweep_kernel, dim3(sweep_grid(items * dim)), dim3(kBlock), 0, stream,
                         tb, s1, s2, item_touch, seg, items, dim, step, p.d[0].h, nullptr, 0);
    }
    if (check_launch("touched_rows_kernel")) return -2;
  }
  if (bias_cfg->algo == COLLIE_OPT_SPARSE_ADAM) {
    hipLaunchKernelGGL(touched_sweep_kernel, dim3(sweep_grid(items)), dim3(kBlock), 0, stream,
                       bias + item_base, bias_state1 + item_base, bias_state2 + item_base, item_touch,
                       row_seg + item_base, items, 1, step, q.d[0].h, nullptr, 0);
    if (check_launch("touched_sweep_kernel")) return -2;
  }
  return 0;
}

// ---- dense-gradient variants (hybrid head) -----------------------------------
namespace collie {
namespace {

// torch.optim.{Adam,SGD,Adagrad}.step on one dense parameter with a dense gradient (the Linear
// weights and biases of the head): one element per thread, torch's formula (apply_update).
__global__ __launch_bounds__(kBlock) void flat_update_kernel(float* __restrict__ param,
                                                             const float* __restrict__ grad,
                                                             float* __restrict__ s1p,
                                                             float* __restrict__ s2p, int64_t n,
                                                             const float* __restrict__ grad_scale,
                                                             const Hyper h) {
  const float scale = grad_scale ? *grad_scale : 1.0f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
    float s1 = s1p ? s1p[i] : 0.0f;
    float s2 = s2p ? s2p[i] : 0.0f;
    param[i] = apply_update(h, param[i], grad[i] * scale, s1, s2);
    if (s1p) s1p[i] = s1;
    if (s2p) s2p[i] = s2;
  }
}

// the same for up to kMultiMax parameters of one optimizer group in ONE launch (the head has ten
// small tensors: ten launches of 2-5 us each otherwise); block b works on tensor t with
// first_block[t] <= b < first_block[t + 1]
constexpr int kMultiMax = 16;
struct MultiFlat {
  int n;
  float* param[kMultiMax];
  const float* grad[kMultiMax];
  float* s1[kMultiMax];
  float* s2[kMultiMax];
  int64_t numel[kMultiMax];
  int first_block[kMultiMax + 1];
};

__global__ __launch_bounds__(kBlock) void flat_update_multi_kernel(const MultiFlat m, const Hyper h) {
  int t = 0;
  while (t + 1 < m.n && static_cast<int>(blockIdx.x) >= m.first_block[t + 1]) ++t;
  const int64_t nblk = m.first_block[t + 1] - m.first_block[t];
  const int64_t stride = nblk * kBlock;
  float* param = m.param[t];
  const float* grad = m.grad[t];
  float* s1p = m.s1[t];
  float* s2p = m.s2[t];
  for (int64_t i = (static_cast<int64_t>(blockIdx.x) - m.first_block[t]) * kBlock + threadIdx.x;
       i < m.numel[t]; i += stride) {
    float s1 = s1p ? s1p[i] : 0.0f;
    float s2 = s2p ? s2p[i] : 0.0f;
    param[i] = apply_update(h, param[i], grad[i], s1, s2);
    if (s1p) s1p[i] = s1;
    if (s2p) s2p[i] = s2;
  }
}

}  // namespace
}  // namespace collie

extern "C" int collie_dense_optim_step_multi(int32_t n, float* const* params,
                                             const float* const* grads, const int64_t* numels,
                                             const collie_optim_cfg* cfg, float* const* state1,
                                             float* const* state2, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(n >= 1 && n <= kMultiMax, "dense_optim_step_multi: 1..%d tensors per call", kMultiMax);
  COLLIE_REQUIRE(params && grads && numels && cfg, "dense_optim_step_multi: NULL pointer");
  COLLIE_REQUIRE(cfg->algo != COLLIE_OPT_SPARSE_ADAM,
                 "SparseAdam does not support dense gradients, please consider Adam instead");
  const bool want1 = cfg->algo != COLLIE_OPT_SGD, want2 = cfg->algo == COLLIE_OPT_ADAM;
  COLLIE_REQUIRE((!want1 || state1) && (!want2 || state2), "dense_optim_step_multi: missing state");
  MultiFlat m;
  memset(&m, 0, sizeof(m));
  m.n = n;
  int blocks = 0;
  for (int t = 0; t < n; ++t) {
    COLLIE_REQUIRE(params[t] && grads[t] && numels[t] >= 1,
                   "dense_optim_step_multi: tensor %d: NULL pointer / empty", t);
    m.param[t] = params[t];
    m.grad[t] = grads[t];
    m.s1[t] = want1 ? state1[t] : nullptr;
    m.s2[t] = want2 ? state2[t] : nullptr;
    COLLIE_REQUIRE((!want1 || m.s1[t]) && (!want2 || m.s2[t]),
                   "dense_optim_step_multi: tensor %d: missing state", t);
    m.numel[t] = numels[t];
    m.first_block[t] = blocks;
    int64_t b = (numels[t] + 4 * kBlock - 1) / (4 * kBlock);     // four elements per thread
    if (b > 1024) b = 1024;
    blocks += static_cast<int>(b);
  }
  m.first_block[n] = blocks;
  Hyper h;
  if (make_hyper(h, cfg, 1, true, m.s1[0], m.s2[0])) return -1;
  hipLaunchKernelGGL(flat_update_multi_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0,
                     stream, m, h);
  return check_launch("flat_update_multi_kernel");
}

extern "C" int collie_dense_optim_step(float* param, const float* grad, int64_t n,
                                       const collie_optim_cfg* cfg, float* state1, float* state2,
                                       const float* grad_scale, collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(param && grad && cfg && n >= 1, "dense_optim_step: NULL pointer / empty tensor");
  COLLIE_REQUIRE(cfg->algo != COLLIE_OPT_SPARSE_ADAM,
                 "SparseAdam does not support dense gradients, please consider Adam instead");
  Hyper h;
  float* s1 = cfg->algo == COLLIE_OPT_SGD ? nullptr : state1;
  float* s2 = cfg->algo == COLLIE_OPT_ADAM ? state2 : nullptr;
  if (make_hyper(h, cfg, 1, true, s1, s2)) return -1;
  hipLaunchKernelGGL(flat_update_kernel, dim3(sweep_grid(n)), dim3(kBlock), 0, stream, param, grad,
                     s1, s2, n, grad_scale, h);
  return check_launch("flat_update_kernel");
}

extern "C" int collie_dense_rows_update(float* table, int64_t num_rows, int32_t dim,
                                        const void* rows, int32_t rows_dtype, const float* vals,
                                        int64_t m, const collie_optim_cfg* cfg, float* state1,
                                        float* state2, int32_t* row_stamp, void* workspace,
                                        size_t workspace_bytes, int32_t* status,
                                        collie_stream_t stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  COLLIE_REQUIRE(table && cfg && (m == 0 || (rows && vals)), "dense_rows_update: NULL pointer");
  COLLIE_REQUIRE(num_rows >= 1 && num_rows < (int64_t{1} << 31) && dim >= 1,
                 "dense_rows_update: bad table shape");
  COLLIE_REQUIRE(m >= 0 && m < (int64_t{1} << 30), "dense_rows_update: bad entry count");
  COLLIE_REQUIRE(cfg->algo != COLLIE_OPT_SPARSE_ADAM,
                 "SparseAdam does not support dense gradients, please consider Adam instead");
  if (m == 0) {
    // an all-zero gradient: no row has an entry, every row takes the zero-gradient update (Adam's
    // moment decay, weight decay) -- the sweep alone, with stamps that match no row
    if (cfg->algo != COLLIE_OPT_ADAM && cfg->weight_decay == 0.0f) return 0;
    COLLIE_REQUIRE(row_stamp != nullptr,
                   "dense Adam / weight decay need row_stamp = int32[rows] scratch");
    float* z1 = cfg->algo == COLLIE_OPT_SGD ? nullptr : state1;
    float* z2 = cfg->algo == COLLIE_OPT_ADAM ? state2 : nullptr;
    Hyper h;
    if (make_hyper(h, cfg, 1, true, z1, z2)) return -1;
    hipLaunchKernelGGL(dense_sweep_kernel, dim3(sweep_grid(num_rows * dim)), dim3(kBlock), 0, stream,
                       table, z1, z2, row_stamp, num_rows, dim, cfg->step, h, nullptr, 0);
    return check_launch("dense_sweep_kernel");
  }
  Carver c(workspace);
  SortWs s;
  carve_sort_ws(c, s, m, dim);
  COLLIE_REQUIRE(workspace && workspace_bytes >= align_up(c.off, 256),
                 "workspace too small: need %zu bytes, got %zu", align_up(c.off, 256),
                 workspace_bytes);
  hipLaunchKernelGGL(coo_entries_kernel, dim3(static_cast<unsigned>((m + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, stream, rows, rows_dtype, m, num_rows, s.ekeys,
                     s.ecodes, status);
  if (check_launch("coo_entries_kernel")) return -2;
  if (sort_and_segment(s, num_rows, 0xffffffffu, stream)) return -1;
  UpdParams p;
  memset(&p, 0, sizeof(p));
  fill_sorted(p, s, num_rows, 0xffffffffu);
  p.D = dim;
  p.step = cfg->step;
  p.stream_rows = beyond_caches(num_rows, dim);
  TableDesc& d = p.d[0];
  d.table = table;
  d.stage = vals;
  d.s1 = cfg->algo == COLLIE_OPT_SGD ? nullptr : state1;
  d.s2 = cfg->algo == COLLIE_OPT_ADAM ? state2 : nullptr;
  if (make_hyper(d.h, cfg, 1, true, d.s1, d.s2)) return -1;
  // rows without an entry still move under Adam (moment decay) and under weight decay
  const bool sweep = cfg->algo == COLLIE_OPT_ADAM || cfg->weight_decay != 0.0f;
  if (sweep) {
    COLLIE_REQUIRE(row_stamp != nullptr,
                   "dense Adam / weight decay need row_stamp = int32[rows] scratch");
    d.row_stamp = row_stamp;
  }
  if (dim == 1) {
    if (!small_coalesce(m))
      hipLaunchKernelGGL(piece_bias_kernel<false>, dim3(piece_bias_grid(m)), dim3(kBlock), 0, stream,
                       p);
    if (launch_bias_rows<false>(p, m, num_rows, stream)) return -2;
  } else if (launch_emb<false>(p, m, stream)) {
    return -2;
  }
  if (sweep) {
    hipLaunchKernelGGL(dense_sweep_kernel, dim3(sweep_grid(num_rows * dim)), dim3(kBlock), 0, stream,
                       d.table, d.s1, d.s2, d.row_stamp, num_rows, dim, p.step, d.h, nullptr, 0);
    if (check_launch("dense_sweep_kernel")) return -2;
  }
  return 0;
}
