// Layout of the caller-provided training workspace (one per model, reused every step).
//
// The workspace must be zero-filled when it is created (the two stamp arrays compare against the
// caller's step id, which starts at 1) and must not be shared between models.
#pragma once

#include "common.h"

namespace collie {

constexpr uint32_t kCodeNone = 0x80000000u;  // entry contributes nothing (zero gradient)
constexpr uint32_t kCodeNeg = 0x40000000u;   // contribution enters with a minus sign
constexpr uint32_t kCodeMask = 0x3fffffffu;  // index of the staged row (sample index b)
// Row-sharded gradient export only (mf_update.hip): user and item rows of the compact tables share
// ONE key space there, so the kind of an entry travels in its code (set after the forward kernel;
// the sample index then has 29 bits: (1 + K) * B < 2^30 and K >= 1 keep B below 2^29).
constexpr uint32_t kCodeItem = 0x20000000u;
constexpr uint32_t kCodeSample = 0x1fffffffu;

constexpr int kSortTile = 2048;       // entries per workgroup per radix pass
constexpr int kMaxRadixBits = 11;
constexpr int kMaxLossBlocks = 2048;  // grid of the fused forward kernel (fixed -> deterministic)

// Sort + segment scratch for `m` (key, code) entries.
struct SortWs {
  int64_t m;
  int64_t ntiles;
  uint32_t* ekeys;      // [m] unsorted keys
  uint32_t* ecodes;     // [m] unsorted payload
  uint32_t* keys[2];    // [m] ping-pong
  uint32_t* vals[2];    // [m]
  uint32_t* hist;       // [2^bits * ntiles]
  uint32_t* totals;     // [4 * 2^kMaxRadixBits]
  uint32_t* seg_start;  // [m + 1]
  uint32_t* blk_heads;  // [ntiles + 1]
  uint32_t* nseg;       // [0] number of segments, [1] index of the first segment with key >= split,
                        // [2] number of pieces of long segments
  // Long segments (a hot row under skewed ids: one row can own a large share of the batch) are cut
  // into pieces of kPieceLen entries that are summed in parallel, then combined in piece order.
  uint32_t* long_base;    // [m] per segment: first piece, or kNoPiece
  uint32_t* piece_start;  // [max_pieces] first entry of the piece
  uint32_t* piece_end;    // [max_pieces]
  float* partial;         // [max_pieces, D] piece sums (bias updates use the first max_pieces floats)
  int64_t max_pieces;
};

// The per-item "looked up in this step" stamps are one BYTE per item: at 4 bytes the stamp array
// of a 10 M-item catalogue (40 MB) is evicted from the Infinity Cache by the row gathers between two
// uses of a line, at 1 byte (10 MB, every line touched ~150 times per step) it stays on-die
// (100 M x 10 M x 128 shape: the forward kernel 1.95 -> 1.82 ms).  Tags cycle through 1..255; the
// forward entry point clears the array whenever the cycle restarts (step % 255 == 0), so a tag
// never meets a stale equal from 255 steps earlier.  0 = never looked up.
constexpr int kTouchCycle = 255;
__host__ __device__ inline uint8_t touch_tag(int step) {
  return static_cast<uint8_t>(step % kTouchCycle + 1);
}

// Captured-step replay (collie_mf_train_schedule_steps / collie_mf_train_advance_step): the kernel
// arguments that change from one step to the next -- the step id, its stamp tag and the optimizers'
// step-dependent scalars -- also exist in device memory, so that a HIP graph of the step, whose
// kernel arguments are frozen at capture, can be replayed for the following steps.  The host fills a
// schedule of the next kSchedLen steps (with the very code that computes the by-value arguments:
// the replayed trajectory is bit-identical); the graph's last node moves `cur` one entry on.
constexpr int kSchedLen = 1024;
constexpr int kDynUserEmb = 0, kDynItemEmb = 1, kDynUserBias = 2, kDynItemBias = 3;
struct StepDyn {
  int32_t step;         // step_id of collie_mf_train_forward / _update
  int32_t tag;          // touch_tag(step)
  float neg_lr[4];      // Hyper::neg_lr per table, index = 2 * is_bias + side
  float bc2_sqrt[4];    // Hyper::bc2_sqrt
  int32_t pad[2];
};
struct DynCtl {
  int32_t cursor;       // schedule entry `cur` holds
  int32_t n;            // entries in the schedule
  int32_t overrun;      // set if a step was replayed past the end of the schedule
};

// Coalesces of at most kSmallMax entries run in one workgroup and cut no segment into pieces
// (rowsort.hip): callers skip the piece kernels for them.
constexpr int kSmallMax = 4096;
inline bool small_coalesce(int64_t m) { return m <= kSmallMax; }
// The sort-free row kernel (mf_update.hip: small_rows_kernel) takes up to this many entries: every
// entry's lane group scans all keys in LDS, which stays cheaper than the ~15 launches of the
// general coalesce up to about this size (B <= 4,096; measured, one call per step: B = 4,096 63 us
// against 84 us, B = 5,461 no longer a gain for callers that update embeddings and biases apart).
constexpr int kSmallRowsMax = 12288;

constexpr int kPieceLen = 256;
constexpr uint32_t kNoPiece = 0xffffffffu;
// every piece but the last of a segment is full, and only segments longer than one piece have any
inline int64_t max_pieces_for(int64_t m) { return 2 * (m / kPieceLen) + 2; }

inline void carve_sort_ws(Carver& c, SortWs& s, int64_t m, int D) {
  s.m = m;
  s.ntiles = (m + kSortTile - 1) / kSortTile;
  s.ekeys = c.take<uint32_t>(m);
  s.ecodes = c.take<uint32_t>(m);
  for (int i = 0; i < 2; ++i) {
    s.keys[i] = c.take<uint32_t>(m);
    s.vals[i] = c.take<uint32_t>(m);
  }
  s.hist = c.take<uint32_t>((size_t{1} << kMaxRadixBits) * static_cast<size_t>(s.ntiles));
  s.totals = c.take<uint32_t>(4 * (size_t{1} << kMaxRadixBits));
  s.seg_start = c.take<uint32_t>(m + 1);
  s.blk_heads = c.take<uint32_t>(s.ntiles + 1);
  s.nseg = c.take<uint32_t>(64);
  s.long_base = c.take<uint32_t>(m);
  s.max_pieces = max_pieces_for(m);
  s.piece_start = c.take<uint32_t>(s.max_pieces);
  s.piece_end = c.take<uint32_t>(s.max_pieces);
  // D floats per piece + one more: the row-sharded gradient export keeps a piece's bias sum at
  // partial[max_pieces * D + piece]
  s.partial = c.take<float>(static_cast<size_t>(s.max_pieces) * (D + 1));
}

// radix plan for keys in [0, nkeys)
struct RadixPlan {
  int passes;
  int bits;  // per pass
};
inline RadixPlan radix_plan(int64_t nkeys) {
  int total = 1;
  while ((int64_t{1} << total) < nkeys) ++total;
  RadixPlan p;
  p.passes = (total + kMaxRadixBits - 1) / kMaxRadixBits;
  p.bits = (total + p.passes - 1) / p.passes;
  if (p.bits < 8) p.bits = 8;
  return p;
}

// The fused training step: 3B entries. [0, B): user rows (key = user id); [B, 3B): two item rows
// per sample, positive and selected negative (key = num_users + item id).
struct TrainWs {
  float* stage_u;        // [B, D]  g * I[pos] - g * I[neg_sel]   -> user-row gradients
  float* stage_i;        // [B, D]  g * U[user]                   -> +pos / -neg_sel item rows
  float* gcoef;          // [B]     g = dL/ds+ (dL/ds-_sel = -g)
  float* loss_partials;  // [2 * kMaxLossBlocks]
  SortWs sort;
  uint8_t* touch_item;   // [num_items] touch_tag(step) of the last forward that looked the item up
  int32_t* seg_item;     // [num_items] step id of the last update in which the item had a segment
  DynCtl* dyn_ctl;       // captured-step replay: see StepDyn
  StepDyn* dyn_cur;      // [1] the scalars of the step being replayed
  StepDyn* dyn_sched;    // [kSchedLen]
  size_t bytes;
};

#ifdef __HIPCC__
// loss = (sum l + sum l^2) / B   (bpr.py:89, hinge.py:85, warp.py:133) from the forward kernel's
// per-block partials, reduced in a fixed order by the first 256 threads of ONE workgroup (every
// thread of the workgroup must call this).  Shared by loss_finalize_kernel and the captured step's
// last kernel, so both give the same bits.
__device__ __forceinline__ void finalize_loss_block(const float* partials, int nblocks, float inv_b,
                                                    float* loss_out, int32_t* status) {
  constexpr int kThreads = 256;
  __shared__ double sh[2][kThreads / kWave];
  double a = 0.0, c = 0.0;
  if (threadIdx.x < kThreads) {
    for (int i = threadIdx.x; i < nblocks; i += kThreads) {
      a += partials[2 * i];
      c += partials[2 * i + 1];
    }
    for (int o = kWave / 2; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, kWave);
      c += __shfl_xor(c, o, kWave);
    }
    if (lane_id() == 0) {
      sh[0][threadIdx.x / kWave] = a;
      sh[1][threadIdx.x / kWave] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tc = 0.0;
    for (int w = 0; w < kThreads / kWave; ++w) {
      ta += sh[0][w];
      tc += sh[1][w];
    }
    const float loss = static_cast<float>((ta + tc) * static_cast<double>(inv_b));
    *loss_out = loss;
    // a non-finite loss term makes the sum non-finite: the flag detect_anomaly polls
    // (trainer.py:333 wraps the fit in torch.autograd.set_detect_anomaly; the fused step has no
    // autograd graph, so the kernel reports it)
    if (status && !isfinite(loss)) atomicOr(status, 2);
  }
}
#endif

inline TrainWs carve_train_ws(void* base, int64_t B, int K, int D, int64_t num_users,
                              int64_t num_items) {
  (void)K;
  (void)num_users;
  Carver c(base);
  TrainWs w;
  // persistent part first: its offsets must not move when the batch size changes
  w.touch_item = c.take<uint8_t>(num_items);
  w.seg_item = c.take<int32_t>(num_items);
  w.dyn_ctl = c.take<DynCtl>(1);
  w.dyn_cur = c.take<StepDyn>(1);
  w.dyn_sched = c.take<StepDyn>(kSchedLen);
  w.stage_u = c.take<float>(static_cast<size_t>(B) * D);
  w.stage_i = c.take<float>(static_cast<size_t>(B) * D);
  w.gcoef = c.take<float>(B);
  w.loss_partials = c.take<float>(2 * kMaxLossBlocks);
  carve_sort_ws(c, w.sort, 3 * B, D);
  w.bytes = align_up(c.off, 256);
  return w;
}

}  // namespace collie
