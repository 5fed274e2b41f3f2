// Layout of the caller-provided training workspace (one per model, reused every step).
#pragma once

#include "common.h"

namespace collie {

constexpr uint32_t kCodeNone = 0x80000000u;  // entry contributes nothing (zero gradient)
constexpr uint32_t kCodeNeg = 0x40000000u;   // contribution enters with a minus sign
constexpr uint32_t kCodeMask = 0x3fffffffu;  // sample index b

constexpr int kSortTile = 2048;      // keys per workgroup per radix pass
constexpr int kMaxLossBlocks = 2048;  // grid of the fused forward kernel (fixed -> deterministic)

struct SideWs {
  int64_t m;           // number of looked-up ids on this side
  int64_t ntiles;      // ceil(m / kSortTile)
  uint32_t* codes;     // [m] sort payload written by the forward kernel
  uint32_t* keys[2];   // [m] ping-pong
  uint32_t* vals[2];   // [m]
  uint32_t* hist;      // [256 * ntiles]
  uint32_t* totals;    // [4 * 256]
  uint32_t* seg_start; // [m + 1]
  uint32_t* blk_heads; // [ntiles + 1]
  uint32_t* nseg;      // [1]
};

struct TrainWs {
  float* stage_u;        // [B, D]  g * (I[pos] - I[neg_sel])   -> user-row gradients
  float* stage_i;        // [B, D]  g * U[user]                 -> +pos / -neg_sel item-row gradients
  float* gcoef;          // [B]     g = dL/ds+ (dL/ds-_sel = -g)
  float* loss_partials;  // [2 * kMaxLossBlocks]
  SideWs side[2];
  size_t bytes;
};

inline TrainWs carve_train_ws(void* base, int64_t B, int K, int D) {
  Carver c(base);
  TrainWs w;
  w.stage_u = c.take<float>(static_cast<size_t>(B) * D);
  w.stage_i = c.take<float>(static_cast<size_t>(B) * D);
  w.gcoef = c.take<float>(B);
  w.loss_partials = c.take<float>(2 * kMaxLossBlocks);
  for (int s = 0; s < 2; ++s) {
    SideWs& sw = w.side[s];
    sw.m = (s == COLLIE_SIDE_USER) ? B : B * (1 + static_cast<int64_t>(K));
    sw.ntiles = (sw.m + kSortTile - 1) / kSortTile;
    sw.codes = c.take<uint32_t>(sw.m);
    for (int i = 0; i < 2; ++i) {
      sw.keys[i] = c.take<uint32_t>(sw.m);
      sw.vals[i] = c.take<uint32_t>(sw.m);
    }
    sw.hist = c.take<uint32_t>(256 * static_cast<size_t>(sw.ntiles));
    sw.totals = c.take<uint32_t>(4 * 256);
    sw.seg_start = c.take<uint32_t>(sw.m + 1);
    sw.blk_heads = c.take<uint32_t>(sw.ntiles + 1);
    sw.nseg = c.take<uint32_t>(64);
  }
  w.bytes = align_up(c.off, 256);
  return w;
}

inline int radix_passes(int64_t rows) {
  int bits = 1;
  while ((int64_t{1} << bits) < rows) ++bits;
  return (bits + 7) / 8;
}

}  // namespace collie
