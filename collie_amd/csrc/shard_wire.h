// Exchange buffers of the row-sharded step (shard_plan.hip, mf_update.hip).
//
// Every buffer that crosses the wire is a matrix of dim-float rows.  The block of one peer holds its
// c embedding (or gradient) rows, followed by ceil(c / dim) rows that carry the c bias values and --
// gradient direction only -- as many rows again with the c owner-side row numbers (int32 bits).  One
// all-to-all with row splits moves embedding rows, biases and row numbers together.
#pragma once

#include "common.h"

namespace collie {

constexpr int kMaxWorld = 16;   // GPUs of one node; keeps the by-value structs inside the kernarg limit

// by value: entry j of the dense list [first[p], first[p + 1]) belongs to peer p, whose block
// starts at buffer row row0[p]
struct WireBlocks {
  int world;
  int64_t first[kMaxWorld + 1];
  int64_t row0[kMaxWorld + 1];
};

// extra_rows_per: 1 = lookup direction (bias rows), 2 = gradient direction (bias rows + id rows)
inline int make_blocks(WireBlocks& w, const int64_t* counts, int world, int dim,
                       int extra_rows_per) {
  COLLIE_REQUIRE(world >= 1 && world <= kMaxWorld, "world must be in [1, %d]", kMaxWorld);
  COLLIE_REQUIRE(counts != nullptr, "peer counts: NULL pointer");
  w.world = world;
  w.first[0] = 0;
  w.row0[0] = 0;
  for (int p = 0; p < world; ++p) {
    COLLIE_REQUIRE(counts[p] >= 0, "peer counts: negative count");
    w.first[p + 1] = w.first[p] + counts[p];
    w.row0[p + 1] = w.row0[p] + counts[p] + extra_rows_per * ((counts[p] + dim - 1) / dim);
  }
  return 0;
}

#ifdef __HIPCC__
__device__ __forceinline__ int block_of(const WireBlocks& w, int64_t j) {
  int p = 0;
  while (p + 1 < w.world && j >= w.first[p + 1]) ++p;
  return p;
}
#endif

// Gradient direction, requester side: the blocks are only known on the device when the export runs
// (how many of the rows requested from a peer carry a gradient depends on the forward pass).
struct GradLayout {
  int64_t row0[kMaxWorld + 1];    // lookup buffer: first compact row of peer p's block
  int64_t dense0[kMaxWorld + 1];  // lookup direction: first entry of peer p in the dense request list
  int64_t gfirst[kMaxWorld + 1];  // first gradient segment of peer p
  int64_t wire0[kMaxWorld + 1];   // gradient buffer: first row of peer p's block
};

}  // namespace collie
