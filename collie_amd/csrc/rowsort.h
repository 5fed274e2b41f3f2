// Internal interface of rowsort.hip.
#pragma once

#include "train_ws.h"

namespace collie {

// Sorts the m (ekeys, ecodes) entries of `s` by key (keys in [0, nkeys)), stable, and writes the
// segment heads: s.seg_start[0..nseg], s.nseg[0] = nseg, s.nseg[1] = index of the first segment
// whose key is >= split (= nseg if none).
// seg_of (optional, [m]): the segment index of every sorted entry.
int sort_and_segment(const SortWs& s, int64_t nkeys, uint32_t split, hipStream_t stream,
                     uint32_t* seg_of = nullptr);

inline const uint32_t* sorted_keys(const SortWs& s, int64_t nkeys) {
  return s.keys[(radix_plan(nkeys).passes - 1) & 1];
}
inline const uint32_t* sorted_vals(const SortWs& s, int64_t nkeys) {
  return s.vals[(radix_plan(nkeys).passes - 1) & 1];
}

}  // namespace collie
