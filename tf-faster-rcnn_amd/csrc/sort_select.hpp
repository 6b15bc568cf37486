// sort_select.hpp -- segmented "threshold / top-K / sort by score" shared by top_k, RoIBBox,
// combined NMS and the Decoder (implemented in sort_select.hip).
#pragma once

#include "common.hpp"

namespace frcnn {

// Where the scores of a segment live.  Segment s = (image b, class cls) = (s / c, s % c);
// score(b, i, cls) = scores[b * stride_b + i * elem_stride + cls].
struct ScoreSource {
  const float *scores;
  long long stride_b;  // floats per image
  int elem_stride;     // floats between consecutive elements of one class (== c)
  int mask_bg;         // Decoder (models/faster_rcnn.py:82-83): zero the whole row when argmax over L == 0
  int L;               // row length for mask_bg
};

__device__ __forceinline__ float load_score(const ScoreSource &s, int b, int i, int cls) {
  const float *row = s.scores + (long long)b * s.stride_b + (long long)i * s.elem_stride;
  if (s.mask_bg) {
    // [TF-recall] tf.argmax: first maximal index (Eigen ArgMax: strict '>' from lowest()).
    float best = -3.402823466e+38f;
    int am = 0;
    for (int l = 0; l < s.L; ++l) {
      float v = row[l];
      if (v > best) {
        best = v;
        am = l;
      }
    }
    return (am != 0) ? row[cls] : 0.0f;
  }
  return row[cls];
}

// Maximum number of candidates the in-LDS sort holds (per segment).
static const int kSortCap16 = 14336;  // uint16 element indices (n <= 65536): 10 B per slot
static const int kSortCap32 = 12032;  // uint32 element indices: 12 B per slot

// Host-side launcher.  For every segment: candidates = elements whose score passes `thr` (strict '>',
// only when use_thr), ordered by (score desc, index asc); the first min(K, #candidates) element
// indices are written to sorted_idx[s * out_stride + r], their scores to sorted_val (optional) and
// the count to counts[s] (optional).  Returns FRCNN_* codes.
int launch_sort_select(const ScoreSource &src, int B, int c, int n, int K, bool use_thr, float thr,
                       int *sorted_idx, float *sorted_val, int *counts, int out_stride,
                       hipStream_t stream);

}  // namespace frcnn
