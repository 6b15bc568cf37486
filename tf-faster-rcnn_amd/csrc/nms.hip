// nms.hip -- greedy non-max suppression (single- and multi-class) and the three fused layer entry
// points built on it: combined_non_max_suppression, RoIBBox and Decoder.
//
// Replaces tf.image.combined_non_max_suppression (utils/bbox_utils.py:79) as called from
// models/faster_rcnn.py:158-164 (RoIBBox: 1 class, IoU 0.7, R outputs) and :84-90 (Decoder:
// L classes, score > 0.5, IoU 0.5, 200 outputs).
//
// Algorithm (one workgroup per segment = (image, class), 4..16 waves; nms_core.hpp):
//   candidates arrive ordered by (score desc, index asc) from sort_select.  They are consumed in
//   chunks of 64..256.  For a chunk:
//     a. every candidate is tested against the list of boxes kept so far (LDS, broadcast reads);
//        the 64 verdicts of a wave become one 64-bit "dead" word by a wave vote;
//     b. the chunk's own upper-triangular suppression matrix is built as IoU tiles of 64 rows x 64
//        columns: lane = column box, the row box is broadcast from LDS, and the lane collects "row r
//        suppresses me" into a 64-bit COLUMN word (rows that are already dead are skipped);
//     c. wave 0 resolves the greedy order one 64-candidate group at a time in parallel rounds (a candidate
//        hit by a kept row is removed; one with no undecided earlier suppressor is kept), marks the
//        columns its kept rows suppress in the later groups with one vote each, and appends the kept
//        boxes to the kept list.
//   The scan stops as soon as R boxes are kept, so the work is ~(kept x scanned) IoU tests instead of
//   the K^2/2 of a full bitmask, and nothing but the final boxes ever leaves the CU.  (For large R the
//   pair tests go to the whole chip instead: nms_bitmask.hip.)
//   Greedy NMS is inherently serial in the kept list; this kernel is latency/ALU bound and is judged on
//   images/s, not GB/s.
//
// The IoU test is TF's IOUGreaterThanThreshold verbatim (corner-normalised boxes, area <= 0 never
// overlaps, fp32 divide, strict '>'), evaluated on the UNCLIPPED boxes; clipping to [0,1] is applied
// to the emitted boxes only (clip_boxes=True).
#include <stdlib.h>

#include "nms_core.hpp"
#include "sort_select.hpp"

namespace frcnn {

struct NmsParams {
  BoxSource box;
  ScoreSource score;
  int c;           // classes (segments per image)
  int n;           // boxes per image
  int R;           // max kept per segment (= min(max_output_size_per_class, n))
  IouThr iou;
  int use_thr;     // re-check score > score_thr when loading candidates (RoIBBox: top_k did not filter)
  float score_thr;
  const int *sorted_idx;  // (S, sorted_stride)
  const int *counts;      // (S) or NULL (= every segment has n candidates)
  int sorted_stride;
  // direct mode (c == 1): final, padded outputs
  int direct;
  int T;           // max_total_size
  int clip;
  float *out_boxes;    // (B,T,4)
  float *out_scores;   // (B,T) or NULL
  float *out_classes;  // (B,T) or NULL
  int *out_valid;      // (B) or NULL
  // per-segment mode (c > 1): kept lists for the merge kernel
  int *seg_idx;      // (S, R)
  float *seg_score;  // (S, R)
  int *seg_count;    // (S)
};

// LDS (dynamic): see NmsLds in nms_core.hpp.
template <int NT, int CH>
__global__ __launch_bounds__(NT) void nms_kernel(NmsParams p, int Rc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const NmsLds<CH> L = carve_nms_lds<CH>(smem, Rc);
  int *k_idx = L.k_idx;

  const int seg = blockIdx.x;
  const int b = seg / p.c;
  const int cls = seg - b * p.c;
  const int tid = threadIdx.x;

  int count = p.counts ? p.counts[seg] : p.n;
  if (count > p.n) count = p.n;
  const int R = p.R;
  const int *sidx = p.sorted_idx + (long long)seg * p.sorted_stride;

  int nk = 0;
  for (int base = 0; base < count && nk < R; base += CH) {
    const int nc = (count - base < CH) ? (count - base) : CH;
    nk = nms_chunk<NT, CH>(L, nc, nk, R, p.iou, [&](int j, float4 &box, int &ei, bool &bad) {
      ei = sidx[base + j];
      box = load_box(p.box, b, ei, cls);
      if (p.use_thr) bad = !(load_score(p.score, b, ei, cls) > p.score_thr);
    });
  }

  // ---- outputs -----------------------------------------------------------------------------------
  if (p.direct) {
    const int T = p.T;
    const int nvalid = nk < T ? nk : T;
    for (int r = tid; r < T; r += NT) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      float sc = 0.0f;
      if (r < nvalid) {
        const int ei = k_idx[r];
        o = load_box(p.box, b, ei, cls);
        if (p.clip) {
          o.x = clip01(o.x);
          o.y = clip01(o.y);
          o.z = clip01(o.z);
          o.w = clip01(o.w);
        }
        if (p.out_scores) sc = load_score(p.score, b, ei, cls);
      }
      *reinterpret_cast<float4 *>(p.out_boxes + ((long long)b * T + r) * 4) = o;
      if (p.out_scores) p.out_scores[(long long)b * T + r] = sc;
      if (p.out_classes) p.out_classes[(long long)b * T + r] = 0.0f;
    }
    if (tid == 0 && p.out_valid) p.out_valid[b] = nvalid;
  } else {
    for (int r = tid; r < nk; r += NT) {
      const int ei = k_idx[r];
      p.seg_idx[(long long)seg * R + r] = ei;
      p.seg_score[(long long)seg * R + r] = load_score(p.score, b, ei, cls);
    }
    if (tid == 0) p.seg_count[seg] = nk;
  }
}

// Cross-class merge of the per-class kept lists of one image: every entry finds its rank in the
// (score desc, class asc, selection order) ordering by binary searches in the other classes' (already
// score-ordered) lists; ranks < T are emitted, the rest of the T rows are zero padding.
// TF sorts with std::sort (ties unspecified); this build defines ties as above.
struct MergeParams {
  BoxSource box;
  int c, R, T, clip;
  const int *seg_idx;
  const float *seg_score;
  const int *seg_count;
  float *out_boxes;
  float *out_scores;
  float *out_classes;
  int *out_valid;
};

__global__ __launch_bounds__(256) void merge_kernel(MergeParams p, int staged) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int cpad = (p.c + 3) & ~3;
  int *cnt = reinterpret_cast<int *>(smem);        // [c]  kept per class
  int *off = cnt + cpad;                           // [c]  exclusive prefix of cnt
  float *lsc = reinterpret_cast<float *>(off + cpad);  // [c*R] when staged
  __shared__ int s_total;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c = p.c, R = p.R, T = p.T;
  for (int k = tid; k < c; k += 256) {
    int v = p.seg_count[(long long)b * c + k];
    cnt[k] = v < R ? v : R;
  }
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int k = 0; k < c; ++k) {
      off[k] = t;
      t += cnt[k];
    }
    s_total = t;
  }
  const float *gsc = p.seg_score + (long long)b * c * R;
  if (staged) {
    // the kept lists are tiny: stage them in LDS so the binary searches do not chase global loads
    for (int cls = 0; cls < c; ++cls)
      for (int pos = tid; pos < cnt[cls]; pos += 256) lsc[cls * R + pos] = gsc[(long long)cls * R + pos];
  }
  __syncthreads();
  const float *sc = staged ? lsc : gsc;
  const int total = s_total;
  const int nvalid = total < T ? total : T;
  // one entry per thread: entry e of the concatenated lists -> (cls, pos) through the prefix table, so
  // that all of an image's (few hundred) entries are ranked in parallel
  for (int e = tid; e < total; e += 256) {
    int cls = 0;
    {
      int lo = 0, hi = c - 1;  // last class with off[cls] <= e and cnt[cls] > 0
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid - 1;
      }
      cls = lo;
    }
    const int pos = e - off[cls];
    const long long seg = (long long)b * c + cls;
    const float s = sc[(long long)cls * R + pos];
    int rank = pos;
    for (int c2 = 0; c2 < c; ++c2) {
      const int n2 = cnt[c2];
      if (c2 == cls || n2 == 0) continue;
      const float *l2 = sc + (long long)c2 * R;
      // number of entries of class c2 that come before (s, cls): s2 > s, or s2 == s and c2 < cls.
      int lo = 0, hi = n2;
      if (c2 < cls) {
        while (lo < hi) {  // first position with l2[pos] < s
          const int mid = (lo + hi) >> 1;
          if (l2[mid] >= s) lo = mid + 1; else hi = mid;
        }
      } else {
        while (lo < hi) {  // first position with l2[pos] <= s
          const int mid = (lo + hi) >> 1;
          if (l2[mid] > s) lo = mid + 1; else hi = mid;
        }
      }
      rank += lo;
    }
    if (rank < T) {
      const int ei = p.seg_idx[seg * R + pos];
      float4 o = load_box(p.box, b, ei, cls);
      if (p.clip) {
        o.x = clip01(o.x);
        o.y = clip01(o.y);
        o.z = clip01(o.z);
        o.w = clip01(o.w);
      }
      *reinterpret_cast<float4 *>(p.out_boxes + ((long long)b * T + rank) * 4) = o;
      if (p.out_scores) p.out_scores[(long long)b * T + rank] = s;
      if (p.out_classes) p.out_classes[(long long)b * T + rank] = (float)cls;
    }
  }
  for (int r = nvalid + tid; r < T; r += 256) {
    *reinterpret_cast<float4 *>(p.out_boxes + ((long long)b * T + r) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.out_scores) p.out_scores[(long long)b * T + r] = 0.0f;
    if (p.out_classes) p.out_classes[(long long)b * T + r] = 0.0f;
  }
  if (tid == 0 && p.out_valid) p.out_valid[b] = nvalid;
}

// ---- host side ---------------------------------------------------------------------------------
struct NmsWorkspace {
  int *sorted_idx;   // S * n
  int *counts;       // S
  int *seg_idx;      // S * R
  float *seg_score;  // S * R
  int *seg_count;    // S
  size_t bytes;
};

static NmsWorkspace carve_workspace(void *ws, long long S, int n, int R) {
  NmsWorkspace w;
  unsigned char *p = reinterpret_cast<unsigned char *>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char *q = p ? p + off : nullptr;
    off += round_up(bytes, 256);
    return q;
  };
  w.sorted_idx = reinterpret_cast<int *>(take((size_t)S * n * 4));
  w.counts = reinterpret_cast<int *>(take((size_t)S * 4));
  w.seg_idx = reinterpret_cast<int *>(take((size_t)S * R * 4));
  w.seg_score = reinterpret_cast<float *>(take((size_t)S * R * 4));
  w.seg_count = reinterpret_cast<int *>(take((size_t)S * 4));
  w.bytes = off + 256;
  return w;
}

template <int NT, int CH>
static int launch_nms_t(const NmsParams &p, long long S, int Rc, hipStream_t stream) {
  const size_t lds = nms_lds_bytes<CH>(Rc);
  if (p.R > kMaxKept || lds > 160 * 1024)
    return set_error(FRCNN_EUNSUPPORTED, "nms: %d boxes kept per class exceeds the in-LDS kept-list capacity %d", p.R, kMaxKept);
  auto kern = nms_kernel<NT, CH>;
  if (lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "nms: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)S), dim3(NT), lds, stream, p, Rc);
  return check_launch("nms");
}

static int launch_nms(const NmsParams &p, long long S, hipStream_t stream) {
  int Rc = (p.R + 7) & ~7;  // multiple of 4: LDS sub-array alignment
  if (Rc < 8) Rc = 8;
  // Few candidates per segment (Decoder: <= R RoIs per class) -> 4 waves, 64-wide chunks;
  // many (RPN proposals) -> 16 waves, 256-wide chunks.
  if (p.n <= 1024) return launch_nms_t<256, 64>(p, S, Rc, stream);
  return launch_nms_t<1024, 256>(p, S, Rc, stream);
}

static int launch_merge(const MergeParams &m, int B, hipStream_t stream) {
  const size_t head = (size_t)((m.c + 3) & ~3) * 4 * 2;
  const size_t full = head + (size_t)m.c * m.R * 4;
  const int staged = full <= 60 * 1024;
  const size_t lds = staged ? full : head;
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)B), dim3(256), lds, stream, m, staged);
  return check_launch("nms merge");
}

// Shared driver: sort -> nms -> (merge).  `score`/`box` describe the inputs.
static int run_combined(const ScoreSource &score, const BoxSource &box, int B, int n, int c,
                        int max_per_class, int max_total, float iou_thr, bool sort_thr, float score_thr,
                        int pre_topk /* <=0: all */, int clip, float *out_boxes, float *out_scores,
                        float *out_classes, int *out_valid, void *workspace, size_t workspace_bytes,
                        hipStream_t stream, const char *who) {
  const long long S = (long long)B * c;
  const int R = max_per_class < n ? max_per_class : n;
  const int K = (pre_topk > 0) ? pre_topk : n;
  NmsWorkspace w = carve_workspace(workspace, S, K, R > 0 ? R : 1);
  if (!workspace || !aligned_to(workspace, 256) || workspace_bytes < w.bytes)
    return set_error(FRCNN_EINVAL, "%s: workspace must be non-null, 256-byte aligned and >= %zu bytes (got %zu)", who, w.bytes, workspace_bytes);
  if (max_total <= 0) return FRCNN_OK;
  int rc;
  if (R <= 0 || n <= 0) {
    // nothing can be selected: zero padding only
    rc = check_hip(hipMemsetAsync(out_boxes, 0, (size_t)B * max_total * 16, stream), who);
    if (!rc && out_scores) rc = check_hip(hipMemsetAsync(out_scores, 0, (size_t)B * max_total * 4, stream), who);
    if (!rc && out_classes) rc = check_hip(hipMemsetAsync(out_classes, 0, (size_t)B * max_total * 4, stream), who);
    if (!rc && out_valid) rc = check_hip(hipMemsetAsync(out_valid, 0, (size_t)B * 4, stream), who);
    return rc;
  }
  rc = launch_sort_select(score, B, c, n, K, sort_thr, score_thr, w.sorted_idx, nullptr, w.counts, K, stream);
  if (rc) return rc;
  NmsParams p;
  p.box = box;
  p.score = score;
  p.c = c;
  p.n = n;
  p.R = R;
  p.iou = make_iou_thr(iou_thr);
  p.use_thr = sort_thr ? 0 : 1;  // the sort did not filter: re-check when loading
  p.score_thr = score_thr;
  p.sorted_idx = w.sorted_idx;
  p.counts = w.counts;
  p.sorted_stride = K;
  p.direct = (c == 1) ? 1 : 0;
  p.T = max_total;
  p.clip = clip;
  p.out_boxes = out_boxes;
  p.out_scores = out_scores;
  p.out_classes = out_classes;
  p.out_valid = out_valid;
  p.seg_idx = w.seg_idx;
  p.seg_score = w.seg_score;
  p.seg_count = w.seg_count;
  rc = launch_nms(p, S, stream);
  if (rc) return rc;
  if (c > 1) {
    MergeParams m;
    m.box = box;
    m.c = c;
    m.R = R;
    m.T = max_total;
    m.clip = clip;
    m.seg_idx = w.seg_idx;
    m.seg_score = w.seg_score;
    m.seg_count = w.seg_count;
    m.out_boxes = out_boxes;
    m.out_scores = out_scores;
    m.out_classes = out_classes;
    m.out_valid = out_valid;
    rc = launch_merge(m, B, stream);
  }
  return rc;
}

// proposal.hip
int launch_proposal(const float *const *deltas, const float *const *scores, int groups, const float *anchors,
                    const float *variances, int B, int N, int K, int R_out, int R_nms, float iou_thr, float *const *out_boxes,
                    float *const *out_scores, int *const *out_valid, hipStream_t stream, bool small_footprint, bool *handled);

// nms_bitmask.hip
bool roi_bbox_bitmask_applies(int B, int K, int R_nms);
size_t roi_bbox_bitmask_workspace_bytes(int B, int K, int R_nms);
int launch_roi_bbox_bitmask(const float *deltas, const float *scores, const float *anchors, const float *variances, int B,
                            int N, int K, int R_out, int R_nms, float iou_thr, float *out_boxes, float *out_scores,
                            int *out_valid, void *workspace, size_t workspace_bytes, hipStream_t stream, bool overlapped);

// decoder.hip
int launch_decoder(const float *const *rois, const float *const *deltas, const float *const *probs, int groups,
                   const float *variances, int B, int R, int L, int max_total, float score_thr, float iou_thr,
                   float *const *out_boxes, float *const *out_labels, float *const *out_scores, int *const *out_valid,
                   hipStream_t stream, bool small_footprint, bool *handled);

}  // namespace frcnn

using namespace frcnn;

extern "C" {

size_t frcnn_combined_nms_workspace_bytes(int B, int n, int q, int c, int max_per_class, int max_total) {
  (void)q;
  (void)max_total;
  if (B < 0 || n < 0 || c <= 0) return 0;
  const int R = max_per_class < n ? max_per_class : n;
  return carve_workspace(nullptr, (long long)B * c, n, R > 0 ? R : 1).bytes;
}

int frcnn_combined_nms(const float *boxes, const float *scores, int B, int n, int q, int c,
                       int max_per_class, int max_total, float iou_threshold,
                       float score_threshold, int clip_boxes, float *out_boxes,
                       float *out_scores, float *out_classes, int32_t *out_valid,
                       void *workspace, size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(boxes && scores && out_boxes, "combined_nms: null pointer");
  FRCNN_REQUIRE(B >= 0 && n >= 0 && c > 0, "combined_nms: bad shape B=%d n=%d c=%d", B, n, c);
  FRCNN_REQUIRE(q == 1 || q == c, "combined_nms: third dimension of boxes must be 1 or num classes (q=%d, c=%d)", q, c);
  FRCNN_REQUIRE(max_per_class >= 0 && max_total >= 0, "combined_nms: negative output size");
  FRCNN_REQUIRE(aligned_to(boxes, 16) && aligned_to(out_boxes, 16), "combined_nms: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
  ScoreSource ss;
  ss.scores = scores;
  ss.stride_b = (long long)n * c;
  ss.elem_stride = c;
  ss.mask_bg = 0;
  ss.L = 0;
  BoxSource bs;
  bs.boxes = boxes;
  bs.anchors = nullptr;
  bs.deltas = nullptr;
  bs.boxes_stride_b = (long long)n * q * 4;
  bs.anchors_stride_b = 0;
  bs.deltas_stride_b = 0;
  bs.q = q;
  bs.v0 = bs.v1 = bs.v2 = bs.v3 = 1.0f;
  return run_combined(ss, bs, B, n, c, max_per_class, max_total, iou_threshold, true, score_threshold, 0,
                      clip_boxes, out_boxes, out_scores, out_classes, out_valid, workspace, workspace_bytes,
                      (hipStream_t)stream, "combined_nms");
}

size_t frcnn_roi_bbox_workspace_bytes(int B, int N, int K, int R) {
  if (B < 0 || N < 0 || K < 0) return 0;
  const int Rr = R < K ? R : K;
  size_t bytes = carve_workspace(nullptr, B, K, Rr > 0 ? Rr : 1).bytes;
  if (roi_bbox_bitmask_applies(B, K, Rr)) {
    const size_t bm = roi_bbox_bitmask_workspace_bytes(B, K, Rr);
    if (bm > bytes) bytes = bm;
  }
  return bytes;
}

// overlapped: the call comes from the grouped entry point, i.e. from a runtime that keeps several batches in flight on
// several streams (RoIPathPipeline): schedules then prefer workgroups that can become resident beside a running crop.
static int roi_bbox_impl(const float *deltas, const float *scores, const float *anchors,
                         const float *variances, int B, int N, int K, int R, float iou_threshold,
                         float *out_boxes, float *out_scores, int32_t *out_valid, void *workspace,
                         size_t workspace_bytes, frcnn_stream_t stream, bool overlapped) {
  FRCNN_REQUIRE(deltas && scores && anchors && variances && out_boxes, "roi_bbox: null pointer");
  FRCNN_REQUIRE(B >= 0 && N > 0 && K > 0 && R >= 0, "roi_bbox: bad shape B=%d N=%d K=%d R=%d", B, N, K, R);
  FRCNN_REQUIRE(K <= N, "roi_bbox: input must have at least k columns (pre_nms_topn=%d, anchors=%d)", K, N);
  FRCNN_REQUIRE(aligned_to(deltas, 16) && aligned_to(anchors, 16) && aligned_to(out_boxes, 16),
                "roi_bbox: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
  {
    // One fused kernel per image (proposal.hip) whenever its LDS budget allows; FRCNN_ROI_BBOX_UNFUSED=1
    // forces the generic sort -> NMS pipeline (used by the parity tests to cover both).
    if (knob(kKnobRoiBboxUnfused, 0) != 1) {
      bool handled = false;
      const int Rn = R < K ? R : K;
      // large post_nms_topn (training): the pair tests go to the whole chip (nms_bitmask.hip)
      if (roi_bbox_bitmask_applies(B, K, Rn))
        return launch_roi_bbox_bitmask(deltas, scores, anchors, variances, B, N, K, R, Rn, iou_threshold, out_boxes,
                                       out_scores, out_valid, workspace, workspace_bytes, (hipStream_t)stream, overlapped);
      int rc = launch_proposal(&deltas, &scores, 1, anchors, variances, B, N, K, R, Rn, iou_threshold, &out_boxes,
                               &out_scores, &out_valid, (hipStream_t)stream, false, &handled);
      if (rc || handled) return rc;
    }
  }
  ScoreSource ss;
  ss.scores = scores;
  ss.stride_b = N;
  ss.elem_stride = 1;
  ss.mask_bg = 0;
  ss.L = 0;
  BoxSource bs;
  bs.boxes = nullptr;
  bs.anchors = anchors;
  bs.deltas = deltas;
  bs.boxes_stride_b = 0;
  bs.anchors_stride_b = 0;
  bs.deltas_stride_b = (long long)N * 4;
  bs.q = 1;
  bs.v0 = variances[0];
  bs.v1 = variances[1];
  bs.v2 = variances[2];
  bs.v3 = variances[3];
  // models/faster_rcnn.py:158-164: NMS over the K gathered candidates (n = K for size_per_class),
  // score_threshold = -inf (strict), clip_boxes = True.
  const float neg_inf = -__builtin_huge_valf();
  // run_combined clamps max_per_class against n; the candidate set has K elements.
  const int Rr = R < K ? R : K;
  return run_combined(ss, bs, B, N, 1, Rr, R, iou_threshold, false, neg_inf, K, 1, out_boxes, out_scores,
                      nullptr, out_valid, workspace, workspace_bytes, (hipStream_t)stream, "roi_bbox");
}

int frcnn_roi_bbox(const float *deltas, const float *scores, const float *anchors,
                   const float *variances, int B, int N, int K, int R, float iou_threshold,
                   float *out_boxes, float *out_scores, int32_t *out_valid, void *workspace,
                   size_t workspace_bytes, frcnn_stream_t stream) {
  return roi_bbox_impl(deltas, scores, anchors, variances, B, N, K, R, iou_threshold, out_boxes, out_scores, out_valid, workspace,
                       workspace_bytes, stream, false);
}

int frcnn_roi_bbox_multi(const float *const *deltas, const float *const *scores, const float *anchors,
                         const float *variances, int groups, int B, int N, int K, int R, float iou_threshold,
                         float *const *out_boxes, void *workspace, size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(deltas && scores && out_boxes, "roi_bbox_multi: null pointer table");
  FRCNN_REQUIRE(groups >= 1 && groups <= 4, "roi_bbox_multi: groups must be 1..4 (got %d)", groups);
  FRCNN_REQUIRE(anchors && variances, "roi_bbox_multi: null pointer");
  FRCNN_REQUIRE(B >= 0 && N > 0 && K > 0 && R >= 0, "roi_bbox_multi: bad shape B=%d N=%d K=%d R=%d", B, N, K, R);
  FRCNN_REQUIRE(K <= N, "roi_bbox_multi: input must have at least k columns (pre_nms_topn=%d, anchors=%d)", K, N);
  bool ok = aligned_to(anchors, 16);
  for (int g = 0; g < groups; ++g) {
    FRCNN_REQUIRE(deltas[g] && scores[g] && out_boxes[g], "roi_bbox_multi: null pointer in batch %d", g);
    ok = ok && aligned_to(deltas[g], 16) && aligned_to(out_boxes[g], 16);
  }
  FRCNN_REQUIRE(ok, "roi_bbox_multi: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
  const int Rn = R < K ? R : K;
  if (groups > 1 && knob(kKnobRoiBboxUnfused, 0) != 1 && !roi_bbox_bitmask_applies(B, K, Rn)) {
    bool handled = false;
    int rc = launch_proposal(deltas, scores, groups, anchors, variances, B, N, K, R, Rn, iou_threshold, out_boxes, nullptr,
                             nullptr, (hipStream_t)stream, false, &handled);
    if (rc || handled) return rc;
  }
  for (int g = 0; g < groups; ++g) {  // schedules without a grouped form: one call per batch (stream-ordered workspace reuse)
    int rc = roi_bbox_impl(deltas[g], scores[g], anchors, variances, B, N, K, R, iou_threshold, out_boxes[g], nullptr,
                           nullptr, workspace, workspace_bytes, stream, true);
    if (rc) return rc;
  }
  return FRCNN_OK;
}

size_t frcnn_decoder_workspace_bytes(int B, int R, int L, int max_total) {
  if (B < 0 || R < 0 || L <= 0) return 0;
  const int Rr = max_total < R ? max_total : R;
  return carve_workspace(nullptr, (long long)B * L, R, Rr > 0 ? Rr : 1).bytes;
}

int frcnn_decoder(const float *rois, const float *deltas, const float *probs,
                  const float *variances, int B, int R, int L, int max_total,
                  float score_threshold, float iou_threshold, float *out_boxes,
                  float *out_labels, float *out_scores, int32_t *out_valid, void *workspace,
                  size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(rois && deltas && probs && variances && out_boxes, "decoder: null pointer");
  FRCNN_REQUIRE(B >= 0 && R >= 0 && L > 0 && max_total >= 0, "decoder: bad shape B=%d R=%d L=%d T=%d", B, R, L, max_total);
  FRCNN_REQUIRE(aligned_to(rois, 16) && aligned_to(deltas, 16) && aligned_to(out_boxes, 16),
                "decoder: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
  {
    // One fused kernel per image (decoder.hip) whenever its LDS budget allows; FRCNN_DECODER_UNFUSED=1 forces
    // the generic sort -> NMS -> merge pipeline (used by the parity tests to cover both).
    if (knob(kKnobDecoderUnfused, 0) != 1) {
      bool handled = false;
      int rc = launch_decoder(&rois, &deltas, &probs, 1, variances, B, R, L, max_total, score_threshold, iou_threshold,
                              &out_boxes, &out_labels, &out_scores, &out_valid, (hipStream_t)stream, false, &handled);
      if (rc || handled) return rc;
    }
  }
  ScoreSource ss;
  ss.scores = probs;
  ss.stride_b = (long long)R * L;
  ss.elem_stride = L;
  ss.mask_bg = 1;
  ss.L = L;
  BoxSource bs;
  bs.boxes = nullptr;
  bs.anchors = rois;
  bs.deltas = deltas;
  bs.boxes_stride_b = 0;
  bs.anchors_stride_b = (long long)R * 4;
  bs.deltas_stride_b = (long long)R * L * 4;
  bs.q = L;
  bs.v0 = variances[0];
  bs.v1 = variances[1];
  bs.v2 = variances[2];
  bs.v3 = variances[3];
  return run_combined(ss, bs, B, R, L, max_total, max_total, iou_threshold, true, score_threshold, 0, 1,
                      out_boxes, out_scores, out_labels, out_valid, workspace, workspace_bytes,
                      (hipStream_t)stream, "decoder");
}


int frcnn_decoder_multi(const float *const *rois, const float *const *deltas, const float *const *probs,
                        const float *variances, int groups, int B, int R, int L, int max_total, float score_threshold,
                        float iou_threshold, float *const *out_boxes, float *const *out_labels,
                        float *const *out_scores, void *workspace, size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(rois && deltas && probs && out_boxes && out_labels && out_scores, "decoder_multi: null pointer table");
  FRCNN_REQUIRE(groups >= 1 && groups <= 4, "decoder_multi: groups must be 1..4 (got %d)", groups);
  FRCNN_REQUIRE(variances, "decoder_multi: null pointer");
  FRCNN_REQUIRE(B >= 0 && R >= 0 && L > 0 && max_total >= 0, "decoder_multi: bad shape B=%d R=%d L=%d T=%d", B, R, L, max_total);
  bool ok = true;
  for (int g = 0; g < groups; ++g) {
    FRCNN_REQUIRE(rois[g] && deltas[g] && probs[g] && out_boxes[g], "decoder_multi: null pointer in batch %d", g);
    ok = ok && aligned_to(rois[g], 16) && aligned_to(deltas[g], 16) && aligned_to(out_boxes[g], 16);
  }
  FRCNN_REQUIRE(ok, "decoder_multi: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
  if (groups > 1 && knob(kKnobDecoderUnfused, 0) != 1) {
    bool handled = false;
    int rc = launch_decoder(rois, deltas, probs, groups, variances, B, R, L, max_total, score_threshold, iou_threshold,
                            out_boxes, out_labels, out_scores, nullptr, (hipStream_t)stream, false, &handled);
    if (rc || handled) return rc;
  }
  for (int g = 0; g < groups; ++g) {  // no grouped form for this shape: one call per batch
    int rc = frcnn_decoder(rois[g], deltas[g], probs[g], variances, B, R, L, max_total, score_threshold, iou_threshold,
                           out_boxes[g], out_labels[g], out_scores[g], nullptr, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return FRCNN_OK;
}

}  // extern "C"
