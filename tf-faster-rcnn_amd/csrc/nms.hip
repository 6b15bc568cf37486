// nms.hip -- greedy non-max suppression (single- and multi-class) and the three fused layer entry
// points built on it: combined_non_max_suppression, RoIBBox and Decoder.
//
// Replaces tf.image.combined_non_max_suppression (utils/bbox_utils.py:79) as called from
// models/faster_rcnn.py:158-164 (RoIBBox: 1 class, IoU 0.7, R outputs) and :84-90 (Decoder:
// L classes, score > 0.5, IoU 0.5, 200 outputs).
//
// Algorithm (one workgroup per segment = (image, class), 4..16 waves):
//   candidates arrive ordered by (score desc, index asc) from sort_select.  They are consumed in
//   chunks of 256.  For a chunk:
//     a. every candidate is tested against the list of boxes kept so far (LDS, broadcast reads);
//        the 64 verdicts of a wave become one 64-bit "dead" word by __ballot;
//     b. the chunk's own 256x256 upper-triangular suppression matrix is built as IoU tiles of
//        64 rows x 64 columns: lane = column box, the row box is broadcast from LDS, and one
//        __ballot per row yields the 64-bit mask word mask[row][column group] in LDS;
//     c. wave 0 resolves the greedy order on scalar registers: pick the lowest alive bit, OR that
//        row's mask words (v_readlane) into the removed set, repeat; kept boxes are appended to the
//        kept list.
//   The scan stops as soon as R boxes are kept, so the work is ~(kept x scanned) IoU tests instead of
//   the K^2/2 of a full bitmask, and nothing but the final boxes ever leaves the CU.
//   Greedy NMS is inherently serial in the kept list; this kernel is latency/ALU bound and is judged on
//   images/s, not GB/s.
//
// The IoU test is TF's IOUGreaterThanThreshold verbatim (corner-normalised boxes, area <= 0 never
// overlaps, fp32 divide, strict '>'), evaluated on the UNCLIPPED boxes; clipping to [0,1] is applied
// to the emitted boxes only (clip_boxes=True).
#include "sort_select.hpp"

namespace frcnn {

static const int kChunk = 256;

struct NmsParams {
  BoxSource box;
  ScoreSource score;
  int c;           // classes (segments per image)
  int n;           // boxes per image
  int R;           // max kept per segment (= min(max_output_size_per_class, n))
  float iou_thr;
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

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// LDS carve (dynamic): kept list (5 floats + 1 int per entry, Rc entries), chunk (5 floats + 1 int
// per entry, 256 entries), mask 256x4 u64, dead 4 u64, 2 ints.
template <int NT>
__global__ __launch_bounds__(NT) void nms_kernel(NmsParams p, int Rc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(smem);  // [256*4]
  unsigned long long *dead = mask + kChunk * 4;                             // [4]
  float *c_ymin = reinterpret_cast<float *>(dead + 4);
  float *c_xmin = c_ymin + kChunk;
  float *c_ymax = c_xmin + kChunk;
  float *c_xmax = c_ymax + kChunk;
  float *c_area = c_xmax + kChunk;
  int *c_idx = reinterpret_cast<int *>(c_area + kChunk);
  int *s_nk = c_idx + kChunk;  // [4]
  float *k_ymin = reinterpret_cast<float *>(s_nk + 4);
  float *k_xmin = k_ymin + Rc;
  float *k_ymax = k_xmin + Rc;
  float *k_xmax = k_ymax + Rc;
  float *k_area = k_xmax + Rc;
  int *k_idx = reinterpret_cast<int *>(k_area + Rc);

  const int seg = blockIdx.x;
  const int b = seg / p.c;
  const int cls = seg - b * p.c;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int NWV = NT / 64;
  const float thr = p.iou_thr;

  int count = p.counts ? p.counts[seg] : p.n;
  if (count > p.n) count = p.n;
  const int R = p.R;
  const int *sidx = p.sorted_idx + (long long)seg * p.sorted_stride;

  if (tid == 0) s_nk[0] = 0;
  __syncthreads();
  int nk = 0;

  for (int base = 0; base < count && nk < R; base += kChunk) {
    const int nc = (count - base < kChunk) ? (count - base) : kChunk;
    // ---- load the chunk ------------------------------------------------------------------------
    if (tid < kChunk) {
      NBox nb;
      nb.ymin = nb.xmin = nb.ymax = nb.xmax = nb.area = 0.0f;
      int ei = -1;
      bool bad = false;
      if (tid < nc) {
        ei = sidx[base + tid];
        nb = normalise_box(load_box(p.box, b, ei, cls));
        if (p.use_thr) bad = !(load_score(p.score, b, ei, cls) > p.score_thr);
      }
      c_ymin[tid] = nb.ymin;
      c_xmin[tid] = nb.xmin;
      c_ymax[tid] = nb.ymax;
      c_xmax[tid] = nb.xmax;
      c_area[tid] = nb.area;
      c_idx[tid] = ei;
      // candidates that fail the score threshold start out dead
      const unsigned long long badw = __ballot(bad);
      if (lane == 0) dead[w] = badw;
    }
    __syncthreads();

    // ---- a. chunk vs kept list ----------------------------------------------------------------
    if (nk > 0) {
      const int j = tid & (kChunk - 1);
      const int slice = tid >> 8;
      constexpr int NSL = NT >> 8;
      NBox me;
      me.ymin = c_ymin[j];
      me.xmin = c_xmin[j];
      me.ymax = c_ymax[j];
      me.xmax = c_xmax[j];
      me.area = c_area[j];
      bool d = false;
      for (int k = slice; k < nk; k += NSL) {
        NBox kb;
        kb.ymin = k_ymin[k];
        kb.xmin = k_xmin[k];
        kb.ymax = k_ymax[k];
        kb.xmax = k_xmax[k];
        kb.area = k_area[k];
        d = d || nms_iou_gt(me, kb, thr);
        if (__all(d)) break;
      }
      const unsigned long long word = __ballot(d && (j < nc));
      if (lane == 0 && word != 0ull) atomicOr(&dead[j >> 6], word);
    }

    // ---- b. intra-chunk IoU tiles (upper triangle), 16 rows per work item -----------------------
    for (int item = w; item < 40; item += NWV) {
      const int tile = item >> 2;
      const int quarter = item & 3;
      const int rg = (tile >= 9) ? 3 : (tile >= 7) ? 2 : (tile >= 4) ? 1 : 0;
      const int tstart = (rg == 0) ? 0 : (rg == 1) ? 4 : (rg == 2) ? 7 : 9;
      const int cg = rg + (tile - tstart);
      if (rg * 64 >= nc || cg * 64 >= nc) continue;  // wave-uniform
      const int jcol = cg * 64 + lane;
      NBox me;
      me.ymin = c_ymin[jcol];
      me.xmin = c_xmin[jcol];
      me.ymax = c_ymax[jcol];
      me.xmax = c_xmax[jcol];
      me.area = c_area[jcol];
#pragma unroll 4
      for (int ii = 0; ii < 16; ++ii) {
        const int irow = rg * 64 + quarter * 16 + ii;
        NBox rb;
        rb.ymin = c_ymin[irow];
        rb.xmin = c_xmin[irow];
        rb.ymax = c_ymax[irow];
        rb.xmax = c_xmax[irow];
        rb.area = c_area[irow];
        const bool bit = (irow < jcol) && (jcol < nc) && nms_iou_gt(rb, me, thr);
        const unsigned long long word = __ballot(bit);
        if (lane == 0) mask[irow * 4 + cg] = word;
      }
    }
    __syncthreads();

    // ---- c. greedy resolve on wave 0 ------------------------------------------------------------
    if (w == 0) {
      unsigned long long rem[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) rem[g] = uniform64(dead[g]);
      int cnt = nk;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int left = nc - g * 64;
        if (left > 0 && cnt < R) {  // wave-uniform
          unsigned long long row[4];
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) row[w2] = (w2 >= g) ? mask[(g * 64 + lane) * 4 + w2] : 0ull;
          const unsigned long long valid = (left >= 64) ? ~0ull : ((1ull << left) - 1ull);
          unsigned long long alive = uniform64(~rem[g] & valid);
          unsigned long long keptbits = 0ull;
          while (alive != 0ull && cnt < R) {
            const int i = (int)__builtin_ctzll(alive);
            keptbits |= (1ull << i);
            ++cnt;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2)
              if (w2 >= g) rem[w2] |= readlane64(row[w2], i);
            alive &= ~(rem[g] | keptbits);
          }
          if ((keptbits >> lane) & 1ull) {
            const int pos = nk + __popcll(keptbits & lanemask_lt());
            const int j = g * 64 + lane;
            k_ymin[pos] = c_ymin[j];
            k_xmin[pos] = c_xmin[j];
            k_ymax[pos] = c_ymax[j];
            k_xmax[pos] = c_xmax[j];
            k_area[pos] = c_area[j];
            k_idx[pos] = c_idx[j];
          }
          nk += __popcll(keptbits);
        }
      }
      if (lane == 0) s_nk[0] = nk;
    }
    __syncthreads();
    nk = s_nk[0];
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

__global__ __launch_bounds__(256) void merge_kernel(MergeParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *cnt = reinterpret_cast<int *>(smem);  // [c]
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
    for (int k = 0; k < c; ++k) t += cnt[k];
    s_total = t;
  }
  __syncthreads();
  const int total = s_total;
  const int nvalid = total < T ? total : T;
  const long long e_all = (long long)c * R;
  for (long long e = tid; e < e_all; e += 256) {
    const int cls = (int)(e / R);
    const int pos = (int)(e - (long long)cls * R);
    if (pos >= cnt[cls]) continue;
    const long long seg = (long long)b * c + cls;
    const float s = p.seg_score[seg * R + pos];
    int rank = pos;
    for (int c2 = 0; c2 < c; ++c2) {
      const int n2 = cnt[c2];
      if (c2 == cls || n2 == 0) continue;
      const float *l2 = p.seg_score + ((long long)b * c + c2) * R;
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
static const int kMaxKept = 6000;  // kept-list capacity in LDS (24 B per entry)

static size_t nms_lds_bytes(int Rc) {
  return (size_t)kChunk * 4 * 8 + 4 * 8 + (size_t)kChunk * 6 * 4 + 4 * 4 + (size_t)Rc * 6 * 4;
}

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

static int launch_nms(const NmsParams &p, long long S, hipStream_t stream) {
  int Rc = (p.R + 7) & ~7;
  if (Rc < 8) Rc = 8;
  const size_t lds = nms_lds_bytes(Rc);
  if (p.R > kMaxKept || lds > 160 * 1024)
    return set_error(FRCNN_EUNSUPPORTED, "nms: %d boxes kept per class exceeds the in-LDS kept-list capacity %d", p.R, kMaxKept);
  // Few candidates per segment (Decoder: <= R RoIs per class) -> 4 waves; RPN proposals -> 16 waves.
  const bool small = p.n <= 1024;
  dim3 grid((unsigned)S);
  if (small) {
    auto kern = nms_kernel<256>;
    if (lds > 48 * 1024) {
      int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "nms: hipFuncSetAttribute");
      if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p, Rc);
  } else {
    auto kern = nms_kernel<1024>;
    if (lds > 48 * 1024) {
      int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "nms: hipFuncSetAttribute");
      if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, stream, p, Rc);
  }
  return check_launch("nms");
}

static int launch_merge(const MergeParams &m, int B, hipStream_t stream) {
  const size_t lds = round_up((size_t)m.c * 4, 16);
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)B), dim3(256), lds, stream, m);
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
  p.iou_thr = iou_thr;
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
  return carve_workspace(nullptr, B, K, Rr > 0 ? Rr : 1).bytes;
}

int frcnn_roi_bbox(const float *deltas, const float *scores, const float *anchors,
                   const float *variances, int B, int N, int K, int R, float iou_threshold,
                   float *out_boxes, float *out_scores, int32_t *out_valid, void *workspace,
                   size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(deltas && scores && anchors && variances && out_boxes, "roi_bbox: null pointer");
  FRCNN_REQUIRE(B >= 0 && N > 0 && K > 0 && R >= 0, "roi_bbox: bad shape B=%d N=%d K=%d R=%d", B, N, K, R);
  FRCNN_REQUIRE(K <= N, "roi_bbox: input must have at least k columns (pre_nms_topn=%d, anchors=%d)", K, N);
  FRCNN_REQUIRE(aligned_to(deltas, 16) && aligned_to(anchors, 16) && aligned_to(out_boxes, 16),
                "roi_bbox: box pointers must be 16-byte aligned");
  if (B == 0) return FRCNN_OK;
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

}  // extern "C"
