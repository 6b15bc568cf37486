// sort_select.hip -- per-segment threshold / radix-select / stable radix sort of scores.
//
// Replaces tf.nn.top_k (models/faster_rcnn.py:153) and the score-ordered candidate queue inside
// tf.image.combined_non_max_suppression (utils/bbox_utils.py:79): for every segment (image, class)
// the elements whose score passes the threshold are ordered by (score descending, index
// ascending) and the first K of them are emitted.
//
// One 1024-thread workgroup (16 waves) per segment; everything after the first read of the scores
// happens in LDS (up to ~144 KB of the CU's 160 KB):
//   1. if the segment has more elements than fit, a 4x8-bit radix SELECT over the scores (LDS
//      histogram, global re-reads that stay in L2) finds the key of the K-th best element;
//   2. the surviving elements are compacted into LDS in index order (blocked prefix scan), keyed by
//      ~order_key(score) so that ascending key == descending score;
//   3. the slots are ordered by (key, slot) with a bucket sort (block_primitives.hpp: buckets linear in the score first,
//      linear in the key's bit pattern second); when the keys cluster either way (a bucket
//      overflows) a 4-pass, 8-bit, least-significant-digit radix sort permutes 16-bit slot numbers instead.  Stability
//      is what yields "equal scores by ascending index": each wave owns a contiguous slice of the
//      current order, counts its digits with LDS adds, the 256x16 counters are scanned digit-major /
//      wave-minor, and each 64-element chunk ranks itself with an 8-ballot match of equal digits.
// The kernel is latency bound (a few tens of KB per segment); its job is to stay off the host and
// out of HBM, not to saturate bandwidth.
#include "sort_select.hpp"
#include "block_primitives.hpp"

namespace frcnn {

template <typename IdxT>
__global__ __launch_bounds__(1024) void sort_select_kernel(ScoreSource src, int c, int n, int K, int use_thr,
                                                           float thr, int slots, int do_select,
                                                           int *__restrict__ sorted_idx,
                                                           float *__restrict__ sorted_val,
                                                           int *__restrict__ counts, int out_stride, int no_bucket_sort) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem);                 // [slots]
  uint32_t *hist = keys + slots;                                       // [kNW*256]
  int *wsum = reinterpret_cast<int *>(hist + kNW * 256);               // [32]
  int *misc = wsum + 32;                                               // [32]
  IdxT *orig = reinterpret_cast<IdxT *>(misc + 32);                    // [slots]
  uint16_t *permA = reinterpret_cast<uint16_t *>(orig + slots);        // [slots]
  // permB = permA + slots                                             // [slots] (ping-pong half)

  const int seg = blockIdx.x;
  const int b = seg / c;
  const int cls = seg - b * c;
  const int tid = threadIdx.x;

  if (K <= 0 || n <= 0) {
    if (tid == 0 && counts) counts[seg] = 0;
    return;
  }
  FRCNN_STAMP(40);

  auto keyf = [&](int i, uint32_t &k) -> bool {
    const float s = load_score(src, b, i, cls);
    k = ~f_order_key(s);
    return !use_thr || s > thr;
  };

  int V = 0;
  if (!do_select && !use_thr) {
    // ---- top_k of a segment that fits: every element is a candidate, slot == element, coalesced load
    for (int i = tid; i < n; i += kNT) {
      uint32_t k;
      keyf(i, k);
      keys[i] = k;
      orig[i] = (IdxT)i;
      permA[i] = (uint16_t)i;
    }
    V = n;
  } else {
    // ---- 1. radix select of the K-th smallest key (only when the segment does not fit) ---------
    uint32_t T = 0xFFFFFFFFu;
    int need_eq = 0;
    bool filter = false;  // true: keep key < T, and the first need_eq elements with key == T
    if (do_select) {
      bool all;
      radix_select(keyf, n, K, hist, misc, T, need_eq, all);
      filter = !all;
    }
    // ---- 2. stable compaction into LDS: thread t owns elements [t*E, (t+1)*E) -------------------
    const int E = (n + kNT - 1) / kNT;
    const int i0 = tid * E;
    const int i1 = (i0 + E < n) ? (i0 + E) : n;
    int cnt = 0, eqc = 0;
    for (int i = i0; i < i1; ++i) {
      uint32_t k;
      if (keyf(i, k)) {
        if (!filter || k < T)
          ++cnt;
        else if (k == T)
          ++eqc;
      }
    }
    int eq_base = 0, tot = 0;
    if (filter) eq_base = block_excl_scan(eqc, wsum, tot);
    int sel_eq = 0;
    if (filter) {
      sel_eq = need_eq - eq_base;
      sel_eq = sel_eq < 0 ? 0 : (sel_eq > eqc ? eqc : sel_eq);
    }
    int pos = block_excl_scan(cnt + sel_eq, wsum, V);
    // V <= slots by construction (no filter: V <= n == slots; filter: V == K <= slots).
    int eq_seen = 0;
    for (int i = i0; i < i1; ++i) {
      uint32_t k;
      if (keyf(i, k)) {
        bool take = !filter || k < T;
        if (filter && k == T) {
          take = (eq_seen < sel_eq);
          ++eq_seen;
        }
        if (take && pos < slots) {
          keys[pos] = k;
          orig[pos] = (IdxT)i;
          permA[pos] = (uint16_t)pos;
          ++pos;
        }
      }
    }
    if (V > slots) V = slots;  // defensive; cannot happen
  }

  FRCNN_STAMP(41);
  // ---- 3. order the V slots by (key, slot): bucket sort; stable LSD radix sort when the keys cluster ----------------
  if (no_bucket_sort || !(block_bucket_sort_perm(keys, permA, permA + slots, V, hist, wsum, misc, true) ||
                          block_bucket_sort_perm(keys, permA, permA + slots, V, hist, wsum, misc, false))) {
    __syncthreads();
    for (int i = tid; i < V; i += kNT) permA[i] = (uint16_t)i;
    block_sort_perm(keys, permA, slots, V, hist, wsum);
  }
  const uint16_t *perm = permA;
  FRCNN_STAMP(42);

  // ---- 4. emit ------------------------------------------------------------------------------------
  const int nout = (V < K) ? V : K;
  for (int r = tid; r < nout; r += kNT) {
    const int i = (int)orig[perm[r]];
    sorted_idx[(long long)seg * out_stride + r] = i;
    if (sorted_val) sorted_val[(long long)seg * out_stride + r] = load_score(src, b, i, cls);
  }
  if (tid == 0 && counts) counts[seg] = nout;
  FRCNN_STAMP(43);
}

static size_t sort_lds_bytes(int slots, size_t idx_bytes) {
  return (size_t)slots * (4 + idx_bytes + 4) + (size_t)kNW * 256 * 4 + 64 * 4;
}

int launch_sort_select(const ScoreSource &src, int B, int c, int n, int K, bool use_thr, float thr,
                       int *sorted_idx, float *sorted_val, int *counts, int out_stride,
                       hipStream_t stream) {
  if (B <= 0 || c <= 0) return FRCNN_OK;
  const bool idx16 = n <= 65536;
  const int cap = idx16 ? kSortCap16 : kSortCap32;
  const int keff = K < n ? K : n;
  int do_select = 0;
  int slots;
  if (n <= cap) {
    slots = n;
  } else {
    if (keff > cap)
      return set_error(FRCNN_EUNSUPPORTED,
                       "sort_select: %d candidates to keep out of %d elements exceeds the in-LDS sort capacity %d",
                       keff, n, cap);
    do_select = 1;
    slots = keff;
  }
  slots = (slots + 7) & ~7;  // keep the sub-arrays 16-byte aligned
  if (slots < 8) slots = 8;
  const size_t lds = sort_lds_bytes(slots, idx16 ? 2 : 4);
  if (lds > 160 * 1024) return set_error(FRCNN_EUNSUPPORTED, "sort_select: %zu bytes of LDS needed", lds);
  const long long nseg = (long long)B * c;
  if (nseg >= (1ll << 31)) return set_error(FRCNN_EINVAL, "sort_select: too many segments");
  dim3 grid((unsigned)nseg), block(kNT);
  const int no_bucket = knob(kKnobSortRadixOnly, 0) ? 1 : 0;  // FRCNN_SORT_RADIX_ONLY=1: the stable radix sort only
  if (idx16) {
    auto kern = sort_select_kernel<uint16_t>;
    if (lds > 48 * 1024) {
      int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "sort_select: hipFuncSetAttribute");
      if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, src, c, n, keff, use_thr ? 1 : 0, thr, slots, do_select,
                       sorted_idx, sorted_val, counts, out_stride, no_bucket);
  } else {
    auto kern = sort_select_kernel<uint32_t>;
    if (lds > 48 * 1024) {
      int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "sort_select: hipFuncSetAttribute");
      if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, src, c, n, keff, use_thr ? 1 : 0, thr, slots, do_select,
                       sorted_idx, sorted_val, counts, out_stride, no_bucket);
  }
  return check_launch("sort_select");
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_top_k(const float *scores, int B, int N, int K, float *out_values,
                           int32_t *out_indices, frcnn_stream_t stream) {
  FRCNN_REQUIRE(scores, "top_k: null scores");
  FRCNN_REQUIRE(B >= 0 && N >= 0 && K >= 0, "top_k: negative shape");
  FRCNN_REQUIRE(K <= N, "top_k: input must have at least k columns (k=%d, columns=%d)", K, N);
  FRCNN_REQUIRE(out_indices, "top_k: out_indices must not be NULL");
  if (B == 0 || K == 0) return FRCNN_OK;
  ScoreSource src;
  src.scores = scores;
  src.stride_b = N;
  src.elem_stride = 1;
  src.mask_bg = 0;
  src.L = 0;
  return launch_sort_select(src, B, 1, N, K, false, 0.0f, out_indices, out_values, nullptr, K,
                            (hipStream_t)stream);
}
