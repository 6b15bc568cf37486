// proposal.hip -- RoIBBox as ONE kernel: lazy top-K + decode + greedy NMS + clip, one workgroup per image.
//
// Replaces models/faster_rcnn.py:147-166 (tf.nn.top_k -> tf.gather -> combined_non_max_suppression).
//
// TF's NMS pops candidates from a priority queue and stops as soon as post_nms_topn boxes are kept, so
// only a prefix of the 6000 pre-NMS candidates is ever looked at.  This kernel does the same, lazily:
//   repeat until R boxes are kept or pre_nms_topn candidates are consumed
//     1. radix SELECT (4x8-bit LDS histograms over the image's scores) of the next SLICE of the
//        score order, i.e. ranks [consumed, consumed + S);
//     2. stable compaction of that slice into LDS (index order) and a stable LSD radix SORT of it
//        (ties: lower anchor index first, as tf.nn.top_k);
//     3. greedy NMS over the slice in chunks of 256 (nms_core.hpp); candidate boxes are decoded on the
//        fly from (anchor, delta * variance) -- the (B,N,4) decoded tensor, the (B,K) top-k indices and
//        the gathered (B,K,4) boxes of the reference are never materialised;
//   then the kept boxes are decoded once more, clipped to [0,1] and written with zero padding.
// Everything between the RPN head outputs and the (B,R,4) RoIs stays inside one CU: no workspace, no
// intermediate HBM traffic, one launch.  Latency/ALU bound (judged on images/s).
#include <stdlib.h>

#include "block_primitives.hpp"
#include "nms_core.hpp"

namespace frcnn {

// One batch's tensors.  A launch covers up to kProposalGroups batches of one shape (frcnn_roi_bbox_multi: the stage
// pipeline's proposals for the batches whose crops share a launch).
static const int kProposalGroups = 4;
struct ProposalIO {
  const float *deltas;  // (B,N,4)
  const float *scores;  // (B,N)
  float *out_boxes;     // (B,R,4)
  float *out_scores;    // (B,R) or NULL
  int *out_valid;       // (B) or NULL
};

struct ProposalParams {
  ProposalIO io[kProposalGroups];
  int images_per_group;  // B of one batch; workgroup blockIdx.x works on image blockIdx.x % B of batch blockIdx.x / B
  const float *anchors;  // (N,4)
  float v0, v1, v2, v3;
  int N, K, R;
  IouThr iou;
  float score_thr;  // -inf: combined_non_max_suppression's default, strict '>'
};

template <int CH, int NT>
static inline size_t proposal_lds_bytes(int Rc, int S) {
  return round_up(nms_lds_bytes<CH>(Rc), 16) + (size_t)S * (4 + 4 + 2 + 2) + (size_t)(NT / 64) * 256 * 4 + 64 * 4;
}

// cache_keys: the image's N sort keys are held in LDS (after the other arrays) so that the radix-select
// passes and the compaction read LDS instead of re-reading the scores through L2.
// NT = 1024: the whole CU works on the image (lowest latency; what the layers use when called one after the other).
// NT = 256 ("small footprint", diagnostic knob FRCNN_PROPOSAL_THREADS=256): four waves and at most 128 VGPRs, the shape of a
// crop_and_resize workgroup, so that the workgroup becomes resident beside the crop kernel's as soon as one of those
// retires instead of waiting for a crop launch's tail (tools/coresidency_probe.hip).  Measured slower end to end in the
// stage pipeline, so every entry point, the *_multi ones included, launches NT = 1024.
template <int CH, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 4 : 1)))
void proposal_kernel(ProposalParams p, int Rc, int S, int cache_keys) {
  constexpr int kNT = NT, kNW = NT / 64;
  // One workgroup per image: latency bound.  When the stage pipeline runs it beside the chip-filling crop kernel, the
  // crop's waves on this CU must not take its issue slots (they cost the crop 1/16 of its CUs, this kernel 40 %).
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const NmsLds<CH> L = carve_nms_lds<CH>(smem, Rc);
  unsigned char *q = smem + ((nms_lds_bytes<CH>(Rc) + 15) & ~(size_t)15);
  uint32_t *keys = reinterpret_cast<uint32_t *>(q);            // [S]
  uint32_t *orig = keys + S;                                   // [S]
  uint32_t *hist = orig + S;                                   // [kNW*256]
  int *wsum = reinterpret_cast<int *>(hist + kNW * 256);       // [32]
  int *misc = wsum + 32;                                       // [32]
  uint16_t *permA = reinterpret_cast<uint16_t *>(misc + 32);   // [S]
  // permB = permA + S                                         // [S] (ping-pong half)
  uint32_t *kcache = reinterpret_cast<uint32_t *>(permA + 2 * S);  // [N] when cache_keys

  const int tid = threadIdx.x;
  int b = blockIdx.x;
  ProposalIO io = p.io[0];
  {  // which batch (workgroup-uniform selects: no dynamic indexing of the kernel arguments)
    const int g = b / p.images_per_group;
    b -= g * p.images_per_group;
#pragma unroll
    for (int i = 1; i < kProposalGroups; ++i)
      if (g == i) io = p.io[i];
  }
  const int N = p.N, K = p.K, R = p.R;
  const float *scores = io.scores + (long long)b * N;
  const float4 *deltas = reinterpret_cast<const float4 *>(io.deltas) + (long long)b * N;
  const float4 *anchors = reinterpret_cast<const float4 *>(p.anchors);

  FRCNN_STAMP(0);
  if (cache_keys) {
    for (int i = tid; i < N; i += kNT) kcache[i] = ~f_order_key(scores[i]);
    __syncthreads();
  }
  FRCNN_STAMP(1);
  auto keyf = [&](int i, uint32_t &k) -> bool {
    k = cache_keys ? kcache[i] : ~f_order_key(scores[i]);
    return true;  // tf.nn.top_k ranks every element
  };
  auto box_of = [&](int i) -> float4 {
    return decode_box(anchors[i], scale_deltas(deltas[i], p.v0, p.v1, p.v2, p.v3));
  };

  // consumed prefix of the score order = {key < T1} U {key == T1 and index-rank among equals < eq1}
  uint32_t T1 = 0;
  int eq1 = 0;
  int consumed = 0, nk = 0;
  bool exhausted = false;
  const int E = (N + kNT - 1) / kNT;
  const int i0 = tid * E;
  const int i1 = (i0 + E < N) ? (i0 + E) : N;

  while (nk < R && consumed < K && !exhausted) {
    const int s_now = (K - consumed < S) ? (K - consumed) : S;
    // ---- 1. key of rank consumed + s_now -----------------------------------------------------------
    uint32_t T2;
    int eq2;
    bool all;
    radix_select<NT>(keyf, N, consumed + s_now, hist, misc, T2, eq2, all);
    FRCNN_STAMP(2);
    if (all) {  // cannot happen while K <= N; kept for safety
      T2 = 0xFFFFFFFFu;
      eq2 = 0x7FFFFFFF;
      exhausted = true;
    }
    // ---- 2. compact the slice (in2 and not in1) in index order --------------------------------------
    int V;
    if (cache_keys) {
      // keys in LDS: every thread owns a contiguous run of elements
      int e1 = 0, e2 = 0;
      for (int i = i0; i < i1; ++i) {
        uint32_t k;
        keyf(i, k);
        e1 += (k == T1);
        e2 += (k == T2);
      }
      int tot;
      const int base1 = block_excl_scan<NT>(e1, wsum, tot);
      const int base2 = block_excl_scan<NT>(e2, wsum, tot);
      int cnt = 0;
      {
        int r1 = base1, r2 = base2;
        for (int i = i0; i < i1; ++i) {
          uint32_t k;
          keyf(i, k);
          const bool in1 = (k < T1) || (k == T1 && r1 < eq1);
          const bool in2 = (k < T2) || (k == T2 && r2 < eq2);
          r1 += (k == T1);
          r2 += (k == T2);
          cnt += (in2 && !in1);
        }
      }
      int pos = block_excl_scan<NT>(cnt, wsum, V);
      {
        int r1 = base1, r2 = base2;
        for (int i = i0; i < i1; ++i) {
          uint32_t k;
          keyf(i, k);
          const bool in1 = (k < T1) || (k == T1 && r1 < eq1);
          const bool in2 = (k < T2) || (k == T2 && r2 < eq2);
          r1 += (k == T1);
          r2 += (k == T2);
          if (in2 && !in1 && pos < S) {
            keys[pos] = k;
            orig[pos] = (uint32_t)i;
            permA[pos] = (uint16_t)pos;
            ++pos;
          }
        }
      }
    } else {
      // keys in global memory (N beyond the LDS cache, e.g. 35721 anchors at config 5): a contiguous run per THREAD would
      // read them with a stride of N/NT elements between lanes (48 us at config 5).  Here a WAVE owns a contiguous chunk and
      // walks it 64 consecutive elements at a time (one coalesced 256-byte read per round); index order inside a round
      // comes from ballots, across rounds from running counts, across waves from a 16-entry scan.
      const int lane = tid & 63, w = tid >> 6;
      const int CW = ((N + kNW - 1) / kNW + 63) & ~63;  // chunk per wave, a multiple of 64
      const int c0 = w * CW;
      const int rounds = CW >> 6;
      const unsigned long long lt = lanemask_lt();
      // one sweep over the wave's chunk: eight rounds' keys are loaded before the first is used (the ballots make the
      // loop body convergent, which rules out the compiler's own unrolling of a run-time trip count)
      auto sweep = [&](auto body) {
        for (int r = 0; r < rounds; r += 8) {
          uint32_t kk[8];
          bool lv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = c0 + ((r + u) << 6) + lane;
            lv[u] = (r + u < rounds) && (i < N);
            kk[u] = 0xFFFFFFFFu;
            if (lv[u]) keyf(i, kk[u]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) body(c0 + ((r + u) << 6) + lane, kk[u], lv[u]);
        }
      };
      // sweep 1: elements equal to T1 / T2 per wave
      int n1 = 0, n2 = 0;
      sweep([&](int, uint32_t k, bool live) {
        n1 += __popcll(ballot64(live && k == T1));
        n2 += __popcll(ballot64(live && k == T2));
      });
      __syncthreads();  // wsum may still be read by the selection
      if (lane == 0) {
        wsum[w] = n1;
        wsum[kNW + w] = n2;
      }
      __syncthreads();
      int base1 = 0, base2 = 0;
#pragma unroll
      for (int ww = 0; ww < kNW; ++ww) {
        base1 += (ww < w) ? wsum[ww] : 0;
        base2 += (ww < w) ? wsum[kNW + ww] : 0;
      }
      // sweep 2: selected elements per wave
      int nsel = 0;
      {
        int r1 = base1, r2 = base2;
        sweep([&](int, uint32_t k, bool live) {
          const unsigned long long b1 = ballot64(live && k == T1), b2 = ballot64(live && k == T2);
          const int q1 = r1 + __popcll(b1 & lt), q2 = r2 + __popcll(b2 & lt);
          const bool in1 = live && ((k < T1) || (k == T1 && q1 < eq1));
          const bool in2 = live && ((k < T2) || (k == T2 && q2 < eq2));
          nsel += __popcll(ballot64(in2 && !in1));
          r1 += __popcll(b1);
          r2 += __popcll(b2);
        });
      }
      __syncthreads();
      if (lane == 0) wsum[w] = nsel;
      __syncthreads();
      int pos = 0;
      V = 0;
#pragma unroll
      for (int ww = 0; ww < kNW; ++ww) {
        pos += (ww < w) ? wsum[ww] : 0;
        V += wsum[ww];
      }
      // sweep 3: place
      {
        int r1 = base1, r2 = base2;
        sweep([&](int i, uint32_t k, bool live) {
          const unsigned long long b1 = ballot64(live && k == T1), b2 = ballot64(live && k == T2);
          const int q1 = r1 + __popcll(b1 & lt), q2 = r2 + __popcll(b2 & lt);
          const bool in1 = live && ((k < T1) || (k == T1 && q1 < eq1));
          const bool in2 = live && ((k < T2) || (k == T2 && q2 < eq2));
          const bool sel = in2 && !in1;
          const unsigned long long bs = ballot64(sel);
          const int at = pos + __popcll(bs & lt);
          if (sel && at < S) {
            keys[at] = k;
            orig[at] = (uint32_t)i;
            permA[at] = (uint16_t)at;
          }
          pos += __popcll(bs);
          r1 += __popcll(b1);
          r2 += __popcll(b2);
        });
      }
    }
    FRCNN_STAMP(3);
    if (V > S) V = S;  // defensive; V == s_now by construction
    // ---- 3. order the slice ----------------------------------------------------------------------------
    // Every key of the slice lies in [lowest key of the slice, T2]: the bits above the highest one in which the two ends
    // differ are common to the whole slice, so the LSD sort only needs the passes below it (scores of one slice share the
    // exponent and the first mantissa bits: 3 passes instead of 4).
    __syncthreads();  // the compacted keys
    uint32_t kmin = 0xFFFFFFFFu;
    for (int j = tid; j < V; j += kNT) kmin = keys[j] < kmin ? keys[j] : kmin;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)kmin, off, 64);
      kmin = o < kmin ? o : kmin;
    }
    if ((tid & 63) == 0) wsum[tid >> 6] = (int)kmin;  // (free: the barrier above is past the compaction's scans)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kNW; ++i) kmin = (uint32_t)wsum[i] < kmin ? (uint32_t)wsum[i] : kmin;
    const uint32_t diff = kmin ^ T2;
    const int npass = (V > 1 && diff) ? (32 - __builtin_clz(diff) + 7) / 8 : 1;
    const int soff = block_sort_digits<NT>([&](int slot, int pass) -> uint32_t { return keys[slot] >> (8 * pass); }, npass, permA, S,
                                       V, hist, wsum);
    const uint16_t *perm = permA + soff;
    FRCNN_STAMP(4);
    // ---- 4. greedy NMS over the slice ------------------------------------------------------------------
    for (int cbase = 0; cbase < V && nk < R; cbase += CH) {
      const int nc = (V - cbase < CH) ? (V - cbase) : CH;
      nk = nms_chunk<NT, CH>(L, nc, nk, R, p.iou, [&](int j, float4 &box, int &ei, bool &bad) {
        ei = (int)orig[perm[cbase + j]];
        box = box_of(ei);
        bad = !(scores[ei] > p.score_thr);
      });
    }
    consumed += V;
    T1 = T2;
    eq1 = eq2;
    if (V < s_now) exhausted = true;
  }

  // ---- outputs: (B,R,4) clipped, score-ordered, zero padded ----------------------------------------
  for (int r = tid; r < R; r += kNT) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.0f;
    if (r < nk) {
      const int ei = L.k_idx[r];
      o = box_of(ei);
      o.x = clip01(o.x);
      o.y = clip01(o.y);
      o.z = clip01(o.z);
      o.w = clip01(o.w);
      sc = scores[ei];
    }
    *reinterpret_cast<float4 *>(io.out_boxes + ((long long)b * R + r) * 4) = o;
    if (io.out_scores) io.out_scores[(long long)b * R + r] = sc;
  }
  if (tid == 0 && io.out_valid) io.out_valid[b] = nk;
  FRCNN_STAMP(9);
}

template <int CH, int NT>
static int launch_proposal_t(ProposalParams &p, int B, int Rc, int S, hipStream_t stream, bool *handled) {
  size_t lds = proposal_lds_bytes<CH, NT>(Rc, S);
  if (lds > 160 * 1024) return FRCNN_OK;  // does not fit: the caller takes the unfused path
  const int cache_keys = (lds + (size_t)p.N * 4 <= 160 * 1024) ? 1 : 0;
  if (cache_keys) lds += (size_t)p.N * 4;
  auto kern = proposal_kernel<CH, NT>;
  if (lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "roi_bbox: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(NT), lds, stream, p, Rc, S, cache_keys);
  *handled = true;
  return check_launch("roi_bbox (fused proposal kernel)");
}

// *handled stays false when the fused kernel cannot hold the problem in LDS; the caller then takes the
// unfused sort -> NMS path.
int launch_proposal(const float *const *deltas, const float *const *scores, int groups, const float *anchors,
                    const float *variances, int B, int N, int K, int R_out, int R_nms, float iou_thr, float *const *out_boxes,
                    float *const *out_scores, int *const *out_valid, hipStream_t stream, bool small_footprint, bool *handled) {
  *handled = false;
  // R_out rows are written; at most R_nms = min(R_out, K) boxes can be kept.
  if (R_nms != R_out) return FRCNN_OK;  // zero rows beyond K candidates: leave to the generic path
  if (R_nms > kMaxKept) return FRCNN_OK;
  if (groups < 1 || groups > kProposalGroups) return FRCNN_OK;
  int Rc = (R_nms + 7) & ~7;
  if (Rc < 8) Rc = 8;
  // slice: a few times the number of boxes wanted, in whole chunks, at most 4096
  int S = ((2 * R_nms + 255) / 256) * 256;
  if (S < 1024) S = 1024;
  if (S > 4096) S = 4096;
  ProposalParams p;
  for (int g = 0; g < kProposalGroups; ++g) {
    const int s = g < groups ? g : 0;
    p.io[g].deltas = deltas[s];
    p.io[g].scores = scores[s];
    p.io[g].out_boxes = out_boxes[s];
    p.io[g].out_scores = out_scores ? out_scores[s] : nullptr;
    p.io[g].out_valid = out_valid ? out_valid[s] : nullptr;
  }
  p.images_per_group = B;
  B *= groups;  // workgroups of the launch
  p.anchors = anchors;
  p.v0 = variances[0];
  p.v1 = variances[1];
  p.v2 = variances[2];
  p.v3 = variances[3];
  p.N = N;
  p.K = K;
  p.R = R_nms;
  p.iou = make_iou_thr(iou_thr);
  p.score_thr = -__builtin_huge_valf();
  // chunk width: tuning knob (FRCNN_PROPOSAL_CHUNK = 64 | 128 | 256), default 128
  const int ch = knob(kKnobProposalChunk, 128);
  if (knob_set(kKnobProposalThreads)) small_footprint = knob(kKnobProposalThreads, 0) == 256;  // diagnostic override
  if (small_footprint) {
    if (ch == 64) return launch_proposal_t<64, 256>(p, B, Rc, S, stream, handled);
    return launch_proposal_t<128, 256>(p, B, Rc, S, stream, handled);
  }
  if (ch == 64) return launch_proposal_t<64, 1024>(p, B, Rc, S, stream, handled);
  if (ch == 256) return launch_proposal_t<256, 1024>(p, B, Rc, S, stream, handled);
  return launch_proposal_t<128, 1024>(p, B, Rc, S, stream, handled);
}

}  // namespace frcnn
