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

static const int kDecodedIsNormalised = 1 << 30;  // flag bit in a candidate's element-index word (N is far below 2^30)

template <int CH, int NT>
static inline size_t proposal_lds_bytes(int Rc, int S) {
  return round_up(nms_lds_bytes<CH>(Rc), 16) + (size_t)S * (4 + 4 + 2 + 2) + (size_t)(NT / 64) * 256 * 4 + 64 * 4 +
         (size_t)S * (16 + 8 + 4) + (size_t)S / 4 +  // + the slice's decoded boxes, areas, flags, element indices, dead / regular words
         round_up(slice64_lds_bytes<NT>(), 16);
}

// cache_keys = 1: the image's N sort keys are held in LDS (after the other arrays) so that the radix-select
// passes and the compaction read LDS instead of re-reading the scores through L2.
// NT = 1024: the whole CU works on the image (lowest latency; what the layers use when called one after the other).
// NT = 256 ("small footprint", diagnostic knob FRCNN_PROPOSAL_THREADS=256): four waves and at most 128 VGPRs, the shape of a
// crop_and_resize workgroup, so that the workgroup becomes resident beside the crop kernel's as soon as one of those
// retires instead of waiting for a crop launch's tail (tools/coresidency_probe.hip).  Measured slower end to end in the
// stage pipeline, so every entry point, the *_multi ones included, launches NT = 1024.
template <int CH, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 4 : 1)))
void proposal_kernel(ProposalParams p, int Rc, int S, int cache_keys, int fast_slices) {
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
  float4 *sbox = reinterpret_cast<float4 *>(permA + 2 * S);    // [S] the slice in score order: corner-normalised boxes,
  float *sarea = reinterpret_cast<float *>(sbox + S);          // [S] areas,
  int *sflag = reinterpret_cast<int *>(sarea + S);             // [S] flags,
  int *sidx = sflag + S;                                       // [S] element indices,
  unsigned long long *sbad = reinterpret_cast<unsigned long long *>(sidx + S);  // [S/64] "fails the score threshold"
  unsigned long long *sreg = sbad + S / 64;                    // [S/64] "regular box"
  Slice64Lds T64;
  T64.dead = sreg + S / 64;
  T64.colw = T64.dead + 2;
  T64.s_nk = reinterpret_cast<int *>(T64.colw + 2 * 64);
  uint32_t *kcache = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(sreg + S / 64) +
                                                  ((slice64_lds_bytes<NT>() + 15) & ~(size_t)15));  // [N] when cache_keys == 1
  uint16_t *kc16 = reinterpret_cast<uint16_t *>(kcache);           // [N] when cache_keys == 2: the keys' upper halves only

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
  if (tid == 0) misc[9] = 0;  // misc[9] = a bucket of the fast slice sort overflowed
  // The first fast slice's boundary search starts with a histogram of the top key byte over every 8th key: those keys pass
  // through this thread's registers while the key cache is filled, so the histogram is taken here (LDS adds by one lane in
  // eight) and prefix16_select's first pass finds it ready -- two barriers and one sweep less.
  const bool hist_in_fill = cache_keys != 0 && fast_slices != 0 && N <= 64 * kNT && (kNT % 8) == 0;
  if (cache_keys) {
    // eight loads per thread in flight (the compiler unrolls this loop by two and waits for every pair: with inputs that
    // come from HBM, not L2, that is 2 us per pair); the first eight are requested BEFORE the histogram's bins are cleared
    // and the barrier behind that (a barrier waits for LDS only: the loads stay in flight across it).  cache_keys == 2: N
    // beyond what LDS holds as 32-bit keys (config 5: 35721) -- the fast slices decide membership from the upper 16 bits
    // alone, so those are what is cached; the full key is fetched for the members of a slice only.
    float sc[8];
    int i0 = tid;
#pragma unroll
    for (int u = 0; u < 8; ++u) sc[u] = (i0 + u * kNT < N) ? scores[i0 + u * kNT] : 0.0f;
    if (hist_in_fill) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
    }
    while (i0 < N) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * kNT;
        if (i < N) {
          const uint32_t k = ~f_order_key(sc[u]);
          if (cache_keys == 1)
            kcache[i] = k;
          else
            kc16[i] = (uint16_t)(k >> 16);
          if (hist_in_fill && (tid & 7) == 0) atomicAdd(&hist[k >> 24], 1u);  // (i & 7 == tid & 7: NT is a multiple of 8)
        }
      }
      i0 += 8 * kNT;
      if (i0 < N) {
#pragma unroll
        for (int u = 0; u < 8; ++u) sc[u] = (i0 + u * kNT < N) ? scores[i0 + u * kNT] : 0.0f;
      }
    }
  }
  __syncthreads();
  bool fast = fast_slices != 0 && N <= 64 * kNT;  // (a thread's elements are the bits of one 64-bit mask)
  FRCNN_STAMP(1);
  auto keyf = [&](int i, uint32_t &k) -> bool {
    k = (cache_keys == 1) ? kcache[i] : ~f_order_key(scores[i]);
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
    uint32_t T2;
    int eq2;
    bool all;
    int V;
    bool in_place = false;  // the slice stands in keys[] / orig[] in score order (no permutation)
    // ---- fast slices: any key boundary will do ------------------------------------------------------------------------
    // The consumed prefix only has to be a prefix of the score order, not a fixed number of candidates.  (1) The boundary
    // comes from a SAMPLE: two histogram passes over every 8th key that is not consumed yet give the 16-bit key prefix P
    // below which about 11/16 of the slice capacity lies; what P really cuts off is then counted exactly.  (2) One sweep
    // lists the members (upper key halves from LDS, ballots for the positions); (3) the members -- about one per thread --
    // fetch their full keys and are BUCKET-sorted: 4 * NT buckets linear in the key over [lower bound, boundary], one LDS
    // add per member that also hands back its place inside the bucket, a scan, the placement of (key, index), and an
    // insertion sort by (key, index) inside the few buckets that hold more than one member -- the index as second key is
    // tf.nn.top_k's "lower index first", so no index-ordered compaction and no stable sort are needed.  Anything unusual
    // (the cut holds more than the slice's capacity or crosses rank K, a bucket with more than 32 members: many equal or
    // tightly clustered scores) takes the exact selection + stable radix sort below, from then on.
    if (fast) {
      const uint32_t lo16 = T1 >> 16;  // T1 is a multiple of 65536 on this path
      auto k16 = [&](int i) -> uint32_t {
        return (cache_keys == 2) ? (uint32_t)kc16[i] : ((cache_keys == 1) ? kcache[i] : ~f_order_key(scores[i])) >> 16;
      };
      auto key_sample = [&](int j, uint32_t &k) -> bool {
        const uint32_t t = k16(j << 3);
        k = t << 16;
        return t >= lo16;
      };
      uint32_t P, low_byte;
      int cnt_s;
      bool allp;
      const int target = (s_now * 11) >> 4;
      prefix16_select<NT>(key_sample, (N + 7) >> 3, (target >> 3) > 0 ? (target >> 3) : 1, hist, misc, P, cnt_s, allp, low_byte,
                          hist_in_fill && consumed == 0);
      const uint32_t hi16 = allp ? 0xFFFFu : P;
      const uint32_t hi = (hi16 << 16) | 0xFFFFu;
      // One sweep over the upper key halves: every thread notes which of its (strided) elements the cut holds as bits of a
      // 64-bit mask -- no ballots, so the compiler keeps the loads in flight -- and a block scan of the counts gives each
      // thread its place in the member list.  (The kind of key cache is chosen outside the loop: a branch per element
      // would sit between the loads.)
      unsigned long long mine = 0ull;
      auto members_from = [&](auto load16) {
        int c = 0;
        for (int i = tid; i < N; i += kNT, ++c) {
          const uint32_t t = load16(i);
          mine |= (unsigned long long)(t >= lo16 && t <= hi16) << c;
        }
      };
      if (cache_keys == 2)
        members_from([&](int i) -> uint32_t { return (uint32_t)kc16[i]; });
      else if (cache_keys == 1)
        members_from([&](int i) -> uint32_t { return kcache[i] >> 16; });
      else
        members_from([&](int i) -> uint32_t { return ~f_order_key(scores[i]) >> 16; });
      int Vf;
      const int base = block_excl_scan<NT, false>(__popcll(mine), wsum, Vf);  // (prefix16_select's barriers lie behind wsum's last reader)
      FRCNN_STAMP(2);
      if (Vf > s_now || Vf < 1) {
        fast = false;
      } else {
        constexpr int NBK = 4 * kNT;  // = the histogram array
        int *mlist = sidx;                                     // [Vf] member element indices (the slice arrays are free here)
        uint32_t *mkey = reinterpret_cast<uint32_t *>(sbox);   // [Vf] their keys
        // a lower bound of the slice's keys: exact from the second slice on, for the first the sample's lowest top byte
        // (a member below it lands in bucket 0: the mapping stays monotone)
        const uint32_t kmin = T1 > (low_byte << 24) ? T1 : (low_byte << 24);
        const uint32_t span = hi - kmin;
        const float scale = (float)NBK / ((float)span + 1.0f);
        auto bucket_of = [&](uint32_t k) -> int {
          const int bkt = (k > kmin) ? (int)((float)(k - kmin) * scale) : 0;  // monotone in k
          return bkt < NBK - 1 ? bkt : NBK - 1;
        };
        for (int i = tid; i < NBK; i += kNT) hist[i] = 0;
        {
          int pos = base;
          while (mine != 0ull) {
            mlist[pos++] = tid + (int)__builtin_ctzll(mine) * kNT;
            mine &= mine - 1ull;
          }
        }
        __syncthreads();
        for (int j = tid; j < Vf; j += kNT) {
          uint32_t k;
          keyf(mlist[j], k);
          mkey[j] = k;
          permA[j] = (uint16_t)atomicAdd(&hist[bucket_of(k)], 1u);  // place inside the bucket (arrival order)
        }
        __syncthreads();
        {
          const uint32_t a0 = hist[4 * tid], a1 = hist[4 * tid + 1], a2 = hist[4 * tid + 2], a3 = hist[4 * tid + 3];
          int total;
          const int ex = block_excl_scan<NT, false>((int)(a0 + a1 + a2 + a3), wsum, total);  // (two barriers since the scan above)
          hist[4 * tid] = (uint32_t)ex;
          hist[4 * tid + 1] = (uint32_t)ex + a0;
          hist[4 * tid + 2] = (uint32_t)ex + a0 + a1;
          hist[4 * tid + 3] = (uint32_t)ex + a0 + a1 + a2;
          if (((a0 > a1 ? a0 : a1) > 32u) || ((a2 > a3 ? a2 : a3) > 32u)) misc[9] = 1;
        }
        __syncthreads();
        if (misc[9]) {
          fast = false;  // (misc[9] stays set: the fast path is not tried again)
        } else {
          for (int j = tid; j < Vf; j += kNT) {
            const uint32_t k = mkey[j];
            const uint32_t at = hist[bucket_of(k)] + permA[j];
            keys[at] = k;
            orig[at] = (uint32_t)mlist[j];
          }
          __syncthreads();
          // bucket b starts at hist[b] and ends where b + 1 starts
          {
            uint32_t begin = hist[4 * tid];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t end = (4 * tid + u + 1 < NBK) ? hist[4 * tid + u + 1] : (uint32_t)Vf;
              for (uint32_t a = begin + 1; a < end; ++a) {
                const uint32_t kk = keys[a], oo = orig[a];
                uint32_t c = a;
                while (c > begin && (keys[c - 1] > kk || (keys[c - 1] == kk && orig[c - 1] > oo))) {
                  keys[c] = keys[c - 1];
                  orig[c] = orig[c - 1];
                  --c;
                }
                keys[c] = kk;
                orig[c] = oo;
              }
              begin = end;
            }
          }
          V = Vf;
          in_place = true;
          T2 = hi + 1u;  // the consumed prefix is {key < T2}; hi = 0xFFFFFFFF ends the loop below
          eq2 = 0;
          all = hi == 0xFFFFFFFFu;
          if (all || consumed + V >= N) exhausted = true;
          __syncthreads();
          FRCNN_STAMP(4);
        }
      }
    }
    if (!in_place) {
    // ---- 1. key of rank consumed + s_now -----------------------------------------------------------
    radix_select<NT>(keyf, N, consumed + s_now, hist, misc, T2, eq2, all);
    FRCNN_STAMP(2);
    if (all) {  // cannot happen while K <= N; kept for safety
      T2 = 0xFFFFFFFFu;
      eq2 = 0x7FFFFFFF;
      exhausted = true;
    }
    // ---- 2. compact the slice (in2 and not in1) in index order --------------------------------------
    if (cache_keys == 1) {
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
    for (int j = tid; j < V; j += kNT) sidx[j] = (int)orig[perm[j]];
    if (V < s_now) exhausted = true;
    __syncthreads();
    FRCNN_STAMP(4);
    }  // exact slice
    // ---- 4. decode the slice's boxes, all threads at once (one round of global-memory latency per slice) ------------
    {
      const int v_up = (V + CH - 1) / CH * CH;  // whole chunks: the tail is read by the tiles, never used
      for (int j = tid; j < v_up; j += kNT) {   // v_up <= S: S is a multiple of 256; whole waves: v_up is a multiple of 64
        NBox nb;
        nb.ymin = nb.xmin = nb.ymax = nb.xmax = nb.area = 0.0f;
        int ei = -1, fl = 0;
        bool bad = false;
        if (j < V) {
          ei = in_place ? (int)orig[j] : sidx[j];
          const float4 box = box_of(ei);
          bad = !(scores[ei] > p.score_thr);
          nb = normalise_box(box);
          fl = box_flags(box, nb);
          // bit 30 of the index word: the corner-normalised box IS the decoded box, bit for bit (always, unless an anchor
          // has a negative extent or a coordinate is NaN) -- the output pass then takes a kept box from LDS instead of
          // fetching anchor and deltas and decoding it a second time
          const bool same_bits = __float_as_int(nb.ymin) == __float_as_int(box.x) && __float_as_int(nb.xmin) == __float_as_int(box.y) &&
                                 __float_as_int(nb.ymax) == __float_as_int(box.z) && __float_as_int(nb.xmax) == __float_as_int(box.w);
          ei |= same_bits ? kDecodedIsNormalised : 0;
        }
        sbox[j] = make_float4(nb.ymin, nb.xmin, nb.ymax, nb.xmax);
        sarea[j] = nb.area;
        sflag[j] = fl;
        sidx[j] = ei;
        const unsigned long long badw = ballot64(bad);
        const unsigned long long regw = ballot64(fl == (kFlagArea | kFlagRegular));
        if ((tid & 63) == 0) {
          sbad[j >> 6] = badw;
          sreg[j >> 6] = regw;
        }
      }
      __syncthreads();
    }
    FRCNN_STAMP(5);
    // ---- 5. greedy NMS over the slice ------------------------------------------------------------------
    if (CH == 64) {
      KeptLds kept;
      kept.box = L.k_box;
      kept.area = L.k_area;
      kept.flag = L.k_flag;
      kept.idx = L.k_idx;
      nk = nms_slice64<NT>(sbox, sarea, sflag, sidx, sbad, sreg, V, kept, T64, nk, R, p.iou);
    } else
    for (int cbase = 0; cbase < V && nk < R; cbase += CH) {
      const int nc = (V - cbase < CH) ? (V - cbase) : CH;
      NmsLds<CH> Lc = L;
      Lc.c_box = sbox + cbase;
      Lc.c_area = sarea + cbase;
      Lc.c_flag = sflag + cbase;
      Lc.c_idx = sidx + cbase;
      if (tid < CH / 64) {
        L.dead[tid] = sbad[(cbase >> 6) + tid];
        L.reg[tid] = sreg[(cbase >> 6) + tid];
      }
      nk = nms_chunk_resolve<NT, CH>(Lc, nc, nk, R, p.iou);
    }
    consumed += V;
    T1 = T2;
    eq1 = eq2;
  }

  // ---- outputs: (B,R,4) clipped, score-ordered, zero padded ----------------------------------------
  for (int r = tid; r < R; r += kNT) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.0f;
    if (r < nk) {
      const int ew = L.k_idx[r];
      const int ei = ew & (kDecodedIsNormalised - 1);
      o = (ew & kDecodedIsNormalised) ? L.k_box[r] : box_of(ei);
      o.x = clip01(o.x);
      o.y = clip01(o.y);
      o.z = clip01(o.z);
      o.w = clip01(o.w);
      if (io.out_scores) sc = scores[ei];
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
  const int cache_keys = (lds + (size_t)p.N * 4 <= 160 * 1024) ? 1 : (lds + round_up((size_t)p.N * 2, 16) <= 160 * 1024) ? 2 : 0;
  if (cache_keys == 1) lds += (size_t)p.N * 4;
  if (cache_keys == 2) lds += round_up((size_t)p.N * 2, 16);
  auto kern = proposal_kernel<CH, NT>;
  if (lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "roi_bbox: hipFuncSetAttribute");
    if (rc) return rc;
  }
  const int fast_slices = knob(kKnobProposalExactSlices, 0) ? 0 : 1;  // FRCNN_PROPOSAL_EXACT_SLICES=1: exact selection only
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(NT), lds, stream, p, Rc, S, cache_keys, fast_slices);
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
  if (N >= kDecodedIsNormalised) return FRCNN_OK;  // (bit 30 of an element index is a flag inside the kernel)
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
  // chunk width: tuning knob (FRCNN_PROPOSAL_CHUNK = 64 | 128 | 256); 64 since the packed pair tests (cfg2: 44.8 us per 16
  // images against 45.9 at 128 and 47.2 at 256; clustered proposals 53.8 / 59.7 / 64.7)
  const int ch = knob(kKnobProposalChunk, 64);
  if (knob_set(kKnobProposalThreads)) small_footprint = knob(kKnobProposalThreads, 0) == 256;  // diagnostic override
  if (small_footprint) {
    if (ch == 128) return launch_proposal_t<128, 256>(p, B, Rc, S, stream, handled);
    return launch_proposal_t<64, 256>(p, B, Rc, S, stream, handled);
  }
  if (ch == 128) return launch_proposal_t<128, 1024>(p, B, Rc, S, stream, handled);
  if (ch == 256) return launch_proposal_t<256, 1024>(p, B, Rc, S, stream, handled);
  return launch_proposal_t<64, 1024>(p, B, Rc, S, stream, handled);
}

}  // namespace frcnn
