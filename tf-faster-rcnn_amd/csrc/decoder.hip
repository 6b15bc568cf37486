// decoder.hip -- the Decoder layer as ONE kernel: background masking, per-class candidate lists, per-class
// greedy NMS and the cross-class top-T merge, one workgroup per image.
//
// Replaces models/faster_rcnn.py:74-91 (reshape, * variances, tile, get_bboxes_from_deltas, argmax / where,
// combined_non_max_suppression with 200 per class / 200 total, score_threshold, IoU 0.5, clip_boxes).
//
// Per image (1024 threads, everything in LDS):
//   A. one lane per RoI row: first-max argmax over the L probabilities; rows whose argmax is background
//      contribute nothing, otherwise every class with prob > score_threshold is a candidate.  Candidates
//      are compacted in (roi, class) order with a block prefix scan.  With softmax inputs there is at most
//      one candidate per RoI; un-normalised scores may give up to R*L.
//   B. candidates are ordered by (class asc, score desc, roi asc): up to 1024 of them by counting ranks
//      (each lane counts the keys that precede its own; all reads are LDS broadcasts), more by the stable
//      LSD radix sort (4 score passes + 1 class pass).
//   C. the classes are shared out over the NMS waves; each wave runs greedy NMS on its class segments in
//      64-candidate tiles: lane = candidate, box decoded on the fly from (roi, delta * variance), test
//      against the wave's kept list, 64x64 intra-tile mask by __ballot, scalar greedy resolve.
//      The (B,R,L,4) decoded tensor of the reference is never materialised.
//   D. every kept entry finds its rank in the cross-class (score desc, class asc, selection order)
//      ordering by binary searches in the other classes' kept lists; ranks < T are decoded once more,
//      clipped to [0,1] and written; the remaining rows are zero padding.
// Latency bound (~2 MB of traffic per batch of 16): judged on images/s.
#include <stdlib.h>

#include "block_primitives.hpp"
#include "nms_core.hpp"

namespace frcnn {

// One batch's tensors; a launch covers up to kDecoderGroups batches of one shape (frcnn_decoder_multi).
static const int kDecoderGroups = 4;
struct DecoderIO {
  const float *rois;    // (B,R,4)
  const float *deltas;  // (B,R,L,4)
  const float *probs;   // (B,R,L)
  float *out_boxes;     // (B,T,4)
  float *out_labels;    // (B,T) or NULL
  float *out_scores;    // (B,T) or NULL
  int *out_valid;       // (B) or NULL
};

struct DecoderParams {
  DecoderIO io[kDecoderGroups];
  int images_per_group;  // B of one batch
  float v0, v1, v2, v3;
  int R, L, T, Rpc;     // Rpc = min(max_output_size_per_class, R)
  float score_thr;
  IouThr iou;
};

static const int kDecMaxClasses = 256;

// per-NMS-wave LDS: one 64-candidate tile + the kept list of the class being processed
__host__ __device__ static inline size_t dec_wave_bytes(int Rpc4) { return (size_t)(64 + Rpc4) * 24; }

// the small-V schedule's arrays (they share the wave areas' space: a launch takes one schedule or the other per image):
// corner-normalised box, original box, (area, flags), sorted position of a kept entry, slot of a sorted position
__host__ __device__ static inline size_t dec_small_bytes(int capV) { return (size_t)capV * (16 + 16 + 8 + 2 + 2); }

__host__ __device__ static inline size_t dec_area_bytes(int Rpc4, int nms_waves, int capV) {
  const size_t a = (size_t)nms_waves * dec_wave_bytes(Rpc4), b = dec_small_bytes(capV);
  return ((a > b ? a : b) + 15) & ~(size_t)15;
}

__host__ __device__ static inline size_t dec_lds_bytes(int Vcap, int Rpc4, int nms_waves, int nt, int capV) {
  return (size_t)Vcap * (4 + 4 + 2 + 2 + 2) + (size_t)(nt / 64) * 256 * 4 + 64 * 4 + (size_t)3 * (kDecMaxClasses + 4) * 4 +
         dec_area_bytes(Rpc4, nms_waves, capV);
}

// NT = 1024: lowest latency; NT = 256: the small-footprint schedule of the stage pipeline (see proposal.hip).
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 4 : 1)))
void decoder_kernel(DecoderParams p, int Vcap, int Rpc4, int nms_waves, int capV) {
  constexpr int kNT = NT, kNW = NT / 64;
  // One workgroup per image: latency bound.  When the stage pipeline runs it beside the chip-filling crop kernel, the
  // crop's waves on this CU must not take its issue slots (they cost the crop 1/16 of its CUs, this kernel 40 %).
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // wave areas / small-V arrays first (float4 alignment), then the 4-byte and 2-byte arrays
  unsigned char *wave_area = smem;
  uint32_t *key = reinterpret_cast<uint32_t *>(smem + dec_area_bytes(Rpc4, nms_waves, capV));  // [Vcap] ascending = score desc
  uint32_t *pk = key + Vcap;                                  // [Vcap] (class << 24) | roi
  uint32_t *hist = pk + Vcap;                                 // [kNW*256]
  int *wsum = reinterpret_cast<int *>(hist + kNW * 256);      // [32]
  int *misc = wsum + 32;                                      // [32]
  int *ccount = misc + 32;                                    // [260] candidates per class
  int *cstart = ccount + kDecMaxClasses + 4;                  // [260] first sorted position of a class
  int *kcount = cstart + kDecMaxClasses + 4;                  // [260] kept per class
  uint16_t *perm = reinterpret_cast<uint16_t *>(kcount + kDecMaxClasses + 4);  // [2*Vcap]
  uint16_t *klist = perm + 2 * Vcap;                          // [Vcap] kept slots; class c at klist[cstart[c] ..]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int b = blockIdx.x;
  DecoderIO io = p.io[0];
  {  // which batch (workgroup-uniform selects)
    const int g = b / p.images_per_group;
    b -= g * p.images_per_group;
#pragma unroll
    for (int i = 1; i < kDecoderGroups; ++i)
      if (g == i) io = p.io[i];
  }
  const int R = p.R, L = p.L, T = p.T;
  const float *probs = io.probs + (long long)b * R * L;
  const float4 *rois = reinterpret_cast<const float4 *>(io.rois) + (long long)b * R;
  const float4 *deltas = reinterpret_cast<const float4 *>(io.deltas) + (long long)b * R * L;

  auto box_of = [&](int r, int cls) -> float4 {
    return decode_box(rois[r], scale_deltas(deltas[(long long)r * L + cls], p.v0, p.v1, p.v2, p.v3));
  };

  FRCNN_STAMP(20);
  for (int i = tid; i < 3 * (kDecMaxClasses + 4); i += kNT) ccount[i] = 0;  // ccount, cstart, kcount are contiguous
  for (int i = tid; i < capV; i += kNT) hist[i] = 0u;  // the small-V schedule's rank accumulators (untouched until then)
  // The image's class probabilities are staged once, coalesced, in the (not yet used) sort buffer: phase A reads every row
  // twice, and a lane walking its own 84-byte row straight from memory pays two uncoalesced round trips.
  float *sprob = reinterpret_cast<float *>(perm);  // [R*L] <= 4*Vcap bytes
  for (int i0 = tid; i0 < R * L; i0 += 8 * kNT) {  // eight loads per thread in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (i0 + u * kNT < R * L) ? probs[i0 + u * kNT] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * kNT < R * L) sprob[i0 + u * kNT] = v[u];
  }
  __syncthreads();

  // ---- A. candidates, compacted in (roi, class) order -------------------------------------------------
  // A1: one lane per RoI: is its first-max argmax a foreground class?  (flags parked in the kept-list buffer, free until C)
  uint16_t *roi_fg = klist;  // [R] <= Vcap
  for (int r = tid; r < R; r += kNT) {
    const float *row = sprob + r * L;
    float best = -3.402823466e+38f;  // [TF-recall] tf.argmax: first maximal index (strict '>' from lowest())
    int am = 0;
    for (int l = 0; l < L; ++l) {
      const float v = row[l];
      if (v > best) {
        best = v;
        am = l;
      }
    }
    roi_fg[r] = (uint16_t)(am != 0);
  }
  __syncthreads();
  FRCNN_STAMP(21);
  // A2: the R*L (roi, class) elements in flat order, a contiguous run per thread: count (per thread for the compaction, per
  // class for the segments), scan, place
  const int RL = R * L;
  const int E = (RL + kNT - 1) / kNT;
  const int e0 = tid * E;
  const int e1 = (e0 + E < RL) ? (e0 + E) : RL;
  // Runs of up to kRun elements (config 2: 7) are read ONCE, all loads in flight together, and stay in registers across the
  // scan: which of them are candidates is a bit mask, class and RoI of element u follow from the run's start by counting.
  constexpr int kRun = 8;
  const bool in_regs = E <= kRun;  // workgroup-uniform
  float rv[kRun];
  uint32_t cand = 0u;  // bit u: element e0 + u is a candidate
  int cnt = 0;
  const int r_first = (e0 < RL) ? e0 / L : 0;
  const int l_first = (e0 < RL) ? e0 - r_first * L : 0;
  if (in_regs) {
    bool fg[kRun];
    {
      int r = r_first, l = l_first;
#pragma unroll
      for (int u = 0; u < kRun; ++u) {
        const bool in = e0 + u < e1;
        rv[u] = in ? sprob[e0 + u] : 0.0f;
        fg[u] = in && (roi_fg[in ? r : 0] != 0);
        const bool wrap = (l + 1 == L);
        l = wrap ? 0 : l + 1;
        r += wrap ? 1 : 0;
      }
    }
    {
      int l = l_first;
#pragma unroll
      for (int u = 0; u < kRun; ++u) {
        const bool c = fg[u] && (rv[u] > p.score_thr);
        cand |= c ? (1u << u) : 0u;
        if (c) atomicAdd(&ccount[l], 1);
        l = (l + 1 == L) ? 0 : l + 1;
      }
    }
    cnt = __popc(cand);
  } else {
    int r = r_first, l = l_first;
    for (int e = e0; e < e1; ++e) {
      if ((roi_fg[r] != 0) && (sprob[e] > p.score_thr)) {
        ++cnt;
        atomicAdd(&ccount[l], 1);
      }
      if (++l == L) {
        l = 0;
        ++r;
      }
    }
  }
  int V;
  int pos = block_excl_scan<NT, false>(cnt, wsum, V);  // (wsum has not been used yet)
  // class segment starts: exclusive prefix of ccount (wave 0, 4 classes per lane; the adds above are behind the scan's barriers)
  if (w == 0) {
    const int c0 = ccount[4 * lane], c1 = ccount[4 * lane + 1], c2 = ccount[4 * lane + 2], c3 = ccount[4 * lane + 3];
    int inc = c0 + c1 + c2 + c3;
    const int s4 = inc;
    inc = wave_incl_scan(inc);
    const int ex = inc - s4;
    cstart[4 * lane] = ex;
    cstart[4 * lane + 1] = ex + c0;
    cstart[4 * lane + 2] = ex + c0 + c1;
    cstart[4 * lane + 3] = ex + c0 + c1 + c2;
  }
  if (V > Vcap) V = Vcap;  // cannot happen: Vcap >= R*L
  const bool small = V <= capV;  // workgroup-uniform
  // the slots grouped by class, arrival order inside a class (small-V schedule only); hist holds 4 * NT words, of which the
  // rank accumulators and the merge keys of that schedule take the first 3 * capV
  uint16_t *cseg = reinterpret_cast<uint16_t *>(hist + 3 * capV);  // [capV]
  int *cfill = kcount;  // [L] segment fill counters (the NMS overwrites every class's entry)
  __syncthreads();
  if (cnt) {
    int r = r_first, l = l_first;
    if (in_regs) {
#pragma unroll
      for (int u = 0; u < kRun; ++u) {
        if (((cand >> u) & 1u) && pos < Vcap) {
          key[pos] = ~f_order_key(rv[u]);
          pk[pos] = ((uint32_t)l << 24) | (uint32_t)r;
          if (small) cseg[cstart[l] + atomicAdd(&cfill[l], 1)] = (uint16_t)pos;
          ++pos;
        }
        const bool wrap = (l + 1 == L);
        l = wrap ? 0 : l + 1;
        r += wrap ? 1 : 0;
      }
    } else {
      for (int e = e0; e < e1; ++e) {
        const float v = sprob[e];
        if (roi_fg[r] != 0 && v > p.score_thr && pos < Vcap) {
          key[pos] = ~f_order_key(v);
          pk[pos] = ((uint32_t)l << 24) | (uint32_t)r;
          if (small) cseg[cstart[l] + atomicAdd(&cfill[l], 1)] = (uint16_t)pos;
          ++pos;
        }
        if (++l == L) {
          l = 0;
          ++r;
        }
      }
    }
  }
  __syncthreads();
  FRCNN_STAMP(22);

  if (small) {
    // ================= small-V schedule (V <= NT: the reference's softmax inputs give at most one candidate per RoI) =====
    float4 *snorm = reinterpret_cast<float4 *>(wave_area);     // [capV] corner-normalised boxes in (class, score, roi) order
    float4 *sorig = snorm + capV;                              // [capV] the decoded boxes as they are written out
    float2 *saf = reinterpret_cast<float2 *>(sorig + capV);    // [capV] (area, flags)
    uint16_t *kq = reinterpret_cast<uint16_t *>(saf + capV);   // [capV] class c's kept entries: sorted positions, at cstart[c] ..
    uint16_t *sslot = kq + capV;                               // [capV] slot of a sorted position
    uint32_t *racc = hist;                                     // [capV] rank accumulators (both rankings)
    unsigned long long *kcomp = reinterpret_cast<unsigned long long *>(hist + capV);  // [capV] (key, class, selection order)
    // ---- B. rank inside the class: only the m candidates of one class are compared (a few dozen), `parts` threads share
    // a candidate when V is small ----------------------------------------------------------------------------------
    const int Vp = (V + 63) & ~63;
    const int parts = Vp > 0 ? kNT / Vp : 1;  // >= 1 (V <= capV <= NT)
    const int part = Vp > 0 ? tid / Vp : 0;
    const int q = tid - part * Vp;  // position in cseg
    // (racc was cleared at the kernel's start: barriers lie between)
    int slot = 0, cls = 0, s0 = 0;
    float4 roi4 = make_float4(0.f, 0.f, 0.f, 0.f), del4 = roi4;  // the candidate's RoI and deltas (part 0's threads)
    if (part < parts && q < V) {
      slot = (int)cseg[q];
      const uint32_t pw = pk[slot];
      cls = (int)(pw >> 24);
      if (part == 0) {  // requested here, used behind the barrier: the ranking hides the global-memory latency
        const int r = (int)(pw & 0xFFFFFFu);
        roi4 = rois[r];
        del4 = deltas[(long long)r * L + cls];
      }
      s0 = cstart[cls];
      const int m = ccount[cls];
      const uint32_t mykey = key[slot];
      const int j0 = s0 + (int)((long long)m * part / parts), j1 = s0 + (int)((long long)m * (part + 1) / parts);
      int rank = 0;
      for (int j = j0; j < j1; ++j) {
        const int o = (int)cseg[j];
        const uint32_t ok = key[o];
        rank += (ok < mykey) || (ok == mykey && o < slot);  // slots ascend with the roi inside a class
      }
      if (rank) atomicAdd(&racc[q], (uint32_t)rank);
    }
    __syncthreads();
    // ---- decode every candidate once, into its sorted position -----------------------------------------------------
    if (part == 0 && q < V) {
      const int qq = s0 + (int)racc[q];
      racc[q] = 0u;  // (own entry; reused by the cross-class ranking)
      const float4 box = decode_box(roi4, scale_deltas(del4, p.v0, p.v1, p.v2, p.v3));
      const NBox nb = normalise_box(box);
      snorm[qq] = make_float4(nb.ymin, nb.xmin, nb.ymax, nb.xmax);
      sorig[qq] = box;
      saf[qq] = make_float2(nb.area, __int_as_float(box_flags(box, nb)));
      sslot[qq] = (uint16_t)slot;
      kcomp[qq] = ~0ull;  // "not kept"
    }
    __syncthreads();
    FRCNN_STAMP(23);
    // ---- C. per-class greedy NMS, the classes dealt to all waves, boxes from LDS ---------------------------------------
    for (int c = w; c < L; c += kNW) {
      const int cs = cstart[c];
      const int m = ccount[c];
      const int Rk = (p.Rpc < m) ? p.Rpc : m;
      int nk = 0;
      for (int base = 0; base < m && nk < Rk; base += 64) {
        const int nc = (m - base < 64) ? (m - base) : 64;
        const int myq = cs + base + ((lane < nc) ? lane : 0);
        const PBox me = load_pbox(snorm, saf, myq);
        // a. against the kept list of this class
        bool d = false;
        for (int k0 = 0; k0 < nk; k0 += 4) {
          PBox kb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) kb[u] = load_pbox(snorm, saf, (int)kq[cs + ((k0 + u < nk) ? (k0 + u) : (nk - 1))]);
          bool irr = false, dd = false;
#pragma unroll
          for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt_fast(me, kb[u], p.iou, irr);
          if (__builtin_expect(ballot64(irr || p.iou.slow_only) != 0ull, 0)) {
            dd = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt(to_nbox(me), to_nbox(kb[u]), p.iou.thr);
          }
          d = d | dd;
          if (ballot64(!(d || (lane >= nc))) == 0ull) break;
        }
        // b. intra-tile mask: row i suppresses column lane (i < lane)
        unsigned long long own = 0ull;
        for (int i0 = 0; i0 < nc; i0 += 4) {
          PBox rb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) rb[u] = load_pbox(snorm, saf, cs + base + ((i0 + u < nc) ? (i0 + u) : (nc - 1)));
          bool irr = false;
          bool bit[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt_fast(rb[u], me, p.iou, irr);
          if (__builtin_expect(ballot64(irr || p.iou.slow_only) != 0ull, 0)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt(to_nbox(rb[u]), to_nbox(me), p.iou.thr);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned long long word = ballot64(bit[u] & (i0 + u < lane) & (lane < nc));
            if (lane == i0 + u) own = word;
          }
        }
        // c. greedy resolve (scalar), rows that suppress nothing are accepted in bulk
        const unsigned long long rem0 = ballot64(d);
        const unsigned long long nz = ballot64(own != 0ull);
        const unsigned long long valid = (nc >= 64) ? ~0ull : ((1ull << nc) - 1ull);
        unsigned long long rem = uniform64(rem0);
        unsigned long long alive = uniform64(~rem & valid);
        unsigned long long keptbits = 0ull;
        int cnt2 = nk;
        while (alive != 0ull && cnt2 < Rk) {
          const unsigned long long nza = alive & nz;
          const int f = (nza != 0ull) ? (int)__builtin_ctzll(nza) : 64;
          unsigned long long batch = (f >= 64) ? alive : (alive & ((1ull << f) - 1ull));
          int nb = __popcll(batch);
          if (nb > 0) {
            if (cnt2 + nb > Rk) {
              unsigned long long t = 0ull, bb = batch;
              for (int k = cnt2; k < Rk; ++k) {
                const unsigned long long low = bb & (0ull - bb);
                t |= low;
                bb ^= low;
              }
              batch = t;
              nb = Rk - cnt2;
            }
            keptbits |= batch;
            cnt2 += nb;
            alive &= ~batch;
            if (cnt2 >= Rk) break;
          }
          if (f < 64) {
            keptbits |= (1ull << f);
            ++cnt2;
            rem |= readlane64(own, f);
            alive &= ~(rem | keptbits);
          }
        }
        if ((keptbits >> lane) & 1ull) {
          const int kp = nk + __popcll(keptbits & lanemask_lt());
          kq[cs + kp] = (uint16_t)myq;
          // cross-class order: score desc (key asc), class asc, selection order
          kcomp[cs + kp] = ((unsigned long long)key[sslot[myq]] << 32) | ((unsigned long long)c << 16) | (unsigned long long)kp;
        }
        nk += __popcll(keptbits);
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) kcount[c] = nk;
    }
    __syncthreads();
    FRCNN_STAMP(24);
    // ---- D. cross-class merge: every kept entry counts the kept entries that precede it --------------------------------
    if (w == 0) {
      int t4 = kcount[4 * lane] + kcount[4 * lane + 1] + kcount[4 * lane + 2] + kcount[4 * lane + 3];
      t4 = wave_incl_scan(t4);
      if (lane == 63) misc[0] = t4;
    }
    unsigned long long mine = ~0ull;
    if (part < parts && q < V) {
      mine = kcomp[q];
      if (mine != ~0ull) {
        // slice boundaries on even positions: two keys per 16-byte LDS read
        const int j0 = ((int)((long long)V * part / parts)) & ~1;
        const int j1e = (part + 1 == parts) ? V : (((int)((long long)V * (part + 1) / parts)) & ~1);
        int rank = 0;
        int j = j0;
        for (; j + 1 < j1e; j += 2) {
          const ulonglong2 two = *reinterpret_cast<const ulonglong2 *>(kcomp + j);
          rank += (two.x < mine) + (two.y < mine);  // (~0 = not kept: never smaller)
        }
        if (j < j1e) rank += kcomp[j] < mine;
        if (rank) atomicAdd(&racc[q], (uint32_t)rank);
      }
    }
    __syncthreads();
    FRCNN_STAMP(26);
    const int total = misc[0];
    const int nvalid = total < T ? total : T;
    if (part == 0 && q < V && mine != ~0ull) {
      const int rank = (int)racc[q];
      if (rank < T) {
        float4 o = sorig[kq[q]];
        o.x = clip01(o.x);
        o.y = clip01(o.y);
        o.z = clip01(o.z);
        o.w = clip01(o.w);
        *reinterpret_cast<float4 *>(io.out_boxes + ((long long)b * T + rank) * 4) = o;
        // the score back from its key: candidates are positive (prob > score_threshold >= 0)
        if (io.out_scores) io.out_scores[(long long)b * T + rank] = u2f(~(uint32_t)(mine >> 32) & 0x7FFFFFFFu);
        if (io.out_labels) io.out_labels[(long long)b * T + rank] = (float)(int)((mine >> 16) & 0xFFFFull);
      }
    }
    for (int r = nvalid + tid; r < T; r += kNT) {
      *reinterpret_cast<float4 *>(io.out_boxes + ((long long)b * T + r) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (io.out_scores) io.out_scores[(long long)b * T + r] = 0.0f;
      if (io.out_labels) io.out_labels[(long long)b * T + r] = 0.0f;
    }
    if (tid == 0 && io.out_valid) io.out_valid[b] = nvalid;
    FRCNN_STAMP(25);
    return;
  }

  // ================= general schedule (more candidates than threads: un-normalised scores) ==============================
  for (int i = tid; i < L; i += kNT) kcount[i] = 0;
  // ---- B. order by (class asc, score desc, roi asc): stable LSD radix sort (4 score passes + 1 class pass) --------------
  for (int i = tid; i < V; i += kNT) perm[i] = (uint16_t)i;  // (published by the sort's first barrier)
  const int soff = block_sort_digits<NT>(
      [&](int slot, int pass) -> uint32_t { return pass < 4 ? (key[slot] >> (8 * pass)) : (pk[slot] >> 24); }, 5, perm,
      Vcap, V, hist, wsum);
  const uint16_t *sperm = perm + soff;
  __syncthreads();

  FRCNN_STAMP(23);
  // ---- C. per-class greedy NMS, one wave per class segment ------------------------------------------------
  if (w < nms_waves) {
    float4 *cbox = reinterpret_cast<float4 *>(wave_area + (size_t)w * dec_wave_bytes(Rpc4));
    float4 *kbox = cbox + 64;
    float2 *caf = reinterpret_cast<float2 *>(kbox + Rpc4);
    float2 *kaf = caf + 64;
    for (int cls = w; cls < L; cls += nms_waves) {
      const int s0 = cstart[cls];
      const int m = ccount[cls];
      const int Rk = (p.Rpc < m) ? p.Rpc : m;
      int nk = 0;
      for (int base = 0; base < m && nk < Rk; base += 64) {
        const int nc = (m - base < 64) ? (m - base) : 64;
        // load the tile: lane = candidate
        PBox me;
        me.ymin = me.xmin = me.ymax = me.xmax = me.area = 0.0f;
        me.flag = 0;
        int slot = 0;
        if (lane < nc) {
          slot = (int)sperm[s0 + base + lane];
          const float4 box = box_of((int)(pk[slot] & 0xFFFFFFu), cls);
          const NBox nb = normalise_box(box);
          me.ymin = nb.ymin;
          me.xmin = nb.xmin;
          me.ymax = nb.ymax;
          me.xmax = nb.xmax;
          me.area = nb.area;
          me.flag = box_flags(box, nb);
        }
        cbox[lane] = make_float4(me.ymin, me.xmin, me.ymax, me.xmax);
        caf[lane] = make_float2(me.area, __int_as_float(me.flag));
        __builtin_amdgcn_wave_barrier();
        // a. against the kept list of this class
        bool d = false;
        for (int k0 = 0; k0 < nk; k0 += 4) {
          PBox kb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) kb[u] = load_pbox(kbox, kaf, (k0 + u < nk) ? (k0 + u) : (nk - 1));
          bool irr = false, dd = false;
#pragma unroll
          for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt_fast(me, kb[u], p.iou, irr);
          if (__builtin_expect(ballot64(irr || p.iou.slow_only) != 0ull, 0)) {
            dd = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt(to_nbox(me), to_nbox(kb[u]), p.iou.thr);
          }
          d = d | dd;
          if (ballot64(!(d || (lane >= nc))) == 0ull) break;
        }
        // b. intra-tile mask: row i suppresses column lane (i < lane)
        unsigned long long own = 0ull;
        for (int i0 = 0; i0 < nc; i0 += 4) {
          PBox rb[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) rb[u] = load_pbox(cbox, caf, (i0 + u < 64) ? (i0 + u) : 63);
          bool irr = false;
          bool bit[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt_fast(rb[u], me, p.iou, irr);
          if (__builtin_expect(ballot64(irr || p.iou.slow_only) != 0ull, 0)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt(to_nbox(rb[u]), to_nbox(me), p.iou.thr);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned long long word = ballot64(bit[u] & (i0 + u < lane) & (lane < nc));
            if (lane == i0 + u) own = word;
          }
        }
        // c. greedy resolve (scalar), rows that suppress nothing are accepted in bulk
        const unsigned long long rem0 = ballot64(d);
        const unsigned long long nz = ballot64(own != 0ull);
        const unsigned long long valid = (nc >= 64) ? ~0ull : ((1ull << nc) - 1ull);
        unsigned long long rem = uniform64(rem0);
        unsigned long long alive = uniform64(~rem & valid);
        unsigned long long keptbits = 0ull;
        int cnt2 = nk;
        while (alive != 0ull && cnt2 < Rk) {
          const unsigned long long nza = alive & nz;
          const int f = (nza != 0ull) ? (int)__builtin_ctzll(nza) : 64;
          unsigned long long batch = (f >= 64) ? alive : (alive & ((1ull << f) - 1ull));
          int nb = __popcll(batch);
          if (nb > 0) {
            if (cnt2 + nb > Rk) {
              unsigned long long t = 0ull, bb = batch;
              for (int k = cnt2; k < Rk; ++k) {
                const unsigned long long low = bb & (0ull - bb);
                t |= low;
                bb ^= low;
              }
              batch = t;
              nb = Rk - cnt2;
            }
            keptbits |= batch;
            cnt2 += nb;
            alive &= ~batch;
            if (cnt2 >= Rk) break;
          }
          if (f < 64) {
            keptbits |= (1ull << f);
            ++cnt2;
            rem |= readlane64(own, f);
            alive &= ~(rem | keptbits);
          }
        }
        if ((keptbits >> lane) & 1ull) {
          const int kp = nk + __popcll(keptbits & lanemask_lt());
          kbox[kp] = make_float4(me.ymin, me.xmin, me.ymax, me.xmax);
          kaf[kp] = make_float2(me.area, __int_as_float(me.flag));
          klist[s0 + kp] = (uint16_t)slot;
        }
        nk += __popcll(keptbits);
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) kcount[cls] = nk;
    }
  }
  __syncthreads();

  FRCNN_STAMP(24);
  // ---- D. cross-class merge: rank of every kept entry, output ------------------------------------------------
  // koff (exclusive prefix of kcount) reuses ccount; wave 0 scans.
  int *koff = ccount;
  if (w == 0) {
    const int c0 = kcount[4 * lane], c1 = kcount[4 * lane + 1], c2 = kcount[4 * lane + 2], c3 = kcount[4 * lane + 3];
    int inc = c0 + c1 + c2 + c3;
    const int s4 = inc;
    inc = wave_incl_scan(inc);
    const int ex = inc - s4;
    koff[4 * lane] = ex;
    koff[4 * lane + 1] = ex + c0;
    koff[4 * lane + 2] = ex + c0 + c1;
    koff[4 * lane + 3] = ex + c0 + c1 + c2;
    if (lane == 63) misc[0] = inc;
  }
  __syncthreads();
  FRCNN_STAMP(26);
  const int total = misc[0];
  const int nvalid = total < T ? total : T;
  // The rank of a kept entry = its position in its own class + the number of entries of every other class that precede it
  // (binary searches: kept lists are ascending in key = descending score).  One work item per (entry, other class), spread
  // over the whole workgroup, partial ranks summed in LDS (hist and perm are free by now); per-entry loops when the entries
  // do not fit those arrays.
  uint32_t *rank_acc = hist;   // [total]
  uint16_t *ecls = perm;       // [total] class of entry e
  const bool spread = total <= kNW * 256;
  auto class_of = [&](int e) -> int {
    int lo = 0, hi = L - 1;  // last class with koff[cls] <= e
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (koff[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  auto preceding = [&](int cls, uint32_t mykey, int c2) -> int {  // entries of class c2 before (mykey, cls)
    const int n2 = kcount[c2];
    const uint16_t *l2 = klist + cstart[c2];
    int lo = 0, hi = n2;
    if (c2 < cls) {  // equal keys: the lower class goes first
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[l2[mid]] <= mykey) lo = mid + 1; else hi = mid;
      }
    } else {
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[l2[mid]] < mykey) lo = mid + 1; else hi = mid;
      }
    }
    return lo;
  };
  if (spread) {
    for (int e = tid; e < total; e += kNT) {
      const int cls = class_of(e);
      ecls[e] = (uint16_t)cls;
      rank_acc[e] = (uint32_t)(e - koff[cls]);
    }
    __syncthreads();
    const long long items = (long long)total * L;
    for (long long it = tid; it < items; it += kNT) {
      const int e = (int)(it / L);
      const int c2 = (int)(it - (long long)e * L);
      const int cls = (int)ecls[e];
      if (c2 == cls || kcount[c2] == 0) continue;
      const int slot = (int)klist[cstart[cls] + (e - koff[cls])];
      const int before = preceding(cls, key[slot], c2);
      if (before) atomicAdd(&rank_acc[e], (uint32_t)before);
    }
    __syncthreads();
  }
  for (int e = tid; e < total; e += kNT) {
    const int cls = spread ? (int)ecls[e] : class_of(e);
    const int kpos = e - koff[cls];
    const int slot = (int)klist[cstart[cls] + kpos];
    int rank;
    if (spread) {
      rank = (int)rank_acc[e];
    } else {
      const uint32_t mykey = key[slot];
      rank = kpos;
      for (int c2 = 0; c2 < L; ++c2)
        if (c2 != cls && kcount[c2] != 0) rank += preceding(cls, mykey, c2);
    }
    if (rank < T) {
      const int r = (int)(pk[slot] & 0xFFFFFFu);
      float4 o = box_of(r, cls);
      o.x = clip01(o.x);
      o.y = clip01(o.y);
      o.z = clip01(o.z);
      o.w = clip01(o.w);
      *reinterpret_cast<float4 *>(io.out_boxes + ((long long)b * T + rank) * 4) = o;
      if (io.out_scores) io.out_scores[(long long)b * T + rank] = probs[(long long)r * L + cls];
      if (io.out_labels) io.out_labels[(long long)b * T + rank] = (float)cls;
    }
  }
  for (int r = nvalid + tid; r < T; r += kNT) {
    *reinterpret_cast<float4 *>(io.out_boxes + ((long long)b * T + r) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (io.out_scores) io.out_scores[(long long)b * T + r] = 0.0f;
    if (io.out_labels) io.out_labels[(long long)b * T + r] = 0.0f;
  }
  if (tid == 0 && io.out_valid) io.out_valid[b] = nvalid;
  FRCNN_STAMP(25);
}

// *handled stays false when the fused kernel cannot hold the problem in LDS; the caller then takes the
// generic sort -> NMS -> merge pipeline.
int launch_decoder(const float *const *rois, const float *const *deltas, const float *const *probs, int groups,
                   const float *variances, int B, int R, int L, int max_total, float score_thr, float iou_thr,
                   float *const *out_boxes, float *const *out_labels, float *const *out_scores, int *const *out_valid,
                   hipStream_t stream, bool small_footprint, bool *handled) {
  *handled = false;
  if (knob_set(kKnobDecoderThreads)) small_footprint = knob(kKnobDecoderThreads, 0) == 256;  // diagnostic override
  const int nt = small_footprint ? 256 : 1024;
  if (groups < 1 || groups > kDecoderGroups) return FRCNN_OK;
  // Rows whose argmax is background are ZEROED by the reference (models/faster_rcnn.py:83), not dropped: with a negative
  // threshold their 0 scores are candidates (0 > thr).  The fused kernel drops them, so it only takes thresholds >= 0.
  if (!(score_thr >= 0.0f)) return FRCNN_OK;
  if (L > kDecMaxClasses || R >= (1 << 24) || R <= 0 || max_total <= 0) return FRCNN_OK;
  const long long rl = (long long)R * L;
  if (rl > 65535) return FRCNN_OK;
  const int Vcap = (int)((rl + 7) & ~7ll);
  const int Rpc = max_total < R ? max_total : R;
  const int Rpc4 = (Rpc + 3) & ~3;
  const int capV = Vcap < nt ? Vcap : nt;  // the small-V schedule takes images with up to this many candidates
  int nms_waves = small_footprint ? 4 : 8;
  while (nms_waves > 1 && dec_lds_bytes(Vcap, Rpc4, nms_waves, nt, capV) > 160 * 1024) nms_waves >>= 1;
  const size_t lds = dec_lds_bytes(Vcap, Rpc4, nms_waves, nt, capV);
  if (lds > 160 * 1024) return FRCNN_OK;
  DecoderParams p;
  for (int g = 0; g < kDecoderGroups; ++g) {
    const int s = g < groups ? g : 0;
    p.io[g].rois = rois[s];
    p.io[g].deltas = deltas[s];
    p.io[g].probs = probs[s];
    p.io[g].out_boxes = out_boxes[s];
    p.io[g].out_labels = out_labels ? out_labels[s] : nullptr;
    p.io[g].out_scores = out_scores ? out_scores[s] : nullptr;
    p.io[g].out_valid = out_valid ? out_valid[s] : nullptr;
  }
  p.images_per_group = B;
  p.v0 = variances[0];
  p.v1 = variances[1];
  p.v2 = variances[2];
  p.v3 = variances[3];
  p.R = R;
  p.L = L;
  p.T = max_total;
  p.Rpc = Rpc;
  p.score_thr = score_thr;
  p.iou = make_iou_thr(iou_thr);
  auto kern = small_footprint ? decoder_kernel<256> : decoder_kernel<1024>;
  if (lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "decoder: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * groups)), dim3(nt), lds, stream, p, Vcap, Rpc4, nms_waves, capV);
  *handled = true;
  return check_launch("decoder (fused kernel)");
}

}  // namespace frcnn
