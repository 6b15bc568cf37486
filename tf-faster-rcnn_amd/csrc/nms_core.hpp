// nms_core.hpp -- the chunked greedy-NMS step shared by nms.hip (generic combined NMS / Decoder) and
// proposal.hip (fused RoIBBox).  See nms.hip for the algorithm description.
#pragma once

#include "common.hpp"

namespace frcnn {

static const int kMaxKept = 5000;  // kept-list capacity in LDS (28 B per entry)

// ---- TF's "iou > iou_threshold" without the division ------------------------------------------------
// The reference computes q = fl32(inter / uni) and tests q > thr.  With thr' the next fp32 above thr and
// mid = (thr + thr') / 2, that is  inter/uni > mid  (>= when the tie rounds up), so the sign of
// inter - mid * uni decides.  fp32 suffices away from the boundary:
//   t = fl(thr * uni) is within 2^-24 (relative) of thr * uni and mid is within 2^-24 of thr, so
//   inter > t * (1 + 2^-21)  =>  true,      inter < t * (1 - 2^-21)  =>  false,
// and only the pairs inside that 2^-20-wide band (about one in a million near-threshold pairs), irregular
// boxes (non-finite coordinates, area outside [1e-30, 5e29]) or unusual thresholds (outside [1e-6, 1e6])
// are re-evaluated with the reference formulation, division included (nms_iou_gt).  The fast and the
// reference forms therefore agree on every input; tests/test_host.py checks the band logic on the CPU.
// The PACKED form of the filter (nms_iou_gt_fast2: both boxes regular) goes one step further and never forms the union:
// with s = fl(area_i + area_j), the sum the reference forms too, uni = fl(s - inter) and the real ratio r = inter / (s - inter),
//   q = r * (1 + d2) / (1 + d1), |d1|, |d2| <= 2^-24,   and   r > thr  <=>  inter > s * k,  k = thr / (1 + thr);
//   inter > s * k * (1 + e)  =>  r > thr * (1 + e) / (1 - thr * e) >= thr * (1 + e),   and likewise below.
// With k_hi >= k * (1 + 2^-21) and k_lo <= k * (1 - 2^-21) (formed in double, rounded outwards), and one more 2^-24 for the
// product fl(s * k_hi):  inter > fl(s * k_hi) => q > thr,   inter < fl(s * k_lo) => q <= thr  -- two multiplies per pair where
// the form above needs a subtraction and three.  tests/test_host.py checks both forms against the division.
struct IouThr {
  float thr;
  int slow_only;
  float k_hi, k_lo;
};

static inline IouThr make_iou_thr(float thr) {
  IouThr t;
  t.thr = thr;
  t.slow_only = !(thr >= 1e-6f && thr <= 1e6f);
  t.k_hi = t.k_lo = 0.0f;
  if (!t.slow_only) {
    const double k = (double)thr / (1.0 + (double)thr);
    t.k_hi = nextafterf((float)(k * (1.0 + 4.76837158203125e-07)), 2.0f);
    t.k_lo = nextafterf((float)(k * (1.0 - 4.76837158203125e-07)), 0.0f);
  }
  return t;
}

// candidate flags
static const int kFlagArea = 1;     // area passes TF's guard, i.e. !(area <= 0)
static const int kFlagRegular = 2;  // finite coordinates and area in [1e-30, 5e29]: the band filter may decide pairs of
                                    // such boxes (their union then lies in [1e-30, 1e30]); anything else is re-evaluated
                                    // with the reference formulation

__device__ __forceinline__ int box_flags(float4 b, const NBox &nb) {
  const bool finite = (fabsf(b.x) <= 3.402823466e+38f) && (fabsf(b.y) <= 3.402823466e+38f) &&
                      (fabsf(b.z) <= 3.402823466e+38f) && (fabsf(b.w) <= 3.402823466e+38f);
  const bool regular = finite && (nb.area >= 1e-30f) && (nb.area <= 5e29f);
  return (!(nb.area <= 0.0f) ? kFlagArea : 0) | (regular ? kFlagRegular : 0);
}

// Packed candidate: corner-normalised box + area + flags (LDS: one float4 and one float2 per candidate,
// so a broadcast row costs a ds_read_b128 and a ds_read_b64 instead of six ds_read_b32).
struct PBox {
  float ymin, xmin, ymax, xmax, area;
  int flag;
};

__device__ __forceinline__ PBox load_pbox(const float4 *box, const float2 *af, int i) {
  const float4 b = box[i];
  const float2 a = af[i];
  PBox p;
  p.ymin = b.x;
  p.xmin = b.y;
  p.ymax = b.z;
  p.xmax = b.w;
  p.area = a.x;
  p.flag = __float_as_int(a.y);
  return p;
}

__device__ __forceinline__ PBox load_pbox(const float4 *box, const float *area, const int *flag, int i) {
  const float4 b = box[i];
  PBox p;
  p.ymin = b.x;
  p.xmin = b.y;
  p.ymax = b.z;
  p.xmax = b.w;
  p.area = area[i];
  p.flag = flag[i];
  return p;
}

__device__ __forceinline__ NBox to_nbox(const PBox &p) {
  NBox n;
  n.ymin = p.ymin;
  n.xmin = p.xmin;
  n.ymax = p.ymax;
  n.xmax = p.xmax;
  n.area = p.area;
  return n;
}

// Straight-line filter form of nms_iou_gt; `irr` is OR-ed with "this pair must be re-evaluated with
// nms_iou_gt" (the caller does that once per group of tests, which keeps the group free of branches so
// that independent tests overlap).
__device__ __forceinline__ bool nms_iou_gt_fast(const PBox &i, const PBox &j, const IouThr &t, bool &irr) {
  // finite coordinates (anything else lacks kFlagRegular and is re-evaluated): v_max/v_min equal std::max/std::min up
  // to the sign of zero, which cannot change the outcome (it only reaches comparisons and products with exact zeros)
  float ih = hw_min(i.ymax, j.ymax) - hw_max(i.ymin, j.ymin);
  float iw = hw_min(i.xmax, j.xmax) - hw_max(i.xmin, j.xmin);
  ih = hw_max(ih, 0.0f);
  iw = hw_max(iw, 0.0f);
  const float inter = ih * iw;
  float uni = i.area + j.area;
  uni = uni - inter;
  const float tt = t.thr * uni;
  const float hi = tt * 1.000000476837158203125f;  // 1 + 2^-21
  const float lo = tt * 0.999999523162841796875f;  // 1 - 2^-21
  const bool yes = inter > hi;
  const bool no = inter < lo;
  // max(area_i, area_j) <= uni <= area_i + area_j: two regular boxes have a union in [1e-30, 1e30]
  // (a regular box always has the area flag too: the AND of two flag words is 0, kFlagArea or kFlagArea | kFlagRegular)
  const int both = i.flag & j.flag;
  const bool area_ok = both != 0;
  const bool regular = both == (kFlagArea | kFlagRegular);
  // irregular or inside the band: exact re-evaluation needed (unless TF's area guard already says "no")
  irr = irr | (area_ok & !(regular & (yes | no)));
  return yes & area_ok;
}

// The band filter for TWO row boxes (a, b) against one box `me`, all three REGULAR (finite, area in [1e-30, 5e29]: no
// flags to look at), as packed fp32: the subtractions, products and scalings of the two tests share v_pk_add_f32 /
// v_pk_mul_f32 (10 VALU instructions per pair instead of 19; only max / min have no packed form), in the union-free form
// described at IouThr.  a_area2 = (area a, area b).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nms_iou_gt_fast2(const float4 &me, float me_area, const float4 &a, const float4 &b, v2f ab_area,
                                                 const IouThr &t, bool &yes_a, bool &yes_b, bool &irr) {
  v2f mny, mxy, mnx, mxx;
  mny.x = hw_min(me.z, a.z);
  mny.y = hw_min(me.z, b.z);
  mxy.x = hw_max(me.x, a.x);
  mxy.y = hw_max(me.x, b.x);
  mnx.x = hw_min(me.w, a.w);
  mnx.y = hw_min(me.w, b.w);
  mxx.x = hw_max(me.y, a.y);
  mxx.y = hw_max(me.y, b.y);
  v2f ih = mny - mxy;
  const v2f iw = mnx - mxx;
  ih.x = hw_max(ih.x, 0.0f);
  ih.y = hw_max(ih.y, 0.0f);
  // iw is NOT clamped at 0: with ih >= 0 a negative iw gives inter <= 0 (or -inf, or NaN from 0 * -inf), and the sum of two
  // regular areas is > 0, so the pair comes out as "no" (inter < lo, lo > 0) -- which is what the clamped product 0 gives
  // -- or, for the NaN, as undecided and goes to the exact test.
  const v2f inter = ih * iw;
  const v2f me2 = {me_area, me_area};
  const v2f sum = me2 + ab_area;
  const v2f k_hi = {t.k_hi, t.k_hi};
  const v2f k_lo = {t.k_lo, t.k_lo};
  const v2f hi = sum * k_hi;
  const v2f lo = sum * k_lo;
  yes_a = inter.x > hi.x;
  yes_b = inter.y > hi.y;
  const bool no_a = inter.x < lo.x, no_b = inter.y < lo.y;
  irr = irr | !(yes_a | no_a) | !(yes_b | no_b);
}

// LDS layout for chunks of CH candidates (CH = 64 * G).
template <int CH>
struct NmsLds {
  unsigned long long *mask;  // [CH * G] intra-chunk column words, mask[column * G + row group]
  unsigned long long *dead;  // [4] per 64-candidate group: candidates removed so far
  unsigned long long *reg;   // [4] per group: candidates whose flags are kFlagArea | kFlagRegular
  float4 *c_box;             // [CH] chunk boxes (corner-normalised)
  float *c_area;             // [CH]
  int *c_flag;               // [CH]
  int *c_idx;                // [CH] element index of each chunk candidate
  int *s_nk;                 // [4]: [0] the kept count handed back, [1] every kept box so far is regular
  float4 *k_box;             // [Rc] kept boxes; the entries from the kept count up to the next multiple of 4 are null boxes
  float *k_area;             // [Rc]
  int *k_flag;               // [Rc]
  int *k_idx;                // [Rc]
};

template <int CH>
__host__ __device__ static inline size_t nms_lds_bytes(int Rc) {
  return (size_t)CH * (CH / 64) * 8 + 8 * 8 + (size_t)CH * 28 + 4 * 4 + (size_t)Rc * 28;
}

// Rc must be a multiple of 4 (keeps every sub-array 16-byte aligned).
template <int CH>
__device__ __forceinline__ NmsLds<CH> carve_nms_lds(unsigned char *smem, int Rc) {
  NmsLds<CH> L;
  L.mask = reinterpret_cast<unsigned long long *>(smem);
  L.dead = L.mask + CH * (CH / 64);
  L.reg = L.dead + 4;
  L.c_box = reinterpret_cast<float4 *>(L.reg + 4);
  L.k_box = L.c_box + CH;
  L.c_area = reinterpret_cast<float *>(L.k_box + Rc);
  L.k_area = L.c_area + CH;  // (16-byte aligned: CH and Rc are multiples of 4)
  L.c_flag = reinterpret_cast<int *>(L.k_area + Rc);
  L.k_flag = L.c_flag + CH;
  L.c_idx = L.k_flag + Rc;
  L.k_idx = L.c_idx + CH;
  L.s_nk = L.k_idx + Rc;
  return L;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  l = __builtin_amdgcn_readfirstlane(l);  // the lane select must live in an SGPR
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// The greedy order inside a block of 64 candidates, in parallel rounds: col = the earlier rows of the block that suppress
// this lane's candidate, und0 (wave-uniform) = the candidates not yet decided.  A candidate hit by a KEPT row is removed, a
// candidate none of whose still-undecided earlier rows suppresses it is kept; the lowest undecided candidate always falls in
// one of the two classes, so every round makes progress and the fixed point is the sequential greedy result.  Returns the
// kept candidates (wave-uniform).  The undecided / kept sets live in scalar registers: a lane only ANDs them with its
// column word, and whether the lane itself is still undecided is applied to the ballots, not to the lanes.
__device__ __forceinline__ unsigned long long resolve_rounds(unsigned long long col, unsigned long long und0) {
  const uint32_t col_lo = (uint32_t)col, col_hi = (uint32_t)(col >> 32);
  unsigned long long und = und0, keptbits = 0ull;
  while (und != 0ull) {
    const bool hit = ((col_lo & (uint32_t)keptbits) | (col_hi & (uint32_t)(keptbits >> 32))) != 0u;
    und &= ~ballot64(hit);
    const bool blocked = ((col_lo & (uint32_t)und) | (col_hi & (uint32_t)(und >> 32))) != 0u;
    const unsigned long long k = und & ~ballot64(blocked);
    keptbits |= k;
    und &= ~k;
  }
  return keptbits;
}

// ---- pair-test helpers shared by the chunk step and the slice loop ------------------------------------------------------
struct KeptLds {
  float4 *box;  // [Rc] (Rc a multiple of 4; null boxes from the kept count up to the next multiple of 4)
  float *area;
  int *flag;
  int *idx;
};

// "some kept box of this wave's share suppresses me": the share is groups `first`, first + step, ... of four kept boxes.
__device__ __forceinline__ bool kept_list_hits(const KeptLds &K, const PBox &me, bool live, int first, int step, int nk,
                                               bool all_regular, const IouThr &thr) {
  bool d = false;
  const int ngroups = (nk + 3) >> 2;
  if (all_regular) {
    // No ballot and no branch inside the loop (the next group's LDS reads overlap this group's arithmetic).  The band filter
    // never says "suppresses" wrongly, it can only leave a pair undecided: if that happened to a live lane anywhere in the
    // wave's share, the share is gone through again with the reference formulation and OR-ed on top.
    const float4 meb = make_float4(me.ymin, me.xmin, me.ymax, me.xmax);
    bool irr = false;
    for (int g4 = first; g4 < ngroups; g4 += step) {
      const float4 *kb = K.box + 4 * g4;
      const float4 b0 = kb[0], b1 = kb[1], b2 = kb[2], b3 = kb[3];
      const float4 ar = *reinterpret_cast<const float4 *>(K.area + 4 * g4);
      const v2f a01 = {ar.x, ar.y}, a23 = {ar.z, ar.w};
      bool y0, y1, y2, y3;
      nms_iou_gt_fast2(meb, me.area, b0, b1, a01, thr, y0, y1, irr);
      nms_iou_gt_fast2(meb, me.area, b2, b3, a23, thr, y2, y3, irr);
      d = d | y0 | y1 | y2 | y3;
    }
    if (__builtin_expect(ballot64(irr && live) != 0ull, 0)) {
      const NBox men = to_nbox(me);
      for (int g4 = first; g4 < ngroups; g4 += step) {
        for (int u = 0; u < 4; ++u) {
          const float4 bb = K.box[4 * g4 + u];
          NBox kn;
          kn.ymin = bb.x;
          kn.xmin = bb.y;
          kn.ymax = bb.z;
          kn.xmax = bb.w;
          kn.area = K.area[4 * g4 + u];
          d = d | nms_iou_gt(men, kn, thr.thr);
        }
      }
    }
  } else {
    for (int g4 = first; g4 < ngroups; g4 += step) {
      PBox kb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kb[u] = load_pbox(K.box, K.area, K.flag, 4 * g4 + u);  // (null boxes carry flag 0)
      bool irr = false, dd = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt_fast(me, kb[u], thr, irr);
      if (__builtin_expect(ballot64(irr || thr.slow_only) != 0ull, 0)) {
        dd = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) dd = dd | nms_iou_gt(to_nbox(me), to_nbox(kb[u]), thr.thr);
      }
      d = d | dd;
      if (ballot64(live && !d) == 0ull) break;
    }
  }
  return d;
}

// bits u = "row irow0 + u suppresses me" for four consecutive rows (IoU only: the caller masks order and validity)
__device__ __forceinline__ uint32_t tile_rows4(const float4 *box, const float *area, const int *flag, int irow0, const PBox &me,
                                               bool packed, const IouThr &thr) {
  bool bit[4];
  if (packed) {  // wave-uniform: the four rows and every live column are regular
    const float4 meb = make_float4(me.ymin, me.xmin, me.ymax, me.xmax);
    const float4 r0 = box[irow0], r1 = box[irow0 + 1], r2 = box[irow0 + 2], r3 = box[irow0 + 3];
    const float4 ar = *reinterpret_cast<const float4 *>(area + irow0);
    const v2f a01 = {ar.x, ar.y}, a23 = {ar.z, ar.w};
    bool irr = false;
    nms_iou_gt_fast2(meb, me.area, r0, r1, a01, thr, bit[0], bit[1], irr);
    nms_iou_gt_fast2(meb, me.area, r2, r3, a23, thr, bit[2], bit[3], irr);
    if (__builtin_expect(ballot64(irr) != 0ull, 0)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt(to_nbox(load_pbox(box, area, flag, irow0 + u)), to_nbox(me), thr.thr);
    }
  } else {
    PBox rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = load_pbox(box, area, flag, irow0 + u);
    bool irr = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt_fast(rb[u], me, thr, irr);
    if (__builtin_expect(ballot64(irr || thr.slow_only) != 0ull, 0)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt(to_nbox(rb[u]), to_nbox(me), thr.thr);
    }
  }
  return (uint32_t)bit[0] | ((uint32_t)bit[1] << 1) | ((uint32_t)bit[2] << 2) | ((uint32_t)bit[3] << 3);
}

// One chunk of nc (<= CH) candidates, in priority order, whose packed boxes (L.c_box, L.c_af), element indices (L.c_idx)
// and initial dead words (L.dead[g]: candidates that fail the score threshold) are already in LDS -- written by this
// workgroup, but not necessarily published yet: the function starts with a barrier.  L.c_* may point anywhere in LDS (a
// slice decoded ahead of time); entries [nc, CH) are read but never influence a result.  nk = boxes kept so far; returns
// the new count (<= R).  Must be called by all NT threads of the block.
template <int NT, int CH>
__device__ __forceinline__ int nms_chunk_resolve(const NmsLds<CH> &L, int nc, int nk, int R, const IouThr &thr);

// The same with the chunk loaded here: cand(j, box, ei, bad) fills candidate j's box (original corner order), element
// index and "fails the score threshold" flag (called by threads j < nc).
template <int NT, int CH, class CandF>
__device__ __forceinline__ int nms_chunk(const NmsLds<CH> &L, int nc, int nk, int R, const IouThr &thr, CandF cand) {
  static_assert(NT >= CH && NT % CH == 0, "block must cover the chunk");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < CH) {
    NBox nb;
    nb.ymin = nb.xmin = nb.ymax = nb.xmax = nb.area = 0.0f;
    int ei = -1, fl = 0;
    bool bad = false;
    if (tid < nc) {
      float4 box;
      cand(tid, box, ei, bad);
      nb = normalise_box(box);
      fl = box_flags(box, nb);
    }
    L.c_box[tid] = make_float4(nb.ymin, nb.xmin, nb.ymax, nb.xmax);
    L.c_area[tid] = nb.area;
    L.c_flag[tid] = fl;
    L.c_idx[tid] = ei;
    const unsigned long long badw = ballot64(bad);  // candidates that fail the score threshold start dead
    const unsigned long long regw = ballot64(fl == (kFlagArea | kFlagRegular));
    if (lane == 0) {
      L.dead[w] = badw;
      L.reg[w] = regw;
    }
  }
  return nms_chunk_resolve<NT, CH>(L, nc, nk, R, thr);
}

template <int NT, int CH>
__device__ __forceinline__ int nms_chunk_resolve(const NmsLds<CH> &L, int nc, int nk, int R, const IouThr &thr) {
  static_assert(CH == 64 || CH == 128 || CH == 256, "chunk must be 1, 2 or 4 waves wide");
  static_assert(NT >= CH && NT % CH == 0, "block must cover the chunk");
  constexpr int G = CH / 64;
  constexpr int NWV = NT / 64;
  constexpr int NSL = NT / CH;                  // slices of the kept list handled in parallel
  constexpr int TILES = G * (G + 1) / 2;        // upper-triangular 64x64 tiles
  // row splits per tile: the largest power of two (<= 16) that keeps the work items within 4 per wave
  constexpr int SPLIT = (TILES * 16 <= 4 * NWV) ? 16 : (TILES * 8 <= 4 * NWV) ? 8 : (TILES * 4 <= 4 * NWV) ? 4
                        : (TILES * 2 <= 4 * NWV) ? 2 : 1;
  constexpr int ROWS = 64 / SPLIT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < CH * G; i += NT) L.mask[i] = 0ull;  // column words, OR-ed together in phase b
  __syncthreads();
  FRCNN_STAMP(10);

  // ---- a. chunk vs kept list, four kept boxes per step -------------------------------------------
  if (nk > 0) {
    const int j = tid & (CH - 1);
    const int slice = tid / CH;
    if ((j & ~63) < nc) {  // wave-uniform: this wave's 64 candidates exist
      const PBox me = load_pbox(L.c_box, L.c_area, L.c_flag, j);
      const unsigned long long valid_w = (nc - (j & ~63) >= 64) ? ~0ull : ((1ull << (nc - (j & ~63))) - 1ull);
      // every kept box and every candidate of this wave regular: packed band filter (nms_iou_gt_fast2); the wave's share of
      // the kept list is the groups slice, slice + NSL, ... of four kept boxes (null boxes pad the list to a multiple of 4)
      const bool all_regular = L.s_nk[1] != 0 && (uniform64(~L.reg[j >> 6]) & valid_w) == 0ull && !thr.slow_only;
      KeptLds kept;
      kept.box = L.k_box;
      kept.area = L.k_area;
      kept.flag = L.k_flag;
      kept.idx = L.k_idx;
      const bool d = kept_list_hits(kept, me, j < nc, slice, NSL, nk, all_regular, thr);
      const unsigned long long word = ballot64(d && (j < nc));
      if (lane == 0 && word != 0ull) atomicOr(&L.dead[j >> 6], word);
    }
    __syncthreads();  // phase b skips the rows that are dead by now
  }
  FRCNN_STAMP(11);

  // ---- b. intra-chunk IoU tiles (upper triangle) as COLUMN words ----------------------------------
  // lane = column candidate; mask[col * G + rg] collects over the rows of row group rg the bits "row suppresses col and
  // comes before it".  A work item is ROWS rows of one tile; rows already dead are skipped four at a time (a suppressed
  // candidate suppresses nobody), partial words are OR-ed in LDS.
  for (int item = w; item < TILES * SPLIT; item += NWV) {
    const int tile = item / SPLIT;
    const int part = item - tile * SPLIT;
    // tile -> (rg, cg), cg >= rg, row-major over the upper triangle
    int rg = 0, rem = tile;
#pragma unroll
    for (int r = 0; r < G; ++r) {
      if (rg == r && rem >= G - rg) {
        rem -= G - rg;
        ++rg;
      }
    }
    const int cg = rg + rem;
    if (rg * 64 >= nc || cg * 64 >= nc) continue;  // wave-uniform
    const int jcol = cg * 64 + lane;
    const PBox me = load_pbox(L.c_box, L.c_area, L.c_flag, jcol);
    const unsigned long long alive_rows = uniform64(~L.dead[rg]);
    const unsigned long long reg_rows = uniform64(L.reg[rg]);
    const bool cols_regular = ballot64(me.flag != (kFlagArea | kFlagRegular) && jcol < nc) == 0ull && !thr.slow_only;
    static_assert(ROWS % 4 == 0, "rows per item must be a multiple of 4");
    unsigned long long bits = 0ull;
    for (int i0 = 0; i0 < ROWS; i0 += 4) {
      const int rl0 = part * ROWS + i0;  // row offset inside the group
      if (((alive_rows >> rl0) & 0xFull) == 0ull) continue;  // wave-uniform
      const int irow0 = rg * 64 + rl0;
      const bool packed = cols_regular && ((reg_rows >> rl0) & 0xFull) == 0xFull;  // wave-uniform
      const uint32_t nib = tile_rows4(L.c_box, L.c_area, L.c_flag, irow0, me, packed, thr);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool hit = ((nib >> u) & 1u) && (irow0 + u < jcol) && (jcol < nc);
        bits |= (unsigned long long)hit << (rl0 + u);
      }
    }
    if (bits != 0ull) atomicOr(&L.mask[jcol * G + rg], bits);
  }
  __syncthreads();
  FRCNN_STAMP(12);

  // ---- c. greedy resolve on wave 0, one 64-candidate group at a time, in parallel rounds ------------------
  // Lane i knows which earlier rows of its group suppress it (col).  A candidate hit by a KEPT row is removed; a
  // candidate none of whose still-undecided earlier rows suppresses it is kept (whatever those rows turn out to be, they
  // do not touch it).  The lowest undecided candidate always falls in one of the two classes, so every round makes
  // progress; the fixed point is the sequential greedy result, reached in (longest suppression chain) rounds.
  if (w == 0) {
    unsigned long long deadw[G];
#pragma unroll
    for (int g = 0; g < G; ++g) deadw[g] = uniform64(L.dead[g]);
    int cnt = nk;
    int kreg = (nk == 0) ? 1 : L.s_nk[1];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int left = nc - g * 64;
      if (left > 0 && cnt < R) {  // wave-uniform
        const unsigned long long valid = (left >= 64) ? ~0ull : ((1ull << left) - 1ull);
        const unsigned long long col = L.mask[(g * 64 + lane) * G + g];
        unsigned long long keptbits = resolve_rounds(col, uniform64(~deadw[g] & valid));
        if (cnt + __popcll(keptbits) > R) {  // the greedy scan stops at R: keep the first R - cnt of them
          unsigned long long t = 0ull, bb = keptbits;
          for (int k = cnt; k < R; ++k) {
            const unsigned long long low = bb & (0ull - bb);
            t |= low;
            bb ^= low;
          }
          keptbits = t;
        }
        // the kept rows suppress into the later groups of the chunk
#pragma unroll
        for (int w2 = g + 1; w2 < G; ++w2)
          if (w2 * 64 < nc) deadw[w2] |= ballot64((L.mask[(w2 * 64 + lane) * G + g] & keptbits) != 0ull);
        if ((keptbits >> lane) & 1ull) {
          const int pos = cnt + __popcll(keptbits & lanemask_lt());
          const int j = g * 64 + lane;
          L.k_box[pos] = L.c_box[j];
          L.k_area[pos] = L.c_area[j];
          L.k_flag[pos] = L.c_flag[j];
          L.k_idx[pos] = L.c_idx[j];
        }
        cnt += __popcll(keptbits);
        if ((keptbits & ~uniform64(L.reg[g])) != 0ull) kreg = 0;
      }
    }
    // null boxes up to the next multiple of 4 (the packed kept-list tests read whole groups of four; the kept-list
    // arrays hold a multiple of 4 entries)
    if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) {
      L.k_box[cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
      L.k_area[cnt + lane] = 0.0f;
      L.k_flag[cnt + lane] = 0;
    }
    if (lane == 0) {
      L.s_nk[0] = cnt;
      L.s_nk[1] = kreg;
    }
  }
  __syncthreads();
  FRCNN_STAMP(13);
  return L.s_nk[0];
}

// ---- a whole pre-decoded slice in chunks of 64, two barriers per chunk (the fused RoIBBox kernel) --------------------------
// The slice's candidates stand in LDS in priority order: sbox / sarea / sflag / sidx, per 64 the words sbad ("fails the score
// threshold") and sreg ("regular box"); the arrays are filled (zero boxes) up to a multiple of 64.  Against nms_chunk_resolve
// at CH = 64: the intra-chunk tile of chunk c + 1 does not depend on chunk c's outcome, so waves 1.. compute it while wave 0
// resolves chunk c (the tile waves OR their pieces into double-buffered column words, cleared a phase ahead, so that the
// resolve reads one word per column); the kept-list tests of a chunk OR their ballots into a word that wave 0 preset with sbad.
struct Slice64Lds {
  unsigned long long *dead;  // [2]  dead word of the current / next chunk
  unsigned long long *colw;  // [2][64]  column words of the current / next chunk's tile (OR-ed together by the tile waves)
  int *s_nk;                 // [4]  [0] kept count, [1] every kept box so far is regular
};
template <int NT>
__host__ __device__ static inline size_t slice64_lds_bytes() {
  return 2 * 8 + 2 * 64 * 8 + 4 * 4;
}

// Returns the new kept count.  Must be called by all NT threads; the slice arrays must be published (a barrier behind
// their last write).  V <= the arrays' filled length; nk may be > 0 (earlier slices).
template <int NT>
__device__ __forceinline__ int nms_slice64(const float4 *sbox, const float *sarea, const int *sflag, const int *sidx,
                                           const unsigned long long *sbad, const unsigned long long *sreg, int V,
                                           const KeptLds &K, const Slice64Lds &T, int nk, int R, const IouThr &thr) {
  constexpr int NWV = NT / 64;
  constexpr int ROWS = 64 / NWV;  // rows of a tile per work item: 4 (16 waves) ... 16 (4 waves)
  static_assert(NWV >= 4 && NWV <= 16 && ROWS % 4 == 0, "nms_slice64: 4 to 16 waves");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nchunks = (V + 63) >> 6;
  const int kRegular = kFlagArea | kFlagRegular;

  // rows [item * ROWS, +ROWS) of chunk c's diagonal tile against column `lane` -> part[c & 1][item][lane]
  auto tile_item = [&](int c, int item) {
    const int cbase = c << 6;
    const int nc = (V - cbase < 64) ? (V - cbase) : 64;
    const int r0 = item * ROWS;
    uint32_t bits = 0;
    if (r0 + 1 < nc) {  // wave-uniform: some row of the item has a later live column
      const PBox me = load_pbox(sbox, sarea, sflag, cbase + lane);
      const unsigned long long regw = uniform64(sreg[c]);
      const bool cols_regular = ballot64(me.flag != kRegular && lane < nc) == 0ull && !thr.slow_only;
#pragma unroll
      for (int i0 = 0; i0 < ROWS; i0 += 4) {
        const bool packed = cols_regular && ((regw >> (r0 + i0)) & 0xFull) == 0xFull;
        bits |= tile_rows4(sbox, sarea, sflag, cbase + r0 + i0, me, packed, thr) << i0;
      }
      // row r0 + u counts for column `lane` when it comes before it; columns beyond the chunk's end count for nothing
      const int before = lane - r0;  // rows r0 .. r0 + before - 1 precede the column
      const uint32_t order = before <= 0 ? 0u : (before >= ROWS ? ((1u << ROWS) - 1u) : ((1u << before) - 1u));
      bits &= (lane < nc) ? order : 0u;
    }
    if (bits != 0u) atomicOr(&T.colw[(c & 1) * 64 + lane], (unsigned long long)bits << r0);
  };

  if (tid < 128) T.colw[tid] = 0ull;
  __syncthreads();
  if (nchunks > 0) tile_item(0, w);
  if (tid == 0) {
    T.dead[0] = sbad[0];
    T.s_nk[0] = nk;
    if (nk == 0) T.s_nk[1] = 1;
  }
  __syncthreads();
  for (int c = 0; c < nchunks && nk < R; ++c) {
    const int cbase = c << 6;
    const int nc = (V - cbase < 64) ? (V - cbase) : 64;
    if (w == 1) T.colw[((c + 1) & 1) * 64 + lane] = 0ull;  // the next chunk's words (last read two barriers ago)
    // ---- a. the chunk against the kept list: wave w takes the groups w, w + NWV, ... of four kept boxes -------------
    if (nk > 0) {
      const PBox me = load_pbox(sbox, sarea, sflag, cbase + lane);
      const unsigned long long valid_w = (nc >= 64) ? ~0ull : ((1ull << nc) - 1ull);
      const unsigned long long already = uniform64(sbad[c]);
      const bool live = ((valid_w & ~already) >> lane) & 1ull;
      const bool all_regular = T.s_nk[1] != 0 && (uniform64(~sreg[c]) & valid_w) == 0ull && !thr.slow_only;
      const bool d = kept_list_hits(K, me, live, w, NWV, nk, all_regular, thr);
      const unsigned long long word = ballot64(d && live);
      if (lane == 0 && word != 0ull) atomicOr(&T.dead[c & 1], word);
    }
    __syncthreads();
    FRCNN_STAMP(11);
    if (w == 0) {
      // ---- c. greedy resolve of the chunk in parallel rounds (see nms_chunk_resolve) ---------------------------------
      // Everything the resolve reads from LDS is requested at once, ahead of the rounds -- the dead word, the column
      // word, and the lane's own candidate in case it is kept (the compiler would otherwise wait for each in turn, five
      // round trips: it cannot tell that the kept list and the slice arrays do not overlap).
      const int j = cbase + lane;
      unsigned long long deadv = T.dead[c & 1];
      unsigned long long col = T.colw[(c & 1) * 64 + lane];
      unsigned long long regv = sreg[c];
      float4 cb = sbox[j];
      float ca = sarea[j];
      int cf = sflag[j], ci = sidx[j];
      asm volatile("" : "+v"(deadv), "+v"(col), "+v"(regv), "+v"(cb.x), "+v"(cb.y), "+v"(cb.z), "+v"(cb.w), "+v"(ca), "+v"(cf), "+v"(ci));
      const unsigned long long deadw = uniform64(deadv);
      const unsigned long long valid = (nc >= 64) ? ~0ull : ((1ull << nc) - 1ull);
      unsigned long long keptbits = resolve_rounds(col, ~deadw & valid);
      int cnt = nk;
      if (cnt + __popcll(keptbits) > R) {  // the greedy scan stops at R: keep the first R - cnt of them
        unsigned long long t = 0ull, bb = keptbits;
        for (int k = cnt; k < R; ++k) {
          const unsigned long long low = bb & (0ull - bb);
          t |= low;
          bb ^= low;
        }
        keptbits = t;
      }
      if ((keptbits >> lane) & 1ull) {
        const int pos = cnt + __popcll(keptbits & lanemask_lt());
        K.box[pos] = cb;
        K.area[pos] = ca;
        K.flag[pos] = cf;
        K.idx[pos] = ci;
      }
      cnt += __popcll(keptbits);
      if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) {  // null boxes up to the next multiple of 4
        K.box[cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        K.area[cnt + lane] = 0.0f;
        K.flag[cnt + lane] = 0;
      }
      if (lane == 0) {
        T.s_nk[0] = cnt;
        if ((keptbits & ~uniform64(regv)) != 0ull) T.s_nk[1] = 0;
        if (c + 1 < nchunks) T.dead[(c + 1) & 1] = sbad[c + 1];
      }
      FRCNN_STAMP(14);
    } else if (c + 1 < nchunks) {
      // ---- b. meanwhile: the next chunk's tile, items 0 .. NWV - 1 over the waves 1 .. NWV - 1 (wave 1 takes two) ------
      tile_item(c + 1, w - 1);
      if (w == 1) tile_item(c + 1, NWV - 1);
    }
    __syncthreads();
    FRCNN_STAMP(13);
    nk = T.s_nk[0];
  }
  return nk;
}

}  // namespace frcnn
