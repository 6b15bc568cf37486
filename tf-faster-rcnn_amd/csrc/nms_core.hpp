// nms_core.hpp -- the chunked greedy-NMS step shared by nms.hip (generic combined NMS / Decoder) and
// proposal.hip (fused RoIBBox).  See nms.hip for the algorithm description.
#pragma once

#include "common.hpp"

namespace frcnn {

static const int kMaxKept = 5000;  // kept-list capacity in LDS (28 B per entry)

// ---- exact, division-free form of TF's "iou > iou_threshold" ------------------------------------------
// With q = fl32(inter / uni) (round to nearest even) and thr' the next fp32 above thr:
//   q > thr  <=>  q >= thr'  <=>  inter/uni >= mid (or > mid), mid = (thr + thr') / 2,
// where the tie case inter/uni == mid rounds to the even neighbour: up (to thr') iff the bit pattern of
// thr is odd.  For uni > 0 this is inter >= mid * uni, and mid * uni is EXACT in fp64 (25 x 24 significant
// bits), so the comparison reproduces the fp32 division bit for bit without dividing (checked against
// the division on near-threshold pairs in tests/test_host.py).  Anything irregular (uni not a positive
// finite number, non-finite threshold, non-finite box coordinates) takes the reference formulation.
struct IouThr {
  double mid;
  float thr;
  int tie_up;
  int slow_only;
};

static inline IouThr make_iou_thr(float thr) {
  IouThr t;
  t.thr = thr;
  t.slow_only = !(thr == thr) || thr > 3.402823466e+38f || thr < -3.402823466e+38f;
  float z = (thr == 0.0f) ? 0.0f : thr;  // -0 -> +0
  uint32_t bits;
  __builtin_memcpy(&bits, &z, 4);
  t.tie_up = (int)(bits & 1u);
  float nxt = __builtin_nextafterf(z, __builtin_huge_valf());
  t.mid = t.slow_only ? 0.0 : ((double)z + (double)nxt) * 0.5;
  return t;
}

// candidate flags
static const int kFlagArea = 1;     // area passes TF's guard, i.e. !(area <= 0)
static const int kFlagSpecial = 2;  // a coordinate is NaN or infinite: use the reference formulation

__device__ __forceinline__ int box_flags(float4 b, const NBox &nb) {
  const bool finite = (fabsf(b.x) <= 3.402823466e+38f) && (fabsf(b.y) <= 3.402823466e+38f) &&
                      (fabsf(b.z) <= 3.402823466e+38f) && (fabsf(b.w) <= 3.402823466e+38f);
  return (!(nb.area <= 0.0f) ? kFlagArea : 0) | (finite ? 0 : kFlagSpecial);
}

// i, j: corner-normalised boxes; fi, fj: their flags.  Same value as nms_iou_gt(i, j, thr) for every input.
__device__ __forceinline__ bool nms_iou_gt_fast(const NBox &i, int fi, const NBox &j, int fj, const IouThr &t) {
  // finite coordinates: v_max/v_min equal std::max/std::min up to the sign of zero, which cannot change
  // the outcome (it only reaches comparisons and products with exact zeros)
  float ih = __builtin_fminf(i.ymax, j.ymax) - __builtin_fmaxf(i.ymin, j.ymin);
  float iw = __builtin_fminf(i.xmax, j.xmax) - __builtin_fmaxf(i.xmin, j.xmin);
  ih = __builtin_fmaxf(ih, 0.0f);
  iw = __builtin_fmaxf(iw, 0.0f);
  const float inter = ih * iw;
  float uni = i.area + j.area;
  uni = uni - inter;
  const double pr = t.mid * (double)uni;
  const double di = (double)inter;
  bool res = t.tie_up ? (di >= pr) : (di > pr);
  res = res && ((fi & fj & kFlagArea) != 0);
  // positive normal or subnormal uni (class mask 0x180), regular threshold, finite coordinates
  const bool regular = __builtin_amdgcn_classf(uni, 0x180) && !t.slow_only && (((fi | fj) & kFlagSpecial) == 0);
  if (__builtin_expect(!regular, 0)) res = nms_iou_gt(i, j, t.thr);
  return res;
}

// LDS layout for chunks of CH candidates (CH = 64 * G).
template <int CH>
struct NmsLds {
  unsigned long long *mask;  // [CH * G] intra-chunk suppression words, mask[row * G + column group]
  unsigned long long *dead;  // [4]
  float *c_ymin, *c_xmin, *c_ymax, *c_xmax, *c_area;  // [CH] chunk boxes (corner-normalised)
  int *c_idx;                                         // [CH] element index of each chunk candidate
  int *c_flag;                                        // [CH]
  int *s_nk;                                          // [4]
  float *k_ymin, *k_xmin, *k_ymax, *k_xmax, *k_area;  // [Rc] kept boxes
  int *k_idx;                                         // [Rc]
  int *k_flag;                                        // [Rc]
};

template <int CH>
__host__ __device__ static inline size_t nms_lds_bytes(int Rc) {
  return (size_t)CH * (CH / 64) * 8 + 4 * 8 + (size_t)CH * 7 * 4 + 4 * 4 + (size_t)Rc * 7 * 4;
}

template <int CH>
__device__ __forceinline__ NmsLds<CH> carve_nms_lds(unsigned char *smem, int Rc) {
  NmsLds<CH> L;
  L.mask = reinterpret_cast<unsigned long long *>(smem);
  L.dead = L.mask + CH * (CH / 64);
  L.c_ymin = reinterpret_cast<float *>(L.dead + 4);
  L.c_xmin = L.c_ymin + CH;
  L.c_ymax = L.c_xmin + CH;
  L.c_xmax = L.c_ymax + CH;
  L.c_area = L.c_xmax + CH;
  L.c_idx = reinterpret_cast<int *>(L.c_area + CH);
  L.c_flag = L.c_idx + CH;
  L.s_nk = L.c_flag + CH;
  L.k_ymin = reinterpret_cast<float *>(L.s_nk + 4);
  L.k_xmin = L.k_ymin + Rc;
  L.k_ymax = L.k_xmin + Rc;
  L.k_xmax = L.k_ymax + Rc;
  L.k_area = L.k_xmax + Rc;
  L.k_idx = reinterpret_cast<int *>(L.k_area + Rc);
  L.k_flag = L.k_idx + Rc;
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

// One chunk of nc (<= CH) candidates, in priority order.  cand(j, box, ei, bad) fills candidate j's box
// (original corner order), element index and "fails the score threshold" flag (called by threads j < nc).
// nk = boxes kept so far; returns the new count (<= R).  Must be called by all NT threads of the block.
template <int NT, int CH, class CandF>
__device__ __forceinline__ int nms_chunk(const NmsLds<CH> &L, int nc, int nk, int R, const IouThr &thr, CandF cand) {
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

  // ---- load the chunk --------------------------------------------------------------------------
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
    L.c_ymin[tid] = nb.ymin;
    L.c_xmin[tid] = nb.xmin;
    L.c_ymax[tid] = nb.ymax;
    L.c_xmax[tid] = nb.xmax;
    L.c_area[tid] = nb.area;
    L.c_idx[tid] = ei;
    L.c_flag[tid] = fl;
    const unsigned long long badw = __ballot(bad);  // candidates that fail the score threshold start dead
    if (lane == 0) L.dead[w] = badw;
  }
  __syncthreads();
  FRCNN_STAMP(10);

  // ---- a. chunk vs kept list ------------------------------------------------------------------
  if (nk > 0) {
    const int j = tid & (CH - 1);
    const int slice = tid / CH;
    if ((j & ~63) < nc) {  // wave-uniform: this wave's 64 candidates exist
      NBox me;
      me.ymin = L.c_ymin[j];
      me.xmin = L.c_xmin[j];
      me.ymax = L.c_ymax[j];
      me.xmax = L.c_xmax[j];
      me.area = L.c_area[j];
      const int mf = L.c_flag[j];
      bool d = false;
      for (int k0 = slice; k0 < nk; k0 += 4 * NSL) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * NSL;
          if (k < nk) {  // wave-uniform
            NBox kb;
            kb.ymin = L.k_ymin[k];
            kb.xmin = L.k_xmin[k];
            kb.ymax = L.k_ymax[k];
            kb.xmax = L.k_xmax[k];
            kb.area = L.k_area[k];
            d = d | nms_iou_gt_fast(me, mf, kb, L.k_flag[k], thr);
          }
        }
        if (__all(d || (j >= nc))) break;
      }
      const unsigned long long word = __ballot(d && (j < nc));
      if (lane == 0 && word != 0ull) atomicOr(&L.dead[j >> 6], word);
    }
  }
  FRCNN_STAMP(11);

  // ---- b. intra-chunk IoU tiles (upper triangle), ROWS rows per work item ------------------------
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
    NBox me;
    me.ymin = L.c_ymin[jcol];
    me.xmin = L.c_xmin[jcol];
    me.ymax = L.c_ymax[jcol];
    me.xmax = L.c_xmax[jcol];
    me.area = L.c_area[jcol];
    const int mf = L.c_flag[jcol];
#pragma unroll 8
    for (int ii = 0; ii < ROWS; ++ii) {
      const int irow = rg * 64 + part * ROWS + ii;
      NBox rb;
      rb.ymin = L.c_ymin[irow];
      rb.xmin = L.c_xmin[irow];
      rb.ymax = L.c_ymax[irow];
      rb.xmax = L.c_xmax[irow];
      rb.area = L.c_area[irow];
      const bool bit = (irow < jcol) & (jcol < nc) & nms_iou_gt_fast(rb, L.c_flag[irow], me, mf, thr);
      const unsigned long long word = __ballot(bit);
      if (lane == 0) L.mask[irow * G + cg] = word;
    }
  }
  __syncthreads();
  FRCNN_STAMP(12);

  // ---- c. greedy resolve on wave 0 (scalar registers) ---------------------------------------------
  // Only a row's OWN column-group word feeds the serial chain (it decides who is alive later in the same
  // group); the words for later groups are OR-ed into L.dead[] once per group by the kept lanes.
  if (w == 0) {
    int cnt = nk;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int left = nc - g * 64;
      if (left > 0 && cnt < R) {  // wave-uniform
        unsigned long long rem = uniform64(L.dead[g]);
        unsigned long long row[G];
#pragma unroll
        for (int w2 = 0; w2 < G; ++w2)
          row[w2] = (w2 >= g && w2 * 64 < nc) ? L.mask[(g * 64 + lane) * G + w2] : 0ull;
        const unsigned long long own = row[g];
        // rows that suppress nothing inside the group are accepted in bulk: only the others need a serial step
        const unsigned long long nz = __ballot(own != 0ull);
        const unsigned long long valid = (left >= 64) ? ~0ull : ((1ull << left) - 1ull);
        unsigned long long alive = uniform64(~rem & valid);
        unsigned long long keptbits = 0ull;
        while (alive != 0ull && cnt < R) {
          const unsigned long long nza = alive & nz;
          const int f = (nza != 0ull) ? (int)__builtin_ctzll(nza) : 64;
          unsigned long long batch = (f >= 64) ? alive : (alive & ((1ull << f) - 1ull));
          int nb = __popcll(batch);
          if (nb > 0) {
            if (cnt + nb > R) {  // keep only the first R - cnt of them
              unsigned long long t = 0ull, bb = batch;
              for (int k = cnt; k < R; ++k) {
                const unsigned long long low = bb & (0ull - bb);
                t |= low;
                bb ^= low;
              }
              batch = t;
              nb = R - cnt;
            }
            keptbits |= batch;
            cnt += nb;
            alive &= ~batch;
            if (cnt >= R) break;
          }
          if (f < 64) {
            keptbits |= (1ull << f);
            ++cnt;
            rem |= readlane64(own, f);
            alive &= ~(rem | keptbits);
          }
        }
        if ((keptbits >> lane) & 1ull) {
#pragma unroll
          for (int w2 = 0; w2 < G; ++w2)
            if (w2 > g && row[w2] != 0ull) atomicOr(&L.dead[w2], row[w2]);
          const int pos = nk + __popcll(keptbits & lanemask_lt());
          const int j = g * 64 + lane;
          L.k_ymin[pos] = L.c_ymin[j];
          L.k_xmin[pos] = L.c_xmin[j];
          L.k_ymax[pos] = L.c_ymax[j];
          L.k_xmax[pos] = L.c_xmax[j];
          L.k_area[pos] = L.c_area[j];
          L.k_idx[pos] = L.c_idx[j];
          L.k_flag[pos] = L.c_flag[j];
        }
        nk += __popcll(keptbits);
      }
    }
    if (lane == 0) L.s_nk[0] = nk;
  }
  __syncthreads();
  FRCNN_STAMP(13);
  return L.s_nk[0];
}

}  // namespace frcnn
