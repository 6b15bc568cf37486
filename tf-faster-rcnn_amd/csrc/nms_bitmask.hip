// nms_bitmask.hip -- RoIBBox for LARGE post_nms_topn (training: train_nms_topn = 1500..2000): the pair tests of the
// greedy NMS spread over the whole chip.
//
// Replaces models/faster_rcnn.py:153-164 like proposal.hip; same results, different schedule.  The fused
// one-workgroup-per-image kernel (proposal.hip) tests every 128-candidate chunk against the kept list on ONE CU: at
// R = 2000 that is ~3 M pair tests per image, 1.6 ms per 16 images (0.38 ms with this schedule).  Here:
//   1. sort_select orders the top-K candidates (score desc, index asc) -- sort_select.hip;
//   2. `bm_prepare_kernel` decodes them once into packed boxes (corner-normalised box, area, flags);
//   3. per PHASE (a band of kBandBlocks 64-candidate column blocks, i.e. 2048 candidates):
//      `bm_tiles_kernel` (whole grid): 64x64 tiles of suppression words; lane = column box, and the lane's word
//         collects over the tile's 64 rows the bits "IoU(row, col) > thr and row before col" (COLUMN words: no ballot, no
//         cross-lane traffic in the tile; the scan only ever asks "is this column hit by one of these kept rows").  Rows are (a) the boxes KEPT in earlier bands, compacted (a box that
//         was suppressed can suppress nobody, so only ~R of the cb0*64 earlier candidates are rows), and (b) the band's own
//         candidates up to the column's block (undecided yet);
//      `bm_scan_kernel` (one workgroup per image): ORs the (a) words into the band's `removed` words, then walks the
//         band's blocks in order: wave 0 resolves the greedy order inside a block in parallel rounds from the block's
//         diagonal tile, then every wave marks in one later block of the band the columns hit by the block's kept rows
//         (one AND and one ballot).
//      An image that has its R boxes (or runs out of candidates) writes its outputs and marks itself done; later phases
//      return at once for it, so the work is proportional to the candidates actually consumed, in bands of 2048.
//      Launch order: a band's (b) tiles do not depend on the scans before it, so they are computed one band AHEAD, by
//      extra workgroups of the previous band's scan launch (`bm_scan_tiles_kernel`: the scan is one latency-bound
//      workgroup per image, the rest of the chip is free); what runs between two scans is only the (a) tiles.
// All state between kernels lives in the caller's workspace; nothing synchronises with the host.
#include <stdlib.h>

#include "nms_core.hpp"
#include "sort_select.hpp"

namespace frcnn {

static const int kBandBlocks = 32;  // column blocks (of 64 candidates) per phase

struct BitmaskParams {
  const float *deltas;   // (B,N,4)
  const float *scores;   // (B,N)
  const float *anchors;  // (N,4)
  float v0, v1, v2, v3;
  int N, K, R_nms, R_out;
  IouThr iou;
  float score_thr;
  int nb;                // 64-candidate blocks = ceil(K / 64)
  // workspace (per image strides in elements)
  const int *sorted_idx;        // (B,K)
  float4 *pbox;                 // (B, nb*64)
  float2 *paf;                  // (B, nb*64)   (area, flags as int bits)
  int Rx;                       // R_nms rounded up to 64: row capacity of xmask
  // column words: element [cb][g*64 + lane] belongs to column box cb*64 + lane, bit r = "row r of row group g suppresses it"
  unsigned long long *xmask;    // (B, nb, Rx)              row groups = the boxes kept in earlier bands, 64 at a time
  unsigned long long *mask;     // (B, nb, kBandBlocks*64)  row groups = the blocks of cb's own band (group = rb - band start)
  unsigned long long *removed;  // (B, nb)
  int *state;                   // (B, 4): nk, done
  int *kidx;                    // (B, R_nms) sorted positions of the kept boxes
  float *out_boxes;             // (B,R_out,4)
  float *out_scores;            // (B,R_out) or NULL
  int *out_valid;               // (B) or NULL
};

// grid (nb/4 rounded up, B), block 256: wave w <-> block 4*blockIdx.x + w.
__global__ __launch_bounds__(256) void bm_prepare_kernel(BitmaskParams p) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int blk = blockIdx.x * 4 + w;
  if (blk >= p.nb) return;
  const int Kp = p.nb * 64;
  const int j = blk * 64 + lane;
  NBox nb;
  nb.ymin = nb.xmin = nb.ymax = nb.xmax = nb.area = 0.0f;
  int fl = 0;
  bool bad = true;  // positions beyond K never exist
  if (j < p.K) {
    const int ei = p.sorted_idx[(long long)b * p.K + j];
    const float4 a = reinterpret_cast<const float4 *>(p.anchors)[ei];
    const float4 d = reinterpret_cast<const float4 *>(p.deltas)[(long long)b * p.N + ei];
    const float4 box = decode_box(a, scale_deltas(d, p.v0, p.v1, p.v2, p.v3));
    nb = normalise_box(box);
    fl = box_flags(box, nb);
    bad = !(p.scores[(long long)b * p.N + ei] > p.score_thr);
  }
  p.pbox[(long long)b * Kp + j] = make_float4(nb.ymin, nb.xmin, nb.ymax, nb.xmax);
  p.paf[(long long)b * Kp + j] = make_float2(nb.area, __int_as_float(fl));
  const unsigned long long badw = ballot64(bad);
  if (lane == 0) {
    p.removed[(long long)b * p.nb + blk] = badw;
    if (blk == 0) {
      p.state[b * 4 + 0] = 0;
      p.state[b * 4 + 1] = 0;
    }
  }
}

// This lane's column box against the 64 rows of a tile: bit r of the result = "row r suppresses the column".  Fully
// unrolled so that the bit positions are immediates (a select and an OR per test, on one half of the word).
template <bool DIAG>
__device__ __forceinline__ unsigned long long tile_column_word(const float4 *rows_box, const float2 *rows_af, const PBox &me,
                                                               const IouThr &thr, int lane) {
  unsigned lo = 0u, hi = 0u;
#pragma unroll
  for (int r0 = 0; r0 < 64; r0 += 4) {
    PBox rbx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rbx[u] = load_pbox(rows_box, rows_af, r0 + u);
    bool irr = false;
    bool bit[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt_fast(rbx[u], me, thr, irr);
    if (__builtin_expect(ballot64(irr || thr.slow_only) != 0ull, 0)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bit[u] = nms_iou_gt(to_nbox(rbx[u]), to_nbox(me), thr.thr);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u;
      const bool hit = DIAG ? (bit[u] && (r < lane)) : bit[u];
      if (r < 32)
        lo |= hit ? (1u << r) : 0u;
      else
        hi |= hit ? (1u << (r - 32)) : 0u;
    }
  }
  return ((unsigned long long)hi << 32) | lo;
}

// One 64x64 tile by one wave: column block cb of image b against row group g -- for g < gx the g-th 64 of the boxes KEPT in
// earlier bands (cross tile -> xmask), otherwise block rb = cb0 + (g - gx) of the column's own band (-> mask; the diagonal
// tile rb == cb only counts rows before the column).  s_box / s_af: 64 entries of this wave's own LDS.
__device__ __forceinline__ void bm_tile(const BitmaskParams &p, int b, int cb, int g, int cb0, int gx, int lane, float4 *s_box,
                                        float2 *s_af) {
  const int Kp = p.nb * 64;
  const float4 *pbox = p.pbox + (long long)b * Kp;
  const float2 *paf = p.paf + (long long)b * Kp;
  const bool cross = g < gx;
  int rb = 0, row_pos;  // row_pos: this lane's row as a position in the sorted order
  if (cross) {
    const int nk_prev = p.state[b * 4 + 0];
    if (g * 64 >= nk_prev) return;  // wave-uniform
    const int kpos = g * 64 + lane;
    row_pos = (kpos < nk_prev) ? p.kidx[(long long)b * p.R_nms + kpos] : -1;
  } else {
    rb = cb0 + (g - gx);
    if (rb > cb) return;  // wave-uniform
    row_pos = rb * 64 + lane;
  }
  float4 rbox = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 raf = make_float2(0.f, 0.f);  // flags 0: fails the area guard, overlaps nobody
  if (row_pos >= 0) {
    rbox = pbox[row_pos];
    raf = paf[row_pos];
  }
  s_box[lane] = rbox;
  s_af[lane] = raf;
  const int jcol = cb * 64 + lane;
  const PBox me = load_pbox(pbox, paf, jcol);
  // wave-private LDS rows: LDS operations of a wave complete in order; keep the compiler from hoisting the reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // rows of earlier bands and of earlier blocks all precede the column; inside the diagonal tile only rows before it count
  unsigned long long word;
  if (!cross && rb == cb)
    word = tile_column_word<true>(s_box, s_af, me, p.iou, lane);
  else
    word = tile_column_word<false>(s_box, s_af, me, p.iou, lane);
  if (cross)
    p.xmask[((long long)b * p.nb + cb) * p.Rx + g * 64 + lane] = word;
  else
    p.mask[((long long)b * p.nb + cb) * (kBandBlocks * 64) + (rb - cb0) * 64 + lane] = word;
}

// grid (band column blocks, ceil(row groups / 4), B), block 256: wave w <-> row group g = 4*blockIdx.y + w.
// Groups g < gx are 64 kept boxes of earlier bands (gx = host-side bound on their number / 64); the others are the band's
// own blocks rb = cb0 + g - gx (not launched / skipped with cross_only: those tiles are computed one band ahead, beside
// the previous band's scan).
__global__ __launch_bounds__(256) void bm_tiles_kernel(BitmaskParams p, int cb0, int gx, int cross_only) {
  __shared__ float4 s_box[4][64];
  __shared__ float2 s_af[4][64];
  const int b = blockIdx.z;
  if (p.state[b * 4 + 1]) return;  // image finished in an earlier phase
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cb = cb0 + blockIdx.x;
  const int g = blockIdx.y * 4 + w;
  if (cb >= p.nb) return;
  if (cross_only && g >= gx) return;
  bm_tile(p, b, cb, g, cb0, gx, lane, s_box[w], s_af[w]);
}

// One workgroup of NW waves per image (NW = 16, or 4: a 256-thread workgroup of at most 128 VGPRs becomes resident beside
// the workgroups of a running crop launch, a 1024-thread one only in that launch's tail -- the stage pipeline's train mode
// runs this kernel under the previous batch's crop).  LDS: removed[nb] | kidx[R_nms] | diag[kBandBlocks * 64].
template <int NW>
__device__ __forceinline__ void bm_scan_image(const BitmaskParams &p, int b, int cb0, int last_phase,
                                              unsigned long long *s_words, unsigned long long *s_kept, int *s_nk) {
  if (p.state[b * 4 + 1]) return;
  constexpr int NT = NW * 64;
  constexpr int NF = kBandBlocks / NW;  // later blocks of the band one wave folds into per step
  static_assert(NW * NF == kBandBlocks && kBandBlocks <= 32, "the waves' folds cover every later block of the band");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nb = p.nb, R = p.R_nms;
  int cb1 = cb0 + kBandBlocks;
  if (cb1 > nb) cb1 = nb;
  constexpr int BW = kBandBlocks * 64;  // rows of a band
  unsigned long long *removed = s_words;
  int *kidx = reinterpret_cast<int *>(s_words + nb);
  const unsigned long long *mask = p.mask + (long long)b * nb * BW;
  int nk = p.state[b * 4 + 0];
  for (int i = tid; i < nb; i += NT) removed[i] = p.removed[(long long)b * nb + i];
  for (int i = tid; i < nk; i += NT) kidx[i] = p.kidx[(long long)b * R + i];
  __syncthreads();

  // (i) the boxes kept in earlier bands suppress into this band: one wave per column block; a column is removed if any of
  // its words over the kept rows is non-zero (row slots past nk hold boxes that overlap nobody)
  if (cb0 > 0 && nk > 0) {
    const unsigned long long *xm = p.xmask + (long long)b * nb * p.Rx;
    for (int cb = cb0 + w; cb < cb1; cb += NW) {
      unsigned long long acc = 0ull;
      const unsigned long long *col = xm + (long long)cb * p.Rx;
#pragma unroll 4
      for (int k0 = 0; k0 < nk; k0 += 64) acc |= col[k0 + lane];
      const unsigned long long hitw = ballot64(acc != 0ull);
      if (lane == 0) removed[cb] |= hitw;
    }
    __syncthreads();
  }

  // (ii) the band's blocks in order.  Nothing a block needs from memory depends on the blocks before it, so the loads run
  // one block ahead of the serial chain: the band's diagonal tiles sit in LDS, the words to fold after block
  // cb + 1 are requested while block cb is resolved.
  unsigned long long *s_diag = reinterpret_cast<unsigned long long *>(kidx + ((R + 1) & ~1));  // [kBandBlocks * 64]
  for (int i = tid; i < (cb1 - cb0) * 64; i += NT) {  // diagonal tile of block cb0 + i/64: row group i/64 of its own column
    const int blk = i >> 6;
    s_diag[i] = mask[(long long)(cb0 + blk) * BW + blk * 64 + (i & 63)];
  }
  unsigned long long fn[NF];  // wave w folds into blocks cb + 1 + w + j * NW, j < NF
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int c = cb0 + 1 + w + j * NW;
    fn[j] = (c < cb1) ? mask[(long long)c * BW + lane] : 0ull;
  }
  __syncthreads();
  for (int cb = cb0; cb < cb1 && nk < R; ++cb) {
    unsigned long long f[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int c = cb + 1 + w + j * NW;
      f[j] = fn[j];
      fn[j] = (c + 1 < cb1) ? mask[(long long)(c + 1) * BW + (cb + 1 - cb0) * 64 + lane] : 0ull;
    }
    if (w == 0) {
      // Greedy order inside the block, in parallel rounds instead of one serial step per suppressing row: lane i knows
      // which earlier rows of the block suppress it (col).  A candidate hit by a KEPT row is removed; a candidate none of
      // whose still-undecided earlier rows suppresses it is kept (whatever those rows turn out to be, they do not touch
      // it).  The lowest undecided candidate always falls in one of the two classes, so every round makes progress; the
      // fixed point is the sequential greedy result, reached in (longest suppression chain) rounds -- a handful.
      const unsigned long long col = s_diag[(cb - cb0) * 64 + lane];
      unsigned long long keptbits = resolve_rounds(col, uniform64(~removed[cb]));
      int cnt = nk + __popcll(keptbits);
      if (cnt > R) {  // the greedy scan stops at R: keep the first R - nk of the block's kept candidates
        unsigned long long t = 0ull, bb = keptbits;
        for (int k = nk; k < R; ++k) {
          const unsigned long long low = bb & (0ull - bb);
          t |= low;
          bb ^= low;
        }
        keptbits = t;
        cnt = R;
      }
      if ((keptbits >> lane) & 1ull) kidx[nk + __popcll(keptbits & lanemask_lt())] = cb * 64 + lane;
      if (lane == 0) {
        s_kept[cb & 1] = keptbits;
        s_nk[cb & 1] = cnt;
      }
    }
    // ONE barrier per block: wave 0 folds this block's kept rows into the NEXT block (its own first fold, cb + 1) as soon as
    // it has them, so it can resolve that block right after the barrier; the other waves fold into the blocks from cb + 2 on
    // while it does -- those words are not read before a later barrier.  (Two waves may touch one word in the same interval:
    // LDS atomics.)
    __syncthreads();
    const unsigned long long kb = s_kept[cb & 1];
    nk = s_nk[cb & 1];
    // the columns of the later blocks that one of this block's kept rows suppresses: one AND and one ballot per block
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int c = cb + 1 + w + j * NW;
      if (c < cb1) {
        const unsigned long long h = ballot64((f[j] & kb) != 0ull);
        if (lane == 0 && h != 0ull) atomicOr(&removed[c], h);
      }
    }
  }
  __syncthreads();  // the last block's folds and kept indices

  // (iii) finished (R boxes, or no candidates left): outputs; otherwise save the state for the next phase
  const bool finished = (nk >= R) || last_phase || (cb1 >= nb);
  if (finished) {
    const int Ro = p.R_out;
    for (int r = tid; r < Ro; r += NT) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      float sc = 0.0f;
      if (r < nk) {
        const int ei = p.sorted_idx[(long long)b * p.K + kidx[r]];
        const float4 a = reinterpret_cast<const float4 *>(p.anchors)[ei];
        const float4 d = reinterpret_cast<const float4 *>(p.deltas)[(long long)b * p.N + ei];
        o = decode_box(a, scale_deltas(d, p.v0, p.v1, p.v2, p.v3));
        o.x = clip01(o.x);
        o.y = clip01(o.y);
        o.z = clip01(o.z);
        o.w = clip01(o.w);
        sc = p.scores[(long long)b * p.N + ei];
      }
      *reinterpret_cast<float4 *>(p.out_boxes + ((long long)b * Ro + r) * 4) = o;
      if (p.out_scores) p.out_scores[(long long)b * Ro + r] = sc;
    }
    if (tid == 0) {
      if (p.out_valid) p.out_valid[b] = nk;
      p.state[b * 4 + 1] = 1;
    }
  } else {
    for (int i = tid; i < nb; i += NT) p.removed[(long long)b * nb + i] = removed[i];
    for (int i = tid; i < nk; i += NT) p.kidx[(long long)b * R + i] = kidx[i];
    if (tid == 0) p.state[b * 4 + 0] = nk;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void bm_scan_kernel(BitmaskParams p, int cb0, int last_phase) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_words[];
  __shared__ unsigned long long s_kept[2];  // double-buffered: wave 0 runs one block ahead of the waves that read them
  __shared__ int s_nk[2];
  bm_scan_image<NW>(p, blockIdx.x, cb0, last_phase, s_words, s_kept, s_nk);
}

// The scan of band [cb0, ...) beside the NEXT band's own tiles, which do not depend on it: workgroups 0 .. B-1 scan, the
// others are NW waves with one (image, column block, row block) tile of band [nb0, nb0 + nbn) each (row block <= column
// block; the other half of the square returns at once).  An image that finishes in this very scan has its next-band tiles
// computed for nothing; images finished earlier are skipped.
template <int NW>
__global__ __launch_bounds__(NW * 64) void bm_scan_tiles_kernel(BitmaskParams p, int cb0, int B, int nb0, int nbn) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_words[];
  __shared__ unsigned long long s_kept[2];
  __shared__ int s_nk[2];
  __shared__ float4 s_box[NW][64];
  __shared__ float2 s_af[NW][64];
  if ((int)blockIdx.x < B) {
    bm_scan_image<NW>(p, blockIdx.x, cb0, 0, s_words, s_kept, s_nk);
    return;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long t = ((long long)blockIdx.x - B) * NW + w;
  const int per_image = nbn * nbn;
  const int b = (int)(t / per_image);
  if (b >= B) return;
  const int rem = (int)(t - (long long)b * per_image);
  const int cbi = rem / nbn, rbi = rem - cbi * nbn;
  if (rbi > cbi) return;
  if (p.state[b * 4 + 1]) return;  // finished in an earlier phase
  bm_tile(p, b, nb0 + cbi, rbi, nb0, 0, lane, s_box[w], s_af[w]);
}

struct BitmaskWorkspace {
  int *sorted_idx;
  int *counts;
  float4 *pbox;
  float2 *paf;
  unsigned long long *xmask, *mask, *removed;
  int *state, *kidx;
  size_t bytes;
};

static BitmaskWorkspace carve_bitmask(void *ws, int B, int K, int R) {
  BitmaskWorkspace w;
  unsigned char *p = reinterpret_cast<unsigned char *>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char *q = p ? p + off : nullptr;
    off += round_up(bytes, 256);
    return q;
  };
  const size_t nb = (size_t)(K + 63) / 64, Kp = nb * 64;
  w.sorted_idx = reinterpret_cast<int *>(take((size_t)B * K * 4));
  w.counts = reinterpret_cast<int *>(take((size_t)B * 4));
  w.pbox = reinterpret_cast<float4 *>(take((size_t)B * Kp * 16));
  w.paf = reinterpret_cast<float2 *>(take((size_t)B * Kp * 8));
  const size_t Rx = (size_t)((R > 0 ? R : 1) + 63) / 64 * 64;
  w.xmask = reinterpret_cast<unsigned long long *>(take((size_t)B * nb * Rx * 8));
  w.mask = reinterpret_cast<unsigned long long *>(take((size_t)B * nb * kBandBlocks * 64 * 8));
  w.removed = reinterpret_cast<unsigned long long *>(take((size_t)B * nb * 8));
  w.state = reinterpret_cast<int *>(take((size_t)B * 16));
  w.kidx = reinterpret_cast<int *>(take((size_t)B * (R > 0 ? R : 1) * 4));
  w.bytes = off + 256;
  return w;
}

// R_nms at or above which frcnn_roi_bbox takes this path (FRCNN_ROI_BBOX_BITMASK_MIN_R overrides; 0 disables).
int roi_bbox_bitmask_min_r() {
  return knob(kKnobBitmaskMinR, 512);  // measured crossover at B=16, K=6000: R=500 149 us either way; R=700 302 (fused) / 153; R=1500 1653 / 368
}

bool roi_bbox_bitmask_applies(int B, int K, int R_nms) {
  const int min_r = roi_bbox_bitmask_min_r();
  const long long nb = (K + 63) / 64;
  return min_r > 0 && R_nms >= min_r && R_nms <= 16384 && B <= 65535 && nb <= 1024 &&
         (size_t)(nb * 8 + (size_t)R_nms * 4 + 8 + kBandBlocks * 64 * 8) <= 150 * 1024;
}

size_t roi_bbox_bitmask_workspace_bytes(int B, int K, int R_nms) { return carve_bitmask(nullptr, B, K, R_nms).bytes; }

int launch_roi_bbox_bitmask(const float *deltas, const float *scores, const float *anchors, const float *variances, int B,
                            int N, int K, int R_out, int R_nms, float iou_thr, float *out_boxes, float *out_scores,
                            int *out_valid, void *workspace, size_t workspace_bytes, hipStream_t stream, bool overlapped) {
  BitmaskWorkspace w = carve_bitmask(workspace, B, K, R_nms);
  if (!workspace || !aligned_to(workspace, 256) || workspace_bytes < w.bytes)
    return set_error(FRCNN_EINVAL, "roi_bbox: workspace must be non-null, 256-byte aligned and >= %zu bytes (got %zu)", w.bytes,
                     workspace_bytes);
  ScoreSource ss;
  ss.scores = scores;
  ss.stride_b = N;
  ss.elem_stride = 1;
  ss.mask_bg = 0;
  ss.L = 0;
  int rc = launch_sort_select(ss, B, 1, N, K, false, 0.0f, w.sorted_idx, nullptr, w.counts, K, stream);
  if (rc) return rc;
  BitmaskParams p;
  p.deltas = deltas;
  p.scores = scores;
  p.anchors = anchors;
  p.v0 = variances[0];
  p.v1 = variances[1];
  p.v2 = variances[2];
  p.v3 = variances[3];
  p.N = N;
  p.K = K;
  p.R_nms = R_nms;
  p.R_out = R_out;
  p.iou = make_iou_thr(iou_thr);
  p.score_thr = -__builtin_huge_valf();
  p.nb = (K + 63) / 64;
  p.sorted_idx = w.sorted_idx;
  p.pbox = w.pbox;
  p.paf = w.paf;
  p.Rx = (R_nms + 63) / 64 * 64;
  p.xmask = w.xmask;
  p.mask = w.mask;
  p.removed = w.removed;
  p.state = w.state;
  p.kidx = w.kidx;
  p.out_boxes = out_boxes;
  p.out_scores = out_scores;
  p.out_valid = out_valid;
  hipLaunchKernelGGL(bm_prepare_kernel, dim3((unsigned)((p.nb + 3) / 4), (unsigned)B), dim3(256), 0, stream, p);
  rc = check_launch("roi_bbox (bitmask prepare)");
  if (rc) return rc;
  const size_t lds = (size_t)p.nb * 8 + (size_t)((R_nms + 1) & ~1) * 4 + (size_t)kBandBlocks * 64 * 8;
  // FRCNN_BITMASK_THREADS = 256 | 1024: the scans' workgroup size (same results).  By default 1024 for a call on its own
  // (config 3's shard of 8 images: 148 against 203 us) and 256 when the call comes from a runtime that overlaps stages
  // (frcnn_roi_bbox_multi): see bm_scan_image.
  const bool small_wg = knob(kKnobBitmaskThreads, overlapped ? 256 : 1024) != 1024;
  auto scan = small_wg ? bm_scan_kernel<4> : bm_scan_kernel<16>;
  auto scan_tiles = small_wg ? bm_scan_tiles_kernel<4> : bm_scan_tiles_kernel<16>;
  const int nw = small_wg ? 4 : 16;
  if (lds > 48 * 1024) {
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(scan), lds, "roi_bbox: hipFuncSetAttribute");
    if (rc) return rc;
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(scan_tiles), lds, "roi_bbox: hipFuncSetAttribute");
    if (rc) return rc;
  }
  // Schedule: the first band's own tiles; then per band [cross tiles against the boxes kept so far] and [scan ‖ the next
  // band's own tiles] -- a band's own tiles do not depend on the scans before it, so they run beside the previous scan
  // (one workgroup per image, latency bound) instead of in front of their own.
  {
    const int nb0 = kBandBlocks < p.nb ? kBandBlocks : p.nb;
    hipLaunchKernelGGL(bm_tiles_kernel, dim3((unsigned)nb0, (unsigned)((nb0 + 3) / 4), (unsigned)B), dim3(256), 0, stream, p, 0, 0, 0);
    rc = check_launch("roi_bbox (bitmask tiles)");
    if (rc) return rc;
  }
  for (int cb0 = 0; cb0 < p.nb; cb0 += kBandBlocks) {
    const int cb1 = (cb0 + kBandBlocks < p.nb) ? cb0 + kBandBlocks : p.nb;
    // at most min(R_nms, cb0 * 64) boxes were kept before this band
    const int kept_bound = (R_nms < cb0 * 64) ? R_nms : cb0 * 64;
    const int gx = (kept_bound + 63) / 64;
    if (gx > 0) {
      hipLaunchKernelGGL(bm_tiles_kernel, dim3((unsigned)(cb1 - cb0), (unsigned)((gx + 3) / 4), (unsigned)B), dim3(256), 0, stream,
                         p, cb0, gx, 1);
      rc = check_launch("roi_bbox (bitmask cross tiles)");
      if (rc) return rc;
    }
    if (cb1 >= p.nb) {
      hipLaunchKernelGGL(scan, dim3((unsigned)B), dim3((unsigned)nw * 64u), lds, stream, p, cb0, 1);
    } else {
      const int nbn = (cb1 + kBandBlocks < p.nb) ? kBandBlocks : p.nb - cb1;  // column blocks of the next band
      const long long tile_wgs = ((long long)B * nbn * nbn + nw - 1) / nw;
      hipLaunchKernelGGL(scan_tiles, dim3((unsigned)(B + tile_wgs)), dim3((unsigned)nw * 64u), lds, stream, p, cb0, B, cb1, nbn);
    }
    rc = check_launch("roi_bbox (bitmask scan)");
    if (rc) return rc;
  }
  return FRCNN_OK;
}

}  // namespace frcnn
