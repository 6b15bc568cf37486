// iou_map.hip -- boxes x ground-truth IoU: the materialised map (drop-in generate_iou_map) and
// the fused map+reductions used by RPN target assignment.
//
// Replaces utils/bbox_utils.py:138-160 and the reductions of utils/train_utils.py:177-179.
//
// Materialised map: HBM-write bound, N*M*4 bytes per image (3.69 MB at N=9216, M=100).  The
// output is written as one flat run of float4 (when M % 4 == 0), lane e -> float4 e of the
// image's (N,M) slab, so stores are full 1 KiB wave lines; gt boxes and their areas sit in LDS
// (structure of arrays, so the four gts of a lane are one ds_read_b128 per coordinate).
//
// Fused reductions: no map is written (75.7 KB of traffic per image instead of 3.69 MB), so the
// kernel is ALU bound (~12 flop + 1 IEEE divide per pair).  One lane owns one box and walks
// the gts from LDS, keeping the row max / first argmax in registers; the per-gt argmax over boxes
// is a 64-lane butterfly max on an order-preserving key, merged per block with an LDS 64-bit
// max and per grid with one 64-bit global atomic max per (block, gt).  Keys are
// (order_key(iou) << 32) | ~box_index, so the maximum is "largest IoU, then smallest index" =
// first-maximum-wins, independent of arrival order (deterministic).
#include "common.hpp"

namespace frcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// utils/bbox_utils.py:154-160 for one pair; barea/garea from :152-153.
// NONAN: the caller has established that no coordinate of either box is NaN.  Then the reference's maximum / minimum
// (written as the ternaries below, which is how they treat NaN) equal the hardware v_max_f32 / v_min_f32 up to the sign of
// a zero, and a zero of either sign gives the same dx / dy after the clamp -- one instruction instead of two per min/max in
// a kernel that sits on the VALU roof.
template <bool NONAN>
__device__ __forceinline__ void iou_terms(float by1, float bx1, float by2, float bx2, float barea, float gy1, float gx1,
                                          float gy2, float gx2, float garea, float &inter, float &uni) {
  float x_top, y_top, x_bot, y_bot;
  if (NONAN) {
    x_top = hw_max(bx1, gx1);
    y_top = hw_max(by1, gy1);
    x_bot = hw_min(bx2, gx2);
    y_bot = hw_min(by2, gy2);
  } else {
    x_top = (bx1 > gx1) ? bx1 : gx1;
    y_top = (by1 > gy1) ? by1 : gy1;
    x_bot = (bx2 < gx2) ? bx2 : gx2;
    y_bot = (by2 < gy2) ? by2 : gy2;
  }
  float dx = x_bot - x_top;
  float dy = y_bot - y_top;
  dx = (dx > 0.0f) ? dx : 0.0f;
  dy = (dy > 0.0f) ? dy : 0.0f;
  inter = dx * dy;
  uni = barea + garea;
  uni = uni - inter;
}

template <bool NONAN>
__device__ __forceinline__ float iou_pair(float by1, float bx1, float by2, float bx2, float barea,
                                          float gy1, float gx1, float gy2, float gx2, float garea) {
  float inter, uni;
  iou_terms<NONAN>(by1, bx1, by2, bx2, barea, gy1, gx1, gy2, gx2, garea, inter, uni);
  return inter / uni;
}

// LDS layout (floats): gy1[Mp] gx1[Mp] gy2[Mp] gx2[Mp] garea[Mp], Mp = M rounded up to 4.
// Returns (to every thread, after a barrier) whether any gt coordinate of the image is NaN: the area is NaN whenever a
// coordinate is (it may also be NaN for infinite boxes, which merely selects the slower exact path).
__device__ __forceinline__ bool stage_gt(float *lds, const float *__restrict__ gt, int M, int Mp) {
  int nan = 0;
  for (int m = threadIdx.x; m < Mp; m += blockDim.x) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) g = *reinterpret_cast<const float4 *>(gt + (long long)m * 4);
    float gh = g.z - g.x;
    float gw = g.w - g.y;
    const float ga = gh * gw;
    lds[m] = g.x;
    lds[Mp + m] = g.y;
    lds[2 * Mp + m] = g.z;
    lds[3 * Mp + m] = g.w;
    lds[4 * Mp + m] = ga;
    nan |= (ga != ga);
  }
  return __syncthreads_or(nan) != 0;
}

// grid: (tiles, B).  Each block writes TILE_F4 consecutive float4 of image b's (N, M) slab.
template <bool VEC4>
__global__ __launch_bounds__(256) void iou_map_kernel(const float *__restrict__ bboxes,
                                                      long long bboxes_stride_b,  // floats
                                                      const float *__restrict__ gt, int N, int M,
                                                      float *__restrict__ out, int per_block) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int Mp = (M + 3) & ~3;
  const bool gt_nan = stage_gt(lds, gt + (long long)b * M * 4, M, Mp);  // includes the barrier
  const float *bb = bboxes + (long long)b * bboxes_stride_b;
  float *ob = out + (long long)b * N * M;
  // Index arithmetic is 32-bit and incremental (one division per lane per block, none per element): with the
  // correctly rounded divide of every pair the kernel sits close to the VALU roof, so every instruction counts.
  const unsigned W = VEC4 ? (unsigned)(M >> 2) : (unsigned)M;  // elements (float4 or float) per box row
  const unsigned total = (unsigned)N * W;                      // < 2^31 (host-checked)
  const unsigned e0 = blockIdx.x * (unsigned)per_block;
  const unsigned e1 = (e0 + (unsigned)per_block < total) ? e0 + (unsigned)per_block : total;
  unsigned e = e0 + threadIdx.x;
  unsigned n = e / W;
  unsigned m = e - n * W;
  const unsigned dn = 256u / W, dm = 256u - dn * W;
  if (VEC4) {
    f32x4 *__restrict__ o4 = reinterpret_cast<f32x4 *>(ob);
    for (; e < e1; e += 256) {
      const unsigned m0 = m << 2;
      const float4 bx = *reinterpret_cast<const float4 *>(bb + (size_t)n * 4);
      const float bh = bx.z - bx.x;
      const float bw = bx.w - bx.y;
      const float barea = bh * bw;
      const f32x4 gy1 = *reinterpret_cast<const f32x4 *>(lds + m0);
      const f32x4 gx1 = *reinterpret_cast<const f32x4 *>(lds + Mp + m0);
      const f32x4 gy2 = *reinterpret_cast<const f32x4 *>(lds + 2 * Mp + m0);
      const f32x4 gx2 = *reinterpret_cast<const f32x4 *>(lds + 3 * Mp + m0);
      const f32x4 ga = *reinterpret_cast<const f32x4 *>(lds + 4 * Mp + m0);
      f32x4 v;
      if (!gt_nan && ballot64(barea != barea) == 0ull) {  // wave-uniform: no NaN coordinate anywhere in this wave's pairs
        v.x = iou_pair<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.x, gx1.x, gy2.x, gx2.x, ga.x);
        v.y = iou_pair<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.y, gx1.y, gy2.y, gx2.y, ga.y);
        v.z = iou_pair<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.z, gx1.z, gy2.z, gx2.z, ga.z);
        v.w = iou_pair<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.w, gx1.w, gy2.w, gx2.w, ga.w);
      } else {
        v.x = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.x, gx1.x, gy2.x, gx2.x, ga.x);
        v.y = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.y, gx1.y, gy2.y, gx2.y, ga.y);
        v.z = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.z, gx1.z, gy2.z, gx2.z, ga.z);
        v.w = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.w, gx1.w, gy2.w, gx2.w, ga.w);
      }
      __builtin_nontemporal_store(v, o4 + e);
      n += dn;
      m += dm;
      if (m >= W) {
        m -= W;
        ++n;
      }
    }
  } else {
    for (; e < e1; e += 256) {
      const float4 bx = *reinterpret_cast<const float4 *>(bb + (size_t)n * 4);
      const float bh = bx.z - bx.x;
      const float bw = bx.w - bx.y;
      const float barea = bh * bw;
      ob[e] = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, lds[m], lds[Mp + m], lds[2 * Mp + m], lds[3 * Mp + m],
                              lds[4 * Mp + m]);
      n += dn;
      m += dm;
      if (m >= W) {
        m -= W;
        ++n;
      }
    }
  }
}

// Materialised map, M % 4 == 0.  grid: (ceil(N/64), B), block 256.  A wave owns 16 consecutive boxes and walks their
// rows four float4 (16 gts) at a time: lane l <-> (box l/4, float4 l%4 of the group), so four lanes write 64 contiguous
// bytes and the box (load, area) is set up once per lane instead of once per element.
//
// The kernel is VALU bound (a correctly rounded divide is ~14 issue slots of the ~25 per pair), and most pairs of a real
// batch do not intersect at all: zero-padded gt rows, and gts far from the 16 spatially adjacent anchors of a wave.  For
// those the quotient is exactly +0 (inter == +0, union > 0), so a group in which NO lane has an intersection skips the
// four divides (wave-uniform branch).
__global__ __launch_bounds__(256) void iou_map_rows_kernel(const float *__restrict__ bboxes, long long bboxes_stride_b,
                                                           const float *__restrict__ gt, int N, int M,
                                                           float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int Mp = M;  // M % 4 == 0
  const bool gt_nan = stage_gt(lds, gt + (long long)b * M * 4, M, Mp);  // includes the barrier
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + wave * 16 + (lane >> 2);
  const int qsub = lane & 3;
  const bool live = n < N;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) bx = *reinterpret_cast<const float4 *>(bboxes + (long long)b * bboxes_stride_b + (long long)n * 4);
  const float bh = bx.z - bx.x;
  const float bw = bx.w - bx.y;
  const float barea = bh * bw;
  const int W = M >> 2;
  f32x4 *__restrict__ orow = reinterpret_cast<f32x4 *>(out + ((long long)b * N + (live ? n : 0)) * M);
  // no NaN coordinate among this wave's boxes or the image's gts (a NaN coordinate makes the area NaN)
  const bool nonan = !gt_nan && ballot64(barea != barea) == 0ull;
  for (int q0 = 0; q0 < W; q0 += 4) {
    const int q = q0 + qsub;
    const bool ok = live && q < W;
    const int m0 = (q < W ? q : W - 1) << 2;
    const f32x4 gy1 = *reinterpret_cast<const f32x4 *>(lds + m0);
    const f32x4 gx1 = *reinterpret_cast<const f32x4 *>(lds + Mp + m0);
    const f32x4 gy2 = *reinterpret_cast<const f32x4 *>(lds + 2 * Mp + m0);
    const f32x4 gx2 = *reinterpret_cast<const f32x4 *>(lds + 3 * Mp + m0);
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(lds + 4 * Mp + m0);
    f32x4 v;
    if (nonan) {
      float in[4], un[4];
      iou_terms<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.x, gx1.x, gy2.x, gx2.x, ga.x, in[0], un[0]);
      iou_terms<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.y, gx1.y, gy2.y, gx2.y, ga.y, in[1], un[1]);
      iou_terms<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.z, gx1.z, gy2.z, gx2.z, ga.z, in[2], un[2]);
      iou_terms<true>(bx.x, bx.y, bx.z, bx.w, barea, gy1.w, gx1.w, gy2.w, gx2.w, ga.w, in[3], un[3]);
      // every inter is >= +0 here (or NaN from inf * 0, which fails the test): their sum is 0 iff all four are
      const float isum = (in[0] + in[1]) + (in[2] + in[3]);
      const float umin = hw_min(hw_min(un[0], un[1]), hw_min(un[2], un[3]));
      const bool trivial = (isum == 0.0f) && (umin > 0.0f);  // +0 / positive = +0 for all four
      if (ballot64(ok && !trivial) == 0ull) {
        v = (f32x4)(0.0f);
      } else {
        v.x = in[0] / un[0];
        v.y = in[1] / un[1];
        v.z = in[2] / un[2];
        v.w = in[3] / un[3];
      }
    } else {
      v.x = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.x, gx1.x, gy2.x, gx2.x, ga.x);
      v.y = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.y, gx1.y, gy2.y, gx2.y, ga.y);
      v.z = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.z, gx1.z, gy2.z, gx2.z, ga.z);
      v.w = iou_pair<false>(bx.x, bx.y, bx.z, bx.w, barea, gy1.w, gx1.w, gy2.w, gx2.w, ga.w);
    }
    // plain (not nontemporal) store: the 64-B pieces of a line arrive in consecutive groups and merge in L2;
    // nontemporal pieces go out as partial lines (3.6x slower), staging rows in LDS for full-line stores costs more
    // issue slots than it saves (54-58 us against 49.5 at config 4)
    if (ok) orow[q] = v;
  }
}

// ---- fused reductions -----------------------------------------------------------------------
// Maximum over the 64 lanes, returned to all of them (wave-uniform).  DPP steps inside the rows of 16 lanes, row
// broadcasts across them, one v_readlane: about ten VALU instructions, against six dependent ds_bpermute round trips for a
// shuffle butterfly (this runs once per (wave, gt) pair that has an overlap; an LDS 64-bit max per lane on the shared
// column key was tried instead and doubled the kernel's time).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1, 3
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One box against every gt of its image (row max / first argmax in registers, column keys merged in LDS).
template <bool NONAN>
__device__ __forceinline__ void reduce_row(const float *lds, int Mp, int m0, int m1, float4 bx, float barea, bool live, int n,
                                           bool want_cols, unsigned long long *colk, float &best, float &mx, int &am) {
  const unsigned long long live_mask = ballot64(live);
  for (int m = m0; m < m1; ++m) {
    float inter, uni;
    iou_terms<NONAN>(bx.x, bx.y, bx.z, bx.w, barea, lds[m], lds[Mp + m], lds[2 * Mp + m], lds[3 * Mp + m], lds[4 * Mp + m],
                     inter, uni);
    // Zero-padded gts and gts far from this wave's 64 neighbouring boxes intersect none of them: the quotient is exactly
    // +0 for every lane (inter == +0, union > 0), so the divide and the cross-lane maximum are skipped (wave-uniform).
    const bool all_zero = NONAN && ballot64(live && !(inter == 0.0f && uni > 0.0f)) == 0ull;
    const float v = all_zero ? 0.0f : inter / uni;
    if (v > best) {
      best = v;
      am = m;
    }
    if (v > mx) mx = v;
    if (want_cols) {
      if (all_zero) {
        // every live lane holds the same key: the lowest live lane (= smallest box index) reports it
        if (live_mask != 0ull && lane_id() == (int)__builtin_ctzll(live_mask)) {
          const unsigned long long key = ((unsigned long long)f_order_key(0.0f) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)n);
          atomicMax(&colk[m], key);
        }
        continue;
      }
      // key 0 = "no candidate" (dead lanes, NaN, and anything not above lowest()).
      uint32_t k = (live && v > -3.402823466e+38f) ? f_order_key(v) : 0u;
      const uint32_t wmax = wave_max_u32(k);
      const unsigned long long hit = ballot64(k == wmax);
      if (wmax != 0u && lane_id() == (int)__builtin_ctzll(hit)) {
        const unsigned long long key = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)n);
        atomicMax(&colk[m], key);
      }
    }
  }
}

// grid: (ceil(N/256), B); block 256: one box per lane.
// LDS: 5*Mp floats of gt + M uint64 column keys.
__global__ __launch_bounds__(256) void iou_reduce_kernel(const float *__restrict__ bboxes,
                                                         long long bboxes_stride_b,
                                                         const float *__restrict__ gt, int N, int M,
                                                         float *__restrict__ max_iou,
                                                         int *__restrict__ argmax_row,
                                                         unsigned long long *__restrict__ col_keys) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int Mp = (M + 3) & ~3;
  unsigned long long *colk = reinterpret_cast<unsigned long long *>(lds + 5 * Mp);
  if (col_keys)
    for (int m = threadIdx.x; m < M; m += 256) colk[m] = 0ull;
  const bool gt_nan = stage_gt(lds, gt + (long long)b * M * 4, M, Mp);  // includes the barrier

  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < N;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) bx = *reinterpret_cast<const float4 *>(bboxes + (long long)b * bboxes_stride_b + (long long)n * 4);
  const float bh = bx.z - bx.x;
  const float bw = bx.w - bx.y;
  const float barea = bh * bw;
  // [TF-recall] Eigen ArgMax / Max reducers: strict '>' starting from lowest().
  float best = -3.402823466e+38f, mx = -3.402823466e+38f;
  int am = 0;
  if (!gt_nan && ballot64(barea != barea) == 0ull)  // wave-uniform: no NaN coordinate in this wave's boxes or the gts
    reduce_row<true>(lds, Mp, 0, M, bx, barea, live, n, col_keys != nullptr, colk, best, mx, am);
  else
    reduce_row<false>(lds, Mp, 0, M, bx, barea, live, n, col_keys != nullptr, colk, best, mx, am);
  if (live) {
    if (max_iou) max_iou[(long long)b * N + n] = mx;
    if (argmax_row) argmax_row[(long long)b * N + n] = am;
  }
  if (col_keys) {
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += 256) {
      const unsigned long long k = colk[m];
      if (k != 0ull) atomicMax(&col_keys[(long long)b * M + m], k);
    }
  }
}

// The same reductions for SMALL batches (a per-GPU shard of 8 images is 288 workgroups of the kernel above: one wave per
// SIMD, each walking all M gts in one dependent chain -- 46 us at config 4's shard, latency bound).  Here a workgroup is
// 64 boxes x 4 gt quarters: wave q walks gts [q*M/4, (q+1)*M/4) for the same 64 boxes, so four times as many waves share
// the chain, and the per-box partial results are merged in LDS in quarter order (strict '>': the earliest gt among equal
// maxima wins, as in one sweep).  grid: (ceil(N/64), B); block 256.
// LDS: 5*Mp floats of gt + M uint64 column keys + 3 * 256 words of partial results.
__global__ __launch_bounds__(256) void iou_reduce_split_kernel(const float *__restrict__ bboxes, long long bboxes_stride_b,
                                                               const float *__restrict__ gt, int N, int M,
                                                               float *__restrict__ max_iou, int *__restrict__ argmax_row,
                                                               unsigned long long *__restrict__ col_keys) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int Mp = (M + 3) & ~3;
  unsigned long long *colk = reinterpret_cast<unsigned long long *>(lds + 5 * Mp);
  float *part = reinterpret_cast<float *>(colk + (M > 0 ? M : 1));  // [3][256]: best, mx, am
  if (col_keys)
    for (int m = threadIdx.x; m < M; m += 256) colk[m] = 0ull;
  const bool gt_nan = stage_gt(lds, gt + (long long)b * M * 4, M, Mp);  // includes the barrier

  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const bool live = n < N;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) bx = *reinterpret_cast<const float4 *>(bboxes + (long long)b * bboxes_stride_b + (long long)n * 4);
  const float bh = bx.z - bx.x;
  const float bw = bx.w - bx.y;
  const float barea = bh * bw;
  const int per = (M + 3) >> 2;
  const int m0 = (q * per < M) ? q * per : M;
  const int m1 = (m0 + per < M) ? m0 + per : M;
  float best = -3.402823466e+38f, mx = -3.402823466e+38f;
  int am = 0;
  if (!gt_nan && ballot64(barea != barea) == 0ull)  // the four waves hold the same boxes: the same decision in all of them
    reduce_row<true>(lds, Mp, m0, m1, bx, barea, live, n, col_keys != nullptr, colk, best, mx, am);
  else
    reduce_row<false>(lds, Mp, m0, m1, bx, barea, live, n, col_keys != nullptr, colk, best, mx, am);
  part[threadIdx.x] = best;
  part[256 + threadIdx.x] = mx;
  part[512 + threadIdx.x] = __int_as_float(am);
  __syncthreads();
  if (q == 0 && live) {
    // quarter 0 is this wave's own (best, mx, am); later quarters only win with a strictly larger value
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float bk = part[k * 64 + lane];
      const float mk = part[256 + k * 64 + lane];
      if (bk > best) {
        best = bk;
        am = __float_as_int(part[512 + k * 64 + lane]);
      }
      if (mk > mx) mx = mk;
    }
    if (max_iou) max_iou[(long long)b * N + n] = mx;
    if (argmax_row) argmax_row[(long long)b * N + n] = am;
  }
  if (col_keys) {
    for (int m = threadIdx.x; m < M; m += 256) {
      const unsigned long long k = colk[m];
      if (k != 0ull) atomicMax(&col_keys[(long long)b * M + m], k);
    }
  }
}

__global__ __launch_bounds__(256) void col_keys_to_index_kernel(const unsigned long long *__restrict__ keys,
                                                                int *__restrict__ argmax_col, int total) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  argmax_col[i] = col_key_index(keys[i]);
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_iou_map(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B, int N,
                  int M, float *out, frcnn_stream_t stream) {
  FRCNN_REQUIRE(bboxes && gt_boxes && out, "iou_map: null pointer");
  FRCNN_REQUIRE(B >= 0 && N >= 0 && M >= 0, "iou_map: negative shape");
  FRCNN_REQUIRE(aligned_to(bboxes, 16) && aligned_to(gt_boxes, 16), "iou_map: box pointers must be 16-byte aligned");
  FRCNN_REQUIRE(M <= 3072, "iou_map: M=%d gt boxes per image exceeds the LDS staging limit 3072", M);
  FRCNN_REQUIRE(B <= 65535, "iou_map: B=%d exceeds 65535", B);
  if (B == 0 || N == 0 || M == 0) return FRCNN_OK;
  const int Mp = (M + 3) & ~3;
  const size_t lds_bytes = (size_t)5 * Mp * sizeof(float);
  const bool vec4 = (M % 4 == 0) && aligned_to(out, 16);
  const long long total = vec4 ? (long long)N * (M / 4) : (long long)N * M;
  FRCNN_REQUIRE(total < (1ll << 31) - 65536, "iou_map: N*M per image too large");
  // Small tiles written in address order stream best (tools/crop_bench.hip: 4-16 KB per workgroup); 8 elements per
  // lane amortise the gt staging.  FRCNN_IOU_TILE overrides the per-lane count (diagnostic).
  const int per_lane = knob(kKnobIouTile, 8) > 0 ? knob(kKnobIouTile, 8) : 8;
  const int per_block = 256 * per_lane;
  const long long tiles = (total + per_block - 1) / per_block;
  dim3 grid((unsigned)tiles, (unsigned)B), block(256);
  const long long stride = bboxes_batched ? (long long)N * 4 : 0;
  if (vec4) {
    if (knob(kKnobIouMapFlat, 0) != 0)  // diagnostic A/B
      hipLaunchKernelGGL(iou_map_kernel<true>, grid, block, lds_bytes, (hipStream_t)stream, bboxes, stride, gt_boxes, N, M,
                         out, per_block);
    else
      hipLaunchKernelGGL(iou_map_rows_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)B), block, lds_bytes,
                         (hipStream_t)stream, bboxes, stride, gt_boxes, N, M, out);
  } else
    hipLaunchKernelGGL(iou_map_kernel<false>, grid, block, lds_bytes, (hipStream_t)stream, bboxes, stride,
                       gt_boxes, N, M, out, per_block);
  return check_launch("iou_map");
}

size_t frcnn_iou_reduce_workspace_bytes(int B, int N, int M) {
  (void)N;
  if (B < 0 || M < 0) return 0;
  return round_up((size_t)B * (size_t)M * sizeof(unsigned long long), 256) + 256;
}

int frcnn_iou_map_reduce(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B,
                         int N, int M, float *max_iou, int32_t *argmax_row, int32_t *argmax_col,
                         void *workspace, size_t workspace_bytes, frcnn_stream_t stream) {
  return frcnn::iou_map_reduce_impl(bboxes, bboxes_batched, gt_boxes, B, N, M, max_iou, argmax_row, argmax_col, workspace,
                                    workspace_bytes, (hipStream_t)stream, true);
}

}  // extern "C"

// finalize = false: the per-gt argmax stays in its 64-bit key form in the workspace (B*M words, see col_key_index) and
// argmax_col is not written -- frcnn_rpn_targets' kernel decodes the keys itself, one launch less.
int frcnn::iou_map_reduce_impl(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B, int N, int M,
                               float *max_iou, int32_t *argmax_row, int32_t *argmax_col, void *workspace,
                               size_t workspace_bytes, hipStream_t stream, bool finalize) {
  FRCNN_REQUIRE(bboxes && gt_boxes, "iou_map_reduce: null pointer");
  FRCNN_REQUIRE(B >= 0 && N >= 0 && M >= 0, "iou_map_reduce: negative shape");
  FRCNN_REQUIRE(aligned_to(bboxes, 16) && aligned_to(gt_boxes, 16), "iou_map_reduce: box pointers must be 16-byte aligned");
  FRCNN_REQUIRE(M <= 2048, "iou_map_reduce: M=%d gt boxes per image exceeds the LDS staging limit 2048", M);
  FRCNN_REQUIRE(B <= 65535, "iou_map_reduce: B=%d exceeds 65535", B);
  if (B == 0) return FRCNN_OK;
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *keys = nullptr;
  if (argmax_col && M > 0) {
    FRCNN_REQUIRE(workspace && aligned_to(workspace, 256), "iou_map_reduce: workspace must be non-null and 256-byte aligned");
    FRCNN_REQUIRE(workspace_bytes >= frcnn_iou_reduce_workspace_bytes(B, N, M), "iou_map_reduce: workspace too small");
    keys = reinterpret_cast<unsigned long long *>(workspace);
    int rc = check_hip(hipMemsetAsync(keys, 0, (size_t)B * M * sizeof(unsigned long long), s), "iou_map_reduce memset");
    if (rc) return rc;
  }
  if (N > 0) {
    const int Mp = (M + 3) & ~3;
    const size_t lds_bytes = (size_t)5 * Mp * sizeof(float) + (size_t)(M > 0 ? M : 1) * sizeof(unsigned long long);
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)B), block(256);
    const long long stride = bboxes_batched ? (long long)N * 4 : 0;
    // 64 boxes x 4 gt quarters per workgroup whenever there are gts to split (measured at config 4: 8 images 46 -> 20 us,
    // 64 images 104 -> 83 us); FRCNN_IOU_REDUCE_SPLIT = 0 | 1 forces one form
    const bool split = knob(kKnobIouReduceSplit, M >= 8 ? 1 : 0) != 0 && M > 0 &&
                       lds_bytes + 3 * 256 * 4 <= 48 * 1024;
    // (Tried in round 3 and removed: the per-gt argmax taken from an LDS tile of the workgroup's 64 x M quotients by a
    // transposed pass, lane = gt, instead of the cross-lane maximum inside the pair loop -- bit-identical, 22.8 us against
    // 20.0 at config 4's shard of 8 images: the cross-lane work only runs for the third of the (wave, gt) steps that have an
    // overlap at all, the transposed pass reads every quotient.)
    if (split)
      hipLaunchKernelGGL(iou_reduce_split_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)B), block, lds_bytes + 3 * 256 * 4, s,
                         bboxes, stride, gt_boxes, N, M, max_iou, argmax_row, keys);
    else
      hipLaunchKernelGGL(iou_reduce_kernel, grid, block, lds_bytes, s, bboxes, stride, gt_boxes, N, M, max_iou,
                         argmax_row, keys);
    int rc = check_launch("iou_map_reduce");
    if (rc) return rc;
  }
  if (keys && finalize) {
    const int total = B * M;
    hipLaunchKernelGGL(col_keys_to_index_kernel, dim3((total + 255) / 256), dim3(256), 0, s, keys, argmax_col, total);
    return check_launch("iou_map_reduce finalize");
  }
  return FRCNN_OK;
}
