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
__device__ __forceinline__ float iou_pair(float by1, float bx1, float by2, float bx2, float barea,
                                          float gy1, float gx1, float gy2, float gx2, float garea) {
  float x_top = (bx1 > gx1) ? bx1 : gx1;
  float y_top = (by1 > gy1) ? by1 : gy1;
  float x_bot = (bx2 < gx2) ? bx2 : gx2;
  float y_bot = (by2 < gy2) ? by2 : gy2;
  float dx = x_bot - x_top;
  float dy = y_bot - y_top;
  dx = (dx > 0.0f) ? dx : 0.0f;
  dy = (dy > 0.0f) ? dy : 0.0f;
  float inter = dx * dy;
  float uni = barea + garea;
  uni = uni - inter;
  return inter / uni;
}

// LDS layout (floats): gy1[Mp] gx1[Mp] gy2[Mp] gx2[Mp] garea[Mp], Mp = M rounded up to 4.
__device__ __forceinline__ void stage_gt(float *lds, const float *__restrict__ gt, int M, int Mp) {
  for (int m = threadIdx.x; m < Mp; m += blockDim.x) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) g = *reinterpret_cast<const float4 *>(gt + (long long)m * 4);
    float gh = g.z - g.x;
    float gw = g.w - g.y;
    lds[m] = g.x;
    lds[Mp + m] = g.y;
    lds[2 * Mp + m] = g.z;
    lds[3 * Mp + m] = g.w;
    lds[4 * Mp + m] = gh * gw;
  }
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
  stage_gt(lds, gt + (long long)b * M * 4, M, Mp);
  __syncthreads();
  const float *bb = bboxes + (long long)b * bboxes_stride_b;
  float *ob = out + (long long)b * N * M;
  if (VEC4) {
    const int M4 = M >> 2;
    const long long total = (long long)N * M4;
    const long long e0 = (long long)blockIdx.x * per_block;
    const long long e1 = (e0 + per_block < total) ? e0 + per_block : total;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
      const int n = (int)(e / M4);
      const int m0 = ((int)(e - (long long)n * M4)) << 2;
      const float4 bx = *reinterpret_cast<const float4 *>(bb + (long long)n * 4);
      const float bh = bx.z - bx.x;
      const float bw = bx.w - bx.y;
      const float barea = bh * bw;
      const f32x4 gy1 = *reinterpret_cast<const f32x4 *>(lds + m0);
      const f32x4 gx1 = *reinterpret_cast<const f32x4 *>(lds + Mp + m0);
      const f32x4 gy2 = *reinterpret_cast<const f32x4 *>(lds + 2 * Mp + m0);
      const f32x4 gx2 = *reinterpret_cast<const f32x4 *>(lds + 3 * Mp + m0);
      const f32x4 ga = *reinterpret_cast<const f32x4 *>(lds + 4 * Mp + m0);
      f32x4 v;
      v.x = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, gy1.x, gx1.x, gy2.x, gx2.x, ga.x);
      v.y = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, gy1.y, gx1.y, gy2.y, gx2.y, ga.y);
      v.z = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, gy1.z, gx1.z, gy2.z, gx2.z, ga.z);
      v.w = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, gy1.w, gx1.w, gy2.w, gx2.w, ga.w);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(ob) + e);
    }
  } else {
    const long long total = (long long)N * M;
    const long long e0 = (long long)blockIdx.x * per_block;
    const long long e1 = (e0 + per_block < total) ? e0 + per_block : total;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) {
      const int n = (int)(e / M);
      const int m = (int)(e - (long long)n * M);
      const float4 bx = *reinterpret_cast<const float4 *>(bb + (long long)n * 4);
      const float bh = bx.z - bx.x;
      const float bw = bx.w - bx.y;
      const float barea = bh * bw;
      ob[e] = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, lds[m], lds[Mp + m], lds[2 * Mp + m],
                       lds[3 * Mp + m], lds[4 * Mp + m]);
    }
  }
}

// ---- fused reductions -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = (o > v) ? o : v;
  }
  return v;
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
  stage_gt(lds, gt + (long long)b * M * 4, M, Mp);
  if (col_keys)
    for (int m = threadIdx.x; m < M; m += 256) colk[m] = 0ull;
  __syncthreads();

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
  for (int m = 0; m < M; ++m) {
    const float v = iou_pair(bx.x, bx.y, bx.z, bx.w, barea, lds[m], lds[Mp + m], lds[2 * Mp + m],
                             lds[3 * Mp + m], lds[4 * Mp + m]);
    if (v > best) {
      best = v;
      am = m;
    }
    if (v > mx) mx = v;
    if (col_keys) {
      // key 0 = "no candidate" (dead lanes, NaN, and anything not above lowest()).
      uint32_t k = (live && v > -3.402823466e+38f) ? f_order_key(v) : 0u;
      const uint32_t wmax = wave_max_u32(k);
      const unsigned long long hit = __ballot(k == wmax);
      if (wmax != 0u && lane_id() == (int)__builtin_ctzll(hit)) {
        const unsigned long long key = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)n);
        atomicMax(&colk[m], key);
      }
    }
  }
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

__global__ __launch_bounds__(256) void col_keys_to_index_kernel(const unsigned long long *__restrict__ keys,
                                                                int *__restrict__ argmax_col, int total) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const unsigned long long k = keys[i];
  // key 0: no box had an IoU above lowest() -> Eigen's ArgMax keeps its initial index 0.
  argmax_col[i] = (k == 0ull) ? 0 : (int)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
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
  // ~16 elements per lane per block keeps the grid well above 256 CUs x 8 at the reference sizes.
  const int per_block = 256 * 16;
  const long long tiles = (total + per_block - 1) / per_block;
  FRCNN_REQUIRE(tiles < (1ll << 31), "iou_map: too large");
  dim3 grid((unsigned)tiles, (unsigned)B), block(256);
  const long long stride = bboxes_batched ? (long long)N * 4 : 0;
  if (vec4)
    hipLaunchKernelGGL(iou_map_kernel<true>, grid, block, lds_bytes, (hipStream_t)stream, bboxes, stride,
                       gt_boxes, N, M, out, per_block);
  else
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
    hipLaunchKernelGGL(iou_reduce_kernel, grid, block, lds_bytes, s, bboxes, stride, gt_boxes, N, M, max_iou,
                       argmax_row, keys);
    int rc = check_launch("iou_map_reduce");
    if (rc) return rc;
  }
  if (keys) {
    const int total = B * M;
    hipLaunchKernelGGL(col_keys_to_index_kernel, dim3((total + 255) / 256), dim3(256), 0, s, keys, argmax_col, total);
    return check_launch("iou_map_reduce finalize");
  }
  return FRCNN_OK;
}

}  // extern "C"
