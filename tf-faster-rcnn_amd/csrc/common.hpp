// common.hpp -- shared host/device helpers for libfrcnn_hip.so (gfx950 only).
//
// Numerics contract: every fp32 operation on a parity-relevant value is a single IEEE-754
// binary32 operation in the reference's order.  The library is built with -ffp-contract=off;
// nothing here may use fmaf, fast-math intrinsics or reassociation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/frcnn_hip.h"

namespace frcnn {

// ---- error plumbing (api.hip) ---------------------------------------------------------------
int set_error(int code, const char *fmt, ...);
int check_hip(hipError_t e, const char *what);
int check_launch(const char *what);
// Raise `kernel`'s dynamic-LDS limit to at least `bytes` on the current device.  hipFuncSetAttribute is called only when
// the limit this library last set for (kernel, device) is below `bytes`: nothing per launch in the steady state.
int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what);

#define FRCNN_REQUIRE(cond, ...)                                  \
  do {                                                            \
    if (!(cond)) return ::frcnn::set_error(FRCNN_EINVAL, __VA_ARGS__); \
  } while (0)

// ---- schedule overrides (api.hip) ------------------------------------------------------------
// Diagnostic / test knobs that force one of several equivalent schedules.  Each is read ONCE from the environment variable
// of the same name when the library is loaded and can be changed afterwards only through frcnn_debug_set_schedule():
// no entry point calls getenv.
enum Knob {
  kKnobCropSlice,        // FRCNN_CROP_SLICE            0|1
  kKnobCropRows,         // FRCNN_CROP_ROWS             0|1
  kKnobDecoderThreads,   // FRCNN_DECODER_THREADS       256|1024
  kKnobRoiBboxUnfused,   // FRCNN_ROI_BBOX_UNFUSED      1
  kKnobDecoderUnfused,   // FRCNN_DECODER_UNFUSED       1
  kKnobBitmaskMinR,      // FRCNN_ROI_BBOX_BITMASK_MIN_R n (0 = never)
  kKnobProposalChunk,    // FRCNN_PROPOSAL_CHUNK        64|128|256
  kKnobProposalThreads,  // FRCNN_PROPOSAL_THREADS      256|1024
  kKnobIouTile,          // FRCNN_IOU_TILE              elements per lane of the flat IoU-map kernel
  kKnobIouMapFlat,       // FRCNN_IOU_MAP_FLAT          1
  kKnobCropPrefetch,     // FRCNN_CROP_PREFETCH         workgroups per XCD of the feature-map warm-up pass (0 = none)
  kKnobCropSplit,        // FRCNN_CROP_SPLIT            channel slices of frcnn_roi_pooling's large-image schedule (0 | 2 | 4 | 8)
  kKnobCropSort,         // FRCNN_CROP_SORT             0 = keep the RoIs in their given order there
  kKnobIouReduceSplit,   // FRCNN_IOU_REDUCE_SPLIT      0|1: the 64 boxes x 4 gt quarters form of the fused IoU reductions
  kKnobProposalExactSlices,  // FRCNN_PROPOSAL_EXACT_SLICES 1: the fused RoIBBox kernel cuts its slices by exact rank only
  kKnobSortRadixOnly,    // FRCNN_SORT_RADIX_ONLY       1: sort_select_kernel orders its slots by the stable radix sort only
  kKnobBitmaskThreads,   // FRCNN_BITMASK_THREADS       256|1024: workgroup size of the whole-chip bitmask schedule's scan kernels
  kKnobCropStage,        // FRCNN_CROP_STAGE            0|1: frcnn_roi_pooling's LDS-staged-rows schedule (default: beyond 4 MiB per image)
  kKnobCount
};
bool knob_set(Knob k);
int knob(Knob k, int unset_value);  // the knob's value, or unset_value

static inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- bit helpers ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// Order-preserving float -> uint32 (larger float => larger key); -0 folded onto +0; NaNs sort by
// bit pattern.  Used for every "largest score first / first maximum" decision.
__device__ __forceinline__ uint32_t f_order_key(float f) {
  if (f == 0.0f) f = 0.0f;
  uint32_t u = f2u(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// std::min / std::max exactly as the TF C++ kernels spell them (NaN behaviour included).
__device__ __forceinline__ float std_minf(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_maxf(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float clip01(float v) { return std_maxf(std_minf(v, 1.0f), 0.0f); }

// ---- transcendental restatements ------------------------------------------------------------
// tf.exp on CPU = Eigen's generic pexp_float (Cephes expf), non-FMA code path.  Written out op
// by op so that the result is bit-identical wherever it is compiled without contraction.
__device__ __forceinline__ float cephes_expf(float x0) {
  float x = (x0 < 88.3762626647950f) ? x0 : 88.3762626647950f;
  x = (x > -88.3762626647949f) ? x : -88.3762626647949f;
  float t = x * 1.44269504088896341f;
  t = t + 0.5f;
  float m = floorf(t);
  float r = x;
  t = m * 0.693359375f;
  float z = m * -2.12194440e-4f;
  r = r - t;
  r = r - z;
  float r2 = r * r;
  float y = 1.9875691500E-4f;
  y = y * r;
  y = y + 1.3981999507E-3f;
  y = y * r;
  y = y + 8.3334519073E-3f;
  y = y * r;
  y = y + 4.1665795894E-2f;
  y = y * r;
  y = y + 1.6666665459E-1f;
  y = y * r;
  y = y + 5.0000001201E-1f;
  y = y * r2;
  y = y + r;
  y = y + 1.0f;
  int e = (int)m + 127;
  float scale = u2f((uint32_t)e << 23);
  float res = y * scale;
  return (res > x0) ? res : ((x0 != x0) ? res : x0);
}

// tf.math.log on CPU = Eigen's generic plog_float (Cephes logf), non-FMA code path.
__device__ __forceinline__ float cephes_logf(float x0) {
  if (x0 != x0 || x0 < 0.0f) return u2f(0x7fc00000u);
  if (x0 == 0.0f) return u2f(0xff800000u);
  if (x0 == u2f(0x7f800000u)) return x0;
  float x = (x0 > 1.17549435e-38f) ? x0 : 1.17549435e-38f;
  uint32_t bits = f2u(x);
  float e = (float)(int)(bits >> 23);
  e = e - 126.0f;
  x = u2f((bits & ~0x7f800000u) | 0x3f000000u);
  bool lt = x < 0.707106781186547524f;
  float tmp = lt ? x : 0.0f;
  x = x - 1.0f;
  e = e - (lt ? 1.0f : 0.0f);
  x = x + tmp;
  float x2 = x * x;
  float x3 = x2 * x;
  float y = 7.0376836292E-2f * x;
  y = y + -1.1514610310E-1f;
  float y1 = -1.2420140846E-1f * x;
  y1 = y1 + 1.4249322787E-1f;
  float y2 = 2.0000714765E-1f * x;
  y2 = y2 + -2.4999993993E-1f;
  y = y * x;
  y = y + 1.1676998740E-1f;
  y1 = y1 * x;
  y1 = y1 + -1.6668057665E-1f;
  y2 = y2 * x;
  y2 = y2 + 3.3333331174E-1f;
  y = y * x3;
  y = y + y1;
  y = y * x3;
  y = y + y2;
  y = y * x3;
  y1 = e * -2.12194440e-4f;
  tmp = x2 * 0.5f;
  y = y + y1;
  x = x - tmp;
  y2 = e * 0.693359375f;
  x = x + y;
  x = x + y2;
  return x;
}

// ---- box codec (utils/bbox_utils.py:96-108 and :121-135) --------------------------------------
__device__ __forceinline__ float4 decode_box(float4 a, float4 d) {
  // a = anchor [y1,x1,y2,x2]; d = deltas [dy,dx,dh,dw] already multiplied by the variances.
  float w = a.w - a.y;
  float h = a.z - a.x;
  float t = 0.5f * w;
  float cx = a.y + t;
  t = 0.5f * h;
  float cy = a.x + t;
  float bw = cephes_expf(d.w) * w;
  float bh = cephes_expf(d.z) * h;
  t = d.y * w;
  float bcx = t + cx;
  t = d.x * h;
  float bcy = t + cy;
  t = 0.5f * bh;
  float y1 = bcy - t;
  t = 0.5f * bw;
  float x1 = bcx - t;
  float4 o;
  o.x = y1;
  o.y = x1;
  o.z = bh + y1;
  o.w = bw + x1;
  return o;
}

__device__ __forceinline__ float4 scale_deltas(float4 d, float v0, float v1, float v2, float v3) {
  d.x = d.x * v0;
  d.y = d.y * v1;
  d.z = d.z * v2;
  d.w = d.w * v3;
  return d;
}

__device__ __forceinline__ float4 encode_box(float4 bx, float4 g) {
  float bw = bx.w - bx.y;
  float bh = bx.z - bx.x;
  float t = 0.5f * bw;
  float bcx = bx.y + t;
  t = 0.5f * bh;
  float bcy = bx.x + t;
  float gw = g.w - g.y;
  float gh = g.z - g.x;
  t = 0.5f * gw;
  float gcx = g.y + t;
  t = 0.5f * gh;
  float gcy = g.x + t;
  if (bw == 0.0f) bw = 1e-3f;
  if (bh == 0.0f) bh = 1e-3f;
  float dx = 0.0f, dy = 0.0f, dw = 0.0f, dh = 0.0f;
  if (gw != 0.0f) {
    t = gcx - bcx;
    dx = t / bw;
    t = gw / bw;
    dw = cephes_logf(t);
  }
  if (gh != 0.0f) {
    t = gcy - bcy;
    dy = t / bh;
    t = gh / bh;
    dh = cephes_logf(t);
  }
  float4 o;
  o.x = dy;
  o.y = dx;
  o.z = dh;
  o.w = dw;
  return o;
}

// __ballot takes an int, which costs a v_cndmask + v_cmp to turn a lane mask that already sits in SGPRs into an int and
// back; this form takes the predicate as it is.
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// Inclusive prefix sum over the 64 lanes with DPP: row_shr 1/2/4/8 inside the rows of 16 lanes, then the row totals
// carried over with row_bcast:15 / row_bcast:31 -- six v_add_u32_dpp instead of six dependent ds_bpermute round trips.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source add 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// ---- bare hardware max / min ------------------------------------------------------------------------
// __builtin_fmaxf / fminf lower to llvm.maxnum / minnum, which in IEEE mode canonicalise both inputs first (a v_max_f32
// x, x each): three instructions per max.  Where the caller has established that neither operand is NaN the bare
// v_max_f32 / v_min_f32 give the same value (up to the sign of a zero).
__device__ __forceinline__ float hw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ---- NMS-internal IoU test (TF non_max_suppression_op.cc IOUGreaterThanThreshold) ------------
struct NBox {  // corner-normalised box + its area
  float ymin, xmin, ymax, xmax, area;
};

__device__ __forceinline__ NBox normalise_box(float4 b) {
  NBox o;
  o.ymin = std_minf(b.x, b.z);
  o.xmin = std_minf(b.y, b.w);
  o.ymax = std_maxf(b.x, b.z);
  o.xmax = std_maxf(b.y, b.w);
  float h = o.ymax - o.ymin;
  float w = o.xmax - o.xmin;
  o.area = h * w;
  return o;
}

__device__ __forceinline__ bool nms_iou_gt(const NBox &i, const NBox &j, float thr) {
  if (i.area <= 0.0f || j.area <= 0.0f) return false;
  float iymin = std_maxf(i.ymin, j.ymin);
  float ixmin = std_maxf(i.xmin, j.xmin);
  float iymax = std_minf(i.ymax, j.ymax);
  float ixmax = std_minf(i.xmax, j.xmax);
  float ih = iymax - iymin;
  float iw = ixmax - ixmin;
  ih = std_maxf(ih, 0.0f);
  iw = std_maxf(iw, 0.0f);
  float inter = ih * iw;
  float uni = i.area + j.area;
  uni = uni - inter;
  float iou = inter / uni;
  return iou > thr;
}

// ---- fused IoU reductions (iou_map.hip) ----------------------------------------------------------------
// The per-gt argmax over the boxes is merged across workgroups as one 64-bit atomic max on (order key << 32 | ~index).
// key 0: no box had an IoU above lowest() -> Eigen's ArgMax keeps its initial index 0.
__host__ __device__ __forceinline__ int col_key_index(unsigned long long k) {
  return (k == 0ull) ? 0 : (int)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
}
int iou_map_reduce_impl(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B, int N, int M, float *max_iou,
                        int32_t *argmax_row, int32_t *argmax_col, void *workspace, size_t workspace_bytes, hipStream_t stream,
                        bool finalize);

// ---- diagnostic stamps (only in builds with -DFRCNN_STAMPS; the shipped library has none) ---------
#ifdef FRCNN_STAMPS
extern __device__ long long g_stamps[256];
extern __device__ int g_nstamps;
#define FRCNN_STAMP(tag)                                                  \
  do {                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0 && g_nstamps < 127) {         \
      g_stamps[2 * g_nstamps] = ((long long)(tag) << 48) | ((long long)clock64() & 0xFFFFFFFFFFFFll); \
      g_stamps[2 * g_nstamps + 1] = (long long)wall_clock64();            \
      ++g_nstamps;                                                        \
    }                                                                     \
  } while (0)
// the same for one chosen workgroup (`cond`: block-uniform)
#define FRCNN_STAMP_IF(cond, tag)                                         \
  do {                                                                    \
    if ((cond) && threadIdx.x == 0 && g_nstamps < 127) {                  \
      g_stamps[2 * g_nstamps] = ((long long)(tag) << 48) | ((long long)clock64() & 0xFFFFFFFFFFFFll); \
      g_stamps[2 * g_nstamps + 1] = (long long)wall_clock64();            \
      ++g_nstamps;                                                        \
    }                                                                     \
  } while (0)
#else
#define FRCNN_STAMP(tag) do { } while (0)
#define FRCNN_STAMP_IF(cond, tag) do { } while (0)
#endif

// ---- wave helpers (wave = 64 lanes on gfx950) ---------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << (threadIdx.x & 63u)) - 1ull;
}

// Where to find a box for the sort/NMS machinery: either stored boxes (B,n,q,4) or boxes decoded on
// the fly from anchors + deltas (RoIBBox / Decoder fusion).
struct BoxSource {
  const float *boxes;       // direct mode: (B,n,q,4); NULL in decode mode
  const float *anchors;     // decode mode
  const float *deltas;      // decode mode
  long long boxes_stride_b;   // floats per image
  long long anchors_stride_b; // floats per image (0 = anchors shared)
  long long deltas_stride_b;  // floats per image
  int q;                    // boxes per element (1 or c) -- direct mode: box q-dim, decode mode: delta q-dim
  float v0, v1, v2, v3;     // variances (decode mode)
};

__device__ __forceinline__ float4 load_box(const BoxSource &s, int b, int i, int cls) {
  int qi = (s.q > 1) ? cls : 0;
  if (s.boxes) {
    return *reinterpret_cast<const float4 *>(s.boxes + (long long)b * s.boxes_stride_b +
                                             ((long long)i * s.q + qi) * 4);
  }
  float4 a = *reinterpret_cast<const float4 *>(s.anchors + (long long)b * s.anchors_stride_b + (long long)i * 4);
  float4 d = *reinterpret_cast<const float4 *>(s.deltas + (long long)b * s.deltas_stride_b +
                                               ((long long)i * s.q + qi) * 4);
  d = scale_deltas(d, s.v0, s.v1, s.v2, s.v3);
  return decode_box(a, d);
}

}  // namespace frcnn
