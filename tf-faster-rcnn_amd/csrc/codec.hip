// codec.hip -- anchor generation and the box delta codec (elementwise, one float4 per lane).
//
// All three kernels are pure streaming: 16 B per lane per access, fully coalesced; at the
// reference's sizes (N <= 36k boxes per image) they are launch-latency bound, so they use a flat
// grid with one box per thread and no LDS.
#include "common.hpp"

namespace frcnn {

// utils/bbox_utils.py:53-57: anchors[(gy*G+gx)*A + a] = clip(base[a] + [cy,cx,cy,cx], 0, 1)
__global__ __launch_bounds__(256) void anchors_kernel(const float4 *__restrict__ base, int A,
                                                      const float *__restrict__ coords, int G,
                                                      float4 *__restrict__ out, int total) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int cell = i / A;
  int a = i - cell * A;
  int gy = cell / G;
  int gx = cell - gy * G;
  float cy = coords[gy], cx = coords[gx];
  float4 b = base[a];
  float4 o;
  o.x = b.x + cy;
  o.y = b.y + cx;
  o.z = b.z + cy;
  o.w = b.w + cx;
  // tf.clip_by_value(v, 0, 1) = maximum(minimum(v, 1), 0)
  o.x = (o.x < 1.0f) ? o.x : 1.0f;
  o.y = (o.y < 1.0f) ? o.y : 1.0f;
  o.z = (o.z < 1.0f) ? o.z : 1.0f;
  o.w = (o.w < 1.0f) ? o.w : 1.0f;
  o.x = (o.x > 0.0f) ? o.x : 0.0f;
  o.y = (o.y > 0.0f) ? o.y : 0.0f;
  o.z = (o.z > 0.0f) ? o.z : 0.0f;
  o.w = (o.w > 0.0f) ? o.w : 0.0f;
  out[i] = o;
}

template <bool HAS_VAR>
__global__ __launch_bounds__(256) void decode_kernel(const float4 *__restrict__ anchors,
                                                     long long anchors_stride_b,  // in boxes
                                                     const float4 *__restrict__ deltas,
                                                     float4 *__restrict__ out, long long N,
                                                     long long total, float v0, float v1, float v2,
                                                     float v3) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  long long b = i / N;
  long long n = i - b * N;
  float4 a = anchors[b * anchors_stride_b + n];
  float4 d = deltas[i];
  if (HAS_VAR) d = scale_deltas(d, v0, v1, v2, v3);
  out[i] = decode_box(a, d);
}

template <bool HAS_VAR>
__global__ __launch_bounds__(256) void encode_kernel(const float4 *__restrict__ bboxes,
                                                     long long bboxes_stride_b,
                                                     const float4 *__restrict__ gt,
                                                     float4 *__restrict__ out, long long N,
                                                     long long total, float v0, float v1, float v2,
                                                     float v3) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  long long b = i / N;
  long long n = i - b * N;
  float4 o = encode_box(bboxes[b * bboxes_stride_b + n], gt[i]);
  if (HAS_VAR) {
    o.x = o.x / v0;
    o.y = o.y / v1;
    o.z = o.z / v2;
    o.w = o.w / v3;
  }
  out[i] = o;
}

// The "final host gather" of the sharded deployment for the small per-image outputs: a plain copy kernel
// whose destination may be pinned host memory (device-mapped), so that the gather is one more kernel on the stage's stream
// instead of a copy-engine transfer with its cross-queue synchronisation.
__global__ __launch_bounds__(256) void gather_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16,
                                                          const unsigned char *__restrict__ src_tail,
                                                          unsigned char *__restrict__ dst_tail, int tail) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// Up to kGatherSegments (src, dst, bytes) copies in ONE launch: blockIdx.y selects the segment.
static const int kGatherSegments = 8;
struct GatherSegments {
  const uint4 *src[kGatherSegments];
  uint4 *dst[kGatherSegments];
  long long n16[kGatherSegments];  // 16-byte units (sizes are multiples of 16)
};

__global__ __launch_bounds__(256) void gather_copy_multi_kernel(GatherSegments g) {
  const uint4 *__restrict__ src = g.src[0];
  uint4 *__restrict__ dst = g.dst[0];
  long long n16 = g.n16[0];
#pragma unroll
  for (int i = 1; i < kGatherSegments; ++i) {  // block-uniform selects: no dynamic indexing of the kernel arguments
    if ((int)blockIdx.y == i) {
      src = g.src[i];
      dst = g.dst[i];
      n16 = g.n16[i];
    }
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_generate_anchors(const float *base_anchors, int A, const float *grid_coords, int G,
                           float *out, frcnn_stream_t stream) {
  FRCNN_REQUIRE(base_anchors && grid_coords && out, "generate_anchors: null pointer");
  FRCNN_REQUIRE(A > 0 && G > 0, "generate_anchors: A=%d G=%d must be positive", A, G);
  FRCNN_REQUIRE((long long)A * G * G < (1ll << 31), "generate_anchors: too many anchors");
  FRCNN_REQUIRE(aligned_to(base_anchors, 16) && aligned_to(out, 16), "generate_anchors: pointers must be 16-byte aligned");
  int total = A * G * G;
  hipLaunchKernelGGL(anchors_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(base_anchors), A, grid_coords, G,
                     reinterpret_cast<float4 *>(out), total);
  return check_launch("generate_anchors");
}

int frcnn_decode_boxes(const float *anchors, int anchors_batched, const float *deltas,
                       const float *variances, int64_t B, int64_t N, float *out,
                       frcnn_stream_t stream) {
  FRCNN_REQUIRE(anchors && deltas && out, "decode_boxes: null pointer");
  FRCNN_REQUIRE(B >= 0 && N >= 0, "decode_boxes: negative shape");
  FRCNN_REQUIRE(aligned_to(anchors, 16) && aligned_to(deltas, 16) && aligned_to(out, 16),
                "decode_boxes: pointers must be 16-byte aligned");
  long long total = (long long)B * N;
  if (total == 0) return FRCNN_OK;
  FRCNN_REQUIRE((total + 255) / 256 < (1ll << 31), "decode_boxes: too many boxes");
  dim3 grid((unsigned)((total + 255) / 256));
  long long stride = anchors_batched ? N : 0;
  if (variances) {
    hipLaunchKernelGGL(decode_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(anchors), stride,
                       reinterpret_cast<const float4 *>(deltas), reinterpret_cast<float4 *>(out),
                       (long long)N, total, variances[0], variances[1], variances[2], variances[3]);
  } else {
    hipLaunchKernelGGL(decode_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(anchors), stride,
                       reinterpret_cast<const float4 *>(deltas), reinterpret_cast<float4 *>(out),
                       (long long)N, total, 1.0f, 1.0f, 1.0f, 1.0f);
  }
  return check_launch("decode_boxes");
}

int frcnn_encode_boxes(const float *bboxes, int bboxes_batched, const float *gt_boxes,
                       const float *div_variances, int64_t B, int64_t N, float *out,
                       frcnn_stream_t stream) {
  FRCNN_REQUIRE(bboxes && gt_boxes && out, "encode_boxes: null pointer");
  FRCNN_REQUIRE(B >= 0 && N >= 0, "encode_boxes: negative shape");
  FRCNN_REQUIRE(aligned_to(bboxes, 16) && aligned_to(gt_boxes, 16) && aligned_to(out, 16),
                "encode_boxes: pointers must be 16-byte aligned");
  long long total = (long long)B * N;
  if (total == 0) return FRCNN_OK;
  FRCNN_REQUIRE((total + 255) / 256 < (1ll << 31), "encode_boxes: too many boxes");
  dim3 grid((unsigned)((total + 255) / 256));
  long long stride = bboxes_batched ? N : 0;
  if (div_variances) {
    hipLaunchKernelGGL(encode_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(bboxes), stride,
                       reinterpret_cast<const float4 *>(gt_boxes), reinterpret_cast<float4 *>(out),
                       (long long)N, total, div_variances[0], div_variances[1], div_variances[2],
                       div_variances[3]);
  } else {
    hipLaunchKernelGGL(encode_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(bboxes), stride,
                       reinterpret_cast<const float4 *>(gt_boxes), reinterpret_cast<float4 *>(out),
                       (long long)N, total, 1.0f, 1.0f, 1.0f, 1.0f);
  }
  return check_launch("encode_boxes");
}


int frcnn_gather_copy(const void *src, void *dst, size_t bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(src && dst, "gather_copy: null pointer");
  FRCNN_REQUIRE(aligned_to(src, 16) && aligned_to(dst, 16), "gather_copy: pointers must be 16-byte aligned");
  FRCNN_REQUIRE(bytes < ((size_t)1 << 40), "gather_copy: size too large");
  if (bytes == 0) return FRCNN_OK;
  const long long n16 = (long long)(bytes / 16);
  const int tail = (int)(bytes % 16);
  const long long blocks = (n16 + 255) / 256 > 0 ? (n16 + 255) / 256 : 1;
  hipLaunchKernelGGL(gather_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst), n16,
                     reinterpret_cast<const unsigned char *>(src) + n16 * 16, reinterpret_cast<unsigned char *>(dst) + n16 * 16,
                     tail);
  return check_launch("gather_copy");
}

int frcnn_gather_copy_multi(const void *const *srcs, void *const *dsts, const size_t *bytes, int n, frcnn_stream_t stream) {
  FRCNN_REQUIRE(srcs && dsts && bytes, "gather_copy_multi: null table");
  FRCNN_REQUIRE(n >= 0 && n <= kGatherSegments, "gather_copy_multi: 0..%d segments (got %d)", kGatherSegments, n);
  GatherSegments g = {};
  long long most = 0;
  int used = 0;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] == 0) continue;
    FRCNN_REQUIRE(srcs[i] && dsts[i], "gather_copy_multi: null pointer in segment %d", i);
    FRCNN_REQUIRE(aligned_to(srcs[i], 16) && aligned_to(dsts[i], 16) && bytes[i] % 16 == 0,
                  "gather_copy_multi: segment %d must be 16-byte aligned and a multiple of 16 bytes", i);
    FRCNN_REQUIRE(bytes[i] < ((size_t)1 << 40), "gather_copy_multi: segment %d too large", i);
    g.src[used] = reinterpret_cast<const uint4 *>(srcs[i]);
    g.dst[used] = reinterpret_cast<uint4 *>(dsts[i]);
    g.n16[used] = (long long)(bytes[i] / 16);
    most = g.n16[used] > most ? g.n16[used] : most;
    ++used;
  }
  if (used == 0) return FRCNN_OK;
  long long bx = (most + 255) / 256;
  if (bx > 4096) bx = 4096;  // grid-stride beyond that
  hipLaunchKernelGGL(gather_copy_multi_kernel, dim3((unsigned)bx, (unsigned)used), dim3(256), 0, (hipStream_t)stream, g);
  return check_launch("gather_copy_multi");
}

}  // extern "C"
