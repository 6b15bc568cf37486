// crop_resize.hip -- bilinear crop_and_resize (RoI pooling), the byte-heaviest kernel of the path.
//
// Replaces tf.image.crop_and_resize as called at models/faster_rcnn.py:283-288.
//
// Roofline: HBM-write bound.  Per box the kernel writes ch*cw*C*4 bytes (33.9 MB per image at
// R=300, 7x7, C=576) and reads one feature map per image (2.36 MB), i.e. 36,234,096 algorithmic
// bytes per image at BASELINE config 2.
//
// Mapping:
//   - one 256-thread workgroup per box; the output crop of a box is one contiguous run of
//     ch*cw*C floats, so lane e of the block stores float4 number e, e+256, ... : every store is
//     a full 1 KiB wave line, written once with the nontemporal hint so the streamed output does not
//     evict the feature map from L2;
//   - the 4 bilinear taps of an output pixel are 16 B-per-lane loads from the image's feature
//     map, which stays resident in the XCD's 4 MiB L2: workgroups are remapped so that one XCD
//     (blockIdx % 8) walks a contiguous range of boxes, i.e. one or two images at a time;
//   - per-pixel sampling parameters (tap byte offsets, lerp weights) are computed once per box by the
//     first ch*cw lanes and broadcast through LDS;
//   - the inner loop is kept lean because the CU's vector-memory path is the co-limiter (4 tap loads + 1 store of
//     16 B per lane = 2.78 GB per launch through 64 B/clk/CU; TA busy 75 % in profiles/r01_pmc_crop_*): SGPR-base +
//     32-bit lane offsets (no 64-bit address arithmetic), packed fp32 lerps, 4 outputs (16 taps) in flight per lane
//     at 4 waves per SIMD.
//   Rejected after measurement (tools/crop_bench.hip): a (box, 256-B channel chunk) workgroup cuts L2 reads by 32 %
//   through L1 reuse yet runs 12-20 % slower (more workgroups, same bytes through the vector-memory path); staging the
//   unique taps in LDS caps the bytes in flight per CU below what the HBM stream needs.
//
// Numerics: the sampling coordinates and the three lerps are evaluated in exactly the TF
// kernel's operation order, one rounding per operation (build has contraction off).
#include "common.hpp"

namespace frcnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) PixParam {
  unsigned off_tl, off_tr, off_bl, off_br;  // tap offsets in BYTES relative to the image (32-bit: host checks the size)
  float lx, ly;
  int valid;
  int pad;
};

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  typedef f32x4 type;
};
template <>
struct VecT<1> {
  typedef float type;
};

// Sampling parameters of output pixel p of a box, in the TF kernel's operation order.
__device__ __forceinline__ PixParam pix_param(int p, int H, int W, unsigned pix_bytes, int ch, int cw, float y1, float x1,
                                              float y2, float x2, bool img_ok) {
  const int y = p / cw;
  const int x = p - y * cw;
  const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
  float in_y, in_x;
  if (ch > 1) {
    float t = y2 - y1;
    t = t * hm1;
    const float hs = t / (float)(ch - 1);
    const float t0 = y1 * hm1;
    const float t1 = (float)y * hs;
    in_y = t0 + t1;
  } else {
    float t0 = y1 + y2;
    t0 = 0.5f * t0;
    in_y = t0 * hm1;
  }
  if (cw > 1) {
    float t = x2 - x1;
    t = t * wm1;
    const float ws = t / (float)(cw - 1);
    const float t0 = x1 * wm1;
    const float t1 = (float)x * ws;
    in_x = t0 + t1;
  } else {
    float t0 = x1 + x2;
    t0 = 0.5f * t0;
    in_x = t0 * wm1;
  }
  const bool ok = img_ok && (in_y >= 0.0f) && (in_y <= hm1) && (in_x >= 0.0f) && (in_x <= wm1);
  PixParam prm;
  if (ok) {
    const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
    const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
    prm.ly = in_y - (float)top;
    prm.lx = in_x - (float)left;
    prm.off_tl = (unsigned)(top * W + left) * pix_bytes;
    prm.off_tr = (unsigned)(top * W + right) * pix_bytes;
    prm.off_bl = (unsigned)(bot * W + left) * pix_bytes;
    prm.off_br = (unsigned)(bot * W + right) * pix_bytes;
    prm.valid = 1;
  } else {
    prm.ly = 0.0f;
    prm.lx = 0.0f;
    prm.off_tl = prm.off_tr = prm.off_bl = prm.off_br = 0;
    prm.valid = 0;
  }
  prm.pad = 0;
  return prm;
}

// a - b on the packed-fp32 pipe: v_pk_add_f32 with the neg modifier on b (exactly a - b); the compiler packs fadd/fmul of
// <4 x float> but splits fsub into four v_sub_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vsub(float a, float b) { return a - b; }
__device__ __forceinline__ f32x4 vsub(f32x4 a, f32x4 b) {
  f32x2 lo, hi;
  const f32x2 alo = a.xy, ahi = a.zw, blo = b.xy, bhi = b.zw;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
  f32x4 r;
  r.xy = lo;
  r.zw = hi;
  return r;
}

// One bilinear sample: three lerps, one rounding per operation.  `base` is wave-uniform and the lane offset 32-bit, so
// the loads take the SGPR-base + VGPR-offset form (no 64-bit address arithmetic per tap); subtractions go through vsub (packed).
template <class vec_t>
__device__ __forceinline__ vec_t bilerp(const char *__restrict__ base, const PixParam &prm, unsigned cb, float extrap) {
  const vec_t tl = *reinterpret_cast<const vec_t *>(base + (size_t)(prm.off_tl + cb));
  const vec_t tr = *reinterpret_cast<const vec_t *>(base + (size_t)(prm.off_tr + cb));
  const vec_t bl = *reinterpret_cast<const vec_t *>(base + (size_t)(prm.off_bl + cb));
  const vec_t br = *reinterpret_cast<const vec_t *>(base + (size_t)(prm.off_br + cb));
  vec_t t = vsub(tr, tl);
  t = t * prm.lx;
  const vec_t tp = tl + t;
  t = vsub(br, bl);
  t = t * prm.lx;
  const vec_t bt = bl + t;
  t = vsub(bt, tp);
  t = t * prm.ly;
  vec_t v = tp + t;
  if (!prm.valid) v = (vec_t)(extrap);
  return v;
}

// Feature maps that are cold (in HBM, not in L2 / Infinity Cache) cost the box kernel a third of its speed: its taps are
// scattered 16-B-per-lane loads whose misses queue behind the kernel's own write stream (measured at config 2 with input
// sets rotating through 650 MB: 134 us per batch against 100 us warm).  So the launch carries WARM-UP WORKGROUPS: per XCD
// and per image of the XCD's box range, pf_blocks workgroups that read the image once with full-line coalesced loads and
// discard it.  They sit in the XCD's slot sequence pf_lead boxes ahead of the image's first box -- workgroups are
// dispatched in blockIdx order -- so an image streams into the L2 that is going to sample it while the boxes of the
// previous image are still being written, and no second launch (with its drain and launch gap) is needed.
//
// Slot sequence of one XCD over its boxes [start, start + count): for every image j of the range, in order,
//   [boxes up to p_j)  [pf_blocks warm-up workgroups of image j]      p_0 = start, p_j = max(p_{j-1}, first box of j - pf_lead)
// then the remaining boxes.  Returns the box of this slot, or -1 with (*pf_img, *pf_piece) set for a warm-up slot, or -2
// for a slot past the end.
__device__ __forceinline__ int crop_slot_role(int slot, int start, int count, int npg, int bpi, int pf_blocks, int pf_lead,
                                              int *pf_box) {
  int v = slot;
  int cb = start;  // next box not yet placed in the sequence
  const int end = start + count;
  int first = start;  // first box (of this XCD's range) of the image under consideration
  while (first < end) {
    int p = first - pf_lead;
    if (p < cb) p = cb;
    const int nb = p - cb;
    if (v < nb) return cb + v;
    v -= nb;
    cb = p;
    if (v < pf_blocks) {
      *pf_box = first;
      return -1 - v;  // warm-up piece v of the image that box `first` samples
    }
    v -= pf_blocks;
    const int lb = first % npg;  // box index inside its triple
    first += bpi - (lb % bpi);   // first box of the next image
  }
  return (cb + v < end) ? cb + v : -1000000;
}

// Channel-slice kernel for feature maps that do not fit an XCD's L2 (e.g. 63x63x576 = 9.1 MB per image).  The channels
// are cut into S slices (S = 2, 4 or 8); the 8 XCDs form 8/S lane groups of S XCDs; a lane group walks a contiguous range
// of boxes and each of its XCDs (blockIdx % 8 = lane group * S + slice) samples only its slice, so an XCD's L2 has to
// hold 1/S of an image.  One workgroup = (box, slice): ch*cw segments of SW float4, strided by C floats.
// `order` (optional): the boxes of every image in spatial order (roi_order_kernel); the lane group then sweeps an image
// top to bottom and the boxes in flight at any moment share a band of source rows instead of being spread over the whole
// image (RoIBBox emits them in score order, i.e. spatially at random).  Every box is still written to its own slot.
__global__ __launch_bounds__(256) void crop_resize_slice_kernel(
    const float *__restrict__ image, int B, int H, int W, int C, const float *__restrict__ boxes,
    const int *__restrict__ box_ind, const int *__restrict__ order, int n, int boxes_per_image, int ch, int cw, float extrap,
    float *__restrict__ out, int S, int SW, int pf_blocks, int pf_lead, int pf_linear) {
  __shared__ PixParam pp[256];
  const int tid = threadIdx.x;
  const int grp = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int nlg = 8 / S;
  const int slice = grp % S, lg = grp / S;
  const int q = n / nlg, r = n - q * nlg;
  const int start = (lg < r) ? lg * (q + 1) : r * (q + 1) + (lg - r) * q;
  const int count = q + (lg < r ? 1 : 0);
  const int CV = C >> 2;
  const int c0 = slice * SW;
  const int sw = (CV - c0 < SW) ? (CV - c0) : SW;  // float4 channels of this slice (may be <= 0 for the last ones)
  if (sw <= 0) return;
  int unit;
  if (pf_blocks > 0) {
    // warm-up workgroups in the slot sequence (see crop_slot_role): this XCD's slice of an image, streamed into its L2
    // pf_lead boxes before the image's first box (RoIPooling layout only: box_ind == NULL, order == NULL)
    int pf_box = 0;
    const int role = crop_slot_role(slot, start, count, n, boxes_per_image, pf_blocks, pf_lead, &pf_box);
    if (role <= -1000000) return;
    if (role < 0) {
      const int piece = -1 - role;
      const int img_i = pf_box / boxes_per_image;
      if (img_i >= B) return;
      f32x4 acc = (f32x4)(0.0f);
      if (pf_linear) {
        // the S XCDs of the lane group read one contiguous S-th of the image each (full lines, no address arithmetic): the
        // whole image lands in the memory-side Infinity Cache, from where every XCD's taps are served
        const long long image_bytes = (long long)H * W * C * 4;
        const long long part = (image_bytes / S + 15) & ~15ll;
        const long long lo = (long long)slice * part;
        const long long hi = (lo + part < image_bytes) ? lo + part : image_bytes;
        const char *src = reinterpret_cast<const char *>(image + (long long)img_i * H * W * C);
#pragma unroll 8
        for (long long o = lo + ((long long)piece * 256 + tid) * 16; o + 16 <= hi; o += (long long)pf_blocks * 256 * 16) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(src + o);
          acc.x += v.x;
        }
      } else {
        const char *src = reinterpret_cast<const char *>(image + (long long)img_i * H * W * C) + (size_t)c0 * 16;
        const int total = H * W * sw;  // float4 of the slice
#pragma unroll 8
        for (int e = piece * 256 + tid; e < total; e += pf_blocks * 256) {
          const int px = e / sw, c = e - px * sw;
          const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)px * ((size_t)C * 4) + (size_t)c * 16);
          acc.x += v.x;  // (any use keeps the load; nothing is stored)
        }
      }
      asm volatile("" ::"v"(acc));
      return;
    }
    unit = role;
  } else {
    if (slot >= count) return;
    unit = start + slot;
  }
  const int box = order ? order[unit] : unit;
  const int bimg = box_ind ? box_ind[box] : (box / boxes_per_image);
  const bool img_ok = (bimg >= 0) && (bimg < B);
  const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + (long long)box * 4);
  const int P = ch * cw;
  const char *__restrict__ base =
      reinterpret_cast<const char *>(image + (long long)(img_ok ? bimg : 0) * H * W * C) + (size_t)c0 * 16;
  char *__restrict__ dstb = reinterpret_cast<char *>(out + (long long)box * P * C) + (size_t)c0 * 16;
  const unsigned pix_bytes = (unsigned)C * 4u;

  for (int p0 = 0; p0 < P; p0 += 256) {
    const int np = (P - p0 < 256) ? (P - p0) : 256;
    if (p0 > 0) __syncthreads();
    if (tid < np) pp[tid] = pix_param(p0 + tid, H, W, pix_bytes, ch, cw, bx.x, bx.y, bx.z, bx.w, img_ok);
    __syncthreads();
    const int total = np * sw;
    int p = tid / sw;
    int c = tid - p * sw;
    const int dp = 256 / sw, dc = 256 - dp * sw;
    for (int e0 = tid; e0 < total; e0 += 256 * 4) {
      f32x4 v[4];
      unsigned off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = e0 + u * 256 < total;
        const int pu = in ? p : 0, cu = in ? c : 0;
        v[u] = bilerp<f32x4>(base, pp[pu], (unsigned)cu * 16u, extrap);
        off[u] = (unsigned)(p0 + pu) * pix_bytes + (unsigned)cu * 16u;
        c += dc;
        p += dp;
        if (c >= sw) {
          c -= sw;
          ++p;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * 256 < total) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4 *>(dstb + (size_t)off[u]));
    }
  }
}

// order[b*R + rank] = b*R + i, rank = position of RoI i of image b when the image's RoIs are sorted by vertical centre
// (ties by index; total order on the float bits, so NaN boxes land somewhere instead of breaking the permutation).
// One workgroup per image, R <= 1024 keys in LDS, rank by counting: R*R/256 LDS reads per lane (~3 us at R = 300).
__global__ __launch_bounds__(256) void roi_order_kernel(const float *__restrict__ rois, int R, int *__restrict__ order) {
  __shared__ uint32_t key[1024];
  const int b = blockIdx.x;
  const float *rb = rois + (long long)b * R * 4;
  for (int i = threadIdx.x; i < R; i += 256) {
    const f32x4 bx = *reinterpret_cast<const f32x4 *>(rb + (long long)i * 4);
    key[i] = f_order_key(bx.x + bx.z);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R; i += 256) {
    const uint32_t k = key[i];
    int rank = 0;
    for (int j = 0; j < R; ++j) {
      const uint32_t kj = key[j];
      rank += (kj < k) || (kj == k && j < i);
    }
    order[(long long)b * R + rank] = b * R + i;
  }
}

// ---- source-row ownership (feature maps beyond an XCD's L2) ---------------------------------------------------------------
// An output row of a box samples exactly two source rows (top, bot = top + 1).  The B*R*ch (box, output row) items of a
// launch are ordered per image by `top` (crop_rows_order_kernel: one counting sort per image) and XCD g takes the g-th
// eighth of every image's sorted list: all eight XCDs work on the SAME image, each on its own band of source rows, so an
// XCD's L2 holds about an eighth of an image's rows with all their channels (1.1 MB at 63x63x576) instead of a slice of
// the channels of all rows, every workgroup writes one contiguous run of cw*C floats (16 KB), and a feature-map byte is
// fetched from HBM by one XCD only (plus the row both neighbours share).  The split is by item COUNT, not by row number:
// every XCD gets the same number of items per image whatever the RoIs' spatial distribution.  Warm-up workgroups stream the
// NEXT image's band of rows (a contiguous byte range in NHWC) into the L2 while the current image's items are written.
//
// order[b*n + rank] = item (r*ch + y) of image b at that rank of the top order (rows whose in_y is out of range carry no
// taps and sort last); bounds[(b*8 + g)*2 + {0,1}] = first / last source row XCD g samples in image b (first > last: none).
// One workgroup of 1024 threads per image; LDS: (H + 2) bins | n tops (uint16) | 16 wave sums.
// rowstart[b*(H + 2) + t] = rank of the first item of image b whose top is t (t = H: the rows without taps; t = H + 1: n).
// rec (optional): the sampling parameters of every output pixel in rank order, rec[(b*n + rank)*cw + x], for the kernel that
// stages source rows in LDS: one coalesced 16-byte load per pixel there instead of item -> box -> coordinates.
struct __attribute__((aligned(16))) StageRec {
  int dst;        // position of the pixel's C floats in the output, in float4 units
  float lx, ly;
  unsigned bits;  // left | right << 10 | (bot == top) << 20 | valid << 21   (columns < 1024)
};

__global__ __launch_bounds__(1024) void crop_rows_order_kernel(const float *__restrict__ boxes, int R, int H, int ch,
                                                               int *__restrict__ order, int *__restrict__ bounds,
                                                               int *__restrict__ rowstart, StageRec *__restrict__ rec, int W,
                                                               int cw, int CV, int NS) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = R * ch;
  int *bins = sm;                                                  // [H + 1] counts, then running cursors
  int *wsum = sm + 1024 + 8;                                       // [16]
  unsigned short *tops = reinterpret_cast<unsigned short *>(wsum + 16);  // [n] top of item i, then top at rank
  unsigned short *sorted = tops + ((n + 1) & ~1);                  // [n]
  for (int i = tid; i <= H; i += 1024) bins[i] = 0;
  __syncthreads();
  const float hm1 = (float)(H - 1);
  for (int i = tid; i < n; i += 1024) {
    const int r = i / ch, y = i - r * ch;
    const float y1 = boxes[((long long)b * R + r) * 4 + 0];
    const float y2 = boxes[((long long)b * R + r) * 4 + 2];
    float in_y;  // pix_param's expression, operation for operation
    if (ch > 1) {
      float t = y2 - y1;
      t = t * hm1;
      const float hs = t / (float)(ch - 1);
      const float t0 = y1 * hm1;
      const float t1 = (float)y * hs;
      in_y = t0 + t1;
    } else {
      float t0 = y1 + y2;
      t0 = 0.5f * t0;
      in_y = t0 * hm1;
    }
    const int top = ((in_y >= 0.0f) && (in_y <= hm1)) ? (int)floorf(in_y) : H;
    tops[i] = (unsigned short)top;
    atomicAdd(&bins[top], 1);
  }
  __syncthreads();
  {  // exclusive scan of the H + 1 <= 1024 bins: thread t owns bin t
    const int v = (tid <= H) ? bins[tid] : 0;
    const int inc = wave_incl_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    if (tid <= H) {
      bins[tid] = base + inc - v;
      rowstart[(long long)b * (H + 2) + tid] = base + inc - v;
    }
    if (tid == 0) rowstart[(long long)b * (H + 2) + H + 1] = n;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int top = tops[i];
    const int pos = atomicAdd(&bins[top], 1);  // (any order inside one source row: it only decides which workgroup writes first)
    order[(long long)b * n + pos] = i;
    sorted[pos] = (unsigned short)top;
    if (rec) {
      const int r = i / ch, y = i - r * ch;
      const long long box = (long long)b * R + r;
      const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + box * 4);
      StageRec *out_rec = rec + ((long long)b * n + pos) * cw;
      for (int x = 0; x < cw; ++x) {
        const PixParam prm = pix_param(y * cw + x, H, W, 1u, ch, cw, bx.x, bx.y, bx.z, bx.w, true);  // offsets in pixels
        StageRec o;
        o.dst = (int)(((box * ch + y) * cw + x) * CV);
        o.lx = prm.lx;
        o.ly = prm.ly;
        o.bits = 0u;
        if (prm.valid) {  // (then top is this item's top: the same expression)
          const unsigned left = prm.off_tl - (unsigned)top * (unsigned)W, right = prm.off_tr - (unsigned)top * (unsigned)W;
          o.bits = left | (right << 10) | ((prm.off_bl == prm.off_tl ? 1u : 0u) << 20) | (1u << 21);
        }
        out_rec[x] = o;
      }
    }
  }
  __syncthreads();
  if (rec) {
    // strips for the staged-row kernel: NS runs of consecutive source rows with about n / NS items each; strip k starts at
    // the first row whose items begin at or after rank k * n / NS (bounds[b*16 + k], k = 0 .. NS <= 15; row H + 1 = the end)
    if (tid <= NS) {
      int tk = H + 1;
      if (tid < NS) {
        const int target = (int)(((long long)tid * n) / NS);
        int lo = 0, hi = H + 1;  // smallest t with rowstart(t) >= target; rowstart(t) = end of bin t - 1 after the placement
        while (lo < hi) {
          const int m = (lo + hi) >> 1;
          const int rs = (m == 0) ? 0 : bins[m - 1];
          if (rs >= target) hi = m; else lo = m + 1;
        }
        tk = lo;
      }
      bounds[(long long)b * 16 + tid] = tk;
    }
    return;
  }
  if (tid < 8) {
    const int per = (n + 7) >> 3;
    const int lo = tid * per;
    int hi = lo + per;
    if (hi > n) hi = n;
    int row_lo = 1, row_hi = 0;  // empty
    if (lo < hi && sorted[lo] < H) {
      row_lo = sorted[lo];
      int k = hi - 1;
      row_hi = sorted[k];
      if (row_hi >= H) {  // the share ends in rows without taps: the last row that has some (binary search, tops ascend)
        int a = lo, c = k;  // sorted[a] < H <= sorted[c]
        while (c - a > 1) {
          const int m = (a + c) >> 1;
          if (sorted[m] < H) a = m; else c = m;
        }
        row_hi = sorted[a];
      }
      row_hi = (row_hi + 1 < H) ? row_hi + 1 : H - 1;  // bot = top + 1
    }
    bounds[((long long)b * 8 + tid) * 2 + 0] = row_lo;
    bounds[((long long)b * 8 + tid) * 2 + 1] = row_hi;
  }
}

// One workgroup = `ipw` consecutive items of its XCD's share of one image ((box, output row) each: cw pixels x C channels,
// one contiguous run of the output), ipw * cw <= 256: the items' pixel parameters are computed together by the first
// ipw * cw lanes (item -> box -> coordinates: two dependent loads, paid once per workgroup instead of once per 16 KB), then
// the workgroup streams the ipw runs like the box kernel streams a crop.  A pixel's parameters carry the position of its
// C floats in the output (PixParam::pad, in float4 units), since consecutive items belong to different boxes.
// Slot sequence of an XCD: pf warm-up workgroups for each of the first `lead` images, then per image b: pf warm-up
// workgroups for image b + lead followed by the wpg = ceil(per / ipw) workgroups of the XCD's `per` items of image b.
template <int WPE, int UNROLL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void crop_resize_rowband_kernel(const float *__restrict__ image, int B, int H, int W, int C, const float *__restrict__ boxes,
                                const int *__restrict__ order, const int *__restrict__ bounds, int R, int ch, int cw,
                                float extrap, float *__restrict__ out, int pf, int ipw, int lead) {
  __shared__ PixParam pp[256];
  const int tid = threadIdx.x;
  const int g = blockIdx.x & 7;
  int s = blockIdx.x >> 3;
  const int n = R * ch;
  const int per = (n + 7) >> 3;
  const int wpg = (per + ipw - 1) / ipw;
  int b, idx, bi;
  if (s < pf * lead) {  // warm-up of the first `lead` images, up front
    bi = s / pf;
    idx = -1;
    s -= bi * pf;  // piece
    b = 0;
  } else {
    s -= pf * lead;
    b = s / (pf + wpg);
    s -= b * (pf + wpg);
    idx = s - pf;  // < 0: warm-up piece s of image b + lead
    bi = b + lead;
  }
  if (b >= B) return;
  if (idx < 0) {
    const int piece = s;
    if (bi >= B) return;
    const int row_lo = bounds[((long long)bi * 8 + g) * 2 + 0], row_hi = bounds[((long long)bi * 8 + g) * 2 + 1];
    if (row_lo > row_hi) return;
    const long long row_bytes = (long long)W * C * 4;
    const char *src = reinterpret_cast<const char *>(image) + ((long long)bi * H + row_lo) * row_bytes;
    const long long bytes = (long long)(row_hi - row_lo + 1) * row_bytes;
    f32x4 acc = (f32x4)(0.0f);
#pragma unroll 8
    for (long long o = ((long long)piece * 256 + tid) * 16; o + 16 <= bytes; o += (long long)pf * 256 * 16) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(src + o);
      acc.x += v.x;  // (any use keeps the load; nothing is stored)
    }
    asm volatile("" ::"v"(acc));
    return;
  }
  // ranks [rank0, rank1) of the image's top order
  const int rank0 = g * per + idx * ipw;
  int rank1 = rank0 + ipw;
  const int share_end = ((g + 1) * per < n) ? (g + 1) * per : n;
  if (rank1 > share_end) rank1 = share_end;
  if (rank0 >= rank1) return;
  const int np = (rank1 - rank0) * cw;  // output pixels of this workgroup (<= 256)
  const int CV = C >> 2;
  if (tid < np) {
    const int k = tid / cw, x = tid - k * cw;
    const int item = order[(long long)b * n + rank0 + k];
    const int r = item / ch, y = item - r * ch;
    const long long box = (long long)b * R + r;
    const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + box * 4);
    PixParam prm = pix_param(y * cw + x, H, W, (unsigned)C * 4u, ch, cw, bx.x, bx.y, bx.z, bx.w, true);
    prm.pad = (int)(((box * ch + y) * cw + x) * CV);  // < 2^31: host-checked
    pp[tid] = prm;
  }
  __syncthreads();
  const char *__restrict__ base = reinterpret_cast<const char *>(image + (long long)b * H * W * C);
  f32x4 *__restrict__ out4 = reinterpret_cast<f32x4 *>(out);
  const int total = np * CV;
  int p = tid / CV;
  int c = tid - p * CV;
  const int dp = 256 / CV, dc = 256 - dp * CV;
  for (int e0 = tid; e0 < total; e0 += 256 * UNROLL) {
    f32x4 v[UNROLL];
    unsigned at[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool in = e0 + u * 256 < total;
      const PixParam prm = pp[in ? p : 0];
      v[u] = bilerp<f32x4>(base, prm, (unsigned)(in ? c : 0) * 16u, extrap);
      at[u] = (unsigned)prm.pad + (unsigned)c;
      c += dc;
      p += dp;
      if (c >= CV) {
        c -= CV;
        ++p;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (e0 + u * 256 < total) __builtin_nontemporal_store(v[u], out4 + at[u]);
  }
}

// ---- LDS-staged feature rows ------------------------------------------------------------------------------------------
// With feature maps this large the RoIs span many source pixels, no two output pixels of a crop share a tap, and the tap
// gather -- four 16-byte loads per 16 bytes written, every one of them an L1 miss -- runs into the L2's bandwidth: at
// config 5 the taps alone take as long as the stores alone (110 us each, 165-175 together).  But the 300 RoIs of an image
// overlap: every source pixel is sampled about 240 times.  So the roles are turned round.  A workgroup owns a STRIP of
// consecutive source rows of one image and one part of the channels and walks it top to bottom with two row buffers in LDS
// (row r lives in buffer r & 1): for the pair (t, t + 1) it produces every (RoI, output row) item of the image whose top
// row is t -- the items come sorted by top from crop_rows_order_kernel, their per-pixel parameters precomputed in that
// order -- with the taps read from LDS, then loads row t + 2 over row t.  Every source row travels from memory to a CU
// once per channel part (plus once more at a strip boundary): 0.3 bytes through the L2 per byte written instead of 4.
// The strips of an image hold equal numbers of items (not rows), so every workgroup of the launch has the same work.
// The channels are cut into Q parts so that two rows fit the LDS: an item's output is cw runs of C / Q floats.
// Rows go from global memory straight into LDS (global_load_lds: no registers; the LDS destination of such a load is
// wave-uniform base + lane * 16, which is rows[j] for j = the wave's first j + lane).
// LDS: 2 * W * CQ float4 of rows | ic * cw PixParam (off_* are float4 indices into the row buffers here).
template <int NT>
__global__ __launch_bounds__(NT) void crop_resize_rowstage_kernel(const float *__restrict__ image, int B, int H, int W, int C,
                                                                  const StageRec *__restrict__ rec,
                                                                  const int *__restrict__ rowstart,
                                                                  const int *__restrict__ strips, int R, int ch, int cw,
                                                                  float extrap, float *__restrict__ out, int Q, int NS, int ic) {
  extern __shared__ __attribute__((aligned(16))) f32x4 rows[];
  const int tid = threadIdx.x;
  // NS > 0: strip k of the image's NS item-balanced strips; NS == 0: one source-row pair per workgroup (k = its top row)
  const int per_image = NS > 0 ? NS : H + 1;
  const int q = blockIdx.x % Q;
  const int k = (blockIdx.x / Q) % per_image;
  const int b = blockIdx.x / (Q * per_image);
  if (b >= B) return;
  const int n = R * ch;
  const int t_begin = NS > 0 ? strips[(long long)b * 16 + k] : k;            // rows [t_begin, t_end), t_end <= H + 1
  const int t_end = NS > 0 ? strips[(long long)b * 16 + k + 1] : k + 1;
  const int *rs = rowstart + (long long)b * (H + 2);
  const bool stamped = (b == B / 2) && (k == per_image / 2) && (q == 0);  // (diagnostic builds: one mid-image workgroup's phases)
  (void)stamped;
  const int CV = C >> 2, CQ = CV / Q;
  const int rowlen = W * CQ;  // float4 of one staged row
  PixParam *pp = reinterpret_cast<PixParam *>(rows + 2 * rowlen);
  f32x4 *__restrict__ out4 = reinterpret_cast<f32x4 *>(out);
  const int dp = NT / CQ, dcc = NT - dp * CQ;
  const int lcol = tid / CQ, lc = tid - lcol * CQ;      // this thread's first (column, channel) of a row load
  const int dcol = NT / CQ, dlc = NT - dcol * CQ;
  // Barriers order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also wait for vmcnt(0), i.e.
  // for the acknowledgement of every output store issued so far, once per chunk.  For the same reason a chunk's records are
  // requested before the previous chunk's stores are issued (vector-memory operations retire in order).
  auto lds_barrier = []() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto load_row = [&](int r) {  // row r of this image, channel part q -> buffer r & 1
    const f32x4 *src = reinterpret_cast<const f32x4 *>(image + ((long long)b * H + r) * W * C) + q * CQ;
    f32x4 *dst = rows + (r & 1) * rowlen;
    int col = lcol, c = lc;
    for (int j = tid; j < rowlen; j += NT) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((long long)col * CV + c)),
                                       (__attribute__((address_space(3))) void *)(dst + (j - (tid & 63))), 16, 0, 0);
      c += dlc;
      col += dcol;
      if (c >= CQ) {
        c -= CQ;
        ++col;
      }
    }
  };
  int held0 = -1, held1 = -1;  // the rows in the two buffers (workgroup-uniform)
  for (int t = t_begin; t < t_end; ++t) {
    const int r0 = rs[t], r1 = rs[t + 1];
    if (r0 >= r1) continue;  // nobody samples this row pair
    FRCNN_STAMP_IF(stamped, 60);
    const StageRec *recs = rec + ((long long)b * n + r0) * cw;  // this row pair's pixels, contiguous
    const int npix = (r1 - r0) * cw;
    // the first chunk's records are requested BEFORE the rows, so that all loads of this step are one round trip
    StageRec next = {};
    {
      const int np0 = (npix < ic * cw) ? npix : ic * cw;
      if (tid < np0) next = recs[tid];
    }
    bool loaded = false;
    if (t < H) {  // (t == H: the rows without taps)
      const int need0 = (t & 1) ? t + 1 : t, need1 = (t & 1) ? t : t + 1;  // the rows buffers 0 / 1 must hold
      if (need0 < H && held0 != need0) {
        load_row(need0);
        held0 = need0;
        loaded = true;
      }
      if (need1 < H && held1 != need1) {
        load_row(need1);
        held1 = need1;
        loaded = true;
      }
    }
    const unsigned topbase = (unsigned)((t & 1) * rowlen), botbase = (unsigned)(((t + 1) & 1) * rowlen);
    for (int p0 = 0; p0 < npix; p0 += ic * cw) {
      const int np = (npix - p0 < ic * cw) ? (npix - p0) : ic * cw;
      if (p0 > 0) lds_barrier();  // the previous chunk's parameters are no longer read
      if (tid < np) {
        const StageRec rc = next;
        PixParam prm;
        const unsigned left = (rc.bits & 1023u) * (unsigned)CQ, right = ((rc.bits >> 10) & 1023u) * (unsigned)CQ;
        const unsigned bb = ((rc.bits >> 20) & 1u) ? topbase : botbase;  // bot == top when in_y is integral
        prm.off_tl = topbase + left;
        prm.off_tr = topbase + right;
        prm.off_bl = bb + left;
        prm.off_br = bb + right;
        prm.lx = rc.lx;
        prm.ly = rc.ly;
        prm.valid = (int)((rc.bits >> 21) & 1u);
        prm.pad = rc.dst + q * CQ;
        pp[tid] = prm;
      }
      if (p0 == 0 && loaded) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged rows have landed in LDS
      lds_barrier();
      FRCNN_STAMP_IF(stamped, 62);
      {
        const int pn = p0 + ic * cw + tid;
        if (pn < npix) next = recs[pn];
      }
      const int total = np * CQ;
      int p = tid / CQ;
      int c = tid - p * CQ;
      for (int e0 = tid; e0 < total; e0 += NT * 4) {
        f32x4 v[4];
        unsigned at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool in = e0 + u * NT < total;
          const PixParam prm = pp[in ? p : 0];
          const unsigned cc = in ? (unsigned)c : 0u;
          // the three lerps of bilerp(), one rounding per operation, taps from LDS
          const f32x4 tl = rows[prm.off_tl + cc], tr = rows[prm.off_tr + cc], bl = rows[prm.off_bl + cc], br = rows[prm.off_br + cc];
          f32x4 tt = vsub(tr, tl);
          tt = tt * prm.lx;
          const f32x4 tp = tl + tt;
          tt = vsub(br, bl);
          tt = tt * prm.lx;
          const f32x4 bt = bl + tt;
          tt = vsub(bt, tp);
          tt = tt * prm.ly;
          f32x4 val = tp + tt;
          if (!prm.valid) val = (f32x4)(extrap);
          v[u] = val;
          at[u] = (unsigned)prm.pad + cc;
          c += dcc;
          p += dp;
          if (c >= CQ) {
            c -= CQ;
            ++p;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e0 + u * NT < total) __builtin_nontemporal_store(v[u], out4 + at[u]);
      }
      FRCNN_STAMP_IF(stamped, 63);
    }
    lds_barrier();  // every wave is done with this pair's rows and parameters before the next step overwrites them
  }
}

// ---- the same, streamed: a strip per workgroup, the loads on their own waves -------------------------------------------------
// crop_resize_rowstage_kernel stages a row pair, produces its items and ends: while its rows travel (8-9 us for 145 KB: the
// CU's outstanding-miss budget against the fabric's latency) the CU stores nothing, and on gfx9 a wave that has issued
// stores cannot wait for a later load without waiting for the stores' acknowledgements too (one in-order vmcnt).  Here a
// workgroup walks one of the image's item-balanced STRIPS of rows top to bottom with FOUR row buffers (row r in buffer
// r % 4) and two record buffers, and the waves are specialised:
//   waves 15 / 14 load the even / odd rows (global_load_lds; row t + 3 is requested when the pair (t, t + 1) starts and
//                  awaited when the pair (t + 1, t + 2) ends: two pairs of slack, and a loader never has two rows in flight,
//                  so its wait is a plain vmcnt(0)),
//   wave 13        loads records (global_load_lds; the next step's chunk of per-pixel records),
//   waves 0-12     compute (taps and records from LDS, nontemporal stores; they never wait on vmcnt).
// One s_barrier per step (a chunk of <= ic items of one row pair); every wave runs the same step sequence.  Every source row
// travels to the CU once per channel part (three more at a strip boundary).
// LDS: 4 * W * CQ float4 of rows | 2 * ic * cw StageRec | H + 2 ints (the image's rowstart).
__global__ __launch_bounds__(1024) void crop_resize_rowstream_kernel(const float *__restrict__ image, int B, int H, int W, int C,
                                                                     const StageRec *__restrict__ rec,
                                                                     const int *__restrict__ rowstart,
                                                                     const int *__restrict__ strips, int R, int ch, int cw,
                                                                     float extrap, float *__restrict__ out, int Q, int NS, int ic) {
  extern __shared__ __attribute__((aligned(16))) f32x4 rows[];
  constexpr int kComputeWaves = 13, NTC = kComputeWaves * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x % Q;
  const int k = (blockIdx.x / Q) % NS;
  const int b = blockIdx.x / (Q * NS);
  if (b >= B) return;
  const int n = R * ch;
  const int CV = C >> 2, CQ = CV / Q;
  const int rowlen = W * CQ;
  const int chunk = ic * cw;  // pixels per step
  StageRec *recbuf = reinterpret_cast<StageRec *>(rows + 4 * rowlen);  // [2][chunk]
  int *rs = reinterpret_cast<int *>(recbuf + 2 * chunk);               // [H + 2]
  const int t_begin = strips[(long long)b * 16 + k], t_end = strips[(long long)b * 16 + k + 1];  // rows [t_begin, t_end), t_end <= H + 1
  if (t_begin >= t_end) return;
  const StageRec *recs = rec + (long long)b * n * cw;  // the image's records, in rank order
  f32x4 *__restrict__ out4 = reinterpret_cast<f32x4 *>(out);
  const int row_wave = 15 - 0;  // even rows; odd rows: wave 14

  auto load_row = [&](int r) {  // row r of this image, channel part q -> buffer r % 4 (64 lanes, 16 bytes each per trip)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(image + ((long long)b * H + r) * W * C) + q * CQ;
    f32x4 *dst = rows + (r & 3) * rowlen;
    for (int j0 = 0; j0 < rowlen; j0 += 64) {
      const int j = j0 + lane;
      if (j < rowlen) {
        const int col = j / CQ, c = j - col * CQ;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((long long)col * CV + c)),
                                         (__attribute__((address_space(3))) void *)(dst + j0), 16, 0, 0);
      }
    }
  };
  auto is_loader_of = [&](int r) { return wave == row_wave - (r & 1); };
  auto load_recs = [&](int first_pixel, int count, int buf) {  // wave 13: `count` records from first_pixel -> recbuf[buf]
    StageRec *dst = recbuf + buf * chunk;
    for (int j0 = 0; j0 < count; j0 += 64) {
      if (j0 + lane < count)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(recs + first_pixel + j0 + lane),
                                         (__attribute__((address_space(3))) void *)(dst + j0), 16, 0, 0);
    }
  };
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // prologue: the image's rowstart (every wave needs it to follow the step sequence), the first rows, the first chunk
  for (int i = tid; i < H + 2; i += 1024) rs[i] = rowstart[(long long)b * (H + 2) + i];
  if (wave >= 14) {
    for (int r = t_begin; r < t_begin + 3; ++r)  // rows t, t + 1 of the first pair and the bot row of the second
      if (r < H && (r == t_begin || r <= t_end) && is_loader_of(r)) load_row(r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();  // rs (nothing has been stored yet: the full barrier costs nothing here)
  if (wave == 13) {
    const int r0 = rs[t_begin], r1 = rs[t_begin + 1];
    const int np = ((r1 - r0) * cw < chunk) ? (r1 - r0) * cw : chunk;
    load_recs(r0 * cw, np, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  step_barrier();

  int step = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const int r0 = rs[t], r1 = rs[t + 1];
    const int npix = (r1 - r0) * cw;
    const int nsteps = npix > 0 ? (npix + chunk - 1) / chunk : 1;  // a row nobody samples is one empty step
    const unsigned topbase = (unsigned)((t & 3) * rowlen), botbase = (unsigned)(((t + 1) & 3) * rowlen);
    for (int sidx = 0; sidx < nsteps; ++sidx, ++step) {
      const int p0 = sidx * chunk;
      const int np = (npix - p0 < chunk) ? (npix - p0) : chunk;  // (<= 0 for the empty step)
      const bool last = sidx + 1 == nsteps;
      FRCNN_STAMP_IF((b == B / 2) && (k == NS / 2) && (q == 0), 70 + (np > 0 ? (r1 - r0 > 99 ? 9 : (r1 - r0) / 11) : 0));
      if (wave >= 14) {
        // row t + 3 (the bot row of the pair after next) into the buffer of row t - 1, which nobody reads any more
        if (sidx == 0 && t + 3 < H && t + 2 < t_end && is_loader_of(t + 3)) load_row(t + 3);
        // row t + 2, requested a pair ago by this wave, must be there when the next pair starts
        if (last && is_loader_of(t + 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (wave == 13) {
        // the next step's records (this pair's next chunk, or the next pair's first) into the other buffer
        int nf = -1, nc = 0;
        if (!last) {
          nf = r0 * cw + p0 + chunk;
          nc = (npix - p0 - chunk < chunk) ? (npix - p0 - chunk) : chunk;
        } else if (t + 1 < t_end) {
          const int q0 = rs[t + 1], q1 = rs[t + 2];
          nf = q0 * cw;
          nc = ((q1 - q0) * cw < chunk) ? (q1 - q0) * cw : chunk;
        }
        if (nc > 0) load_recs(nf, nc, (step + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (np > 0) {
        const StageRec *rb = recbuf + (step & 1) * chunk;
        const int total = np * CQ;
        int p = tid / CQ;
        int c = tid - p * CQ;
        const int dp = NTC / CQ, dcc = NTC - dp * CQ;
        for (int e0 = tid; e0 < total; e0 += NTC * 4) {
          f32x4 v[4];
          unsigned at[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool in = e0 + u * NTC < total;
            const StageRec rc = rb[in ? p : 0];
            const unsigned cc = in ? (unsigned)c : 0u;
            const unsigned left = (rc.bits & 1023u) * (unsigned)CQ, right = ((rc.bits >> 10) & 1023u) * (unsigned)CQ;
            const unsigned bb = ((rc.bits >> 20) & 1u) ? topbase : botbase;  // bot == top when in_y is integral
            // the three lerps of bilerp(), one rounding per operation, taps from LDS
            const f32x4 tl = rows[topbase + left + cc], tr = rows[topbase + right + cc], bl = rows[bb + left + cc],
                        br = rows[bb + right + cc];
            f32x4 tt = vsub(tr, tl);
            tt = tt * rc.lx;
            const f32x4 tp = tl + tt;
            tt = vsub(br, bl);
            tt = tt * rc.lx;
            const f32x4 bt = bl + tt;
            tt = vsub(bt, tp);
            tt = tt * rc.ly;
            f32x4 val = tp + tt;
            if (!((rc.bits >> 21) & 1u)) val = (f32x4)(extrap);
            v[u] = val;
            at[u] = (unsigned)(rc.dst + q * CQ) + cc;
            c += dcc;
            p += dp;
            if (c >= CQ) {
              c -= CQ;
              ++p;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (e0 + u * NTC < total) __builtin_nontemporal_store(v[u], out4 + at[u]);
        }
      }
      FRCNN_STAMP_IF((b == B / 2) && (k == NS / 2) && (q == 0), 80);
      step_barrier();
    }
  }
}

#ifndef FRCNN_CROP_WAVES_PER_EU
#define FRCNN_CROP_WAVES_PER_EU 4
#endif
// ROWS = false: one workgroup per box (all ch*cw output pixels); ROWS = true: one workgroup per (box, output row), i.e.
// cw pixels = 16 KB at 7 x 576 -- a single trip of 4 outputs per lane, seven times as many workgroups and a tighter
// write front.  `units` = number of workgroups' worth of work (n or n*ch).
// Up to kCropGroups independent (image, boxes, out) triples of one shape handled by ONE launch (frcnn_crop_and_resize_multi:
// the stage pipeline crops two or more batches per launch, one launch tail and one inter-launch gap for all of them).
static const int kCropGroups = 4;
struct CropGroups {
  const float *image[kCropGroups];
  const float *boxes[kCropGroups];
  float *out[kCropGroups];
  int n_per_group;  // boxes per triple; 0 = a single triple (no group arithmetic)
  // warm-up workgroups (PF kernels only; RoIPooling layout: box_ind == NULL, every image has boxes_per_image boxes)
  int pf_blocks;    // workgroups that stream one image into the XCD's L2
  int pf_lead;      // ... dispatched this many boxes before the image's first box
};

template <int VEC, bool ROWS, int WPE, bool PF = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void crop_resize_kernel(
    CropGroups gr, int B, int H, int W, int C,
    const int *__restrict__ box_ind, int units, int boxes_per_image, int ch, int cw, float extrap) {
  typedef typename VecT<VEC>::type vec_t;
  __shared__ PixParam pp[256];

  // XCD-aware remap: blocks b, b+8, b+16, ... share an XCD (and its L2); give each of the 8
  // groups a contiguous range of boxes so that it works through one image's feature map at a time.
  const int tid = threadIdx.x;
  const int grp = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int q = units >> 3, r = units & 7;
  const int start = (grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q;
  int unit = start + slot;
  if (PF) {
    const int npg = gr.n_per_group > 0 ? gr.n_per_group : units;
    int pf_box = 0;
    const int role = crop_slot_role(slot, start, q + (grp < r ? 1 : 0), npg, boxes_per_image, gr.pf_blocks, gr.pf_lead, &pf_box);
    if (role <= -1000000) return;
    if (role < 0) {
      const int piece = -1 - role;
      const int g = pf_box / npg;
      const int img_i = (pf_box - g * npg) / boxes_per_image;
      const float *image = gr.image[0];
#pragma unroll
      for (int i = 1; i < kCropGroups; ++i)
        if (g == i) image = gr.image[i];
      if (img_i >= B) return;
      const long long image_bytes = (long long)H * W * C * 4;
      const char *src = reinterpret_cast<const char *>(image) + (long long)img_i * image_bytes;
      f32x4 acc = (f32x4)(0.0f);
#pragma unroll 8
      for (long long o = ((long long)piece * 256 + tid) * 16; o + 16 <= image_bytes; o += (long long)gr.pf_blocks * 256 * 16) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + o);
        acc.x += v.x;  // (any use keeps the load; nothing is stored)
      }
      asm volatile("" ::"v"(acc));
      return;
    }
    unit = role;
  }
  int box = ROWS ? unit / ch : unit;
  const int pbeg = ROWS ? (unit - box * ch) * cw : 0;  // first output pixel of this workgroup
  // which triple (workgroup-uniform; selects instead of indexing the kernel-argument arrays dynamically)
  const float *__restrict__ image = gr.image[0];
  const float *__restrict__ boxes = gr.boxes[0];
  float *__restrict__ out = gr.out[0];
  if (gr.n_per_group > 0) {
    const int g = box / gr.n_per_group;
    box -= g * gr.n_per_group;
#pragma unroll
    for (int i = 1; i < kCropGroups; ++i) {
      if (g == i) {
        image = gr.image[i];
        boxes = gr.boxes[i];
        out = gr.out[i];
      }
    }
  }

  const int bimg = box_ind ? box_ind[box] : (box / boxes_per_image);
  const bool img_ok = (bimg >= 0) && (bimg < B);
  const float y1 = boxes[(long long)box * 4 + 0];
  const float x1 = boxes[(long long)box * 4 + 1];
  const float y2 = boxes[(long long)box * 4 + 2];
  const float x2 = boxes[(long long)box * 4 + 3];

  const int CV = C / VEC;
  const int P = ROWS ? cw : ch * cw;  // output pixels of this workgroup
  const vec_t *__restrict__ img =
      reinterpret_cast<const vec_t *>(image + (long long)(img_ok ? bimg : 0) * H * W * C);

  for (int p0 = 0; p0 < P; p0 += 256) {
    const int np = (P - p0 < 256) ? (P - p0) : 256;
    if (p0 > 0) __syncthreads();
    if (tid < np) pp[tid] = pix_param(pbeg + p0 + tid, H, W, (unsigned)C * 4u, ch, cw, y1, x1, y2, x2, img_ok);
    __syncthreads();

    const int total = np * CV;
    const char *__restrict__ base = reinterpret_cast<const char *>(img);
    char *__restrict__ dstb = reinterpret_cast<char *>(out + ((long long)box * ch * cw + pbeg + p0) * C);
    int p = tid / CV;
    int c = tid - p * CV;
    const int dp = 256 / CV, dc = 256 - dp * CV;
#ifndef FRCNN_CROP_UNROLL
#define FRCNN_CROP_UNROLL 4
#endif
    // FRCNN_CROP_UNROLL outputs per lane per trip, all their taps issued before the first lerp.  Unrolled by hand (the
    // packed-subtract asm is convergent, which rules out the compiler's runtime unroll): lanes past the end recompute
    // the last element (in-bounds loads) and skip the store.
    for (int e0 = tid; e0 < total; e0 += 256 * FRCNN_CROP_UNROLL) {
      vec_t v[FRCNN_CROP_UNROLL];
#pragma unroll
      for (int u = 0; u < FRCNN_CROP_UNROLL; ++u) {
        const bool in = e0 + u * 256 < total;
        const PixParam prm = pp[in ? p : 0];
        v[u] = bilerp<vec_t>(base, prm, (unsigned)(in ? c : 0) * (unsigned)sizeof(vec_t), extrap);
        c += dc;
        p += dp;
        if (c >= CV) {
          c -= CV;
          ++p;
        }
      }
#pragma unroll
      for (int u = 0; u < FRCNN_CROP_UNROLL; ++u) {
        const int e = e0 + u * 256;
        if (e < total) {
          vec_t *d = reinterpret_cast<vec_t *>(dstb + (size_t)((unsigned)e * (unsigned)sizeof(vec_t)));
#ifdef FRCNN_CROP_PLAIN_STORE
          *d = v[u];
#else
          __builtin_nontemporal_store(v[u], d);
#endif
        }
      }
    }
  }
}

// Whether a launch over `units` boxes carries warm-up workgroups, and its grid then (0 = no).  FRCNN_CROP_PREFETCH = n
// forces n warm-up workgroups per image (0 = none); by default 192 of them (measured 24 / 48 / 96 / 192 / 288 / 576 ->
// 103 / 116 / 121 / 123 / 123 / 119 k images/s in the stage pipeline, cold inputs) when the batch's feature maps exceed one XCD's
// L2 (a single small image is sampled by every XCD at once and needs none).
static unsigned crop_warmup_grid(CropGroups &gr, int units, int boxes_per_image, int B, long long image_bytes) {
  int pf = knob(kKnobCropPrefetch, -1);
  if (pf < 0) pf = (image_bytes * B > 4ll * 1024 * 1024 && image_bytes >= 256 * 1024) ? 192 : 0;
  int lead = 256;  // twice the boxes resident per XCD (32 CUs x 4 workgroups): measured 64 / 128 / 256 -> 120.7 / 121.4 / 124.3 k images/s
  if (pf >= 10000) {  // diagnostic: lead * 10000 + workgroups
    lead = pf / 10000;
    pf %= 10000;
  }
  if (pf <= 0 || units < 8 || boxes_per_image <= 0) return 0;
  if (pf > 1024) pf = 1024;
  gr.pf_blocks = pf;
  gr.pf_lead = lead;
  const int qmax = (units + 7) / 8;
  long long images_max = (qmax + boxes_per_image - 2) / boxes_per_image + 1;  // images an XCD's range can touch
  // grouped launch whose triples do not end on an image boundary: crop_slot_role restarts its image walk at every triple,
  // so a range that spans g triple boundaries can hold g more (partial) images than the bound above
  if (gr.n_per_group > 0 && gr.n_per_group % boxes_per_image != 0) images_max += kCropGroups - 1;
  const long long slots = qmax + images_max * pf;
  if (slots * 8 >= (1ll << 31)) return 0;
  return (unsigned)(slots * 8);
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_crop_and_resize(const float *image, int B, int H, int W, int C,
                                     const float *boxes, const int32_t *box_ind, int n,
                                     int boxes_per_image, int crop_h, int crop_w,
                                     float extrapolation_value, float *out,
                                     frcnn_stream_t stream) {
  FRCNN_REQUIRE(image && boxes && out, "crop_and_resize: null pointer");
  FRCNN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "crop_and_resize: image shape (%d,%d,%d,%d) must be positive", B, H, W, C);
  FRCNN_REQUIRE(crop_h > 0 && crop_w > 0, "crop_and_resize: crop size (%d,%d) must be positive", crop_h, crop_w);
  FRCNN_REQUIRE(n >= 0, "crop_and_resize: negative box count");
  FRCNN_REQUIRE(box_ind || boxes_per_image > 0, "crop_and_resize: box_ind is NULL and boxes_per_image <= 0");
  FRCNN_REQUIRE((long long)H * W * C < (1ll << 30), "crop_and_resize: one image must hold fewer than 2^30 floats");
  FRCNN_REQUIRE(C < (1 << 20), "crop_and_resize: depth must be below 2^20");
  FRCNN_REQUIRE((long long)crop_h * crop_w < (1ll << 24), "crop_and_resize: crop too large");
  if (n == 0) return FRCNN_OK;
  const bool vec4 = (C % 4 == 0) && aligned_to(image, 16) && aligned_to(out, 16);
  dim3 grid((unsigned)n), block(256);
  // The box kernel keeps the one or two images an XCD is working on in its 4 MiB L2.  An image that does not fit there
  // (e.g. 63x63x576 = 9.1 MB) is read through the channel-slice kernel instead, each XCD holding one eighth of it
  // (measured: 179 against 232 us at config 5; 23 % slower than the box kernel at config 2, whose images fit).
  // FRCNN_CROP_SLICE=0|1 forces the choice (diagnostic).
  const int force_slice = knob(kKnobCropSlice, -1);
  bool use_slice = false;
  if (vec4 && aligned_to(boxes, 16) && (long long)n * 8 < (1ll << 31) && (long long)crop_h * crop_w * C * 4 < (1ll << 32))
    use_slice = force_slice >= 0 ? force_slice != 0 : (double)H * W * C * 4 > 4.0 * 1024 * 1024;
  if (use_slice) {
    const int CV = C / 4, SW = (CV + 7) / 8;
    hipLaunchKernelGGL(crop_resize_slice_kernel, dim3((unsigned)n * 8u), block, 0, (hipStream_t)stream, image, B, H, W, C,
                       boxes, box_ind, (const int *)nullptr, n, boxes_per_image, crop_h, crop_w, extrapolation_value, out, 8, SW, 0, 0, 0);
    return check_launch("crop_and_resize");
  }
  // FRCNN_CROP_ROWS=0|1 forces the (box) / (box, output row) decomposition (diagnostic).
  const bool rows_fit = (long long)n * crop_h < (1ll << 31);
  const bool use_rows = vec4 && rows_fit && (knob(kKnobCropRows, 0) != 0);
  CropGroups gr = {};
  gr.image[0] = image;
  gr.boxes[0] = boxes;
  gr.out[0] = out;
  gr.n_per_group = 0;
  if (use_rows) {
    hipLaunchKernelGGL((crop_resize_kernel<4, true, 6>), dim3((unsigned)(n * crop_h)), block, 0, (hipStream_t)stream, gr,
                       B, H, W, C, box_ind, n * crop_h, boxes_per_image, crop_h, crop_w, extrapolation_value);
  } else if (vec4) {
    unsigned pf_grid = 0;
    if (!box_ind && (pf_grid = crop_warmup_grid(gr, n, boxes_per_image, B, (long long)H * W * C * 4)) != 0)
      hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU, true>), dim3(pf_grid), block, 0, (hipStream_t)stream,
                         gr, B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
    else
      hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU>), grid, block, 0, (hipStream_t)stream, gr,
                         B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
  } else {
    hipLaunchKernelGGL((crop_resize_kernel<1, false, FRCNN_CROP_WAVES_PER_EU>), grid, block, 0, (hipStream_t)stream, gr,
                       B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
  }
  return check_launch("crop_and_resize");
}

// order of the (box, output row) items (source-row ownership, crop heights up to kRowbandMaxCh) | per-image row bounds;
// the RoI order of the sorted channel-slice sweep shares the front of the same area
static const int kRowbandMaxCh = 16;
static const int kStageMaxPixels = 64;  // crop_h * crop_w up to which the LDS-staged form keeps per-pixel records
extern "C" size_t frcnn_roi_pooling_workspace_bytes(int B, int R) {
  if (B <= 0 || R <= 0) return 256;
  return round_up((size_t)B * R * kRowbandMaxCh * sizeof(int), 256) + round_up((size_t)B * 16 * sizeof(int), 256) +
         round_up((size_t)B * 1032 * sizeof(int), 256) +                                   // order | bounds | rowstart (H <= 1023)
         round_up((size_t)B * R * kStageMaxPixels * sizeof(StageRec), 256) + 256;          // | per-pixel records
}

extern "C" int frcnn_roi_pooling(const float *feature_map, int B, int H, int W, int C, const float *rois, int R,
                                 int crop_h, int crop_w, float *out, void *workspace, size_t workspace_bytes,
                                 frcnn_stream_t stream) {
  FRCNN_REQUIRE(feature_map && rois && out, "roi_pooling: null pointer");
  FRCNN_REQUIRE(B >= 0 && R >= 0, "roi_pooling: negative shape");
  if (B == 0 || R == 0) return FRCNN_OK;
  FRCNN_REQUIRE((long long)B * R < (1ll << 28), "roi_pooling: too many RoIs");
  const int n = B * R;
  const long long image_bytes = (long long)H * W * C * 4;
  // Feature maps beyond an XCD's 4 MiB L2 (config 5: 9.1 MB) take the channel-slice kernel: channels cut over S XCDs, S the
  // smallest of 2 / 4 / 8 that brings an XCD's share of an image under 2.5 MB, with warm-up workgroups one image ahead.
  // Measured at config 5 (16 images, COLD inputs, us per launch; profiles/r02_*cfg5*): S = 4 with warm-up 190-194,
  // S = 8 without 217-230 (round 1's schedule), S = 4 without 222, S = 2 214, box kernel 240; warm inputs: S = 4 151
  // without warm-up, 172-175 with.  Sweeping every image's RoIs top to bottom (FRCNN_CROP_SORT=1, roi_order_kernel) is
  // SLOWER (S = 8 279, S = 4 253-266, S = 2 246) although its HBM traffic is ideal (FETCH 160 MB for 146 MB of feature
  // maps, WRITE 537 MB).  FRCNN_CROP_SPLIT = S forces S (0 = never this schedule).
  // A launch of at most 1024 boxes (batch 1-3 at 300 RoIs) does not fill the chip with one workgroup per box -- four fit on a
  // CU -- and is bound by the time ONE workgroup takes for its 113 KB crop: the same split, four workgroups per box, no
  // warm-up, runs the crop of config 1 in 12.6 instead of 16.6 us (B = 2: -10 %, B = 3: -12 %, B = 4: -3 %, B = 8: none).
  // Source-row ownership (crop_resize_rowband_kernel): the default beyond 4 MiB per image when the caller brings the
  // workspace; FRCNN_CROP_ROWBAND = 0 | 1 forces the choice for any size (1 still needs the workspace and the shape limits).
  {
    const int rb = knob(kKnobCropRowband, -1);
    const bool want = rb >= 0 ? rb != 0 : (image_bytes > 4ll * 1024 * 1024 && !knob_set(kKnobCropSplit) && !knob_set(kKnobCropSlice));
    const long long nb = (long long)R * crop_h;
    const bool fits = H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0 && (C % 4 == 0) && aligned_to(feature_map, 16) &&
                      aligned_to(out, 16) && aligned_to(rois, 16) && image_bytes < (1ll << 32) && H <= 1023 &&
                      crop_h <= kRowbandMaxCh && nb <= 24000 && (long long)crop_h * crop_w < (1ll << 24) &&
                      crop_w <= 256 && (long long)B * R * crop_h * crop_w * (C / 4) < (1ll << 31) && workspace &&
                      aligned_to(workspace, 256) &&
                      workspace_bytes >= frcnn_roi_pooling_workspace_bytes(B, R);
    if (want && fits) {
      int *order = reinterpret_cast<int *>(workspace);
      int *bounds = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + round_up((size_t)B * R * kRowbandMaxCh * sizeof(int), 256));
      const size_t lds = (size_t)(1024 + 8 + 16) * 4 + (size_t)(((nb + 1) & ~1ll) * 2) * 2;
      auto okern = crop_rows_order_kernel;
      if (lds > 48 * 1024) {
        int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(okern), lds, "roi_pooling: hipFuncSetAttribute");
        if (rc) return rc;
      }
      int *rowstart = bounds + round_up((size_t)B * 16 * sizeof(int), 256) / sizeof(int);
      StageRec *rec = reinterpret_cast<StageRec *>(rowstart + round_up((size_t)B * 1032 * sizeof(int), 256) / sizeof(int));
      // LDS-staged rows (crop_resize_rowstage_kernel): Q = the fewest channel parts (1 / 2 / 4 / 8) with which the two rows and
      // the parameters fit the CU's LDS.  FRCNN_CROP_ROWBAND = 2 forces it, 1 forces the tap-gather form below.
      const int NT = 1024;
      int ic = 64;
      if (ic * crop_w > NT) ic = NT / crop_w;
      const int CV = C / 4;
      int Q = 0;
      size_t stage_lds = 0;
      const int q_forced = rb >= 2 ? knob(kKnobCropSplit, 0) : 0;  // (diagnostic: FRCNN_CROP_SPLIT = Q with FRCNN_CROP_ROWBAND = 2)
      for (int cand = 1; cand <= 8 && ic > 0 && crop_h * crop_w <= kStageMaxPixels && W <= 1023; cand *= 2) {
        if (CV % cand || cand < q_forced) continue;
        const size_t need = (size_t)2 * W * (CV / cand) * 16 + (size_t)ic * crop_w * sizeof(PixParam);
        if (need <= 160 * 1024) {
          Q = cand;
          stage_lds = need;
          break;
        }
      }
      // the streamed form: four row buffers, two record buffers and the image's rowstart
      int Qs = 0;
      size_t stream_lds = 0;
      for (int cand = 1; cand <= 8 && ic > 0 && Q > 0; cand *= 2) {
        if (CV % cand || cand < q_forced) continue;
        const size_t need = (size_t)4 * W * (CV / cand) * 16 + (size_t)2 * ic * crop_w * sizeof(StageRec) + (size_t)(H + 2) * 4 + 16;
        if (need <= 160 * 1024) {
          Qs = cand;
          stream_lds = need;
          break;
        }
      }
      const bool stage = Q > 0 && (rb >= 0 ? (rb == 2 || (rb == 3 && Qs > 0)) : true);
      // strips per image: FRCNN_CROP_PREFETCH = n (1..15) makes a workgroup walk one of n item-balanced strips of rows with a
      // sliding two-row window instead (diagnostic)
      int NS = 0;  // one row pair per workgroup by default (measured below)
      if (rb == 3 && Qs > 0) NS = (256 + B * Qs - 1) / (B * Qs);  // the streamed form: one strip per CU
      if (knob(kKnobCropPrefetch, -1) > 0) NS = knob(kKnobCropPrefetch, -1);
      if (NS > 15) NS = 15;
      if (rb == 3 && NS < 1) NS = 1;
      hipLaunchKernelGGL(okern, dim3((unsigned)B), dim3(1024), lds, (hipStream_t)stream, rois, R, H, crop_h, order, bounds, rowstart,
                         stage ? rec : (StageRec *)nullptr, W, crop_w, CV, NS);
      int rc = check_launch("roi_pooling (row order)");
      if (rc) return rc;
      if (stage && rb == 3) {  // the streamed form (crop_resize_rowstream_kernel)
        auto skern = crop_resize_rowstream_kernel;
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(skern), stream_lds, "roi_pooling: hipFuncSetAttribute");
        if (rc) return rc;
        hipLaunchKernelGGL(skern, dim3((unsigned)(B * Qs * NS)), dim3(1024), stream_lds, (hipStream_t)stream, feature_map, B, H, W, C,
                           rec, rowstart, bounds, R, crop_h, crop_w, 0.0f, out, Qs, NS, ic);
        return check_launch("roi_pooling (streamed rows)");
      }
      if (stage) {
        auto skern = crop_resize_rowstage_kernel<1024>;
        if (stage_lds > 48 * 1024) {
          rc = ensure_dynamic_lds(reinterpret_cast<const void *>(skern), stage_lds, "roi_pooling: hipFuncSetAttribute");
          if (rc) return rc;
        }
        const long long wgs = (long long)B * Q * (NS > 0 ? NS : H + 1);
        FRCNN_REQUIRE(wgs < (1ll << 31), "roi_pooling: too many workgroups");
        hipLaunchKernelGGL(skern, dim3((unsigned)wgs), dim3(NT), stage_lds, (hipStream_t)stream, feature_map, B, H, W, C,
                           rec, rowstart, bounds, R, crop_h, crop_w, 0.0f, out, Q, NS, ic);
        return check_launch("roi_pooling (staged rows)");
      }
      // warm-up workgroups per (XCD, image): FRCNN_CROP_PREFETCH as for the other schedules (0 = none)
      int pf = knob(kKnobCropPrefetch, B > 1 ? 16 : 0), ipw_forced = 0, lead_forced = 0;
      if (pf >= 10000) {  // diagnostic: lead in images * 1000000 + items per workgroup * 10000 + warm-up workgroups
        lead_forced = pf / 1000000;
        ipw_forced = (pf / 10000) % 100;
        pf %= 10000;
      }
      if (pf > 256) pf = 256;
      if (pf < 0) pf = 0;
      // items per workgroup: a crop's worth (the box kernel's granularity), at most what 256 parameter lanes hold
      int ipw = ipw_forced > 0 ? ipw_forced : crop_h;
      if (ipw * crop_w > 256) ipw = 256 / crop_w;
      const long long per = (nb + 7) / 8;
      const long long wpg = (per + ipw - 1) / ipw;
      int lead = lead_forced > 0 ? lead_forced : 1;
      if (lead > B) lead = B;
      const long long slots = (long long)pf * lead + (long long)B * (pf + wpg);
      FRCNN_REQUIRE(slots * 8 < (1ll << 31), "roi_pooling: too many workgroups");
      hipLaunchKernelGGL((crop_resize_rowband_kernel<FRCNN_CROP_WAVES_PER_EU, 4>), dim3((unsigned)(slots * 8)), dim3(256), 0,
                         (hipStream_t)stream, feature_map, B, H, W, C, rois, order, bounds, R, crop_h, crop_w, 0.0f, out, pf, ipw, lead);
      return check_launch("roi_pooling (row bands)");
    }
  }
  int S = knob(kKnobCropSplit, -1);
  const bool small_launch = S < 0 && image_bytes <= 4ll * 1024 * 1024 && n <= 1024;
  if (S < 0)
    S = small_launch ? 4
                     : (image_bytes <= 4ll * 1024 * 1024 ? 0 : (image_bytes <= 5ll * 1024 * 1024 ? 2 : (image_bytes <= 10ll * 1024 * 1024 ? 4 : 8)));
  const bool shape_ok = H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0 && (C % 4 == 0) && aligned_to(feature_map, 16) &&
                        aligned_to(out, 16) && aligned_to(rois, 16) && (long long)crop_h * crop_w * C * 4 < (1ll << 32) &&
                        image_bytes < (1ll << 32) && (long long)crop_h * crop_w < (1ll << 24);
  if ((S == 2 || S == 4 || S == 8) && shape_ok) {
    const int *order = nullptr;
    const bool sort = knob(kKnobCropSort, 0) != 0 && R <= 1024 && workspace && aligned_to(workspace, 256) &&
                      workspace_bytes >= frcnn_roi_pooling_workspace_bytes(B, R);
    if (sort) {
      int *ord = reinterpret_cast<int *>(workspace);
      hipLaunchKernelGGL(roi_order_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, rois, R, ord);
      int rc = check_launch("roi_pooling (RoI order)");
      if (rc) return rc;
      order = ord;
    }
    const int CV = C / 4, SW = (CV + S - 1) / S, nlg = 8 / S;
    long long slots = (n + nlg - 1) / nlg;
    // warm-up workgroups (unsorted sweep only: a sorted sweep has no "first box of the next image" in slot order that
    // the image-by-image warm-up could lead): FRCNN_CROP_PREFETCH as for the box kernel; by default 32 workgroups per (XCD,
    // image), one image (R boxes) ahead: 16 / 32 / 96 workgroups -> 194 / 190 / 190 us, lead 128 / 200 / 300 -> 221 / 199 / 193
    int pf = order ? 0 : knob(kKnobCropPrefetch, small_launch ? 0 : 32), lead = R < 128 ? 128 : (R > 512 ? 512 : R), linear = 0;
    if (pf >= 100000000) {  // diagnostic: + 100000000 = linear (Infinity Cache) warm-up instead of the XCD's own slice
      linear = 1;
      pf -= 100000000;
    }
    if (pf >= 10000) {
      lead = pf / 10000;
      pf %= 10000;
    }
    if (pf > 1024) pf = 1024;
    if (pf > 0) slots += ((slots + R - 2) / R + 1) * pf;
    FRCNN_REQUIRE(slots * 8 < (1ll << 31), "roi_pooling: too many workgroups");
    hipLaunchKernelGGL(crop_resize_slice_kernel, dim3((unsigned)slots * 8u), dim3(256), 0, (hipStream_t)stream, feature_map, B, H,
                       W, C, rois, (const int *)nullptr, order, n, R, crop_h, crop_w, 0.0f, out, S, SW, pf, lead, linear);
    return check_launch("roi_pooling");
  }
  return frcnn_crop_and_resize(feature_map, B, H, W, C, rois, nullptr, n, R, crop_h, crop_w, 0.0f, out, stream);
}

extern "C" int frcnn_crop_and_resize_multi(const float *const *images, const float *const *boxes, float *const *outs,
                                           int groups, int B, int H, int W, int C, int n, int boxes_per_image, int crop_h,
                                           int crop_w, float extrapolation_value, void *workspace, size_t workspace_bytes,
                                           frcnn_stream_t stream) {
  FRCNN_REQUIRE(images && boxes && outs, "crop_and_resize_multi: null pointer table");
  FRCNN_REQUIRE(groups >= 1 && groups <= kCropGroups, "crop_and_resize_multi: groups must be 1..%d (got %d)", kCropGroups, groups);
  FRCNN_REQUIRE(boxes_per_image > 0, "crop_and_resize_multi: boxes_per_image must be positive");
  FRCNN_REQUIRE(n >= 0, "crop_and_resize_multi: negative box count");
  bool one_launch = groups > 1 && n > 0 && (C % 4 == 0) && (long long)n * groups < (1ll << 28) &&
                    !((double)H * W * C * 4 > 4.0 * 1024 * 1024) && !knob_set(kKnobCropSlice) && !knob_set(kKnobCropRows);
  for (int g = 0; g < groups; ++g) {
    FRCNN_REQUIRE(images[g] && boxes[g] && outs[g], "crop_and_resize_multi: null pointer in triple %d", g);
    one_launch = one_launch && aligned_to(images[g], 16) && aligned_to(outs[g], 16);
  }
  if (!one_launch) {  // shapes the grouped launch is not built for: one launch per triple, same results
    for (int g = 0; g < groups; ++g) {
      // (the RoIPooling layout with extrapolation 0 goes through the layer-level entry point: it picks the schedule for
      // feature maps beyond L2)
      const int rc = (extrapolation_value == 0.0f && n == B * boxes_per_image && B > 0)
                         ? frcnn_roi_pooling(images[g], B, H, W, C, boxes[g], boxes_per_image, crop_h, crop_w, outs[g], workspace,
                                             workspace_bytes, stream)  // (launches on one stream: the triples can share the workspace)
                         : frcnn_crop_and_resize(images[g], B, H, W, C, boxes[g], nullptr, n, boxes_per_image, crop_h, crop_w,
                                                 extrapolation_value, outs[g], stream);
      if (rc) return rc;
    }
    return FRCNN_OK;
  }
  FRCNN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "crop_and_resize_multi: image shape (%d,%d,%d,%d) must be positive", B, H, W, C);
  FRCNN_REQUIRE(crop_h > 0 && crop_w > 0, "crop_and_resize_multi: crop size (%d,%d) must be positive", crop_h, crop_w);
  FRCNN_REQUIRE((long long)H * W * C < (1ll << 30), "crop_and_resize_multi: one image must hold fewer than 2^30 floats");
  FRCNN_REQUIRE(C < (1 << 20), "crop_and_resize_multi: depth must be below 2^20");
  FRCNN_REQUIRE((long long)crop_h * crop_w < (1ll << 24), "crop_and_resize_multi: crop too large");
  CropGroups gr = {};
  for (int g = 0; g < kCropGroups; ++g) {
    const int s = g < groups ? g : 0;
    gr.image[g] = images[s];
    gr.boxes[g] = boxes[s];
    gr.out[g] = outs[s];
  }
  gr.n_per_group = n;
  if (const unsigned pf_grid = crop_warmup_grid(gr, n * groups, boxes_per_image, B, (long long)H * W * C * 4))
    hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU, true>), dim3(pf_grid), dim3(256), 0,
                       (hipStream_t)stream, gr, B, H, W, C, (const int *)nullptr, n * groups, boxes_per_image, crop_h, crop_w,
                       extrapolation_value);
  else
    hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU>), dim3((unsigned)(n * groups)), dim3(256), 0,
                       (hipStream_t)stream, gr, B, H, W, C, (const int *)nullptr, n * groups, boxes_per_image, crop_h, crop_w,
                       extrapolation_value);
  return check_launch("crop_and_resize_multi");
}
