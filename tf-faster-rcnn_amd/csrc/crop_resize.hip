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

// ---- LDS-staged feature rows (feature maps beyond an XCD's L2) -----------------------------------------------------------
// With feature maps this large the RoIs span many source pixels, no two output pixels of a crop share a tap, and the tap
// gather -- four 16-byte loads per 16 bytes written, every one of them an L1 miss -- runs into the L2's bandwidth: at
// config 5 the taps alone take as long as the stores alone (110 us each, 165-175 together, whatever the schedule: channel
// slices per XCD, or a band of source rows per XCD, which brought FETCH_SIZE from 2.35x to 1.25x the feature maps and did
// not change the time).  But the 300 RoIs of an image overlap: every source pixel is sampled about 240 times.  So the
// roles are turned round.  An output row of a box samples exactly two source rows (top, bot = top + 1).  A pre-pass
// (crop_rows_order_kernel, one counting sort per image) orders the image's R * ch (box, output row) items by top and
// writes every output pixel's sampling parameters in that order; then a workgroup owns a source-row pair (t, t + 1) of one
// image and one part of the channels, stages those two rows in LDS ONCE -- straight from global memory
// (global_load_lds: no registers; the LDS destination of such a load is wave-uniform base + lane * 16, which is rows[j]
// for j = the wave's first j + lane) -- and produces every item of the image whose top row is t with its taps read from
// LDS: 0.55 bytes through the L2 per byte written instead of 4.  The channels are cut into the fewest parts Q with which two
// rows fit the LDS: an item's output is cw runs of C / Q floats (1152 B at config 5: nine full lines).
//
// Measured alternatives, all bit-identical (config 5's shard, 16 images, cold inputs; this kernel: 153 us): two workgroups
// per CU with Q = 4 so that one loads while the other computes: 163 (576-byte runs); a workgroup walking an item-balanced
// strip of rows with a sliding two-row window (every row loaded once): 170 -- on gfx9 a wave that has issued stores cannot
// wait for a later load without waiting for the stores' acknowledgements (one in-order vmcnt), so each row costs a full
// drain; the same strip walk with specialised waves (two row-loader waves, a record-loader wave, thirteen compute waves
// that never wait on vmcnt, four row buffers, one s_barrier per row): 177 -- the loaders' requests queue behind the CU's
// own stores in its one in-order memory pipeline, and the barrier couples the store stream to them.
struct __attribute__((aligned(16))) StageRec {
  int dst;        // position of the pixel's C floats in the output, in float4 units
  float lx, ly;
  unsigned bits;  // left | right << 10 | (bot == top) << 20 | valid << 21   (columns < 1024)
};

// One workgroup of 1024 threads per image.  rowstart[b*(H + 2) + t] = rank of the first item of image b whose top row is t
// (t = H: the rows whose in_y is out of range -- no taps, sorted last; t = H + 1: n = R * ch); rank_top[b*n + i] = rank of item
// i = r * ch + y in that order, with its top row in the upper 16 bits.  LDS: 1024 + 8 bins | 16 wave sums | n tops (uint16).
__global__ __launch_bounds__(1024) void crop_rows_order_kernel(const float *__restrict__ boxes, int R, int H, int ch,
                                                               int *__restrict__ rowstart, unsigned *__restrict__ rank_top) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = R * ch;
  int *bins = sm;                                                         // [H + 1] counts, then running cursors
  int *wsum = sm + 1024 + 8;                                              // [16]
  unsigned short *tops = reinterpret_cast<unsigned short *>(wsum + 16);  // [n] top of item i
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
    const int pos = atomicAdd(&bins[top], 1);  // (any order inside one source row: it only decides which lanes write the item)
    rank_top[(long long)b * n + i] = (unsigned)pos | ((unsigned)top << 16);
  }
}

// rec[(b*n + rank)*cw + x] = the sampling parameters of output pixel x of the item at that rank: one thread per output pixel
// of the launch (the order kernel is one workgroup per image; these B*R*ch*cw coordinate computations would triple its time).
__global__ __launch_bounds__(256) void crop_rows_records_kernel(const float *__restrict__ boxes, int B, int R, int H, int W, int ch,
                                                                int cw, int CV, const unsigned *__restrict__ rank_top,
                                                                StageRec *__restrict__ rec) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = R * ch;
  if (e >= (long long)B * n * cw) return;
  const long long item_g = e / cw;  // b*n + i
  const int x = (int)(e - item_g * cw);
  const int b = (int)(item_g / n), i = (int)(item_g - (long long)b * n);
  const int r = i / ch, y = i - r * ch;
  const unsigned rt = rank_top[item_g];
  const int pos = (int)(rt & 0xFFFFu), top = (int)(rt >> 16);
  const long long box = (long long)b * R + r;
  const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + box * 4);
  const PixParam prm = pix_param(y * cw + x, H, W, 1u, ch, cw, bx.x, bx.y, bx.z, bx.w, true);  // offsets in pixels
  StageRec o;
  o.dst = (int)(((box * ch + y) * cw + x) * CV);
  o.lx = prm.lx;
  o.ly = prm.ly;
  o.bits = 0u;
  if (prm.valid) {  // (then its top row is the item's top: the same expression)
    const unsigned left = prm.off_tl - (unsigned)top * (unsigned)W, right = prm.off_tr - (unsigned)top * (unsigned)W;
    o.bits = left | (right << 10) | ((prm.off_bl == prm.off_tl ? 1u : 0u) << 20) | (1u << 21);
  }
  rec[((long long)b * n + pos) * cw + x] = o;
}

// grid: one workgroup of NT threads per (image b, top row t <= H, channel part q); see the index decode for the order.
// LDS: 2 * W * CQ float4 of rows | ic * cw PixParam (off_* are float4 indices into the staged rows here).
// Up to kStageGroups batches of one shape per launch (frcnn_roi_pooling_prepared_multi): image index b of the launch is image
// b % B of batch b / B; every batch has its own feature maps, output and pre-pass results.
static const int kStageGroups = 4;
struct StageGroups {
  const float *image[kStageGroups];
  float *out[kStageGroups];
  const StageRec *rec[kStageGroups];
  const int *rowstart[kStageGroups];
};

template <int NT>
__global__ __launch_bounds__(NT) void crop_resize_rowstage_kernel(StageGroups gr, int groups, int B, int H, int W, int C, int R,
                                                                  int ch, int cw, float extrap, int Q, int ic, int rg, int gpx) {
  extern __shared__ __attribute__((aligned(16))) f32x4 rows[];
  const int tid = threadIdx.x;
  int q, t, b;
  if (rg > 0) {
    // Workgroups are dealt to the 8 XCDs by blockIdx % 8.  Row t is loaded twice, as the top of pair t and as the bottom of
    // pair t - 1: groups of rg consecutive pairs go to ONE XCD, back to back in its dispatch order, so that the second load
    // of a row finds it in that XCD's L2 (group k of an image -> XCD k % 8, so an XCD gets groups from all over the image).
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;
    const int per_image = gpx * rg * Q;  // slots of one image on one XCD
    b = s / per_image;
    const int rem = s - b * per_image;
    const int gl = rem / (rg * Q), in_g = rem - gl * (rg * Q);
    t = (gl * 8 + x) * rg + in_g / Q;
    q = in_g % Q;
    if (t > H) return;
  } else {
    q = blockIdx.x % Q;
    t = (blockIdx.x / Q) % (H + 1);
    b = blockIdx.x / (Q * (H + 1));
  }
  if (b >= B * groups) return;
  // which batch (workgroup-uniform; selects instead of indexing the kernel-argument arrays dynamically)
  const float *__restrict__ image = gr.image[0];
  float *__restrict__ out = gr.out[0];
  const StageRec *__restrict__ rec = gr.rec[0];
  const int *__restrict__ rowstart = gr.rowstart[0];
  if (groups > 1) {
    const int g = b / B;
    b -= g * B;
#pragma unroll
    for (int i = 1; i < kStageGroups; ++i) {
      if (g == i) {
        image = gr.image[i];
        out = gr.out[i];
        rec = gr.rec[i];
        rowstart = gr.rowstart[i];
      }
    }
  }
  const int n = R * ch;
  const int r0 = rowstart[(long long)b * (H + 2) + t], r1 = rowstart[(long long)b * (H + 2) + t + 1];
  if (r0 >= r1) return;  // nobody samples this row pair
  const bool stamped = (b == B / 2) && (t == H / 2) && (q == 0);  // (diagnostic builds: one mid-image workgroup's phases)
  (void)stamped;
  FRCNN_STAMP_IF(stamped, 60);
  const int CV = C >> 2, CQ = CV / Q;
  const int rowlen = W * CQ;  // float4 of one staged row
  PixParam *pp = reinterpret_cast<PixParam *>(rows + 2 * rowlen);
  const StageRec *recs = rec + ((long long)b * n + r0) * cw;  // this row pair's pixels, contiguous
  const int npix = (r1 - r0) * cw;
  // the first chunk's records are requested BEFORE the rows, so that all of the workgroup's loads are one round trip
  StageRec next = {};
  {
    const int np0 = (npix < ic * cw) ? npix : ic * cw;
    if (tid < np0) next = recs[tid];
  }
  if (t < H) {  // (t == H: the rows without taps)
    const int tb = (t + 1 < H) ? t + 1 : t;
    const f32x4 *src_t = reinterpret_cast<const f32x4 *>(image + ((long long)b * H + t) * W * C) + q * CQ;
    const f32x4 *src_b = reinterpret_cast<const f32x4 *>(image + ((long long)b * H + tb) * W * C) + q * CQ;
    int col = tid / CQ, c = tid - col * CQ;
    const int dcol = NT / CQ, dc = NT - dcol * CQ;
    for (int j = tid; j < rowlen; j += NT) {
      const int j0 = j - (tid & 63);
      const long long gi = (long long)col * CV + c;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_t + gi),
                                       (__attribute__((address_space(3))) void *)(rows + j0), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_b + gi),
                                       (__attribute__((address_space(3))) void *)(rows + rowlen + j0), 16, 0, 0);
      c += dc;
      col += dcol;
      if (c >= CQ) {
        c -= CQ;
        ++col;
      }
    }
  }
  f32x4 *__restrict__ out4 = reinterpret_cast<f32x4 *>(out);
  const int dp = NT / CQ, dcc = NT - dp * CQ;
  // Barriers inside this loop order LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also wait for
  // vmcnt(0), i.e. for the acknowledgement of every output store issued so far.  For the same reason the next chunk's
  // records are requested before this chunk's stores are issued (vector-memory operations retire in order).
  auto lds_barrier = []() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int p0 = 0; p0 < npix; p0 += ic * cw) {
    const int np = (npix - p0 < ic * cw) ? (npix - p0) : ic * cw;
    if (p0 > 0) lds_barrier();  // the previous chunk's parameters are no longer read
    FRCNN_STAMP_IF(stamped, 61);
    if (tid < np) {
      const StageRec rc = next;
      PixParam prm;
      const unsigned left = (rc.bits & 1023u) * (unsigned)CQ, right = ((rc.bits >> 10) & 1023u) * (unsigned)CQ;
      const unsigned brow = ((rc.bits >> 20) & 1u) ? 0u : (unsigned)rowlen;  // bot == top when in_y is integral
      prm.off_tl = left;
      prm.off_tr = right;
      prm.off_bl = brow + left;
      prm.off_br = brow + right;
      prm.lx = rc.lx;
      prm.ly = rc.ly;
      prm.valid = (int)((rc.bits >> 21) & 1u);
      prm.pad = rc.dst + q * CQ;
      pp[tid] = prm;
    }
    if (p0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged rows have landed in LDS
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

#ifndef FRCNN_CROP_NT
#define FRCNN_CROP_NT 256  // threads per workgroup of the box kernel (diagnostic builds: 512 / 1024)
#endif
template <int VEC, bool ROWS, int WPE, bool PF = false>
__global__ __launch_bounds__(FRCNN_CROP_NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
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
      for (long long o = ((long long)piece * FRCNN_CROP_NT + tid) * 16; o + 16 <= image_bytes; o += (long long)gr.pf_blocks * FRCNN_CROP_NT * 16) {
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
    const int dp = FRCNN_CROP_NT / CV, dc = FRCNN_CROP_NT - dp * CV;
#ifndef FRCNN_CROP_UNROLL
#define FRCNN_CROP_UNROLL 4
#endif
    // FRCNN_CROP_UNROLL outputs per lane per trip, all their taps issued before the first lerp.  Unrolled by hand (the
    // packed-subtract asm is convergent, which rules out the compiler's runtime unroll): lanes past the end recompute
    // the last element (in-bounds loads) and skip the store.
    for (int e0 = tid; e0 < total; e0 += FRCNN_CROP_NT * FRCNN_CROP_UNROLL) {
      vec_t v[FRCNN_CROP_UNROLL];
#pragma unroll
      for (int u = 0; u < FRCNN_CROP_UNROLL; ++u) {
        const bool in = e0 + u * FRCNN_CROP_NT < total;
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
        const int e = e0 + u * FRCNN_CROP_NT;
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
  dim3 grid((unsigned)n), block(256), box_block(FRCNN_CROP_NT);
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
    hipLaunchKernelGGL((crop_resize_kernel<4, true, 6>), dim3((unsigned)(n * crop_h)), box_block, 0, (hipStream_t)stream, gr,
                       B, H, W, C, box_ind, n * crop_h, boxes_per_image, crop_h, crop_w, extrapolation_value);
  } else if (vec4) {
    unsigned pf_grid = 0;
    if (!box_ind && (pf_grid = crop_warmup_grid(gr, n, boxes_per_image, B, (long long)H * W * C * 4)) != 0)
      hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU, true>), dim3(pf_grid), box_block, 0, (hipStream_t)stream,
                         gr, B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
    else
      hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU>), grid, box_block, 0, (hipStream_t)stream, gr,
                         B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
  } else {
    hipLaunchKernelGGL((crop_resize_kernel<1, false, FRCNN_CROP_WAVES_PER_EU>), grid, box_block, 0, (hipStream_t)stream, gr,
                       B, H, W, C, box_ind, n, boxes_per_image, crop_h, crop_w, extrapolation_value);
  }
  return check_launch("crop_and_resize");
}

// frcnn_roi_pooling's workspace: the RoI order of the sorted channel-slice sweep (B*R ints) at the front; for the
// LDS-staged form the per-image rowstart (H + 2 <= 1032 ints), the items' ranks and the per-pixel records (crop_h * crop_w <=
// kStageMaxPixels)
static const int kStageMaxPixels = 64;
static size_t stage_rowstart_offset(int B, int R) { return round_up((size_t)B * R * sizeof(int), 256); }
static size_t stage_rank_offset(int B, int R) { return stage_rowstart_offset(B, R) + round_up((size_t)B * 1032 * sizeof(int), 256); }
static size_t stage_rec_offset(int B, int R) { return stage_rank_offset(B, R) + round_up((size_t)B * R * kStageMaxPixels * sizeof(int), 256); }
extern "C" size_t frcnn_roi_pooling_workspace_bytes(int B, int R) {
  if (B <= 0 || R <= 0) return 256;
  return stage_rec_offset(B, R) + round_up((size_t)B * R * kStageMaxPixels * sizeof(StageRec), 256) + 256;
}

// The LDS-staged schedule of frcnn_roi_pooling in three pieces, so that the pre-pass (which needs the RoIs only) can run on
// another stream than the crop (frcnn_roi_pooling_prepare / frcnn_roi_pooling_prepared).
struct StagePlan {
  int Q = 0;  // channel parts; 0: this call does not take the staged schedule
  int ic = 0, rg = 0, gpx = 0;
  size_t lds = 0;
};

static StagePlan stage_plan(const float *feature_map, int B, int H, int W, int C, const float *rois, int R, int crop_h, int crop_w,
                            const float *out, const void *workspace, size_t workspace_bytes) {
  StagePlan plan;
  const long long image_bytes = (long long)H * W * C * 4;
  const int rb = knob(kKnobCropStage, -1);
  const bool want = rb >= 0 ? rb != 0 : (image_bytes > 4ll * 1024 * 1024 && !knob_set(kKnobCropSplit) && !knob_set(kKnobCropSlice));
  const long long nb = (long long)R * crop_h;
  const int NT = 1024;
  int ic = 64;  // items per chunk of parameters
  const bool ic_forced = rb > 0 && knob(kKnobCropSort, 0) > 1;
  if (ic_forced) ic = knob(kKnobCropSort, 0);  // (diagnostic: FRCNN_CROP_STAGE=1 FRCNN_CROP_SORT=items per chunk)
  if (crop_w > 0 && ic * crop_w > NT) ic = NT / crop_w;
  const int CV = C / 4;
  const bool fits = B > 0 && R > 0 && H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0 && (C % 4 == 0) && aligned_to(feature_map, 16) &&
                    aligned_to(out, 16) && aligned_to(rois, 16) && image_bytes < (1ll << 32) && H <= 1023 && W <= 1023 &&
                    nb <= 65535 && crop_h * crop_w <= kStageMaxPixels && ic > 0 && (long long)B * nb * crop_w < (1ll << 31) &&
                    (long long)B * R * crop_h * crop_w * CV < (1ll << 31) && workspace && aligned_to(workspace, 256) &&
                    workspace_bytes >= frcnn_roi_pooling_workspace_bytes(B, R);
  if (!(want && fits)) return plan;
  const int q_min = rb > 0 ? knob(kKnobCropSplit, 0) : 0;
  for (int cand = 1; cand <= 8; cand *= 2) {  // the fewest channel parts with which two rows and the parameters fit the LDS
    if (CV % cand || cand < q_min) continue;
    const size_t need = (size_t)2 * W * (CV / cand) * 16 + (size_t)ic * crop_w * sizeof(PixParam);
    if (need <= 160 * 1024) {
      plan.Q = cand;
      plan.lds = need;
      break;
    }
  }
  if (plan.Q == 0) return plan;
  if (!ic_forced && ic > 32) {
    // half the parameter area when that lets one more workgroup share the CU's LDS (one loads its rows while the other
    // stores): config 2's shape forced through this schedule, Q = 2: 120-121 k images/s against 116.7 k in the stage pipeline
    const size_t rows_bytes = (size_t)2 * W * (CV / plan.Q) * 16, lds_cu = 160 * 1024;
    const size_t small = rows_bytes + (size_t)32 * crop_w * sizeof(PixParam);
    if (lds_cu / small > lds_cu / plan.lds) {
      ic = 32;
      plan.lds = small;
    }
  }
  plan.ic = ic;
  // Row pairs per XCD group: the largest of 8 / 4 / 2 that deals the H + 1 pairs to the 8 XCDs within 7 % of evenly, else 0
  // (pairs dealt round-robin, every row fetched twice).  Config 5 (H = 63), 16 images, cold inputs, FETCH_SIZE x 2 / kernel
  // time under the profiler / crop time in the stage pipeline: 0: 298 MB, 145.3 us, 156.6 us; 4: 185 MB, 141.4, 154.9;
  // 8: 166 MB (1.14 x the feature maps), 143.6, 155.4.  FRCNN_CROP_PREFETCH overrides (diagnostic).
  int rg = 0;
  for (int cand = 8; cand >= 2 && !rg; cand >>= 1) {
    const int slots = (((H + 1 + cand - 1) / cand + 7) / 8) * 8 * cand;
    if ((long long)slots * 100 <= (long long)(H + 1) * 107) rg = cand;
  }
  rg = knob(kKnobCropPrefetch, rg);
  plan.rg = rg < 0 ? 0 : (rg > 64 ? 64 : rg);
  plan.gpx = plan.rg > 0 ? ((H + 1 + plan.rg - 1) / plan.rg + 7) / 8 : 0;  // groups of one image per XCD
  const long long wgs = plan.rg > 0 ? (long long)8 * B * plan.gpx * plan.rg * plan.Q : (long long)B * plan.Q * (H + 1);
  if (wgs >= (1ll << 31)) plan.Q = 0;
  return plan;
}

static int stage_prepass(const StagePlan &plan, int B, int H, int W, int C, const float *rois, int R, int crop_h, int crop_w,
                         void *workspace, hipStream_t stream) {
  (void)plan;
  const long long nb = (long long)R * crop_h;
  int *rowstart = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + stage_rowstart_offset(B, R));
  StageRec *rec = reinterpret_cast<StageRec *>(reinterpret_cast<char *>(workspace) + stage_rec_offset(B, R));
  unsigned *rank_top = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(workspace) + stage_rank_offset(B, R));
  const size_t lds = (size_t)(1024 + 8 + 16) * 4 + (size_t)((nb + 1) & ~1ll) * 2;
  auto okern = crop_rows_order_kernel;
  if (lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(okern), lds, "roi_pooling: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(okern, dim3((unsigned)B), dim3(1024), lds, stream, rois, R, H, crop_h, rowstart, rank_top);
  int rc = check_launch("roi_pooling (row order)");
  if (rc) return rc;
  const long long npix_all = (long long)B * nb * crop_w;
  hipLaunchKernelGGL(crop_rows_records_kernel, dim3((unsigned)((npix_all + 255) / 256)), dim3(256), 0, stream, rois, B, R, H, W, crop_h,
                     crop_w, C / 4, rank_top, rec);
  return check_launch("roi_pooling (records)");
}

static int stage_main(const StagePlan &plan, const float *const *feature_maps, float *const *outs, void *const *workspaces, int groups,
                      int B, int H, int W, int C, int R, int crop_h, int crop_w, hipStream_t stream) {
  StageGroups gr = {};
  for (int g = 0; g < groups; ++g) {
    gr.image[g] = feature_maps[g];
    gr.out[g] = outs[g];
    gr.rowstart[g] = reinterpret_cast<const int *>(reinterpret_cast<char *>(workspaces[g]) + stage_rowstart_offset(B, R));
    gr.rec[g] = reinterpret_cast<const StageRec *>(reinterpret_cast<char *>(workspaces[g]) + stage_rec_offset(B, R));
  }
  auto skern = crop_resize_rowstage_kernel<1024>;
  if (plan.lds > 48 * 1024) {
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(skern), plan.lds, "roi_pooling: hipFuncSetAttribute");
    if (rc) return rc;
  }
  const long long Bt = (long long)B * groups;
  const long long wgs = plan.rg > 0 ? 8 * Bt * plan.gpx * plan.rg * plan.Q : Bt * plan.Q * (H + 1);
  if (wgs >= (1ll << 31)) return set_error(FRCNN_EUNSUPPORTED, "roi_pooling: too many workgroups");
  hipLaunchKernelGGL(skern, dim3((unsigned)wgs), dim3(1024), plan.lds, stream, gr, groups, B, H, W, C, R, crop_h, crop_w, 0.0f, plan.Q,
                     plan.ic, plan.rg, plan.gpx);
  return check_launch("roi_pooling (staged rows)");
}

extern "C" int frcnn_roi_pooling_prepare(const float *feature_map, int B, int H, int W, int C, const float *rois, int R, int crop_h,
                                         int crop_w, const float *out, void *workspace, size_t workspace_bytes, int *prepared,
                                         frcnn_stream_t stream) {
  FRCNN_REQUIRE(feature_map && rois && out && prepared, "roi_pooling_prepare: null pointer");
  FRCNN_REQUIRE(B >= 0 && R >= 0, "roi_pooling_prepare: negative shape");
  *prepared = 0;
  if (B == 0 || R == 0) return FRCNN_OK;
  FRCNN_REQUIRE((long long)B * R < (1ll << 28), "roi_pooling_prepare: too many RoIs");
  const StagePlan plan = stage_plan(feature_map, B, H, W, C, rois, R, crop_h, crop_w, out, workspace, workspace_bytes);
  if (plan.Q == 0) return FRCNN_OK;  // another schedule: nothing to prepare, call frcnn_roi_pooling
  int rc = stage_prepass(plan, B, H, W, C, rois, R, crop_h, crop_w, workspace, (hipStream_t)stream);
  if (rc) return rc;
  *prepared = 1;
  return FRCNN_OK;
}

extern "C" int frcnn_roi_pooling_prepared(const float *feature_map, int B, int H, int W, int C, const float *rois, int R, int crop_h,
                                          int crop_w, float *out, void *workspace, size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(feature_map && rois && out, "roi_pooling_prepared: null pointer");
  FRCNN_REQUIRE(B >= 0 && R >= 0, "roi_pooling_prepared: negative shape");
  if (B == 0 || R == 0) return FRCNN_OK;
  FRCNN_REQUIRE((long long)B * R < (1ll << 28), "roi_pooling_prepared: too many RoIs");
  const StagePlan plan = stage_plan(feature_map, B, H, W, C, rois, R, crop_h, crop_w, out, workspace, workspace_bytes);
  if (plan.Q == 0)
    return set_error(FRCNN_EINVAL, "roi_pooling_prepared: these arguments do not take the prepared schedule "
                                   "(frcnn_roi_pooling_prepare must have reported prepared = 1 for the same arguments)");
  void *const wsp = workspace;
  return stage_main(plan, &feature_map, &out, &wsp, 1, B, H, W, C, R, crop_h, crop_w, (hipStream_t)stream);
}

extern "C" int frcnn_roi_pooling_prepared_multi(const float *const *feature_maps, const float *const *rois, float *const *outs,
                                                void *const *workspaces, size_t workspace_bytes, int groups, int B, int H, int W,
                                                int C, int R, int crop_h, int crop_w, frcnn_stream_t stream) {
  FRCNN_REQUIRE(feature_maps && rois && outs && workspaces, "roi_pooling_prepared_multi: null table");
  FRCNN_REQUIRE(groups >= 1 && groups <= kStageGroups, "roi_pooling_prepared_multi: 1..%d batches per launch (got %d)", kStageGroups, groups);
  FRCNN_REQUIRE(B >= 0 && R >= 0, "roi_pooling_prepared_multi: negative shape");
  if (B == 0 || R == 0) return FRCNN_OK;
  FRCNN_REQUIRE((long long)B * R < (1ll << 28), "roi_pooling_prepared_multi: too many RoIs");
  StagePlan plan;
  for (int g = 0; g < groups; ++g) {
    FRCNN_REQUIRE(feature_maps[g] && rois[g] && outs[g], "roi_pooling_prepared_multi: null pointer in batch %d", g);
    // (the plan depends on the shape and the schedule knobs; the pointers only have to pass its alignment checks)
    plan = stage_plan(feature_maps[g], B, H, W, C, rois[g], R, crop_h, crop_w, outs[g], workspaces[g], workspace_bytes);
    if (plan.Q == 0)
      return set_error(FRCNN_EINVAL, "roi_pooling_prepared_multi: batch %d does not take the prepared schedule "
                                     "(frcnn_roi_pooling_prepare must have reported prepared = 1 for the same arguments)", g);
  }
  return stage_main(plan, feature_maps, outs, workspaces, groups, B, H, W, C, R, crop_h, crop_w, (hipStream_t)stream);
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
  // LDS-staged feature rows (crop_resize_rowstage_kernel): the default beyond 4 MiB per image when the caller brings the
  // workspace; FRCNN_CROP_STAGE = 0 | 1 forces the choice for any size (1 still needs the workspace and the shape limits;
  // with it FRCNN_CROP_SPLIT = Q sets a minimum number of channel parts: diagnostic).
  {
    const StagePlan plan = stage_plan(feature_map, B, H, W, C, rois, R, crop_h, crop_w, out, workspace, workspace_bytes);
    if (plan.Q > 0) {
      int rc = stage_prepass(plan, B, H, W, C, rois, R, crop_h, crop_w, workspace, (hipStream_t)stream);
      if (rc) return rc;
      void *const wsp = workspace;
      return stage_main(plan, &feature_map, &out, &wsp, 1, B, H, W, C, R, crop_h, crop_w, (hipStream_t)stream);
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
    hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU, true>), dim3(pf_grid), dim3(FRCNN_CROP_NT), 0,
                       (hipStream_t)stream, gr, B, H, W, C, (const int *)nullptr, n * groups, boxes_per_image, crop_h, crop_w,
                       extrapolation_value);
  else
    hipLaunchKernelGGL((crop_resize_kernel<4, false, FRCNN_CROP_WAVES_PER_EU>), dim3((unsigned)(n * groups)), dim3(FRCNN_CROP_NT), 0,
                       (hipStream_t)stream, gr, B, H, W, C, (const int *)nullptr, n * groups, boxes_per_image, crop_h, crop_w,
                       extrapolation_value);
  return check_launch("crop_and_resize_multi");
}
