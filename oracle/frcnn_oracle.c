/*
 * frcnn_oracle.c -- CPU restatement of the reference hot path.  See frcnn_oracle.h for the
 * status header ("parity unpinned" against TensorFlow; test infrastructure only).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * All citations are relative to /root/reference/.
 */
#include "frcnn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------------------------
 * bit helpers
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* Order-preserving map float -> uint32 (larger float => larger key); -0 is folded onto +0 so
 * that equal values compare equal, NaNs sort by their bit pattern (documented, unpinned). */
static inline uint32_t f_order_key(float f) {
  uint32_t u;
  if (f == 0.0f) f = 0.0f * 0.0f; /* -0 -> +0 */
  u = f2u(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* std::min / std::max exactly as the TF C++ kernels use them (NaN behaviour included). */
static inline float std_minf(float a, float b) { return (b < a) ? b : a; }
static inline float std_maxf(float a, float b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------------------------------
 * [TF-recall] tf.exp on CPU: Eigen generic pexp_float (Cephes expf), non-FMA code path
 * (official TF 2.0 wheels are built without FMA).  utils/bbox_utils.py:100-101.
 * ---------------------------------------------------------------------------------------- */
float orc_expf(float _x) {
  const float exp_hi = 88.3762626647950f;
  const float exp_lo = -88.3762626647949f;
  const float LOG2EF = 1.44269504088896341f;
  const float C1 = 0.693359375f;
  const float C2 = -2.12194440e-4f;
  const float p0 = 1.9875691500E-4f;
  const float p1 = 1.3981999507E-3f;
  const float p2 = 8.3334519073E-3f;
  const float p3 = 4.1665795894E-2f;
  const float p4 = 1.6666665459E-1f;
  const float p5 = 5.0000001201E-1f;
  float x, m, r, tmp, z, r2, y, scale, res;
  int32_t e;
  /* pmax(pmin(_x, hi), lo) with SSE min/max NaN semantics (second operand on NaN). */
  x = (_x < exp_hi) ? _x : exp_hi;
  x = (x > exp_lo) ? x : exp_lo;
  tmp = x * LOG2EF;
  tmp = tmp + 0.5f;
  m = floorf(tmp);
  r = x;
  tmp = m * C1;
  z = m * C2;
  r = r - tmp;
  r = r - z;
  r2 = r * r;
  y = p0;
  y = y * r;
  y = y + p1;
  y = y * r;
  y = y + p2;
  y = y * r;
  y = y + p3;
  y = y * r;
  y = y + p4;
  y = y * r;
  y = y + p5;
  y = y * r2;
  y = y + r;
  y = y + 1.0f;
  /* pldexp: multiply by the float whose exponent field is m + 127. */
  e = (int32_t)m + 127;
  scale = u2f((uint32_t)e << 23);
  res = y * scale;
  /* pmax(res, _x) */
  return (res > _x) ? res : ((_x != _x) ? res : _x);
}

/* [TF-recall] tf.math.log on CPU: Eigen generic plog_float (Cephes logf), non-FMA.
 * utils/bbox_utils.py:133-134. */
float orc_logf(float _x) {
  const float SQRTHF = 0.707106781186547524f;
  const float p0 = 7.0376836292E-2f;
  const float p1 = -1.1514610310E-1f;
  const float p2 = 1.1676998740E-1f;
  const float p3 = -1.2420140846E-1f;
  const float p4 = +1.4249322787E-1f;
  const float p5 = -1.6668057665E-1f;
  const float p6 = +2.0000714765E-1f;
  const float p7 = -2.4999993993E-1f;
  const float p8 = +3.3333331174E-1f;
  const float q1 = -2.12194440e-4f;
  const float q2 = 0.693359375f;
  const float min_norm_pos = 1.17549435e-38f; /* 0x00800000 */
  float x, e, tmp, x2, x3, y, y1, y2;
  uint32_t bits;
  int lt;
  if (_x != _x || _x < 0.0f) return u2f(0x7fc00000u) * 1.0f; /* NaN */
  if (_x == 0.0f) return -INFINITY;
  if (_x == INFINITY) return INFINITY;
  x = (_x > min_norm_pos) ? _x : min_norm_pos;
  bits = f2u(x);
  e = (float)(int32_t)(bits >> 23);
  e = e - 126.0f;
  x = u2f((bits & ~0x7f800000u) | 0x3f000000u); /* mantissa in [0.5,1) */
  lt = x < SQRTHF;
  tmp = lt ? x : 0.0f;
  x = x - 1.0f;
  e = e - (lt ? 1.0f : 0.0f);
  x = x + tmp;
  x2 = x * x;
  x3 = x2 * x;
  y = p0 * x;
  y = y + p1;
  y1 = p3 * x;
  y1 = y1 + p4;
  y2 = p6 * x;
  y2 = y2 + p7;
  y = y * x;
  y = y + p2;
  y1 = y1 * x;
  y1 = y1 + p5;
  y2 = y2 * x;
  y2 = y2 + p8;
  y = y * x3;
  y = y + y1;
  y = y * x3;
  y = y + y2;
  y = y * x3;
  y1 = e * q1;
  tmp = x2 * 0.5f;
  y = y + y1;
  x = x - tmp;
  y2 = e * q2;
  x = x + y;
  x = x + y2;
  return x;
}

/* ------------------------------------------------------------------------------------------
 * anchors
 * ---------------------------------------------------------------------------------------- */
/* utils/bbox_utils.py:24-34.  `scale /= img_size` and `scale ** 2 / ratio` are Python float64;
 * tf.sqrt receives that Python float (-> fp32 tensor); `w * ratio`, `h / 2` are fp32. */
void orc_base_anchors(double img_size, const double *ratios, int nr, const double *scales,
                      int ns, float *out) {
  int si, ri, k = 0;
  for (si = 0; si < ns; ++si) {
    double s = scales[si] / img_size;
    for (ri = 0; ri < nr; ++ri) {
      float arg = (float)(s * s / ratios[ri]);
      float w = sqrtf(arg);
      float h = w * (float)ratios[ri];
      out[k * 4 + 0] = (-h) / 2.0f;
      out[k * 4 + 1] = (-w) / 2.0f;
      out[k * 4 + 2] = h / 2.0f;
      out[k * 4 + 3] = w / 2.0f;
      ++k;
    }
  }
}

/* utils/bbox_utils.py:47-57.  tf.range(0,G)/G is an int32 true-divide => float64, the stride
 * add is float64, then one cast to fp32 (:49); meshgrid 'xy' (:50) => cell index gy*G+gx. */
void orc_generate_anchors(int G, const float *base, int A, float *out) {
  int gy, gx, a, c;
  double stride = 1.0 / (double)G;
  for (gy = 0; gy < G; ++gy) {
    float cy = (float)((double)gy / (double)G + stride / 2.0);
    for (gx = 0; gx < G; ++gx) {
      float cx = (float)((double)gx / (double)G + stride / 2.0);
      float grid[4];
      grid[0] = cy;
      grid[1] = cx;
      grid[2] = cy;
      grid[3] = cx;
      for (a = 0; a < A; ++a) {
        float *o = out + ((size_t)(gy * G + gx) * A + a) * 4;
        for (c = 0; c < 4; ++c) {
          float v = base[a * 4 + c] + grid[c];
          /* tf.clip_by_value(v, 0, 1) = maximum(minimum(v, 1), 0) */
          v = (v < 1.0f) ? v : 1.0f;
          v = (v > 0.0f) ? v : 0.0f;
          o[c] = v;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * decode / encode
 * ---------------------------------------------------------------------------------------- */
/* utils/bbox_utils.py:96-108, one rounding per Python-level op, in the reference order. */
static inline void decode_one(const float *a, const float *d, float *o) {
  float w = a[3] - a[1];
  float h = a[2] - a[0];
  float t, cx, cy, bw, bh, bcx, bcy, y1, x1;
  t = 0.5f * w;
  cx = a[1] + t;
  t = 0.5f * h;
  cy = a[0] + t;
  bw = orc_expf(d[3]) * w;
  bh = orc_expf(d[2]) * h;
  t = d[1] * w;
  bcx = t + cx;
  t = d[0] * h;
  bcy = t + cy;
  t = 0.5f * bh;
  y1 = bcy - t;
  t = 0.5f * bw;
  x1 = bcx - t;
  o[0] = y1;
  o[1] = x1;
  o[2] = bh + y1;
  o[3] = bw + x1;
}

void orc_decode(const float *anchors, int anchors_batched, const float *deltas,
                const float *variances, int64_t B, int64_t N, float *out) {
  int64_t b;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    int64_t n;
    for (n = 0; n < N; ++n) {
      const float *a = anchors + (anchors_batched ? (b * N + n) : n) * 4;
      const float *dp = deltas + (b * N + n) * 4;
      float d[4];
      int c;
      for (c = 0; c < 4; ++c) d[c] = variances ? dp[c] * variances[c] : dp[c];
      decode_one(a, d, out + (b * N + n) * 4);
    }
  }
}

/* utils/bbox_utils.py:121-135. Output order [dy, dx, dh, dw]. */
static inline void encode_one(const float *bx, const float *g, const float *divv, float *o) {
  float bw = bx[3] - bx[1];
  float bh = bx[2] - bx[0];
  float t, bcx, bcy, gw, gh, gcx, gcy, dx, dy, dw, dh;
  t = 0.5f * bw;
  bcx = bx[1] + t;
  t = 0.5f * bh;
  bcy = bx[0] + t;
  gw = g[3] - g[1];
  gh = g[2] - g[0];
  t = 0.5f * gw;
  gcx = g[1] + t;
  t = 0.5f * gh;
  gcy = g[0] + t;
  if (bw == 0.0f) bw = 1e-3f;
  if (bh == 0.0f) bh = 1e-3f;
  if (gw == 0.0f) {
    dx = 0.0f;
    dw = 0.0f;
  } else {
    t = gcx - bcx;
    dx = t / bw;
    t = gw / bw;
    dw = orc_logf(t);
  }
  if (gh == 0.0f) {
    dy = 0.0f;
    dh = 0.0f;
  } else {
    t = gcy - bcy;
    dy = t / bh;
    t = gh / bh;
    dh = orc_logf(t);
  }
  o[0] = dy;
  o[1] = dx;
  o[2] = dh;
  o[3] = dw;
  if (divv) {
    int c;
    for (c = 0; c < 4; ++c) o[c] = o[c] / divv[c];
  }
}

void orc_encode(const float *bboxes, int bboxes_batched, const float *gt,
                const float *div_variances, int64_t B, int64_t N, float *out) {
  int64_t b, n;
  for (b = 0; b < B; ++b)
    for (n = 0; n < N; ++n)
      encode_one(bboxes + (bboxes_batched ? (b * N + n) : n) * 4, gt + (b * N + n) * 4,
                 div_variances, out + (b * N + n) * 4);
}

/* ------------------------------------------------------------------------------------------
 * IoU map -- utils/bbox_utils.py:150-160.  No epsilon, no area guard (0/0 -> NaN).
 * ---------------------------------------------------------------------------------------- */
static inline float iou_map_one(const float *bx, float barea, const float *g, float garea) {
  float x_top = (bx[1] > g[1]) ? bx[1] : g[1];
  float y_top = (bx[0] > g[0]) ? bx[0] : g[0];
  float x_bot = (bx[3] < g[3]) ? bx[3] : g[3];
  float y_bot = (bx[2] < g[2]) ? bx[2] : g[2];
  float dx = x_bot - x_top;
  float dy = y_bot - y_top;
  float inter, uni;
  dx = (dx > 0.0f) ? dx : 0.0f;
  dy = (dy > 0.0f) ? dy : 0.0f;
  inter = dx * dy;
  uni = barea + garea;
  uni = uni - inter;
  return inter / uni;
}

void orc_iou_map(const float *bboxes, int bboxes_batched, const float *gt, int B, int N, int M,
                 float *out) {
  int b;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    int n, m;
    float *garea = (float *)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    for (m = 0; m < M; ++m) {
      const float *g = gt + ((size_t)b * M + m) * 4;
      float gh = g[2] - g[0];
      float gw = g[3] - g[1];
      garea[m] = gh * gw;
    }
    for (n = 0; n < N; ++n) {
      const float *bx = bboxes + (bboxes_batched ? ((size_t)b * N + n) : (size_t)n) * 4;
      float bh = bx[2] - bx[0];
      float bw = bx[3] - bx[1];
      float barea = bh * bw;
      float *o = out + ((size_t)b * N + n) * M;
      for (m = 0; m < M; ++m) o[m] = iou_map_one(bx, barea, gt + ((size_t)b * M + m) * 4, garea[m]);
    }
    free(garea);
  }
}

/* ------------------------------------------------------------------------------------------
 * sorting helpers
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint32_t key; /* larger first */
  int32_t idx;  /* smaller first on equal key */
} sort_item;

static int cmp_item_desc(const void *pa, const void *pb) {
  const sort_item *a = (const sort_item *)pa;
  const sort_item *b = (const sort_item *)pb;
  if (a->key != b->key) return (a->key > b->key) ? -1 : 1;
  if (a->idx != b->idx) return (a->idx < b->idx) ? -1 : 1;
  return 0;
}

/* [TF-recall] tf.nn.top_k: comparator "value desc, then index asc" (topk_op.cc). */
int orc_top_k(const float *scores, int B, int N, int K, float *vals, int32_t *idx) {
  int b;
  if (K > N || K < 0) return -1;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    int n;
    sort_item *it = (sort_item *)malloc(sizeof(sort_item) * (size_t)(N > 0 ? N : 1));
    for (n = 0; n < N; ++n) {
      it[n].key = f_order_key(scores[(size_t)b * N + n]);
      it[n].idx = n;
    }
    qsort(it, (size_t)N, sizeof(sort_item), cmp_item_desc);
    for (n = 0; n < K; ++n) {
      if (idx) idx[(size_t)b * K + n] = it[n].idx;
      if (vals) vals[(size_t)b * K + n] = scores[(size_t)b * N + it[n].idx];
    }
    free(it);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * [TF-recall] combined_non_max_suppression (non_max_suppression_op.cc, BatchedNonMaxSuppressionOp)
 * reached from utils/bbox_utils.py:79.
 * ---------------------------------------------------------------------------------------- */
static inline int nms_iou_gt(const float *bi, const float *bj, float thr) {
  float ymin_i = std_minf(bi[0], bi[2]);
  float xmin_i = std_minf(bi[1], bi[3]);
  float ymax_i = std_maxf(bi[0], bi[2]);
  float xmax_i = std_maxf(bi[1], bi[3]);
  float ymin_j = std_minf(bj[0], bj[2]);
  float xmin_j = std_minf(bj[1], bj[3]);
  float ymax_j = std_maxf(bj[0], bj[2]);
  float xmax_j = std_maxf(bj[1], bj[3]);
  float hi = ymax_i - ymin_i, wi = xmax_i - xmin_i;
  float hj = ymax_j - ymin_j, wj = xmax_j - xmin_j;
  float area_i = hi * wi;
  float area_j = hj * wj;
  float iymin, ixmin, iymax, ixmax, ih, iw, inter, uni, iou;
  if (area_i <= 0.0f || area_j <= 0.0f) return 0;
  iymin = std_maxf(ymin_i, ymin_j);
  ixmin = std_maxf(xmin_i, xmin_j);
  iymax = std_minf(ymax_i, ymax_j);
  ixmax = std_minf(xmax_i, xmax_j);
  ih = iymax - iymin;
  iw = ixmax - ixmin;
  ih = std_maxf(ih, 0.0f);
  iw = std_maxf(iw, 0.0f);
  inter = ih * iw;
  uni = area_i + area_j;
  uni = uni - inter;
  iou = inter / uni;
  return iou > thr;
}

typedef struct {
  float score;
  int32_t cls;
  int32_t order; /* global insertion order: class-major, then selection order */
  float box[4];
} result_cand;

static int cmp_result_desc(const void *pa, const void *pb) {
  const result_cand *a = (const result_cand *)pa;
  const result_cand *b = (const result_cand *)pb;
  /* TF: std::sort by score desc (ties unspecified); this build: stable. */
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  if (a->order != b->order) return (a->order < b->order) ? -1 : 1;
  return 0;
}

static inline float clip01(float v) { return std_maxf(std_minf(v, 1.0f), 0.0f); }

int orc_combined_nms(const float *boxes, const float *scores, int B, int n, int q, int c,
                     int max_per_class, int max_total, float iou_thr, float score_thr,
                     int clip_boxes, float *out_boxes, float *out_scores, float *out_classes,
                     int32_t *out_valid) {
  int b;
  if (!(q == 1 || q == c) || max_total < 0 || max_per_class < 0) return -1;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    const float *bb = boxes + (size_t)b * n * q * 4;
    const float *ss = scores + (size_t)b * n * c;
    int size_per_class = max_per_class < n ? max_per_class : n;
    size_t cap = (size_t)c * (size_t)(size_per_class > 0 ? size_per_class : 1);
    result_cand *res = (result_cand *)malloc(sizeof(result_cand) * cap);
    sort_item *it = (sort_item *)malloc(sizeof(sort_item) * (size_t)(n > 0 ? n : 1));
    int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)(size_per_class > 0 ? size_per_class : 1));
    int nres = 0, cls, k, i;
    for (cls = 0; cls < c; ++cls) {
      int ncand = 0, nsel = 0;
      int qi = (q > 1) ? cls : 0;
      for (i = 0; i < n; ++i) {
        float s = ss[(size_t)i * c + cls];
        if (s > score_thr) { /* strict; NaN excluded */
          it[ncand].key = f_order_key(s);
          it[ncand].idx = i;
          ++ncand;
        }
      }
      qsort(it, (size_t)ncand, sizeof(sort_item), cmp_item_desc);
      for (k = 0; k < ncand && nsel < size_per_class; ++k) {
        int id = it[k].idx, j, keep = 1;
        const float *bi = bb + ((size_t)id * q + qi) * 4;
        for (j = nsel - 1; j >= 0; --j) {
          if (nms_iou_gt(bi, bb + ((size_t)sel[j] * q + qi) * 4, iou_thr)) {
            keep = 0;
            break;
          }
        }
        if (keep) {
          sel[nsel++] = id;
          res[nres].score = ss[(size_t)id * c + cls];
          res[nres].cls = cls;
          res[nres].order = nres;
          memcpy(res[nres].box, bi, sizeof(float) * 4);
          ++nres;
        }
      }
    }
    qsort(res, (size_t)nres, sizeof(result_cand), cmp_result_desc);
    {
      int valid = nres < max_total ? nres : max_total;
      float *ob = out_boxes + (size_t)b * max_total * 4;
      float *os = out_scores ? out_scores + (size_t)b * max_total : NULL;
      float *oc = out_classes ? out_classes + (size_t)b * max_total : NULL;
      for (k = 0; k < max_total; ++k) {
        if (k < valid) {
          for (i = 0; i < 4; ++i) ob[k * 4 + i] = clip_boxes ? clip01(res[k].box[i]) : res[k].box[i];
          if (os) os[k] = res[k].score;
          if (oc) oc[k] = (float)res[k].cls;
        } else {
          for (i = 0; i < 4; ++i) ob[k * 4 + i] = 0.0f;
          if (os) os[k] = 0.0f;
          if (oc) oc[k] = 0.0f;
        }
      }
      if (out_valid) out_valid[b] = valid;
    }
    free(sel);
    free(it);
    free(res);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * [TF-recall] crop_and_resize (crop_and_resize_op.cc, CPU functor), bilinear.
 * reached from models/faster_rcnn.py:283-288.
 * ---------------------------------------------------------------------------------------- */
int orc_crop_and_resize(const float *image, int B, int H, int W, int C, const float *boxes,
                        const int32_t *box_ind, int n, int ch, int cw, float extrapolation,
                        float *out) {
  int bi, bad = 0;
  for (bi = 0; bi < n; ++bi)
    if (box_ind[bi] < 0 || box_ind[bi] >= B) bad = 1;
  if (bad) return -1;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1) schedule(static)
  for (bi = 0; bi < n; ++bi) {
    const float y1 = boxes[bi * 4 + 0], x1 = boxes[bi * 4 + 1];
    const float y2 = boxes[bi * 4 + 2], x2 = boxes[bi * 4 + 3];
    const float *img = image + (size_t)box_ind[bi] * H * W * C;
    float *o = out + (size_t)bi * ch * cw * C;
    const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
    float height_scale = 0.0f, width_scale = 0.0f;
    int y, x, d;
    if (ch > 1) {
      float t = y2 - y1;
      t = t * hm1;
      height_scale = t / (float)(ch - 1);
    }
    if (cw > 1) {
      float t = x2 - x1;
      t = t * wm1;
      width_scale = t / (float)(cw - 1);
    }
    for (y = 0; y < ch; ++y) {
      float in_y;
      int top, bot;
      float y_lerp;
      if (ch > 1) {
        float t0 = y1 * hm1;
        float t1 = (float)y * height_scale;
        in_y = t0 + t1;
      } else {
        float t0 = y1 + y2;
        t0 = 0.5f * t0;
        in_y = t0 * hm1;
      }
      if (!(in_y >= 0.0f && in_y <= hm1)) { /* TF: in_y < 0 || in_y > H-1; NaN also lands here */
        for (x = 0; x < cw; ++x)
          for (d = 0; d < C; ++d) o[((size_t)y * cw + x) * C + d] = extrapolation;
        continue;
      }
      top = (int)floorf(in_y);
      bot = (int)ceilf(in_y);
      y_lerp = in_y - (float)top;
      for (x = 0; x < cw; ++x) {
        float in_x, x_lerp;
        int left, right;
        const float *ptl, *ptr_, *pbl, *pbr;
        float *po = o + ((size_t)y * cw + x) * C;
        if (cw > 1) {
          float t0 = x1 * wm1;
          float t1 = (float)x * width_scale;
          in_x = t0 + t1;
        } else {
          float t0 = x1 + x2;
          t0 = 0.5f * t0;
          in_x = t0 * wm1;
        }
        if (!(in_x >= 0.0f && in_x <= wm1)) {
          for (d = 0; d < C; ++d) po[d] = extrapolation;
          continue;
        }
        left = (int)floorf(in_x);
        right = (int)ceilf(in_x);
        x_lerp = in_x - (float)left;
        ptl = img + ((size_t)top * W + left) * C;
        ptr_ = img + ((size_t)top * W + right) * C;
        pbl = img + ((size_t)bot * W + left) * C;
        pbr = img + ((size_t)bot * W + right) * C;
        for (d = 0; d < C; ++d) {
          float tl = ptl[d], tr = ptr_[d], bl = pbl[d], br = pbr[d];
          float t, tp, bt;
          t = tr - tl;
          t = t * x_lerp;
          tp = tl + t;
          t = br - bl;
          t = t * x_lerp;
          bt = bl + t;
          t = bt - tp;
          t = t * y_lerp;
          po[d] = tp + t;
        }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * layer compositions
 * ---------------------------------------------------------------------------------------- */
/* models/faster_rcnn.py:139-166 */
int orc_roi_bbox(const float *deltas, const float *scores, const float *anchors,
                 const float *variances, int B, int N, int K, int R, float iou_thr,
                 float *out_boxes, float *out_scores, int32_t *out_valid) {
  float *dec, *pre_boxes, *pre_scores, *cls_tmp;
  int32_t *idx;
  int b, k, rc;
  if (K > N) return -1; /* tf.nn.top_k raises InvalidArgument */
  dec = (float *)malloc(sizeof(float) * (size_t)B * N * 4);
  pre_boxes = (float *)malloc(sizeof(float) * (size_t)B * K * 4);
  pre_scores = (float *)malloc(sizeof(float) * (size_t)B * K);
  idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * K);
  cls_tmp = (float *)malloc(sizeof(float) * (size_t)B * R);
  orc_decode(anchors, 0, deltas, variances, B, N, dec);            /* :149-150 */
  orc_top_k(scores, B, N, K, NULL, idx);                            /* :153 */
  for (b = 0; b < B; ++b)                                           /* :154-155 */
    for (k = 0; k < K; ++k) {
      int32_t i = idx[(size_t)b * K + k];
      memcpy(pre_boxes + ((size_t)b * K + k) * 4, dec + ((size_t)b * N + i) * 4, 16);
      pre_scores[(size_t)b * K + k] = scores[(size_t)b * N + i];
    }
  rc = orc_combined_nms(pre_boxes, pre_scores, B, K, 1, 1, R, R, iou_thr, -INFINITY, 1,
                        out_boxes, out_scores, cls_tmp, out_valid);  /* :158-164 */
  free(cls_tmp);
  free(idx);
  free(pre_scores);
  free(pre_boxes);
  free(dec);
  return rc;
}

/* models/faster_rcnn.py:274-293 */
int orc_roi_pooling(const float *fm, int B, int H, int W, int C, const float *rois, int R,
                    int ph, int pw, float *out) {
  int32_t *ind = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * R);
  int b, r, rc;
  for (b = 0; b < B; ++b)
    for (r = 0; r < R; ++r) ind[(size_t)b * R + r] = b; /* :280-281 */
  rc = orc_crop_and_resize(fm, B, H, W, C, rois, ind, B * R, ph, pw, 0.0f, out);
  free(ind);
  return rc;
}

/* models/faster_rcnn.py:74-91 */
int orc_decoder(const float *rois, const float *deltas, const float *probs,
                const float *variances, int B, int R, int L, int max_total, float score_thr,
                float iou_thr, float *out_boxes, float *out_labels, float *out_scores,
                int32_t *out_valid) {
  size_t nrl = (size_t)B * R * L;
  float *exp_rois = (float *)malloc(sizeof(float) * nrl * 4);
  float *pred = (float *)malloc(sizeof(float) * nrl * 4);
  float *lab = (float *)malloc(sizeof(float) * nrl);
  int b, r, l, rc;
  for (b = 0; b < B; ++b)
    for (r = 0; r < R; ++r) {
      const float *p = probs + ((size_t)b * R + r) * L;
      int am = 0;
      float best = -FLT_MAX; /* [TF-recall] Eigen ArgMax: strict >, starts from lowest() */
      for (l = 0; l < L; ++l) {
        memcpy(exp_rois + (((size_t)b * R + r) * L + l) * 4, rois + ((size_t)b * R + r) * 4, 16); /* :80 */
        if (p[l] > best) {
          best = p[l];
          am = l;
        }
      }
      for (l = 0; l < L; ++l) lab[((size_t)b * R + r) * L + l] = (am != 0) ? p[l] : 0.0f; /* :82-83 */
    }
  orc_decode(exp_rois, 1, deltas, variances, (int64_t)B * R, L, pred); /* :78-81 (per (b,r): L boxes) */
  rc = orc_combined_nms(pred, lab, B, R, L, L, max_total, max_total, iou_thr, score_thr, 1,
                        out_boxes, out_scores, out_labels, out_valid); /* :84-90 */
  free(lab);
  free(pred);
  free(exp_rois);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * RPN target assignment
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t key;
  int32_t idx;
} isort_item;

static int cmp_iitem_desc(const void *pa, const void *pb) {
  const isort_item *a = (const isort_item *)pa;
  const isort_item *b = (const isort_item *)pb;
  if (a->key != b->key) return (a->key > b->key) ? -1 : 1;
  if (a->idx != b->idx) return (a->idx < b->idx) ? -1 : 1;
  return 0;
}

/* utils/train_utils.py:82-88 for one row.  [TF-recall] argsort(DESCENDING) == top_k(full row):
 * ties ascending index; the second argsort inverts the permutation (rank of each element). */
static void select_row(const uint8_t *mask, const int32_t *rnd, int32_t select, int N, uint8_t *out) {
  isort_item *it = (isort_item *)malloc(sizeof(isort_item) * (size_t)(N > 0 ? N : 1));
  int n;
  for (n = 0; n < N; ++n) {
    it[n].key = (mask[n] ? 1 : 0) * rnd[n]; /* :84 */
    it[n].idx = n;
  }
  qsort(it, (size_t)N, sizeof(isort_item), cmp_iitem_desc); /* :85 */
  for (n = 0; n < N; ++n) {
    int i = it[n].idx; /* rank of element i is n (:86) */
    out[i] = (uint8_t)((mask[i] != 0) && (n < select)); /* :87-88 */
  }
  free(it);
}

void orc_select_mask(const uint8_t *mask, const int32_t *rnd, const int32_t *select,
                     int n_select, int B, int N, uint8_t *out) {
  int b;
  for (b = 0; b < B; ++b)
    select_row(mask + (size_t)b * N, rnd + (size_t)b * N, select[n_select == 1 ? 0 : b], N,
               out + (size_t)b * N);
}

/* utils/train_utils.py:170-202 */
void orc_rpn_targets(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                     const int32_t *rnd_pos, const int32_t *rnd_neg, int B, int N, int M,
                     int total_pos, int total_neg, const float *variances, float *out_deltas,
                     float *out_labels, float *dbg_max_iou, int32_t *dbg_argmax_row,
                     int32_t *dbg_argmax_col) {
  int b;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    float *iou = (float *)malloc(sizeof(float) * (size_t)N * (M > 0 ? M : 1));
    float *maxiou = (float *)malloc(sizeof(float) * (size_t)N);
    int32_t *arow = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    int32_t *acol = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    uint8_t *pos = (uint8_t *)malloc((size_t)N);
    uint8_t *pos_sel = (uint8_t *)malloc((size_t)N);
    uint8_t *neg = (uint8_t *)malloc((size_t)N);
    uint8_t *neg_sel = (uint8_t *)malloc((size_t)N);
    const float *g = gt_boxes + (size_t)b * M * 4;
    int n, m, pos_count = 0;
    int32_t neg_count;
    orc_iou_map(anchors, 0, g, 1, N, M, iou); /* :176 (single image slice) */
    for (n = 0; n < N; ++n) { /* :177,:179  [TF-recall] first maximal index, strict > from lowest() */
      float best = -FLT_MAX, mx = -FLT_MAX;
      int am = 0;
      for (m = 0; m < M; ++m) {
        float v = iou[(size_t)n * M + m];
        if (v > best) {
          best = v;
          am = m;
        }
        if (v > mx) mx = v;
      }
      arow[n] = am;
      maxiou[n] = mx;
    }
    for (m = 0; m < M; ++m) { /* :178 */
      float best = -FLT_MAX;
      int am = 0;
      for (n = 0; n < N; ++n) {
        float v = iou[(size_t)n * M + m];
        if (v > best) {
          best = v;
          am = n;
        }
      }
      acol[m] = am;
    }
    for (n = 0; n < N; ++n) pos[n] = (uint8_t)(maxiou[n] > 0.7f); /* :180 */
    for (m = 0; m < M; ++m) /* :181-188 */
      if (gt_labels[(size_t)b * M + m] != -1) pos[acol[m]] = 1;
    select_row(pos, rnd_pos + (size_t)b * N, total_pos, N, pos_sel); /* :189 */
    for (n = 0; n < N; ++n) pos_count += pos_sel[n]; /* :190 */
    neg_count = (total_pos + total_neg) - pos_count; /* :191 */
    for (n = 0; n < N; ++n) neg[n] = (uint8_t)((maxiou[n] < 0.3f) && !pos_sel[n]); /* :193 */
    select_row(neg, rnd_neg + (size_t)b * N, neg_count, N, neg_sel); /* :194 */
    for (n = 0; n < N; ++n) {
      float pl = pos_sel[n] ? 1.0f : -1.0f; /* :195 */
      float nl = neg_sel[n] ? 1.0f : 0.0f;  /* :196 */
      float gsel[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      out_labels[(size_t)b * N + n] = pl + nl; /* :197 */
      if (pos_sel[n] && M > 0) memcpy(gsel, g + (size_t)arow[n] * 4, 16); /* :198-199 */
      encode_one(anchors + (size_t)n * 4, gsel, variances, out_deltas + ((size_t)b * N + n) * 4); /* :200 */
    }
    if (dbg_max_iou) memcpy(dbg_max_iou + (size_t)b * N, maxiou, sizeof(float) * (size_t)N);
    if (dbg_argmax_row) memcpy(dbg_argmax_row + (size_t)b * N, arow, sizeof(int32_t) * (size_t)N);
    if (dbg_argmax_col && M > 0) memcpy(dbg_argmax_col + (size_t)b * M, acol, sizeof(int32_t) * (size_t)M);
    free(neg_sel);
    free(neg);
    free(pos_sel);
    free(pos);
    free(acol);
    free(arow);
    free(maxiou);
    free(iou);
  }
}

/* ------------------------------------------------------------------------------------------
 * RoIDelta -- models/faster_rcnn.py:206-235 (second-stage training targets).
 * roi_bboxes (B,R,4); gt_boxes (B,M,4); gt_labels (B,M) int32; rnd_pos / rnd_neg (B,R) int32 are the
 * random integers of the two randomly_select_xyz_mask calls (:219, :221; utils/train_utils.py:82-83).
 * out_deltas (B, R*L, 4); out_labels (B,R,L) one-hot fp32.
 * ---------------------------------------------------------------------------------------- */
void orc_roi_delta(const float *rois, const float *gt_boxes, const int32_t *gt_labels,
                   const int32_t *rnd_pos, const int32_t *rnd_neg, int B, int R, int M, int L,
                   int total_pos, int total_neg, const float *variances, float *out_deltas,
                   float *out_labels) {
  int b;
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
  for (b = 0; b < B; ++b) {
    float *iou = (float *)malloc(sizeof(float) * (size_t)(R > 0 ? R : 1) * (M > 0 ? M : 1));
    float *maxiou = (float *)malloc(sizeof(float) * (size_t)(R > 0 ? R : 1));
    int32_t *arow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    uint8_t *pos = (uint8_t *)malloc((size_t)(R > 0 ? R : 1));
    uint8_t *pos_sel = (uint8_t *)malloc((size_t)(R > 0 ? R : 1));
    uint8_t *neg = (uint8_t *)malloc((size_t)(R > 0 ? R : 1));
    uint8_t *neg_sel = (uint8_t *)malloc((size_t)(R > 0 ? R : 1));
    const float *g = gt_boxes + (size_t)b * M * 4;
    const float *rb = rois + (size_t)b * R * 4;
    int r, m, l, c;
    orc_iou_map(rb, 0, g, 1, R, M, iou); /* :215 (this image's slice; bboxes are per image here) */
    for (r = 0; r < R; ++r) {            /* :216-217  [TF-recall] first maximal index, strict > from lowest(); NaN never wins */
      float best = -FLT_MAX, mx = -FLT_MAX;
      int am = 0;
      for (m = 0; m < M; ++m) {
        float v = iou[(size_t)r * M + m];
        if (v > best) {
          best = v;
          am = m;
        }
        if (v > mx) mx = v;
      }
      arow[r] = am;
      maxiou[r] = mx;
    }
    for (r = 0; r < R; ++r) pos[r] = (uint8_t)(maxiou[r] > 0.5f);                      /* :218 */
    select_row(pos, rnd_pos + (size_t)b * R, total_pos, R, pos_sel);                   /* :219 */
    for (r = 0; r < R; ++r) neg[r] = (uint8_t)((maxiou[r] < 0.5f) && (maxiou[r] > 0.1f)); /* :220 */
    select_row(neg, rnd_neg + (size_t)b * R, total_neg, R, neg_sel);                   /* :221 */
    for (r = 0; r < R; ++r) {
      float gsel[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      float d[4];
      int32_t lab;
      if (pos_sel[r]) memcpy(gsel, g + (size_t)arow[r] * 4, 16);                         /* :222-223 */
      lab = (pos_sel[r] ? gt_labels[(size_t)b * M + arow[r]] : -1) + (neg_sel[r] ? 1 : 0); /* :224-229 */
      encode_one(rb + (size_t)r * 4, gsel, variances, d);                               /* :230 */
      for (l = 0; l < L; ++l) {
        float oh = (lab == l) ? 1.0f : 0.0f; /* :231 one_hot: out-of-range labels give an all-zero row */
        out_labels[((size_t)b * R + r) * L + l] = oh;
        for (c = 0; c < 4; ++c) /* :232-234 the scatter is a multiply: 0 * delta keeps delta's sign / NaN */
          out_deltas[(((size_t)b * R + r) * L + l) * 4 + c] = oh * d[c];
      }
    }
    free(neg_sel);
    free(neg);
    free(pos_sel);
    free(pos);
    free(arow);
    free(maxiou);
    free(iou);
  }
}
