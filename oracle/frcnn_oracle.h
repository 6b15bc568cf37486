/*
 * frcnn_oracle.h -- CPU restatement (plain C99) of the reference's proposal/RoI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" against real TensorFlow 2.0.  The reference
 * (FurkanOM/tf-faster-rcnn) is pure Python that delegates the arithmetic to the
 * un-vendored third-party dependency tensorflow==2.0.0 (environment.yml:49-52), which is
 * not installed here, and none of the reference's own tests executes the hot path
 * (SURVEY.md section 4, 8c).  Each function below cites the reference file:line it
 * follows; TensorFlow-kernel behaviour is restated from the published algorithm of the
 * named TF op and is marked [TF-recall].
 *
 * Every floating point operation is a single, separately rounded IEEE-754 binary32
 * operation, in the reference's order (build with -ffp-contract=off, no fast-math).
 */
#ifndef FRCNN_ORACLE_H
#define FRCNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* number of OpenMP threads used over images (1 = scalar port). */
void orc_set_threads(int n);
int orc_get_threads(void);

/* [TF-recall] Eigen pexp<float>/plog<float> (Cephes), non-FMA build; tf.exp / tf.math.log. */
float orc_expf(float x);
float orc_logf(float x);

/* utils/bbox_utils.py:13-34 generate_base_anchors. out: (ns*nr, 4). */
void orc_base_anchors(double img_size, const double *ratios, int nr, const double *scales,
                      int ns, float *out);
/* utils/bbox_utils.py:37-57 generate_anchors. base: (A,4); out: (G*G*A, 4). */
void orc_generate_anchors(int G, const float *base, int A, float *out);

/* utils/bbox_utils.py:86-108 get_bboxes_from_deltas.
 * anchors: (N,4) when anchors_batched==0 else (B,N,4); deltas (B,N,4); variances: 4 floats
 * multiplied into the deltas first (models/faster_rcnn.py:149,79) or NULL. out (B,N,4). */
void orc_decode(const float *anchors, int anchors_batched, const float *deltas,
                const float *variances, int64_t B, int64_t N, float *out);

/* utils/bbox_utils.py:111-135 get_deltas_from_bboxes; bboxes (N,4) or (B,N,4), gt (B,N,4).
 * inv_variances: if non-NULL the result is divided by these 4 floats (train_utils.py:200). */
void orc_encode(const float *bboxes, int bboxes_batched, const float *gt,
                const float *div_variances, int64_t B, int64_t N, float *out);

/* utils/bbox_utils.py:138-160 generate_iou_map; bboxes (N,4) or (B,N,4); gt (B,M,4); out (B,N,M). */
void orc_iou_map(const float *bboxes, int bboxes_batched, const float *gt, int B, int N, int M,
                 float *out);

/* [TF-recall] tf.nn.top_k(sorted=True): value desc, ties lower index first.
 * scores (B,N) -> vals (B,K), idx (B,K). Returns 0, or -1 if K>N. */
int orc_top_k(const float *scores, int B, int N, int K, float *vals, int32_t *idx);

/* [TF-recall] tf.image.combined_non_max_suppression (pad_per_class=False).
 * boxes (B,n,q,4), scores (B,n,c); outputs (B,T,4),(B,T),(B,T),(B), T=max_total.
 * Tie rule (unspecified by TF, defined by this build): equal scores pop in ascending box
 * index; the cross-class merge orders equal scores by (class asc, selection order). */
int orc_combined_nms(const float *boxes, const float *scores, int B, int n, int q, int c,
                     int max_per_class, int max_total, float iou_thr, float score_thr,
                     int clip_boxes, float *out_boxes, float *out_scores, float *out_classes,
                     int32_t *out_valid);

/* [TF-recall] tf.image.crop_and_resize(method=bilinear). image (B,H,W,C) NHWC; boxes (n,4);
 * box_ind (n); out (n,ch,cw,C). Returns -1 on a box index outside [0,B). */
int orc_crop_and_resize(const float *image, int B, int H, int W, int C, const float *boxes,
                        const int32_t *box_ind, int n, int ch, int cw, float extrapolation,
                        float *out);

/* models/faster_rcnn.py:130-166 RoIBBox.call.  deltas (B,N,4), scores (B,N), anchors (N,4).
 * out_boxes (B,R,4) (clipped, zero padded). Optional out_scores (B,R), out_valid (B) may be NULL. */
int orc_roi_bbox(const float *deltas, const float *scores, const float *anchors,
                 const float *variances, int B, int N, int K, int R, float iou_thr,
                 float *out_boxes, float *out_scores, int32_t *out_valid);

/* models/faster_rcnn.py:264-293 RoIPooling.call. fm (B,H,W,C); rois (B,R,4); out (B,R,ph,pw,C). */
int orc_roi_pooling(const float *fm, int B, int H, int W, int C, const float *rois, int R,
                    int ph, int pw, float *out);

/* models/faster_rcnn.py:63-91 Decoder.call. rois (B,R,4); deltas (B,R,L*4); probs (B,R,L).
 * outputs boxes (B,T,4), labels (B,T) fp32, scores (B,T), valid (B) (valid may be NULL). */
int orc_decoder(const float *rois, const float *deltas, const float *probs,
                const float *variances, int B, int R, int L, int max_total, float score_thr,
                float iou_thr, float *out_boxes, float *out_labels, float *out_scores,
                int32_t *out_valid);

/* utils/train_utils.py:72-88 randomly_select_xyz_mask with the random integers given.
 * mask (B,N) uint8; rnd (B,N) int32; select: n_select ints (1 or B). out (B,N) uint8. */
void orc_select_mask(const uint8_t *mask, const int32_t *rnd, const int32_t *select,
                     int n_select, int B, int N, uint8_t *out);

/* utils/train_utils.py:149-202 calculate_rpn_actual_outputs with the random integers of both
 * sampling calls given. anchors (N,4); gt_boxes (B,M,4); gt_labels (B,M) int32 (-1 = pad).
 * out_deltas (B,N,4); out_labels (B,N) fp32 in {-1,0,1} (caller reshapes to (B,G,G,A)).
 * Optional debug outputs (may be NULL): max_iou (B,N), argmax_row (B,N), argmax_col (B,M). */
void orc_rpn_targets(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                     const int32_t *rnd_pos, const int32_t *rnd_neg, int B, int N, int M,
                     int total_pos, int total_neg, const float *variances, float *out_deltas,
                     float *out_labels, float *dbg_max_iou, int32_t *dbg_argmax_row,
                     int32_t *dbg_argmax_col);

/* models/faster_rcnn.py:206-235 (RoIDelta.call).  rois (B,R,4); gt_boxes (B,M,4), M >= 1; gt_labels (B,M) int32;
 * rnd_pos / rnd_neg (B,R) int32: the random integers of the two sampling calls.
 * out_deltas (B,R*L,4); out_labels (B,R,L) one-hot fp32. */
void orc_roi_delta(const float *rois, const float *gt_boxes, const int32_t *gt_labels,
                   const int32_t *rnd_pos, const int32_t *rnd_neg, int B, int R, int M, int L,
                   int total_pos, int total_neg, const float *variances, float *out_deltas,
                   float *out_labels);

#ifdef __cplusplus
}
#endif
#endif
