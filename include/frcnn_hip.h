/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) proposal/RoI hot path of
 * FurkanOM/tf-faster-rcnn, hand-written HIP.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - tensors are contiguous, row-major, fp32 / int32 / uint8 as declared;
 *   - boxes are [y1, x1, y2, x2], feature maps NHWC (the reference's layouts);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every entry point
 *     only enqueues work on that stream: no allocation, no synchronisation;
 *   - scratch memory comes from the caller: `*_workspace_bytes()` says how much, the pointer
 *     must be 256-byte aligned;
 *   - return value: 0 = ok, FRCNN_EINVAL / FRCNN_EUNSUPPORTED / FRCNN_EHIP otherwise, with a
 *     thread-local message available from frcnn_last_error().  No exception crosses the ABI.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  The reference has no FFI: its boundary is the Python
 * functions/Keras layers cited, which delegate to TensorFlow 2.0 ops.
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_EINVAL (-1)       /* bad shape / argument (TF: InvalidArgumentError) */
#define FRCNN_EUNSUPPORTED (-2) /* size beyond what the kernels are built for */
#define FRCNN_EHIP (-3)         /* HIP runtime error */

typedef void *frcnn_stream_t;

/* Library version (major*10000 + minor*100 + patch). */
int frcnn_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char *frcnn_last_error(void);

/* Diagnostic: force one of several equivalent schedules (same results; used by the parity tests to cover all of them).
 * `name` is one of the FRCNN_* knob names listed in DESIGN.md section 7; is_set == 0 restores the default.  The library
 * reads the environment variables of the same names once, when it is loaded, and never calls getenv afterwards.  No
 * counterpart in the reference.  Not for use concurrently with other calls. */
int frcnn_debug_set_schedule(const char *name, int is_set, int value);

/* ---- anchors ------------------------------------------------------------------------- */
/* utils/bbox_utils.py:37-57 generate_anchors: broadcast add of the base anchors (A,4)
 * (utils/bbox_utils.py:13-34, computed by the host in fp64/fp32 as the reference does) over the
 * G x G cell centres `grid_coords` (G) and clip to [0,1].  out: (G*G*A, 4). */
int frcnn_generate_anchors(const float *base_anchors, int A, const float *grid_coords, int G,
                           float *out, frcnn_stream_t stream);

/* ---- box codec ----------------------------------------------------------------------- */
/* utils/bbox_utils.py:86-108 get_bboxes_from_deltas.  anchors: (N,4) shared by every batch row
 * when anchors_batched == 0, else (B,N,4).  deltas, out: (B,N,4).  variances: host pointer to
 * 4 floats multiplied into the deltas first (models/faster_rcnn.py:149 / :79), or NULL. */
int frcnn_decode_boxes(const float *anchors, int anchors_batched, const float *deltas,
                       const float *variances, int64_t B, int64_t N, float *out,
                       frcnn_stream_t stream);

/* utils/bbox_utils.py:111-135 get_deltas_from_bboxes.  bboxes (N,4) or (B,N,4); gt_boxes, out
 * (B,N,4).  div_variances: host pointer to 4 floats the result is divided by
 * (utils/train_utils.py:200), or NULL. */
int frcnn_encode_boxes(const float *bboxes, int bboxes_batched, const float *gt_boxes,
                       const float *div_variances, int64_t B, int64_t N, float *out,
                       frcnn_stream_t stream);

/* ---- IoU ----------------------------------------------------------------------------- */
/* utils/bbox_utils.py:138-160 generate_iou_map.  bboxes (N,4) or (B,N,4); gt (B,M,4);
 * out (B,N,M), materialised. */
int frcnn_iou_map(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B, int N,
                  int M, float *out, frcnn_stream_t stream);

/* utils/train_utils.py:176-179: the IoU map fused with its three reductions, map never
 * written.  max_iou (B,N) fp32, argmax_row (B,N) int32 (best gt per box), argmax_col (B,M)
 * int32 (best box per gt).  Any output may be NULL.  Workspace: frcnn_iou_reduce_workspace_bytes. */
size_t frcnn_iou_reduce_workspace_bytes(int B, int N, int M);
int frcnn_iou_map_reduce(const float *bboxes, int bboxes_batched, const float *gt_boxes, int B,
                         int N, int M, float *max_iou, int32_t *argmax_row, int32_t *argmax_col,
                         void *workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* ---- top-k --------------------------------------------------------------------------- */
/* tf.nn.top_k(sorted=True) as used at models/faster_rcnn.py:153 and rpn_predictor.py:64:
 * per row, K largest, descending, equal values by ascending index.  scores (B,N);
 * out_values (B,K) and/or out_indices (B,K) (either may be NULL).  K > N is FRCNN_EINVAL. */
int frcnn_top_k(const float *scores, int B, int N, int K, float *out_values,
                int32_t *out_indices, frcnn_stream_t stream);

/* ---- NMS ----------------------------------------------------------------------------- */
/* utils/bbox_utils.py:60-83 non_max_suppression == tf.image.combined_non_max_suppression
 * (pad_per_class=False).  boxes (B,n,q,4) with q==1 or q==c; scores (B,n,c).
 * out_boxes (B,T,4), out_scores (B,T), out_classes (B,T) fp32, out_valid (B) int32, T=max_total.
 * out_scores / out_classes / out_valid may be NULL. */
size_t frcnn_combined_nms_workspace_bytes(int B, int n, int q, int c, int max_per_class,
                                          int max_total);
int frcnn_combined_nms(const float *boxes, const float *scores, int B, int n, int q, int c,
                       int max_per_class, int max_total, float iou_threshold,
                       float score_threshold, int clip_boxes, float *out_boxes,
                       float *out_scores, float *out_classes, int32_t *out_valid,
                       void *workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* ---- layers -------------------------------------------------------------------------- */
/* models/faster_rcnn.py:130-166 RoIBBox.call, fused: variance scale + decode + per-image top-K
 * + single-class greedy NMS + clip + zero pad.  deltas (B,N,4), scores (B,N), anchors (N,4),
 * variances: host, 4 floats.  out_boxes (B,R,4); out_scores (B,R) and out_valid (B) optional.
 * Schedules (same results): one workgroup per image for R < 512, the whole-chip bitmask NMS from
 * there on (training's train_nms_topn); the latter keeps its tiles in the workspace, which is why the
 * workspace query grows with K*R (about 3 MB per image at K=6000, R=2000). */
size_t frcnn_roi_bbox_workspace_bytes(int B, int N, int K, int R);
int frcnn_roi_bbox(const float *deltas, const float *scores, const float *anchors,
                   const float *variances, int B, int N, int K, int R, float iou_threshold,
                   float *out_boxes, float *out_scores, int32_t *out_valid, void *workspace,
                   size_t workspace_bytes, frcnn_stream_t stream);

/* frcnn_roi_bbox for `groups` (1..4) batches of one shape in ONE launch: deltas / scores / out_boxes are HOST arrays
 * of `groups` device pointers.  No counterpart in the reference (which calls the layer once per batch,
 * models/faster_rcnn.py:318); for serving loops that keep several batches in flight (RoIPathPipeline): the one
 * workgroup per image of `groups` batches become resident together.  Same results as `groups` frcnn_roi_bbox calls;
 * schedules without a grouped form are executed that way, with the workgroup sizes that can become resident beside a
 * running crop launch (the whole-chip bitmask schedule's scans: 256 instead of 1024 threads). */
int frcnn_roi_bbox_multi(const float *const *deltas, const float *const *scores, const float *anchors,
                         const float *variances, int groups, int B, int N, int K, int R,
                         float iou_threshold, float *const *out_boxes, void *workspace,
                         size_t workspace_bytes, frcnn_stream_t stream);

/* tf.image.crop_and_resize(method="bilinear") as called at models/faster_rcnn.py:283-288.
 * image (B,H,W,C); boxes (n,4); box_ind (n) int32, or NULL meaning box i belongs to image
 * i / boxes_per_image (the RoIPooling layout, models/faster_rcnn.py:280-282).  out (n,ch,cw,C).
 * An index outside [0,B) yields extrapolation_value rows (TF raises; the host wrapper checks). */
int frcnn_crop_and_resize(const float *image, int B, int H, int W, int C, const float *boxes,
                          const int32_t *box_ind, int n, int boxes_per_image, int crop_h,
                          int crop_w, float extrapolation_value, float *out,
                          frcnn_stream_t stream);

/* models/faster_rcnn.py:264-293 RoIPooling.call as one entry point: RoI r of image b samples image b (:280-282),
 * bilinear crop_and_resize to (crop_h, crop_w) with extrapolation value 0 (:283-288).  feature_map (B,H,W,C); rois (B,R,4);
 * out (B,R,crop_h,crop_w,C).  Same results as frcnn_crop_and_resize(box_ind = NULL, boxes_per_image = R).  The optional
 * workspace (frcnn_roi_pooling_workspace_bytes; NULL = none) lets feature maps larger than an XCD's L2 (4 MiB) take the
 * source-row-ownership schedule: the (RoI, output row) items of every image are ordered by the source row they sample and
 * each XCD takes an eighth of that order, i.e. a band of source rows; every output is still written to its own slot. */
size_t frcnn_roi_pooling_workspace_bytes(int B, int R);
int frcnn_roi_pooling(const float *feature_map, int B, int H, int W, int C, const float *rois, int R,
                      int crop_h, int crop_w, float *out, void *workspace, size_t workspace_bytes,
                      frcnn_stream_t stream);

/* frcnn_roi_pooling in two calls, for callers that overlap batches on several streams (models/faster_rcnn.py:264-293, same
 * results).  The source-row-ownership schedule starts with a pre-pass that needs the RoIs only: frcnn_roi_pooling_prepare
 * runs it on `stream` -- typically the stream that produced the RoIs -- into `workspace` and sets *prepared = 1; the crop
 * itself is then frcnn_roi_pooling_prepared with the SAME arguments on any stream ordered after the first call, and
 * `workspace` must stay untouched in between (one workspace per batch in flight).  *prepared = 0 (nothing was launched)
 * means these arguments take another schedule: call frcnn_roi_pooling instead. */
int frcnn_roi_pooling_prepare(const float *feature_map, int B, int H, int W, int C, const float *rois, int R,
                              int crop_h, int crop_w, const float *out, void *workspace, size_t workspace_bytes,
                              int *prepared, frcnn_stream_t stream);
int frcnn_roi_pooling_prepared(const float *feature_map, int B, int H, int W, int C, const float *rois, int R,
                               int crop_h, int crop_w, float *out, void *workspace, size_t workspace_bytes,
                               frcnn_stream_t stream);
/* `groups` (1..4) prepared batches of one shape in ONE launch of the crop kernel (host arrays of `groups` device pointers;
 * every batch with the workspace its frcnn_roi_pooling_prepare call filled, each of workspace_bytes).  Same results as
 * `groups` frcnn_roi_pooling_prepared calls. */
int frcnn_roi_pooling_prepared_multi(const float *const *feature_maps, const float *const *rois, float *const *outs,
                                     void *const *workspaces, size_t workspace_bytes, int groups, int B, int H,
                                     int W, int C, int R, int crop_h, int crop_w, frcnn_stream_t stream);

/* The same operation (models/faster_rcnn.py:283-288, RoIPooling layout: box_ind == NULL) for `groups` (1..4)
 * independent batches of one shape in ONE launch: images / boxes / outs are HOST arrays of `groups` device pointers
 * ((B,H,W,C), (n,4), (n,ch,cw,C) each).  The reference calls the layer once per batch; a serving loop that has
 * several batches in flight (RoIPathPipeline) pays one launch tail and one inter-launch gap for all of them.
 * Results are those of `groups` frcnn_crop_and_resize calls; shapes the grouped launch is not built for are
 * executed that way -- for the RoIPooling layout through frcnn_roi_pooling, which is what the optional workspace
 * (frcnn_roi_pooling_workspace_bytes(B, boxes_per_image); NULL = none) is handed to. */
int frcnn_crop_and_resize_multi(const float *const *images, const float *const *boxes,
                                float *const *outs, int groups, int B, int H, int W, int C, int n,
                                int boxes_per_image, int crop_h, int crop_w,
                                float extrapolation_value, void *workspace, size_t workspace_bytes,
                                frcnn_stream_t stream);

/* models/faster_rcnn.py:63-91 Decoder.call, fused: per-class variance scale + decode,
 * background-argmax masking, per-class NMS, cross-class top-T.  rois (B,R,4); deltas (B,R,L*4);
 * probs (B,R,L).  out_boxes (B,T,4), out_labels (B,T) fp32, out_scores (B,T), out_valid (B)
 * optional. */
size_t frcnn_decoder_workspace_bytes(int B, int R, int L, int max_total);
int frcnn_decoder(const float *rois, const float *deltas, const float *probs,
                  const float *variances, int B, int R, int L, int max_total,
                  float score_threshold, float iou_threshold, float *out_boxes,
                  float *out_labels, float *out_scores, int32_t *out_valid, void *workspace,
                  size_t workspace_bytes, frcnn_stream_t stream);

/* frcnn_decoder for `groups` (1..4) batches of one shape in ONE launch (HOST arrays of device pointers), the
 * Decoder-side companion of frcnn_roi_bbox_multi / frcnn_crop_and_resize_multi (models/faster_rcnn.py:361-365 is
 * called once per batch in the reference).  Same results as `groups` frcnn_decoder calls. */
int frcnn_decoder_multi(const float *const *rois, const float *const *deltas,
                        const float *const *probs, const float *variances, int groups, int B, int R,
                        int L, int max_total, float score_threshold, float iou_threshold,
                        float *const *out_boxes, float *const *out_labels, float *const *out_scores,
                        void *workspace, size_t workspace_bytes, frcnn_stream_t stream);

/* ---- RPN target assignment ----------------------------------------------------------- */
/* utils/train_utils.py:72-88 randomly_select_xyz_mask with the random integers supplied by the
 * caller (the reference draws them from TF's unseeded Philox stream).  mask (B,N) uint8;
 * random_mask (B,N) int32; select: device int32, n_select == 1 or B entries.  out (B,N) uint8. */
int frcnn_select_mask(const uint8_t *mask, const int32_t *random_mask, const int32_t *select,
                      int n_select, int B, int N, uint8_t *out, frcnn_stream_t stream);
/* The same call (utils/train_utils.py:72-88) with the random integers drawn in-kernel: U[1, 10 * max(select)) as :82-83,
 * from a counter-based hash of (seed, image, element); the maximum over `select` is taken on the device, so nothing is
 * read back to the host.  Where tf.random.uniform raises (10 * max(select) <= 1) every `select` is <= 0 and the output is
 * all zero. */
int frcnn_select_mask_seeded(const uint8_t *mask, const int32_t *select, int n_select, uint64_t seed, int B,
                             int N, uint8_t *out, frcnn_stream_t stream);

/* utils/train_utils.py:149-202 calculate_rpn_actual_outputs.  anchors (N,4); gt_boxes (B,M,4);
 * gt_labels (B,M) int32 (-1 = padding).  rnd_pos / rnd_neg: (B,N) int32 random integers of the
 * two sampling calls, or both NULL to draw them in-kernel from `seed` (counter-based hash,
 * U[1, 10*total_pos) and U[1, 10*(total_pos+total_neg))).  out_deltas (B,N,4); out_labels (B,N)
 * fp32 in {-1,0,1} (== (B,G,G,A) in memory). */
size_t frcnn_rpn_targets_workspace_bytes(int B, int N, int M);
int frcnn_rpn_targets(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                      const int32_t *rnd_pos, const int32_t *rnd_neg, uint64_t seed, int B, int N,
                      int M, int total_pos, int total_neg, const float *variances,
                      float *out_deltas, float *out_labels, void *workspace,
                      size_t workspace_bytes, frcnn_stream_t stream);

/* frcnn_rpn_targets without a workspace argument and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving / feeder loops, harnesses that
 * forbid hipMalloc).  The scratch (frcnn_rpn_targets_workspace_bytes(B,N,M) bytes: max IoU, both argmaxes, column keys)
 * lives in a file-scope device array of FRCNN_RPN_TARGETS_STATIC_MAX_BYTES that every GPU the library runs on holds until
 * the process ends; shapes needing more are FRCNN_EUNSUPPORTED.  All calls on one GPU share that array: calls that
 * overlap on one GPU (different streams, different threads) race -- serialise them on one stream.  Same results as
 * frcnn_rpn_targets.
 * Graph capture: the FIRST call on a device resolves the array's address (hipGetSymbolAddress) and may raise the kernels'
 * dynamic-LDS limit (hipFuncSetAttribute); neither is legal while a stream is capturing.  Make one eager call per device,
 * with the largest N and M that will be captured, before capturing; later calls issue kernel launches and one
 * hipMemsetAsync only. */
#define FRCNN_RPN_TARGETS_STATIC_MAX_BYTES ((size_t)8 << 20) /* covers B=64, N=9216, M=100 (4.8 MB) and B=16, N=35721 */
int frcnn_rpn_targets_static(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                             const int32_t *rnd_pos, const int32_t *rnd_neg, uint64_t seed, int B, int N,
                             int M, int total_pos, int total_neg, const float *variances,
                             float *out_deltas, float *out_labels, frcnn_stream_t stream);

/* models/faster_rcnn.py:195-235 RoIDelta.call (second-stage training targets): RoI x gt IoU (:215), row max /
 * first argmax (:216-217), positives > 0.5 and negatives in (0.1, 0.5) sub-sampled to total_pos / total_neg
 * (:218-221), gathered gt box / label (:222-229), encode / variances (:230), one-hot labels and the per-class
 * delta "scatter" (:231-234).  rois (B,R,4); gt_boxes (B,M,4), M >= 1; gt_labels (B,M) int32.  rnd_pos / rnd_neg:
 * (B,R) int32 random integers of the two sampling calls, or both NULL to draw them in-kernel from `seed`.
 * out_deltas (B,R*L,4); out_labels (B,R,L) one-hot fp32. */
size_t frcnn_roi_delta_workspace_bytes(int B, int R, int M);
int frcnn_roi_delta(const float *rois, const float *gt_boxes, const int32_t *gt_labels,
                    const int32_t *rnd_pos, const int32_t *rnd_neg, uint64_t seed, int B, int R,
                    int M, int total_labels, int total_pos, int total_neg, const float *variances,
                    float *out_deltas, float *out_labels, void *workspace, size_t workspace_bytes,
                    frcnn_stream_t stream);

/* ---- host gather ----------------------------------------------------------------------- */
/* Copy `bytes` bytes from device memory to `dst`, which may be pinned host memory mapped into the device (hipHostMalloc):
 * the "final host gather" of the small per-image outputs in a deployment sharded over GPUs (there is no counterpart in the
 * reference, which is single-device; Keras' predict() returns host arrays: faster_rcnn_predictor.py:56).  A kernel on
 * `stream`, not a copy-engine transfer: it orders like any other stage.  Both pointers 16-byte aligned. */
int frcnn_gather_copy(const void *src, void *dst, size_t bytes, frcnn_stream_t stream);
/* The same for n (0..8) separate tensors in ONE launch: srcs / dsts / bytes are HOST arrays of n entries; every segment
 * 16-byte aligned and a multiple of 16 bytes.  (RoIs, boxes, labels and scores of a batch are four tensors.) */
int frcnn_gather_copy_multi(const void *const *srcs, void *const *dsts, const size_t *bytes, int n,
                            frcnn_stream_t stream);

/* ---- host gather runtime ----------------------------------------------------------------- */
/* The final host gather of a batch sharded over the GPUs of one node, one object per rank (no counterpart in the
 * reference: it is single-device; Keras' predict() returns host arrays, faster_rcnn_predictor.py:56).  A step's small
 * outputs travel in two hops: frcnn_host_gather_submit launches ONE copy kernel on `stream` (srcs[i] -> stage[i], device
 * memory -> this rank's pinned staging, the frcnn_gather_copy_multi rules: at most 8 segments, 16-byte aligned, multiples
 * of 16 bytes), records the slot's event behind it and returns; a worker thread owned by the object waits for that event
 * and copies stage[i] -> dsts[i] (host memory, normally this rank's rows of a POSIX-shared buffer that every rank maps).
 * `slot` (0..slots-1) names the staging area the step uses: a submit blocks only while the job that last used the same
 * slot has not been published (back-pressure; a running loop that walks the slots round robin never waits).
 * frcnn_host_gather_flush blocks until every submitted job is in its destination.  These are the only entry points of
 * the library that own host resources (events, a thread) or may block; all of them are host-side, none is a stream
 * operation.  The device current at frcnn_host_gather_create owns the object; submit from one thread at a time. */
typedef struct frcnn_host_gather frcnn_host_gather_t;
int frcnn_host_gather_create(int slots, frcnn_host_gather_t **out);
int frcnn_host_gather_submit(frcnn_host_gather_t *h, int slot, const void *const *srcs, void *const *stage,
                             void *const *dsts, const size_t *bytes, int n, frcnn_stream_t stream);
int frcnn_host_gather_flush(frcnn_host_gather_t *h);
/* Diagnostics: seconds the worker spent copying, seconds submit spent blocked on back-pressure, jobs published.  Any
 * pointer may be NULL. */
int frcnn_host_gather_stats(frcnn_host_gather_t *h, double *publish_seconds, double *submit_wait_seconds,
                            long long *jobs);
/* Publishes what is queued, stops the worker, releases the events.  NULL is a no-op. */
int frcnn_host_gather_destroy(frcnn_host_gather_t *h);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
