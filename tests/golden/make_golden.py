"""Generate the committed golden vectors of the hot path (small seeded cases) with the CPU oracle.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

The reference cannot run here (TensorFlow is not installed) and ships no fixture for this path
(SURVEY.md 4, 8c), so these vectors pin the BUILD's behaviour (oracle == HIP path), not TensorFlow's:
"parity unpinned" against TF.  Hand-derivable known answers live in tests/test_oracle.py instead.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
import synth  # noqa: E402


def main() -> None:
    B, G, C, L, K, R, M = 2, 8, 8, 5, 64, 16, 4
    hp = synth.hyper_params(feature_map_shape=G, pre_nms_topn=K, test_nms_topn=R)
    anchors = oracle.generate_anchors(hp)
    deltas, scores = synth.rpn_outputs(0, B, G)
    fm = synth.feature_map(0, B, G, C)
    hd, hpb = synth.head_outputs(0, B, R, L)
    rois, roi_scores, roi_valid = oracle.roi_bbox(deltas, scores, anchors, hp["variances"], K, R, 0.7, return_all=True)
    pooled = oracle.roi_pooling(fm, rois, hp["pooling_size"])
    det_boxes, det_labels, det_scores, det_valid = oracle.decoder(rois, hd, hpb, hp["variances"], L, return_valid=True)
    np.savez_compressed(os.path.join(HERE, "inference_path.npz"), anchors=anchors, deltas=deltas, scores=scores, fm=fm,
                        head_deltas=hd, head_probs=hpb, rois=rois, roi_scores=roi_scores, roi_valid=roi_valid,
                        pooled=pooled, det_boxes=det_boxes, det_labels=det_labels, det_scores=det_scores,
                        det_valid=det_valid)
    gt, labels = synth.gt_boxes(0, B, M)
    rp, rn = synth.sampling_ints(0, B, anchors.shape[0])
    hp_t = dict(hp, total_pos_bboxes=8, total_neg_bboxes=8)
    tdel, tlab, miou, arow, acol = oracle.calculate_rpn_actual_outputs(anchors, gt, labels, hp_t, (rp, rn), debug=True)
    iou = oracle.generate_iou_map(anchors, gt)
    np.savez_compressed(os.path.join(HERE, "rpn_targets.npz"), anchors=anchors, gt_boxes=gt, gt_labels=labels, rnd_pos=rp,
                        rnd_neg=rn, iou_map=iou, max_iou=miou, argmax_row=arow, argmax_col=acol, bbox_deltas=tdel,
                        bbox_labels=tlab)
    # RoIDelta: RoIs = the proposals above plus jittered copies of the gts (so positives and negatives exist)
    Lr, Rr = 5, 24
    rng = np.random.default_rng(3)
    rd_rois = np.concatenate([rois, np.clip(np.repeat(gt, 2, axis=1) + rng.normal(0, 0.05, (B, 2 * M, 4)), 0, 1)], 1).astype(np.float32)
    rd_rois[:, Rr - 2:] = 0.0
    rd_lab = np.where(labels >= 0, labels % (Lr - 1) + 1, labels).astype(np.int32)
    rdp = rng.integers(1, 40, (B, Rr), dtype=np.int32)
    rdn = rng.integers(1, 40, (B, Rr), dtype=np.int32)
    hp_r = dict(hp, total_labels=Lr, total_pos_bboxes=4, total_neg_bboxes=4)
    rdd, rdl = oracle.roi_delta(rd_rois, gt, rd_lab, hp_r, (rdp, rdn))
    np.savez_compressed(os.path.join(HERE, "roi_delta.npz"), rois=rd_rois, gt_boxes=gt, gt_labels=rd_lab, rnd_pos=rdp,
                        rnd_neg=rdn, roi_bbox_deltas=rdd, roi_bbox_labels=rdl)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
