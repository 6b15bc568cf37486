"""A second, independent restatement of the reference's hot-path arithmetic in NumPy fp32.

Used only to cross-check the C oracle (``oracle/frcnn_oracle.c``): two restatements written separately
from the reference's lines must agree bit for bit.  Every operation is a separately rounded float32
NumPy op in the reference's order.  Citations are relative to the reference repository root.
"""

from __future__ import annotations

import numpy as np

f32 = np.float32


def cephes_exp(x):
    """[TF-recall] Eigen pexp_float, non-FMA path (what tf.exp runs on CPU)."""
    x = np.asarray(x, dtype=f32)
    x0 = x
    x = np.minimum(x, f32(88.3762626647950))
    x = np.maximum(x, f32(-88.3762626647949))
    m = np.floor(x * f32(1.44269504088896341) + f32(0.5)).astype(f32)
    r = x - m * f32(0.693359375)
    r = r - m * f32(-2.12194440e-4)
    r2 = r * r
    y = np.full_like(r, f32(1.9875691500E-4))
    for c in (1.3981999507E-3, 8.3334519073E-3, 4.1665795894E-2, 1.6666665459E-1, 5.0000001201E-1):
        y = y * r + f32(c)
    y = y * r2 + r
    y = y + f32(1.0)
    scale = ((m.astype(np.int32) + 127).astype(np.uint32) << np.uint32(23)).view(f32)
    return np.maximum(y * scale, x0)


def cephes_log(x):
    """[TF-recall] Eigen plog_float, non-FMA path (positive finite inputs only)."""
    x = np.asarray(x, dtype=f32)
    x = np.maximum(x, f32(1.17549435e-38))
    bits = x.view(np.uint32)
    e = (bits >> np.uint32(23)).astype(np.int32).astype(f32) - f32(126.0)
    x = ((bits & np.uint32(0x807FFFFF)) | np.uint32(0x3F000000)).view(f32)
    lt = x < f32(0.707106781186547524)
    tmp = np.where(lt, x, f32(0))
    x = x - f32(1.0)
    e = e - np.where(lt, f32(1), f32(0))
    x = x + tmp
    x2 = x * x
    x3 = x2 * x
    y = f32(7.0376836292E-2) * x + f32(-1.1514610310E-1)
    y1 = f32(-1.2420140846E-1) * x + f32(1.4249322787E-1)
    y2 = f32(2.0000714765E-1) * x + f32(-2.4999993993E-1)
    y = y * x + f32(1.1676998740E-1)
    y1 = y1 * x + f32(-1.6668057665E-1)
    y2 = y2 * x + f32(3.3333331174E-1)
    y = y * x3 + y1
    y = y * x3 + y2
    y = y * x3
    y1 = e * f32(-2.12194440e-4)
    tmp = x2 * f32(0.5)
    y = y + y1
    x = x - tmp
    y2 = e * f32(0.693359375)
    x = x + y
    return x + y2


def generate_anchors(hp):
    """utils/bbox_utils.py:13-57."""
    base = []
    for scale in hp["anchor_scales"]:
        scale = scale / hp["img_size"]
        for ratio in hp["anchor_ratios"]:
            w = np.sqrt(f32(scale ** 2 / ratio))
            h = f32(w * f32(ratio))
            base.append([-h / f32(2), -w / f32(2), h / f32(2), w / f32(2)])
    base = np.asarray(base, dtype=f32)
    G = hp["feature_map_shape"]
    coords = (np.arange(G) / G + (1 / G) / 2).astype(f32)
    gx, gy = np.meshgrid(coords, coords)
    grid = np.stack([gy.ravel(), gx.ravel(), gy.ravel(), gx.ravel()], -1)
    anchors = (base.reshape(1, -1, 4) + grid.reshape(-1, 1, 4)).reshape(-1, 4)
    return np.clip(anchors, f32(0), f32(1)).astype(f32)


def get_bboxes_from_deltas(anchors, deltas):
    """utils/bbox_utils.py:96-108."""
    a, d = np.asarray(anchors, f32), np.asarray(deltas, f32)
    w = a[..., 3] - a[..., 1]
    h = a[..., 2] - a[..., 0]
    cx = a[..., 1] + f32(0.5) * w
    cy = a[..., 0] + f32(0.5) * h
    bw = cephes_exp(d[..., 3]) * w
    bh = cephes_exp(d[..., 2]) * h
    bcx = d[..., 1] * w + cx
    bcy = d[..., 0] * h + cy
    y1 = bcy - f32(0.5) * bh
    x1 = bcx - f32(0.5) * bw
    return np.stack([y1, x1, bh + y1, bw + x1], -1).astype(f32)


def generate_iou_map(bboxes, gt):
    """utils/bbox_utils.py:150-160; bboxes (N,4) or (B,N,4), gt (B,M,4)."""
    b, g = np.asarray(bboxes, f32), np.asarray(gt, f32)
    if b.ndim == 2:
        b = b[None]
    by1, bx1, by2, bx2 = [b[..., i:i + 1] for i in range(4)]
    gy1, gx1, gy2, gx2 = [g[..., i:i + 1] for i in range(4)]
    garea = ((gy2 - gy1) * (gx2 - gx1))[..., 0]
    barea = ((by2 - by1) * (bx2 - bx1))[..., 0]
    t = lambda v: np.transpose(v, (0, 2, 1))  # noqa: E731
    x_top = np.maximum(bx1, t(gx1))
    y_top = np.maximum(by1, t(gy1))
    x_bot = np.minimum(bx2, t(gx2))
    y_bot = np.minimum(by2, t(gy2))
    inter = np.maximum(x_bot - x_top, f32(0)) * np.maximum(y_bot - y_top, f32(0))
    union = barea[..., None] + garea[:, None, :] - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / union).astype(f32)


def crop_and_resize(image, boxes, box_ind, crop, extrapolation=0.0):
    """[TF-recall] crop_and_resize_op.cc CPU functor, bilinear."""
    image = np.asarray(image, f32)
    _, H, W, C = image.shape
    ch, cw = crop
    out = np.empty((len(boxes), ch, cw, C), f32)
    hm1, wm1 = f32(H - 1), f32(W - 1)
    for n, (bx, bi) in enumerate(zip(np.asarray(boxes, f32), box_ind)):
        y1, x1, y2, x2 = bx
        hs = (y2 - y1) * hm1 / f32(ch - 1) if ch > 1 else f32(0)
        ws = (x2 - x1) * wm1 / f32(cw - 1) if cw > 1 else f32(0)
        for y in range(ch):
            in_y = y1 * hm1 + f32(y) * hs if ch > 1 else f32(0.5) * (y1 + y2) * hm1
            if not (in_y >= 0 and in_y <= hm1):
                out[n, y] = extrapolation
                continue
            top, bot = int(np.floor(in_y)), int(np.ceil(in_y))
            ly = f32(in_y - f32(top))
            for x in range(cw):
                in_x = x1 * wm1 + f32(x) * ws if cw > 1 else f32(0.5) * (x1 + x2) * wm1
                if not (in_x >= 0 and in_x <= wm1):
                    out[n, y, x] = extrapolation
                    continue
                left, right = int(np.floor(in_x)), int(np.ceil(in_x))
                lx = f32(in_x - f32(left))
                tl, tr = image[bi, top, left], image[bi, top, right]
                bl, br = image[bi, bot, left], image[bi, bot, right]
                tp = tl + (tr - tl) * lx
                bt = bl + (br - bl) * lx
                out[n, y, x] = tp + (bt - tp) * ly
    return out


def nms_iou(bi, bj):
    """[TF-recall] IOUGreaterThanThreshold's IoU (returns 0 for area <= 0)."""
    bi, bj = np.asarray(bi, f32), np.asarray(bj, f32)
    ymin_i, xmin_i, ymax_i, xmax_i = min(bi[0], bi[2]), min(bi[1], bi[3]), max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j, ymax_j, xmax_j = min(bj[0], bj[2]), min(bj[1], bj[3]), max(bj[0], bj[2]), max(bj[1], bj[3])
    ai = f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i)
    aj = f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j)
    if ai <= 0 or aj <= 0:
        return f32(0)
    ih = max(f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f32(0))
    iw = max(f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f32(0))
    inter = f32(ih * iw)
    return f32(inter / f32(f32(ai + aj) - inter))


def single_class_nms(boxes, scores, max_out, iou_thr, score_thr=-np.inf):
    """Greedy NMS of one class: returns the selected indices in selection order (ties: lower index)."""
    order = sorted((i for i in range(len(scores)) if scores[i] > score_thr), key=lambda i: (-scores[i], i))
    sel = []
    for i in order:
        if len(sel) >= max_out:
            break
        if all(not (nms_iou(boxes[i], boxes[j]) > f32(iou_thr)) for j in reversed(sel)):
            sel.append(i)
    return sel


def get_deltas_from_bboxes(bboxes, gt):
    """utils/bbox_utils.py:121-135, vectorised like the reference (tf.where selects after computing both sides)."""
    b, g = np.asarray(bboxes, f32), np.asarray(gt, f32)
    bw = b[..., 3] - b[..., 1]
    bh = b[..., 2] - b[..., 0]
    bcx = b[..., 1] + f32(0.5) * bw
    bcy = b[..., 0] + f32(0.5) * bh
    gw = g[..., 3] - g[..., 1]
    gh = g[..., 2] - g[..., 0]
    gcx = g[..., 1] + f32(0.5) * gw
    gcy = g[..., 0] + f32(0.5) * gh
    bw = np.where(bw == 0, f32(1e-3), bw)
    bh = np.where(bh == 0, f32(1e-3), bh)
    with np.errstate(all="ignore"):
        dx = np.where(gw == 0, f32(0), (gcx - bcx) / bw)
        dy = np.where(gh == 0, f32(0), (gcy - bcy) / bh)
        dw = np.where(gw == 0, f32(0), cephes_log(np.where(gw == 0, f32(1), gw / bw)))
        dh = np.where(gh == 0, f32(0), cephes_log(np.where(gh == 0, f32(1), gh / bh)))
    return np.stack([dy, dx, dh, dw], axis=-1).astype(f32)


def randomly_select_xyz_mask(mask, select, rnd):
    """utils/train_utils.py:82-88 with the random integers given: double argsort, descending, ties by lower index."""
    mask = np.asarray(mask, bool)
    key = mask.astype(np.int32) * np.asarray(rnd, np.int32)
    order = np.argsort(-key.astype(np.int64), axis=-1, kind="stable")
    ranks = np.argsort(order, axis=-1, kind="stable")
    sel = np.asarray(select).reshape(-1, 1)
    return mask & (ranks < sel)


def roi_delta(rois, gt, labels, total_labels, total_pos, total_neg, variances, rnd_pos, rnd_neg):
    """models/faster_rcnn.py:215-234, op for op (positive, NaN-free gt coordinates only: np.maximum propagates NaN)."""
    rois, gt, labels = np.asarray(rois, f32), np.asarray(gt, f32), np.asarray(labels, np.int32)
    iou = generate_iou_map(rois, gt)
    # [TF-recall] argmax / reduce_max skip NaN (strict > from lowest())
    filled = np.where(np.isnan(iou), -np.inf, iou)
    arg = np.argmax(filled, axis=2)
    merged = np.where(np.isinf(filled.max(axis=2)), f32(-3.402823466e38), filled.max(axis=2)).astype(f32)
    pos = randomly_select_xyz_mask(merged > f32(0.5), [total_pos], rnd_pos)
    neg = randomly_select_xyz_mask((merged < f32(0.5)) & (merged > f32(0.1)), [total_neg], rnd_neg)
    gmap = np.take_along_axis(gt, arg[..., None], axis=1)
    egt = np.where(pos[..., None], gmap, f32(0))
    lmap = np.take_along_axis(labels, arg, axis=1)
    elab = np.where(pos, lmap, -1) + neg.astype(np.int32)
    d = (get_deltas_from_bboxes(rois, egt) / np.asarray(variances, f32)).astype(f32)
    onehot = (elab[..., None] == np.arange(total_labels)).astype(f32)
    with np.errstate(invalid="ignore"):
        out = onehot[..., None] * d[..., None, :]
    B, R = rois.shape[:2]
    return out.reshape(B, R * total_labels, 4).astype(f32), onehot
