"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's detection AP
(utils/eval_det.py:74-302 eval_det_cls / eval_det_multiprocessing, with get_iou_obb = utils/box_util.py:122-147 box3d_iou,
:37-84 polygon_clip, :92-105 convex_hull_intersection through scipy's qhull, :108-113 box3d_vol).

PINNED: tests/golden/eval_det.npz holds a synthetic validation set (boxes from the reference's get_3d_box) and the recall /
precision / AP the reference's own eval_det_multiprocessing returns for it at IoU 0.25 and 0.5 (oracle/make_golden.py:eval_cases).
"""
import numpy as np
from scipy.spatial import ConvexHull


def clip_polygon(subject, clip):
    """Sutherland-Hodgman, counter-clockwise polygons as lists of (x, y); None when nothing is left (box_util.py:37-84)."""
    out = list(subject)
    a = clip[-1]
    for b in clip:
        src, out = out, []
        inside = lambda p: (b[0] - a[0]) * (p[1] - a[1]) > (b[1] - a[1]) * (p[0] - a[0])  # noqa: E731
        s = src[-1]
        for e in src:
            if inside(e) != inside(s):
                dc = (a[0] - b[0], a[1] - b[1])
                dp = (s[0] - e[0], s[1] - e[1])
                n1 = a[0] * b[1] - a[1] * b[0]
                n2 = s[0] * e[1] - s[1] * e[0]
                n3 = 1.0 / (dc[0] * dp[1] - dc[1] * dp[0])
                out.append(((n1 * dp[0] - n2 * dc[0]) * n3, (n1 * dp[1] - n2 * dc[1]) * n3))
            if inside(e):
                out.append(e)
            s = e
        a = b
        if not out:
            return None
    return out


def box3d_vol(c):
    edge = lambda i, j: np.sqrt(np.sum((c[i] - c[j]) ** 2))  # noqa: E731
    return edge(0, 1) * edge(1, 2) * edge(0, 4)


def box3d_iou(c1, c2):
    """float64 corners [8,3] -> 3-D IoU (box_util.py:122-147)."""
    r1 = [(c1[i, 0], c1[i, 2]) for i in (3, 2, 1, 0)]
    r2 = [(c2[i, 0], c2[i, 2]) for i in (3, 2, 1, 0)]
    with np.errstate(all="ignore"):  # nearly coincident boxes divide by ~0 in the reference too
        poly = clip_polygon(r1, r2)
    area = 0.0
    if poly is not None:
        try:
            area = ConvexHull(poly).volume
        except Exception:  # qhull refuses degenerate polygons: the reference counts them as no overlap
            area = 0.0
    inter = area * max(0.0, min(c1[0, 1], c2[0, 1]) - max(c1[4, 1], c2[4, 1]))
    return inter / (box3d_vol(c1) + box3d_vol(c2) - inter)


def voc_ap(rec, prec):
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def eval_class(pred, gt, ovthresh, stable=True):
    """pred {img: [(corners, score)]}, gt {img: [corners]} of ONE class -> (rec, prec, ap, tp flags in ranked order)."""
    boxes = {img: np.array(b, dtype=np.float64).reshape(-1, 8, 3) for img, b in gt.items()}
    taken = {img: [False] * len(b) for img, b in boxes.items()}
    npos = sum(len(b) for b in boxes.values())
    imgs, conf, bb = [], [], []
    for img, dets in pred.items():
        for box, score in dets:
            imgs.append(img), conf.append(score), bb.append(np.asarray(box, np.float64))
    order = np.argsort(-np.array(conf), kind="stable" if stable else None)
    tp = np.zeros(len(order))
    for d, k in enumerate(order):
        cand = boxes.get(imgs[k], np.zeros((0, 8, 3)))
        best, arg = -np.inf, -1
        for j in range(len(cand)):
            iou = box3d_iou(bb[k], cand[j])
            if iou > best:
                best, arg = iou, j
        if best > ovthresh and not taken[imgs[k]][arg]:
            tp[d] = 1.0
            taken[imgs[k]][arg] = True
    ctp, cfp = np.cumsum(tp), np.cumsum(1.0 - tp)
    rec = ctp / float(npos) if npos else np.zeros_like(ctp)
    prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec), tp


def eval_det(pred_all, gt_all, ovthresh=0.25):
    """{img: [(class, corners, score)]}, {img: [(class, corners)]} -> (rec, prec, ap) per class; a class without any detection
    reports 0 (eval_det.py:294-298)."""
    pred, gt = {}, {}
    for img, dets in pred_all.items():
        for c, box, score in dets:
            pred.setdefault(c, {}).setdefault(img, []).append((box, score))
            gt.setdefault(c, {}).setdefault(img, [])
    for img, boxes in gt_all.items():
        for c, box in boxes:
            gt.setdefault(c, {}).setdefault(img, []).append(box)
    rec, prec, ap = {}, {}, {}
    for c in gt:
        if c in pred:
            rec[c], prec[c], ap[c], _ = eval_class(pred[c], gt[c], ovthresh)
        else:
            rec[c] = prec[c] = ap[c] = 0
    return rec, prec, ap
