"""Average precision of 3-D detections with the box matching on the device (reference utils/eval_det.py:74-302; SURVEY.md
§8f rank 4).

The reference walks every detection in Python and clips it against every ground-truth box of its image and class with a
Python Sutherland-Hodgman loop + a qhull call per pair (minutes for a validation set, hence its ``Pool(10)``).  Here ONE
launch (``vdetr_box3d_iou_max_f64``) gives every detection its best ground-truth box for all classes and images at once;
the greedy "first detection in confidence order claims the box" rule is a scatter-min over ranks; the precision / recall
curves and the VOC AP are a few numpy lines on the host, as in the reference.  Same inputs and outputs as the reference's
``eval_det`` / ``eval_det_multiprocessing`` (dictionaries of host lists, as ``parse_predictions`` produces them).

Detections with exactly equal confidence are ranked by a STABLE sort in insertion order (the reference's ``np.argsort`` leaves
that order to numpy's sort implementation).  Only ``get_iou_func=get_iou_obb`` (box3d_iou on corners), the one the reference's
APCalculator uses, is supported.  No CPU path: a missing GPU / library raises.
"""
import numpy as np
import torch

from . import _lib as L

AREA_RNG = (0.17, 0.44)  # eval_det.py:90: volume bounds of the S / M / L splits


def voc_ap(rec, prec, use_07_metric=False):
    """eval_det.py:23-56."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = np.max(prec[rec >= t]) if np.sum(rec >= t) else 0
            ap = ap + p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # precision envelope
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _volumes(corners):
    """eval_det.py:64-69 box3d_vol_batch."""
    a = np.sqrt((corners[:, 0, 2] - corners[:, 1, 2]) ** 2)
    b = np.sqrt((corners[:, 1, 0] - corners[:, 2, 0]) ** 2)
    c = np.sqrt((corners[:, 0, 1] - corners[:, 4, 1]) ** 2)
    return a * b * c


def _size_filter(vol, size):
    if size == "S":
        return vol < AREA_RNG[0]
    if size == "M":
        return np.logical_and(vol > AREA_RNG[0], vol < AREA_RNG[1])
    if size == "L":
        return vol > AREA_RNG[1]
    return np.ones(vol.shape, bool)


def match_detections(pred_corners, pred_img, pred_cls, pred_score, gt_corners, gt_img, gt_cls, num_images, ovthresh, device="cuda"):
    """Flat arrays (detections in insertion order; ground truth grouped by ascending image index) ->
    (order, tp) : ``order`` ranks the detections by (class, descending score, insertion), ``tp[order]`` marks true positives."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("eval_det: CPU not supported")
    P, G = len(pred_score), len(gt_cls)
    begin = np.zeros(num_images + 1, np.int32)
    np.add.at(begin, np.asarray(gt_img, np.int64) + 1, 1)
    begin = np.cumsum(begin).astype(np.int32)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    pc, pi, pk = t(pred_corners.reshape(P, 8, 3), np.float32), t(pred_img, np.int32), t(pred_cls, np.int32)
    gc = t(gt_corners.reshape(G, 8, 3), np.float32) if G else torch.zeros((1, 8, 3), device=dev)
    gk = t(gt_cls, np.int32) if G else torch.zeros(1, dtype=torch.int32, device=dev)
    gb = t(begin, np.int32)
    ovmax = torch.empty(P, dtype=torch.float64, device=dev)
    jmax = torch.empty(P, dtype=torch.int32, device=dev)
    L.check(L.lib().vdetr_box3d_iou_max_f64(L.ptr(pc), L.ptr(pi), L.ptr(pk), P, L.ptr(gc), L.ptr(gk), L.ptr(gb), L.ptr(ovmax),
                                            L.ptr(jmax), L.stream_ptr()), "box3d_iou_max")
    # rank: class-major, then descending confidence, ties in insertion order (two stable sorts)
    score = t(pred_score, np.float64)
    o1 = torch.sort(-score, stable=True)[1]
    order = o1[torch.sort(pk[o1].long(), stable=True)[1]]
    rank = torch.empty(P, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(P, device=dev)
    hit = ovmax > ovthresh
    claim = torch.full((max(G, 1),), P, dtype=torch.int64, device=dev)
    claim.scatter_reduce_(0, jmax.long()[hit], rank[hit], reduce="amin")   # the best-ranked detection of every box
    tp = hit.clone()
    tp[hit] = claim[jmax.long()[hit]] == rank[hit]
    return order.cpu().numpy(), tp.cpu().numpy()


def eval_det_multiprocessing(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None, size="", device="cuda"):
    """eval_det.py:241-302 (and :186-237 eval_det): pred_all {img_id: [(classname, corners [8,3], score)]},
    gt_all {img_id: [(classname, corners [8,3])]} -> (rec, prec, ap) dictionaries keyed by class name."""
    if get_iou_func is not None and getattr(get_iou_func, "__name__", "") != "get_iou_obb":
        raise NotImplementedError("eval_det: only get_iou_obb (box3d_iou on 8 corners) is built")
    img_index = {}
    for img_id in list(pred_all.keys()) + list(gt_all.keys()):
        img_index.setdefault(img_id, len(img_index))
    names = {}                                   # class name -> integer, in the reference's `gt` dictionary order
    pc, pi, pk, ps = [], [], [], []
    for img_id, dets in pred_all.items():
        for classname, bbox, score in dets:
            names.setdefault(classname, len(names))
            pc.append(bbox), pi.append(img_index[img_id]), pk.append(names[classname]), ps.append(score)
    gc, gi, gk = [], [], []
    for img_id, boxes in gt_all.items():
        for classname, bbox in boxes:
            names.setdefault(classname, len(names))
            gc.append(bbox), gi.append(img_index[img_id]), gk.append(names[classname])
    arrays = (np.asarray(pc, np.float32).reshape(-1, 8, 3), np.asarray(pi, np.int64), np.asarray(pk, np.int64), np.asarray(ps, np.float64),
              np.asarray(gc, np.float32).reshape(-1, 8, 3), np.asarray(gi, np.int64), np.asarray(gk, np.int64))
    return evaluate_flat(*arrays, len(img_index), list(names), ovthresh, use_07_metric, size, device)


def evaluate_flat(pc, pi, pk, ps, gc, gi, gk, num_images, class_names, ovthresh=0.25, use_07_metric=False, size="", device="cuda"):
    """The same evaluation on flat arrays: detections (corners [P,8,3], image index, class index, score) in insertion order,
    ground truth (corners [G,8,3], image index, class index); ``class_names[k]`` is the key class index k is reported under."""
    if size != "":
        keep_p, keep_g = _size_filter(_volumes(pc), size), _size_filter(_volumes(gc), size)
        pc, pi, pk, ps, gc, gi, gk = pc[keep_p], pi[keep_p], pk[keep_p], ps[keep_p], gc[keep_g], gi[keep_g], gk[keep_g]
    g_order = np.argsort(gi, kind="stable")      # ground truth grouped by image for the kernel
    gc, gi, gk = gc[g_order], gi[g_order], gk[g_order]
    if len(ps):
        order, tp = match_detections(pc, pi, pk, ps, gc, gi, gk, num_images, ovthresh, device)
    else:
        order, tp = np.zeros(0, np.int64), np.zeros(0, bool)
    cls_sorted, tp_sorted = pk[order], tp[order]
    npos_all = np.bincount(gk, minlength=len(class_names))
    has_pred = np.bincount(pk, minlength=len(class_names)) > 0
    rec, prec, ap = {}, {}, {}
    for c, classname in enumerate(class_names):
        if not has_pred[c]:                      # eval_det.py:294-298
            rec[classname] = prec[classname] = ap[classname] = 0
            continue
        sel = tp_sorted[cls_sorted == c]
        tpc, fpc = np.cumsum(sel.astype(np.float64)), np.cumsum((~sel).astype(np.float64))
        npos = int(npos_all[c])
        r = tpc / float(npos) if npos else np.zeros_like(tpc)
        p = tpc / np.maximum(tpc + fpc, np.finfo(np.float64).eps)
        rec[classname], prec[classname], ap[classname] = r, p, voc_ap(r, p, use_07_metric)
    return rec, prec, ap


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None):
    """eval_det.py:186-237, the single-process form: same result (it raises KeyError there for a ground-truth class without
    any detection; here that class reports 0 as in the multiprocessing form)."""
    return eval_det_multiprocessing(pred_all, gt_all, ovthresh, use_07_metric, get_iou_func)
