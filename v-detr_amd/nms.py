"""Greedy 3-D NMS of box predictions on the device (reference utils/nms.py:78-162, used by parse_predictions in
utils/ap_calculator.py:165-220).

``nms_3d_faster`` / ``nms_3d_faster_samecls`` keep the reference's call shape (one scene, boxes as rows
``(x1, y1, z1, x2, y2, z2, score[, cls])``, returns the picked indices best first); ``batched_nms_3d`` is the form the
evaluation loop wants: corners, scores and classes of a whole batch stay on the device, the result is the ``pred_mask`` of
ap_calculator.py:165-220.  Scores are visited in the order of a STABLE ascending arg-sort taken from its end; numpy's default
arg-sort is not stable, so boxes with exactly equal scores may be visited in a different order there.  No CPU path.
"""
import torch

from . import _lib as L


def batched_nms_3d(corners, scores, classes=None, valid=None, iou_threshold=0.25, old_type=False):
    """corners [B,K,8,3] f32, scores [B,K] f32, classes [B,K] int or None (class-agnostic), valid [B,K] bool or None
    (nonempty_box_mask) -> keep [B,K] bool."""
    L.require_gpu(corners, "corners")
    L.require_float(corners, "corners")
    L.require_float(scores, "scores")
    B, K = scores.shape
    assert corners.shape == (B, K, 8, 3)
    corners, scores = corners.detach().contiguous(), scores.detach().contiguous()
    cls = classes.detach().to(torch.int32).contiguous() if classes is not None else None
    val = valid.detach().to(torch.uint8).contiguous() if valid is not None else None
    order = torch.sort(scores, dim=1, stable=True)[1].contiguous()
    keep = torch.empty((B, K), dtype=torch.uint8, device=scores.device)
    lib = L.lib()
    nbytes = lib.vdetr_nms3d_workspace_bytes(B, K)
    ws = L.workspace(nbytes, scores.device)
    L.check(lib.vdetr_nms3d_f32(L.ptr(corners), L.ptr(scores), L.ptr(cls), L.ptr(val), L.ptr(order), B, K, float(iou_threshold),
                                int(bool(old_type)), L.ptr(keep), L.ptr(ws), nbytes, L.stream_ptr()), "nms3d")
    return keep.bool()


def _rows_to_corners(boxes):
    """(x1,y1,z1,x2,y2,z2) rows -> 8 corners whose min / max are exactly those extents"""
    lo, hi = boxes[:, 0:3], boxes[:, 3:6]
    c = lo[:, None, :].repeat(1, 8, 1)
    c[:, 4:] = hi[:, None, :]
    return c


def _picked(boxes, classes, overlap_threshold, old_type):
    b = boxes.to(torch.float32)
    keep = batched_nms_3d(_rows_to_corners(b)[None], b[None, :, 6].contiguous(), classes, None, overlap_threshold, old_type)[0]
    order = torch.sort(b[:, 6], stable=True)[1].flip(0)
    return order[keep[order]]


def nms_3d_faster(boxes, overlap_threshold, old_type=False):
    """utils/nms.py:78-118 for a GPU tensor of rows (x1,y1,z1,x2,y2,z2,score): indices of the kept boxes, best first."""
    return _picked(boxes, None, overlap_threshold, old_type)


def nms_3d_faster_samecls(boxes, overlap_threshold, old_type=False):
    """utils/nms.py:121-162: rows (x1,y1,z1,x2,y2,z2,score,cls); only boxes of the same class suppress each other."""
    return _picked(boxes, boxes[None, :, 7].to(torch.int32), overlap_threshold, old_type)
