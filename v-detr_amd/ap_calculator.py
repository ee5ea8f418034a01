"""parse_predictions on the device (reference utils/ap_calculator.py:48-282).

The reference copies every prediction tensor to the host, loops over scenes x boxes in Python to build the NMS rows, runs the
numpy NMS per scene and then builds the per-class detection lists with a Python comprehension over classes x boxes.  Here
the empty-box test (``vdetr_box_point_count_f32``: no (B,N,K) flag tensor), the NMS (``vdetr_nms3d_f32``, all scenes in
three launches) and the confidence test stay on the GPU; only the kept boxes are assembled into the reference's list format
after ONE device->host copy.  No CPU path: CPU tensors raise.

``config_dict`` is the reference's (``get_ap_config_dict``, ap_calculator.py:285-321).  ``rotated_nms`` is not supported (the
reference prints a box and then fails on an undefined ``pred_mask``).
"""
import itertools

import numpy as np
import torch

from . import _lib as L
from .nms import batched_nms_3d


def get_ap_config_dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True,
                       per_class_proposal=True, use_cls_confidence_only=False, conf_thresh=0.0, no_nms=False, dataset_config=None,
                       empty_pt_thre=5, rotated_nms=False, angle_nms=False, angle_conf=False):
    """ap_calculator.py:285-321, same keys and defaults."""
    return dict(remove_empty_box=remove_empty_box, use_3d_nms=use_3d_nms, nms_iou=nms_iou, use_old_type_nms=use_old_type_nms,
                cls_nms=cls_nms, per_class_proposal=per_class_proposal, use_cls_confidence_only=use_cls_confidence_only,
                conf_thresh=conf_thresh, no_nms=no_nms, dataset_config=dataset_config, empty_pt_thre=empty_pt_thre,
                rotated_nms=rotated_nms, angle_nms=angle_nms, angle_conf=angle_conf)


def box_point_counts(points, boxes):
    """points [B,N,3] f32, boxes [B,K,7] f32 (centre xyz, sizes, yaw) -> [B,K] int32 number of points inside each box
    (mmcv points_in_boxes_all(...).sum over the points, ap_calculator.py:90-91)."""
    L.require_gpu(points, "points")
    L.require_float(points, "points")
    L.require_float(boxes, "boxes")
    B, N, _ = points.shape
    K = boxes.shape[1]
    assert points.shape[2] == 3 and boxes.shape == (B, K, 7)
    points, boxes = points.detach().contiguous(), boxes.detach().contiguous()
    counts = torch.zeros((B, K), dtype=torch.int32, device=points.device)
    L.check(L.lib().vdetr_box_point_count_f32(L.ptr(points), L.ptr(boxes), B, N, K, L.ptr(counts), L.stream_ptr()), "box_point_count")
    return counts


def nonempty_box_mask(point_cloud, predicted_boxes_CSA, obj_prob, empty_pt_thre, max_points=40000):
    """ap_calculator.py:78-93: boxes with at least ``empty_pt_thre`` of (at most 40000 randomly chosen) points inside; a scene
    with no such box keeps its most confident one.  -> [B,K] bool"""
    xyz = point_cloud[..., :3]
    if xyz.shape[1] > max_points:
        xyz = xyz[:, torch.randperm(xyz.shape[1], device=xyz.device)[:max_points]]
    mask = box_point_counts(xyz, predicted_boxes_CSA) >= int(empty_pt_thre)
    none = ~mask.any(dim=1)
    best = obj_prob.argmax(dim=1)
    mask[torch.arange(mask.shape[0], device=mask.device), best] |= none
    return mask


def _plane_corners(corners):
    """corners for the 2-D NMS (ap_calculator.py:116-146: x and z extents): the y axis becomes a dummy extent [0, 1], so the
    3-D volumes and intersections are the 2-D areas times exactly 1"""
    c = corners.clone()
    c[..., 1] = c[..., 2]
    c[..., 2] = 0.0
    c[..., 4:, 2] = 1.0
    return c


def prediction_masks(predicted_boxes, sem_cls_probs, objectness_probs, angle_probs, point_cloud, config_dict,
                     predicted_boxes_CSA=None):
    """The device half of parse_predictions: -> dict(pred_mask [B,K] bool after the empty-box test and the NMS,
    keep [B,K] bool = pred_mask & (obj_prob > conf_thresh), pred_sem_cls [B,K], nonempty [B,K] bool)."""
    L.require_gpu(predicted_boxes, "predicted_boxes")
    cfg = config_dict
    if cfg.get("rotated_nms"):
        raise NotImplementedError("rotated_nms: the reference has no working path either (ap_calculator.py:112-113)")
    corners = predicted_boxes.detach().float()
    obj = objectness_probs.detach().float()
    sem = sem_cls_probs.detach().float()
    B, K = obj.shape
    pred_sem_cls = sem.argmax(-1)
    if cfg["remove_empty_box"]:
        if predicted_boxes_CSA is None or point_cloud is None:
            raise ValueError("remove_empty_box needs point_cloud and predicted_boxes_CSA (centre, size, angle rows)")
        nonempty = nonempty_box_mask(point_cloud, predicted_boxes_CSA.detach().float(), obj, cfg["empty_pt_thre"])
    else:
        nonempty = torch.ones((B, K), dtype=torch.bool, device=obj.device)
    if cfg.get("no_nms"):
        pred_mask = nonempty
    elif not cfg["use_3d_nms"]:
        pred_mask = batched_nms_3d(_plane_corners(corners), obj, None, nonempty, cfg["nms_iou"], cfg["use_old_type_nms"])
    elif not cfg["cls_nms"]:
        pred_mask = batched_nms_3d(corners, obj, None, nonempty, cfg["nms_iou"], cfg["use_old_type_nms"])
    else:
        score = obj * angle_probs.detach().float() if cfg.get("angle_nms") else obj
        pred_mask = batched_nms_3d(corners, score, pred_sem_cls, nonempty, cfg["nms_iou"], cfg["use_old_type_nms"])
    keep = pred_mask & (obj > cfg["conf_thresh"])
    return dict(pred_mask=pred_mask, keep=keep, pred_sem_cls=pred_sem_cls, nonempty=nonempty)


def parse_predictions(predicted_boxes, sem_cls_probs, objectness_probs, angle_probs, point_cloud, config_dict,
                      predicted_boxes_CSA=None):
    """ap_calculator.py:48-282, same arguments and the same result: a list (batch) of lists of
    ``(class, corners [8,3] float32 array, score)`` in the reference's order (class-major for the per-class forms)."""
    cfg = config_dict
    m = prediction_masks(predicted_boxes, sem_cls_probs, objectness_probs, angle_probs, point_cloud, cfg, predicted_boxes_CSA)
    sem = sem_cls_probs.detach().float()
    obj = objectness_probs.detach().float()
    per_class = cfg.get("angle_conf") or cfg["per_class_proposal"]
    if per_class:
        assert cfg["use_cls_confidence_only"] is False
        score = sem * obj[..., None]                                    # [B,K,C]
        if cfg.get("angle_conf"):
            score = score * angle_probs.detach().float()[..., None]
    elif cfg["use_cls_confidence_only"]:
        score = sem.gather(-1, m["pred_sem_cls"][..., None])[..., 0]    # [B,K]
    else:
        score = obj
    keep, cls, corners, score = (t.cpu().numpy() for t in (m["keep"], m["pred_sem_cls"], predicted_boxes.detach().float(), score))
    out = []
    for i in range(keep.shape[0]):
        js = np.nonzero(keep[i])[0]
        rows = list(corners[i][js])                      # one [8,3] view per kept box
        if per_class:
            ncls = cfg["dataset_config"].num_semcls
            per_cls = np.ascontiguousarray(score[i][js].T)  # [C, kept]
            cur = []
            for ii in range(ncls):
                cur += zip(itertools.repeat(ii), rows, list(per_cls[ii]))
            out.append(cur)
        else:
            out.append(list(zip(cls[i][js].tolist(), rows, list(score[i][js]))))
    return out
