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


def detections_flat(predicted_boxes, sem_cls_probs, objectness_probs, angle_probs, point_cloud, config_dict,
                    predicted_boxes_CSA=None):
    """parse_predictions without the Python lists: per scene (corners [n,8,3] f32, classes [m], box index [m], scores [m]) as
    numpy arrays, the m detections in the order of the reference's list (class-major for the per-class forms)."""
    cfg = config_dict
    m = prediction_masks(predicted_boxes, sem_cls_probs, objectness_probs, angle_probs, point_cloud, cfg, predicted_boxes_CSA)
    sem = sem_cls_probs.detach().float()
    obj = objectness_probs.detach().float()
    per_class = cfg.get("angle_conf") or cfg["per_class_proposal"]
    if per_class:
        assert cfg["use_cls_confidence_only"] is False
        score = sem * obj[..., None]
        if cfg.get("angle_conf"):
            score = score * angle_probs.detach().float()[..., None]
    elif cfg["use_cls_confidence_only"]:
        score = sem.gather(-1, m["pred_sem_cls"][..., None])[..., 0]
    else:
        score = obj
    keep, cls, corners, score = (t.cpu().numpy() for t in (m["keep"], m["pred_sem_cls"], predicted_boxes.detach().float(), score))
    out = []
    for i in range(keep.shape[0]):
        js = np.nonzero(keep[i])[0]
        if per_class:
            ncls = cfg["dataset_config"].num_semcls
            out.append((corners[i][js], np.repeat(np.arange(ncls), len(js)), np.tile(np.arange(len(js)), ncls),
                        np.ascontiguousarray(score[i][js][:, :ncls].T).ravel()))
        else:
            out.append((corners[i][js], cls[i][js].astype(np.int64), np.arange(len(js)), score[i][js]))
    return out


class APCalculator(object):
    """ap_calculator.py:324-529 with the same constructor, methods and result dictionaries.  ``step`` keeps the detections of
    a batch as flat arrays (no per-detection Python tuples) and ``compute_metrics`` hands them to the device matcher
    (eval_det.evaluate_flat); ``accumulate`` still accepts the reference's lists of tuples."""

    def __init__(self, dataset_config, ap_iou_thresh=[0.25, 0.5], class2type_map=None, exact_eval=False, ap_config_dict=None,
                 no_nms=False, args=None):
        self.ap_iou_thresh = ap_iou_thresh
        if ap_config_dict is None:
            ap_config_dict = get_ap_config_dict(
                dataset_config=dataset_config, remove_empty_box=exact_eval, no_nms=no_nms, use_3d_nms=not args.no_3d_nms,
                nms_iou=args.nms_iou, empty_pt_thre=args.empty_pt_thre, conf_thresh=args.conf_thresh, rotated_nms=args.rotated_nms,
                angle_nms=args.angle_nms, angle_conf=args.angle_conf, use_old_type_nms=args.use_old_type_nms,
                cls_nms=not args.no_cls_nms, per_class_proposal=not args.no_per_class_proposal,
                use_cls_confidence_only=args.use_cls_confidence_only)
        self.ap_config_dict = ap_config_dict
        self.class2type_map = class2type_map
        self.reset()

    def reset(self):
        self.gt_map_cls = {}     # {scan_id: [(classname, bbox)]}            (filled by accumulate() only)
        self.pred_map_cls = {}   # {scan_id: [(classname, bbox, score)]}
        self._flat_pred = {}     # {scan_id: (corners [n,8,3], classes [m], box index [m], scores [m])}  (filled by step())
        self._flat_gt = {}       # {scan_id: (classes [g], corners [g,8,3])}
        self.scan_cnt = 0

    def make_gt_list(self, gt_box_corners, gt_box_sem_cls_labels, gt_box_present):
        return [[(gt_box_sem_cls_labels[i, j].item(), gt_box_corners[i, j]) for j in range(gt_box_corners.shape[1])
                 if gt_box_present[i, j] == 1] for i in range(gt_box_corners.shape[0])]

    def step_meter(self, outputs, targets):
        if "outputs" in outputs:
            outputs = outputs["outputs"]
        csa = torch.cat((outputs["center_unnormalized"].detach(), outputs["size_unnormalized"].detach(),
                         outputs["angle_continuous"].detach().unsqueeze(-1)), dim=-1)
        self.step(predicted_box_corners=outputs["box_corners"], sem_cls_probs=outputs["sem_cls_prob"],
                  objectness_probs=outputs["objectness_prob"], angle_probs=outputs["angle_prob"],
                  point_cloud=targets["point_clouds"], gt_box_corners=targets["gt_box_corners"],
                  gt_box_sem_cls_labels=targets["gt_box_sem_cls_label"], gt_box_present=targets["gt_box_present"],
                  predicted_box_CSA=csa)

    def step(self, predicted_box_corners, sem_cls_probs, objectness_probs, angle_probs, point_cloud, gt_box_corners,
             gt_box_sem_cls_labels, gt_box_present, predicted_box_CSA):
        gt_c = gt_box_corners.detach().float().cpu().numpy()
        gt_k = gt_box_sem_cls_labels.detach().cpu().numpy()
        gt_p = gt_box_present.detach().cpu().numpy() == 1
        flat = detections_flat(predicted_box_corners, sem_cls_probs, objectness_probs, angle_probs, point_cloud,
                               self.ap_config_dict, predicted_box_CSA)
        for i, det in enumerate(flat):
            self._flat_pred[self.scan_cnt] = det
            self._flat_gt[self.scan_cnt] = (gt_k[i][gt_p[i]].astype(np.int64), gt_c[i][gt_p[i]])
            self.scan_cnt += 1

    def accumulate(self, batch_pred_map_cls, batch_gt_map_cls):
        assert len(batch_pred_map_cls) == len(batch_gt_map_cls)
        for pred, gt in zip(batch_pred_map_cls, batch_gt_map_cls):
            self.gt_map_cls[self.scan_cnt] = gt
            self.pred_map_cls[self.scan_cnt] = pred
            self.scan_cnt += 1

    def _flat_arrays(self):
        pc, pi, pk, ps, gc, gi, gk = [], [], [], [], [], [], []
        for scan in range(self.scan_cnt):
            if scan in self._flat_pred:
                corners, cls, idx, score = self._flat_pred[scan]
                g_cls, g_corners = self._flat_gt[scan]
                pc.append(corners[idx]), pk.append(cls), ps.append(score.astype(np.float64))
            else:
                dets, gts = self.pred_map_cls[scan], self.gt_map_cls[scan]
                pc.append(np.asarray([d[1] for d in dets], np.float32).reshape(-1, 8, 3))
                pk.append(np.asarray([d[0] for d in dets], np.int64)), ps.append(np.asarray([d[2] for d in dets], np.float64))
                g_cls = np.asarray([g[0] for g in gts], np.int64)
                g_corners = np.asarray([g[1] for g in gts], np.float32).reshape(-1, 8, 3)
            pi.append(np.full(len(pk[-1]), scan, np.int64))
            gc.append(g_corners), gk.append(g_cls), gi.append(np.full(len(g_cls), scan, np.int64))
        cat = lambda xs, shape, dt: np.concatenate(xs) if xs else np.zeros(shape, dt)  # noqa: E731
        return (cat(pc, (0, 8, 3), np.float32), cat(pi, 0, np.int64), cat(pk, 0, np.int64), cat(ps, 0, np.float64),
                cat(gc, (0, 8, 3), np.float32), cat(gi, 0, np.int64), cat(gk, 0, np.int64))

    def compute_metrics(self, size=""):
        from collections import OrderedDict

        from .eval_det import evaluate_flat
        pc, pi, pk, ps, gc, gi, gk = self._flat_arrays()
        # classes are reported under their integer label; the reference's dictionary holds every class that has a detection
        # or a ground-truth box
        present = np.union1d(np.unique(pk), np.unique(gk)).astype(np.int64)
        remap = np.full(int(present.max()) + 1 if present.size else 1, -1, np.int64)
        remap[present] = np.arange(present.size)
        overall_ret = OrderedDict()
        for thr in self.ap_iou_thresh:
            rec, prec, ap = evaluate_flat(pc, pi, remap[pk], ps, gc, gi, remap[gk], self.scan_cnt, [int(c) for c in present],
                                          ovthresh=thr, size=size)
            ret = OrderedDict()
            for key in sorted(ap.keys()):
                clsname = self.class2type_map[key] if self.class2type_map else str(key)
                ret["%s Average Precision" % (clsname)] = ap[key]
            ap_vals = np.array(list(ap.values()), dtype=np.float32)
            ap_vals[np.isnan(ap_vals)] = 0
            ret["mAP"] = ap_vals.mean()
            rec_list = []
            for key in sorted(ap.keys()):
                clsname = self.class2type_map[key] if self.class2type_map else str(key)
                last = rec[key][-1] if isinstance(rec[key], np.ndarray) and rec[key].size else 0   # no detection: 0 (:471-476)
                ret["%s Recall" % (clsname)] = last
                rec_list.append(last)
            ret["AR"] = np.mean(rec_list)
            overall_ret[thr] = ret
        return overall_ret

    def __str__(self):
        return self.metrics_to_str(self.compute_metrics())

    def metrics_to_str(self, overall_ret, per_class=True):
        lines = [", ".join(f"mAP{x:.2f}" for x in self.ap_iou_thresh) + ": " +
                 ", ".join(f"{overall_ret[x]['mAP'] * 100:.2f}" for x in self.ap_iou_thresh),
                 ", ".join(f"AR{x:.2f}" for x in self.ap_iou_thresh) + ": " +
                 ", ".join(f"{overall_ret[x]['AR'] * 100:.2f}" for x in self.ap_iou_thresh)]
        if per_class:
            for thr in self.ap_iou_thresh:
                lines += ["-" * 5, f"IOU Thresh={thr}"]
                lines += [f"{k}: {v * 100:.2f}" for k, v in overall_ret[thr].items() if k not in ("mAP", "AR")]
        return "\n".join(lines)

    def metrics_to_dict(self, overall_ret):
        out = {}
        for thr in self.ap_iou_thresh:
            out[f"mAP_{thr}"] = overall_ret[thr]["mAP"] * 100
            out[f"AR_{thr}"] = overall_ret[thr]["AR"] * 100
        return out
