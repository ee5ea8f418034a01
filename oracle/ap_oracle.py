"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's parse_predictions
(utils/ap_calculator.py:48-282): empty-box removal, NMS variant selection, confidence test and the per-scene detection lists.

PINNED: tests/golden/parse_predictions.npz holds inputs and the detection lists produced by the reference's own
parse_predictions imported in the build container (oracle/make_golden.py:ap_cases) for every NMS / scoring variant.  One
function stays UNPINNED there: mmcv.ops.points_in_boxes_all (mmcv-full 1.6.1, README.md:54-59, not installed) is replaced
in that import by the restatement of its published check_pt_in_box3d (oracle/criterion_oracle.py:points_in_boxes_all), as for
the criterion fixtures.
"""
import numpy as np

from . import nms_oracle as NO


def box_point_counts(points, boxes):
    """points [N,3] f32, boxes [K,7] f32 (centre xyz, sizes, yaw) -> [K] number of points inside (float32 arithmetic in
    mmcv's order; the caller's bottom-centre shift of ap_calculator.py:80 included)."""
    p = np.asarray(points, np.float32)[:, None, :]
    b = np.asarray(boxes, np.float32)[None, :, :]
    two = np.float32(2)
    zb = b[..., 2] - b[..., 5] / two
    zc = zb + b[..., 5] / two
    inz = ~(np.abs(p[..., 2] - zc) > b[..., 5] / two)
    sx, sy = p[..., 0] - b[..., 0], p[..., 1] - b[..., 1]
    ca, sa = np.cos(-b[..., 6]).astype(np.float32), np.sin(-b[..., 6]).astype(np.float32)
    lx, ly = sx * ca - sy * sa, sx * sa + sy * ca
    hx, hy = b[..., 3] / two, b[..., 4] / two
    return (inz & (lx > -hx) & (lx < hx) & (ly > -hy) & (ly < hy)).sum(0)


def prediction_mask(corners, sem_cls_probs, obj_prob, angle_probs, points, cfg, boxes_csa=None, stable=False):
    """-> (pred_mask [B,K] bool, pred_sem_cls [B,K])"""
    B, K = obj_prob.shape
    pred_sem_cls = np.argmax(sem_cls_probs, -1)
    nonempty = np.ones((B, K), bool)
    if cfg["remove_empty_box"]:
        for i in range(B):
            nonempty[i] = box_point_counts(points[i][:, :3], boxes_csa[i]) >= cfg["empty_pt_thre"]
            if not nonempty[i].any():
                nonempty[i, obj_prob[i].argmax()] = True
    if cfg.get("no_nms"):
        return nonempty, pred_sem_cls
    mask = np.zeros((B, K), bool)
    for i in range(B):
        idx = np.nonzero(nonempty[i])[0]
        c = corners[i, idx]
        if not cfg["use_3d_nms"]:
            rows = np.stack([c[:, :, 0].min(1), c[:, :, 2].min(1), c[:, :, 0].max(1), c[:, :, 2].max(1), obj_prob[i, idx]], 1)
            pick = NO.nms_2d(rows, cfg["nms_iou"], cfg["use_old_type_nms"], stable=stable)
        elif not cfg["cls_nms"]:
            pick = NO.nms_3d(NO.extents_with_score(c, obj_prob[i, idx]), cfg["nms_iou"], old_type=cfg["use_old_type_nms"],
                             stable=stable)
        else:
            score = obj_prob[i, idx] * angle_probs[i, idx] if cfg.get("angle_nms") else obj_prob[i, idx]
            pick = NO.nms_3d(NO.extents_with_score(c, score, pred_sem_cls[i, idx]), cfg["nms_iou"], same_class=True,
                             old_type=cfg["use_old_type_nms"], stable=stable)
        mask[i, idx[pick]] = True
    return mask, pred_sem_cls


def parse_predictions(corners, sem_cls_probs, obj_prob, angle_probs, points, cfg, boxes_csa=None, num_semcls=None, stable=False):
    """numpy arrays in, the reference's list of lists of (class, corners [8,3], score) out."""
    mask, pred_sem_cls = prediction_mask(corners, sem_cls_probs, obj_prob, angle_probs, points, cfg, boxes_csa, stable)
    out = []
    for i in range(obj_prob.shape[0]):
        js = [j for j in range(obj_prob.shape[1]) if mask[i, j] and obj_prob[i, j] > cfg["conf_thresh"]]
        if cfg.get("angle_conf"):
            out.append([(ii, corners[i, j], sem_cls_probs[i, j, ii] * obj_prob[i, j] * angle_probs[i, j])
                        for ii in range(num_semcls) for j in js])
        elif cfg["per_class_proposal"]:
            out.append([(ii, corners[i, j], sem_cls_probs[i, j, ii] * obj_prob[i, j]) for ii in range(num_semcls) for j in js])
        elif cfg["use_cls_confidence_only"]:
            out.append([(int(pred_sem_cls[i, j]), corners[i, j], sem_cls_probs[i, j, pred_sem_cls[i, j]]) for j in js])
        else:
            out.append([(int(pred_sem_cls[i, j]), corners[i, j], obj_prob[i, j]) for j in js])
    return out
