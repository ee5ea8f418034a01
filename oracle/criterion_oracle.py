"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): torch-CPU restatement of the reference's set criterion
(criterion.py) -- SURVEY.md §8(f) rank 1, the first row next to the hot path.

What is restated, with the reference lines it follows:
  * pairwise GIoU of axis-aligned boxes given as 8 camera-frame corners     utils/box_util.py:441-520 (helpers), :523-600
  * matcher cost matrix + assignment                                       criterion.py:100-228
  * focal / angle / centre / size / GIoU / cardinality losses               criterion.py:77-98, 262-509
  * ground-truth repetition + compaction (repeat_num)                       criterion.py:511-600
  * stage loop, binary first stage, weights                                 criterion.py:602-708
  * seed-point classification loss                                          criterion.py:270-327

Third-party pieces that are NOT under /root/reference:
  * scipy.optimize.linear_sum_assignment (requirements.txt:9, scipy==1.5.1): called directly here (scipy is in the image),
    restated in oracle/lsa_oracle.py.
  * mmcv.ops.points_in_boxes_all (README.md:54-59, mmcv-full==1.6.1): absent from the image.  Restated from its published
    semantics (mmcv/ops/csrc/common/cuda/points_in_boxes_cuda_kernel.cuh, check_pt_in_box3d): a box is
    (x, y, z_bottom, dx, dy, dz, yaw); a point is inside iff |z - (z_bottom + dz/2)| <= dz/2 and, after rotating
    (x - cx, y - cy) by -yaw, -dx/2 < lx < dx/2 and -dy/2 < ly < dy/2 (strict).  PARITY UNPINNED for this one function.
  * rotated boxes (any gt angle > 0, criterion.py:616) replace the axis-aligned footprint overlap by the area of the
    Sutherland-Hodgman clip of the two footprint quadrilaterals (box_util.py:393-439, 566-589); restated in
    `clip_area` with the reference's predicates and formulas (differentiable w.r.t. the prediction's corners).

PINNED: tests/golden/criterion_*.npz hold inputs / losses / assignments / gradients produced by the reference's own
criterion.py imported in the build container (oracle/make_golden.py, mmcv stubbed with the restatement above);
tests/test_oracle_criterion.py checks this file against them.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

DEFAULT_WEIGHTS = dict(  # main.py:118-137
    matcher_cls_cost=3.0, matcher_giou_cost=2.0, matcher_center_cost=1.0, matcher_objectness_cost=0.0,
    matcher_size_cost=0.5, matcher_anglecls_cost=0.0, matcher_anglereg_cost=0.0,
    loss_giou_weight=2.0, loss_sem_cls_weight=3.0, loss_angle_cls_weight=0.1, loss_angle_reg_weight=0.5,
    loss_center_weight=1.0, loss_size_weight=0.5, point_cls_loss_weight=0.05, loss_no_object_weight=0.0)


# ------------------------------------------------------------------------------------------------ geometry
def _edge(c, i, j):
    return ((c[..., i, :] - c[..., j, :]) ** 2).sum(-1).clamp(min=1e-6).sqrt()


def _inside(c1, c2, p):
    """box_util.py:405-407: p strictly on the left of the directed clip edge c1 -> c2"""
    return bool((c2[0] - c1[0]) * (p[1] - c1[1]) > (c2[1] - c1[1]) * (p[0] - c1[0]))


def _cross_point(c1, c2, s, e):
    """box_util.py:393-402: intersection of the lines c1c2 and se"""
    dcx, dcy = c1[0] - c2[0], c1[1] - c2[1]
    dpx, dpy = s[0] - e[0], s[1] - e[1]
    n1 = c1[0] * c2[1] - c1[1] * c2[0]
    n2 = s[0] * e[1] - s[1] * e[0]
    n3 = 1.0 / (dcx * dpy - dcy * dpx)
    return torch.stack([(n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3])


def clip_area(subject, clip):
    """Twice the area (before the reference's final *0.5, applied by the caller) of `subject` [4,2] clipped by the convex
    `clip` [4,2] (box_util.py:410-439 + the shoelace sum of :583-588); 0 if nothing is left."""
    poly = [subject[i] for i in range(subject.shape[0])]
    c1 = clip[-1]
    for c2 in clip:
        src, poly = poly, []
        s = src[-1]
        for e in src:
            if _inside(c1, c2, e):
                if not _inside(c1, c2, s):
                    poly.append(_cross_point(c1, c2, s, e))
                poly.append(e)
            elif _inside(c1, c2, s):
                poly.append(_cross_point(c1, c2, s, e))
            s = e
        c1 = c2
        if not poly:
            return subject.new_zeros(())
    xs = torch.stack([v[0] for v in poly])
    ys = torch.stack([v[1] for v in poly])
    return (torch.dot(xs, torch.roll(ys, 1)) - torch.dot(ys, torch.roll(xs, 1))).abs()


def pairwise_giou(c1, c2, nactual, rotated=False):
    """c1 [B,P,8,3], c2 [B,G,8,3] camera-frame corners (y down), nactual [B] -> GIoU [B,P,G]; columns >= nactual are 0.
    rotated=False: axis-aligned footprint overlap (box_util.py:545-556); True: polygon clip where that overlap is > 0."""
    B, P, G = c1.shape[0], c1.shape[1], c2.shape[1]
    top = torch.minimum(c1[:, :, 0, 1, None], c2[:, None, :, 0, 1])          # y is negative-up: top = min
    bot = torch.maximum(c1[:, :, 4, 1, None], c2[:, None, :, 4, 1])
    height = (top - bot).clamp(min=0)
    lo = torch.maximum(c1[:, :, None, 2][..., [0, 2]], c2[:, None, :, 2][..., [0, 2]])   # corner 2 = (-l/2, ., -w/2)
    hi = torch.minimum(c1[:, :, None, 0][..., [0, 2]], c2[:, None, :, 0][..., [0, 2]])   # corner 0 = (+l/2, ., +w/2)
    wh = (hi - lo).clamp(min=0)
    valid = (torch.arange(G)[None, :] < nactual[:, None]).to(c1.dtype)[:, None, :]        # [B,1,G]
    area = wh[..., 0] * wh[..., 1] * valid
    if rotated:
        order = [3, 2, 1, 0]
        r1, r2 = c1[:, :, order][..., [0, 2]], c2[:, :, order][..., [0, 2]]    # footprints, counter-clockwise (:539-544)
        rows = []
        for b in range(B):
            cols = []
            for p in range(P):
                cols.append(torch.stack([0.5 * clip_area(r1[b, p], r2[b, g].detach())
                                         if g < int(nactual[b]) and float(area[b, p, g]) != 0.0 else area.new_zeros(())
                                         for g in range(G)]))
            rows.append(torch.stack(cols))
        area = torch.stack(rows)
    # enclosing axis-aligned box (box_util.py:466-505; y flipped, hence the swapped min/max on that axis)
    mn1, mx1 = c1.min(2).values, c1.max(2).values
    mn2, mx2 = c2.min(2).values, c2.max(2).values
    ex = (torch.maximum(mx1[:, :, None, 0], mx2[:, None, :, 0]) - torch.minimum(mn1[:, :, None, 0], mn2[:, None, :, 0])).abs()
    ey = (torch.minimum(-mx1[:, :, None, 1], -mx2[:, None, :, 1]) - torch.maximum(-mn1[:, :, None, 1], -mn2[:, None, :, 1])).abs()
    ez = (torch.maximum(mx1[:, :, None, 2], mx2[:, None, :, 2]) - torch.minimum(mn1[:, :, None, 2], mn2[:, None, :, 2])).abs()
    enclosing = ex * ey * ez
    v1 = (_edge(c1, 0, 1) * _edge(c1, 1, 2) * _edge(c1, 0, 4)).clamp(min=1e-8)
    v2 = (_edge(c2, 0, 1) * _edge(c2, 1, 2) * _edge(c2, 0, 4)).clamp(min=1e-8)
    total = v1[:, :, None] + v2[:, None, :]
    good = ((enclosing > 2e-8) & (total > 4e-8)).to(c1.dtype)
    inter = area * height
    union = (total - inter).clamp(min=1e-8)
    giou = inter / union - (1 - union / enclosing)
    return giou * good * valid


def points_in_boxes_all(points, boxes):
    """points [B,N,3], boxes [B,G,7] = (cx, cy, z_bottom, dx, dy, dz, yaw) -> int32 [B,N,G] (see the header)."""
    p = points[:, :, None, :]
    b = boxes[:, None, :, :]
    zc = b[..., 2] + b[..., 5] / 2
    inz = (p[..., 2] - zc).abs() <= b[..., 5] / 2
    sx, sy = p[..., 0] - b[..., 0], p[..., 1] - b[..., 1]
    ca, sa = torch.cos(-b[..., 6]), torch.sin(-b[..., 6])
    lx = sx * ca - sy * sa
    ly = sx * sa + sy * ca
    inxy = (lx > -b[..., 3] / 2) & (lx < b[..., 3] / 2) & (ly > -b[..., 4] / 2) & (ly < b[..., 4] / 2)
    return (inz & inxy).to(torch.int32)


# ------------------------------------------------------------------------------------------------ targets
_PER_BOX = ("gt_box_corners", "gt_box_centers", "gt_box_centers_normalized", "gt_box_sem_cls_label", "gt_box_present",
            "gt_box_sizes", "gt_box_sizes_normalized", "gt_box_angles", "gt_angle_class_label", "gt_angle_residual_label")


def count_boxes(targets, world_average=None):
    """criterion.py:660-666.  ``world_average`` stands for utils.dist.all_reduce_average (identity on one rank)."""
    nactual = targets["gt_box_present"].sum(1).long()
    total = nactual.sum()
    avg = world_average(total) if world_average is not None else total
    return nactual, float(max(float(avg), 1.0)), int(total)


def repeat_targets(targets, times):
    """criterion.py:511-600: every per-box field tiled ``times`` x along the box axis, present boxes moved to the front
    (stable), the rest zeroed."""
    out = dict(targets)
    for k in _PER_BOX:
        if k not in targets:
            continue
        t = targets[k]
        t = t.repeat(1, times, *([1] * (t.dim() - 2)))
        out[k] = t
    present = out["gt_box_present"] > 0
    order = torch.argsort((~present).to(torch.int8), dim=1, stable=True)
    count = present.sum(1)
    keep = torch.arange(present.shape[1])[None, :] < count[:, None]
    for k in _PER_BOX:
        if k not in out:
            continue
        t = out[k]
        idx = order.reshape(order.shape + (1,) * (t.dim() - 2)).expand_as(t)
        t = torch.gather(t, 1, idx)
        out[k] = t * keep.reshape(keep.shape + (1,) * (t.dim() - 2)).to(t.dtype)
    return out


# ------------------------------------------------------------------------------------------------ matcher
def huber(e, delta=1.0):
    a = e.abs()
    q = a.clamp(max=delta)
    return 0.5 * q * q + delta * (a - q)


def pair_terms(o, t):
    """The three pairwise matrices single_output_forward attaches to the outputs (criterion.py:618-631)."""
    rotated = bool((t["gt_box_angles"] > 0).any())                                    # criterion.py:616
    giou = pairwise_giou(o["box_corners"], t["gt_box_corners"], t["nactual_gt"], rotated)
    pc, ps = o["pre_box_center_unnormalized"][:, :, None], o["pre_box_size_unnormalized"][:, :, None]
    want_c = (t["gt_box_centers"][:, None] - pc) / (ps + 1e-5)
    center = (o["center_reg"][:, :, None] - want_c).abs().sum(-1)
    want_s = torch.log((t["gt_box_sizes"][:, None] + 1e-5) / (ps + 1e-5))
    size = (o["size_reg"][:, :, None] - want_s).abs().sum(-1)
    return giou, center, size


def match_costs(o, t, giou, center, size, w, focal=True):
    """final_cost [B,P,G] of criterion.py:122-196 (fp32, terms added in the reference's order)."""
    P = o["sem_cls_prob"].shape[1]
    lab = t["gt_box_sem_cls_label"][:, None, :].expand(-1, P, -1)
    if focal:
        p = o["sem_cls_prob"].sigmoid()
        neg = 0.75 * p ** 2.0 * (-(1 - p + 1e-8).log())
        pos = 0.25 * (1 - p) ** 2.0 * (-(p + 1e-8).log())
        cls = torch.gather(pos - neg, 2, lab)
    else:
        cls = -torch.gather(o["sem_cls_prob"], 2, lab)
    alab = t["gt_angle_class_label"][:, None, :].expand(-1, P, -1)
    acls = -torch.gather(o["angle_logits"], 2, alab)
    nbin = o["angle_residual_normalized"].shape[-1]
    res = torch.gather(o["angle_residual_normalized"], 2, alab)
    areg = huber(res - (t["gt_angle_residual_label"] / (np.pi / nbin))[:, None, :])
    obj = -o["objectness_prob"][:, :, None]
    return (w["matcher_cls_cost"] * cls + w["matcher_objectness_cost"] * obj + w["matcher_center_cost"] * center
            + w["matcher_giou_cost"] * (-giou) + w["matcher_size_cost"] * size + w["matcher_anglecls_cost"] * acls
            + w["matcher_anglereg_cost"] * areg).detach()


def assign(cost, nactual):
    """criterion.py:198-221: Hungarian matching per scene on the first nactual columns."""
    B, P, _ = cost.shape
    inds = torch.zeros((B, P), dtype=torch.int64)
    mask = torch.zeros((B, P), dtype=torch.float32)
    c = cost.numpy()
    for b in range(B):
        n = int(nactual[b])
        if n > 0:
            rows, cols = linear_sum_assignment(c[b, :, :n])
            inds[b, torch.from_numpy(rows)] = torch.from_numpy(cols)
            mask[b, torch.from_numpy(rows)] = 1
    return inds, mask


# ------------------------------------------------------------------------------------------------ losses
def focal_sum(logits, labels, alpha):
    """sigmoid_focal_loss(...)*P of criterion.py:77-98 with the one-hot built as in :290-301: = sum over every logit.
    labels [B,P] in [0, C]; C = no object."""
    C = logits.shape[-1]
    onehot = F.one_hot(labels, C + 1)[..., :C].to(logits.dtype)
    p = logits.sigmoid()
    ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    pt = p * onehot + (1 - p) * (1 - onehot)
    loss = ce * (1 - pt) ** 2
    loss = (alpha * onehot + (1 - alpha) * (1 - onehot)) * loss
    return loss.mean(1).sum() * logits.shape[1]


def ce_mean(logits, labels, no_object_weight):
    """F.cross_entropy(..., weight=ones with the LAST class at `no_object_weight`, reduction="mean") of criterion.py:240-246,
    366-371: the weighted mean over every row."""
    wt = torch.ones(logits.shape[-1])
    wt[-1] = no_object_weight
    return F.cross_entropy(logits.transpose(2, 1), labels, wt, reduction="mean")


def stage_losses(o, t, w, focal_alpha=0.25):
    """single_output_forward (criterion.py:602-657) for one stage.  Returns (weighted total, dict of weighted parts,
    (inds, mask)).  focal_alpha=None selects the cross-entropy class loss (cls_loss="celoss")."""
    focal = focal_alpha is not None
    giou, center, size = pair_terms(o, t)
    cost = match_costs(o, t, giou.detach(), center.detach(), size.detach(), w, focal=focal)
    inds, mask = assign(cost, t["nactual_gt"])
    nb = t["num_boxes"]
    parts = {}
    if t["num_boxes_replica"] > 0:
        logits = o["sem_cls_logits"]
        lab = torch.gather(t["gt_box_sem_cls_label"], 1, inds)
        if focal:
            lab = torch.where(mask > 0, lab, torch.full_like(lab, logits.shape[-1]))
            parts["loss_sem_cls"] = focal_sum(logits, lab, focal_alpha) / nb
        else:  # unmatched rows carry the last ("no object") class; weighted mean, not divided by num_boxes (:360-371)
            lab = torch.where(mask > 0, lab, torch.full_like(lab, logits.shape[-1] - 1))
            parts["loss_sem_cls"] = ce_mean(logits, lab, w["loss_no_object_weight"])
        alab = torch.gather(t["gt_angle_class_label"], 1, inds)
        nbin = o["angle_logits"].shape[-1]
        parts["loss_angle_cls"] = (F.cross_entropy(o["angle_logits"].transpose(2, 1), alab, reduction="none") * mask).sum() / nb
        res = torch.gather(o["angle_residual_normalized"], 2, alab[..., None])[..., 0]
        want = torch.gather(t["gt_angle_residual_label"] / (np.pi / nbin), 1, inds)
        parts["loss_angle_reg"] = (huber(res - want) * mask).sum() / nb
        parts["loss_center"] = (torch.gather(center, 2, inds[..., None])[..., 0] * mask).sum() / nb
        gsz = torch.gather(t["gt_box_sizes"], 1, inds[..., None].expand(-1, -1, 3))
        want_s = torch.log((gsz + 1e-5) / (o["pre_box_size_unnormalized"] + 1e-5))
        parts["loss_size"] = ((want_s - o["size_reg"]).abs().sum(-1) * mask).sum() / nb
        parts["loss_giou"] = (torch.gather(1 - giou, 2, inds[..., None])[..., 0] * mask).sum() / nb
    else:
        zero = o["sem_cls_logits"].sum() * 0.0
        for k in ("loss_sem_cls", "loss_angle_cls", "loss_angle_reg", "loss_center", "loss_size", "loss_giou"):
            parts[k] = zero
    with torch.no_grad():
        lg = o["sem_cls_logits"]
        pred_objects = (lg.argmax(-1) != lg.shape[-1] - 1).sum(1)
        parts["loss_cardinality"] = F.l1_loss(pred_objects.float(), t["nactual_gt"].float())
    total = 0
    for k in ("loss_giou", "loss_sem_cls", "loss_angle_cls", "loss_angle_reg", "loss_center", "loss_size"):  # criterion.py:720-728
        wk = w[k + "_weight"]
        if wk > 0:
            parts[k] = parts[k] * wk
            total = total + parts[k]
    return total, parts, (inds, mask)


def point_cls_loss(enc, t, w, focal_alpha=0.25):
    """criterion.py:270-327 (focal branch; focal_alpha=None: the cross-entropy branch :309-322)."""
    logits = enc["point_cls_logits"]
    if t["num_boxes_replica"] == 0:
        return logits.sum() * 0.0, None
    boxes = torch.cat((t["gt_box_centers"], t["gt_box_sizes"], t["gt_box_angles"][..., None]), -1).clone()
    boxes[..., 2] = boxes[..., 2] - boxes[..., 5] / 2
    inside = points_in_boxes_all(enc["seed_xyz"], boxes)
    G = inside.shape[-1]
    inside = inside * (torch.arange(G)[None, None, :] < t["nactual_gt"][:, None, None]).to(inside.dtype)
    vol = t["gt_box_sizes"].prod(-1)
    score = inside * vol[:, None, :]
    score = torch.where(score == 0, torch.full_like(score, 1000.0), score)
    score = torch.cat((score, torch.full_like(score[..., :1], 100.0)), -1)
    pick = score.argmin(-1)
    matched = pick != G
    pick = torch.where(matched, pick, torch.zeros_like(pick))
    lab = torch.gather(t["gt_box_sem_cls_label"], 1, pick)
    if focal_alpha is None:
        lab = torch.where(matched, lab, torch.full_like(lab, logits.shape[-1] - 1))
        return ce_mean(logits, lab, w["loss_no_object_weight"]), lab
    lab = torch.where(matched, lab, torch.full_like(lab, logits.shape[-1]))
    return focal_sum(logits, lab, focal_alpha) / t["num_boxes"], lab


def set_criterion(outputs, targets, weights=None, repeat_num=5, is_bilable=True, focal_alpha=0.25, world_average=None):
    """SetCriterion.forward (criterion.py:659-708).  Returns (loss, loss_dict, assignments per stage)."""
    w = dict(DEFAULT_WEIGHTS)
    w.update(weights or {})
    t = dict(targets)
    t["nactual_gt"], t["num_boxes"], t["num_boxes_replica"] = count_boxes(t, world_average)

    def rep(x):
        if repeat_num > 1:
            r = repeat_targets(x, repeat_num)
            r["nactual_gt"], r["num_boxes"], r["num_boxes_replica"] = count_boxes(r, world_average)
            return r
        return x

    loss, parts, a = stage_losses(outputs["outputs"], rep(t), w, focal_alpha)
    loss_dict, assigns = dict(parts), {"outputs": a}
    for k, o in enumerate(outputs.get("aux_outputs", [])):
        if k == 0 and is_bilable:
            tb = dict(t)
            tb["gt_box_sem_cls_label"] = torch.zeros_like(t["gt_box_sem_cls_label"])
            l, p, a = stage_losses(o, tb, w, focal_alpha)
        else:
            l, p, a = stage_losses(o, rep(t), w, focal_alpha)
        loss = loss + l
        assigns[k] = a
        for kk, v in p.items():
            loss_dict[f"{kk}_{k}"] = v
    if "enc_outputs" in outputs:
        enc = dict(outputs["enc_outputs"])
        enc["seed_xyz"] = outputs["seed_xyz"]
        l, lab = point_cls_loss(enc, t, w, focal_alpha)
        l = l * w["point_cls_loss_weight"]
        loss = loss + l
        loss_dict["enc_point_cls_loss"] = l
        assigns["point_labels"] = lab
    return loss, loss_dict, assigns
