"""Set criterion of V-DETR on the device (reference criterion.py).

Same entry points as the reference -- ``build_criterion(args, dataset_config)``, ``SetCriterion.forward(outputs, targets)
-> (loss, loss_dict)``, ``Matcher.forward(outputs, targets)`` -- with the work behind them moved into five HIP kernels
(v-detr_amd/csrc/criterion.hip, C-ABI in include/vdetr_hip.h):

  reference (per stage, 9 stages per step)                         here
  -------------------------------------------------------------    ------------------------------------------------------
  three [B,P,G] matrices from ~60 torch launches (:618-631)        vdetr_match_cost_f32: one launch, cost stored box-major
  final_cost.cpu() + scipy linear_sum_assignment + H2D (:198-221)  vdetr_lsa_f64: ONE launch for all stages and scenes
  gathers from the matrices + autograd through them (:329-509)     vdetr_set_loss_f32: losses and gradients, one launch
  repeat_ground_truth: deepcopy + ~25 masked copies (:511-600)     vdetr_gt_prepare_f32: one launch per step
  mmcv points_in_boxes_all + argmin (:270-289)                     vdetr_point_labels_f32

No host synchronisation anywhere (the reference has >= 30 per step: ``.item()``, ``.cpu()``, Python slicing with device
integers), so the whole training step including the criterion can be captured in one hipGraph.  ``nactual_gt`` and
``num_boxes`` stay device tensors.

Scope: ``cls_loss`` "focalloss_<alpha>" (default) or "celoss" (3DETR's weighted cross entropy), ``iou_type="giou"``; the
mmcv-based diou / iou variants raise.  Rotated ground
truth (any ``gt_box_angles`` > 0, criterion.py:616) switches the GIoU's footprint overlap to the polygon clip of
box_util.py:566-589 through a device flag -- the reference decides that with ``.item()``.  There is no CPU path: CPU
tensors raise.
"""
import ctypes

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L

_DIFF = ("sem_cls_logits", "center_reg", "size_reg", "box_corners", "angle_logits", "angle_residual_normalized")
LOSS_NAMES = ("loss_sem_cls", "loss_angle_cls", "loss_angle_reg", "loss_center", "loss_size", "loss_giou", "loss_cardinality")


def pack_ground_truth(targets):
    """The per-box target fields as records [B, G, VDETR_GT_FLOATS] (layout: include/vdetr_hip.h VDETR_GT_*)."""
    c = targets["gt_box_corners"]
    L.require_gpu(c, "gt_box_corners")
    B, G = c.shape[:2]
    f = torch.float32
    cols = [c.reshape(B, G, 24).to(f), targets["gt_box_centers"].to(f), targets["gt_box_sizes"].to(f),
            targets["gt_box_angles"].to(f)[..., None], targets["gt_box_sem_cls_label"].to(f)[..., None],
            targets["gt_angle_class_label"].to(f)[..., None], targets["gt_angle_residual_label"].to(f)[..., None],
            targets["gt_box_present"].to(f)[..., None], c.new_zeros((B, G, 1), dtype=f)]
    return torch.cat(cols, dim=-1).contiguous()


class PreparedTargets:
    """What every stage needs of the ground truth, resident on the device: records (plain and repeated), box counts and
    the two ``num_boxes`` normalisers (criterion.py:592-600, 660-666)."""

    def __init__(self, targets, repeat_num):
        gt = pack_ground_truth(targets)
        B, G, _ = gt.shape
        dev = gt.device
        self.B, self.G, self.repeat = B, G, max(int(repeat_num), 1)
        self.gt = gt
        self.nactual = torch.empty(B, dtype=torch.int64, device=dev)
        self.sums = torch.empty(4, dtype=torch.float32, device=dev)  # box counts (plain, repeated), rotated flag, pad
        if self.repeat > 1:
            self.gt_rep = torch.empty((B, G * self.repeat, L.VDETR_GT_FLOATS), dtype=torch.float32, device=dev)
            self.nactual_rep = torch.empty(B, dtype=torch.int64, device=dev)
        else:
            self.gt_rep, self.nactual_rep = None, None
        L.check(L.lib().vdetr_gt_prepare_f32(L.ptr(gt), B, G, self.repeat, L.ptr(self.gt_rep), L.ptr(self.nactual),
                                             L.ptr(self.nactual_rep), L.ptr(self.sums), L.stream_ptr()), "gt_prepare")
        # all_reduce_average(nactual.sum()) then clamp(min=1)  (criterion.py:593, 661; utils/dist.py)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            counts = self.sums[:2]
            dist.all_reduce(counts)
            counts /= dist.get_world_size()
        self.num_boxes = self.sums[:2].clamp(min=1.0)
        self.rotated = self.sums[2:3]  # device flag: any gt angle > 0 on this rank (criterion.py:616 is evaluated per rank)
        if self.repeat == 1:
            self.gt_rep, self.nactual_rep = self.gt, self.nactual

    def stage(self, repeated):
        """-> (records, slots, nactual, num_boxes[1]) of the repeated or the plain list"""
        if repeated and self.repeat > 1:
            return self.gt_rep, self.G * self.repeat, self.nactual_rep, self.num_boxes[1:2]
        return self.gt, self.G, self.nactual, self.num_boxes[0:1]


class Matcher(nn.Module):
    """criterion.py:100-228.  ``forward`` matches ONE stage (API parity); SetCriterion batches all stages in one solve."""

    def __init__(self, cls_loss, cost_class, cost_objectness, cost_giou, cost_center, cost_size, args):
        super().__init__()
        self.cls_loss = cls_loss
        self.cost_class, self.cost_objectness, self.cost_giou = cost_class, cost_objectness, cost_giou
        self.cost_center, self.cost_size = cost_center, cost_size
        self.matcher_anglecls_cost = args.matcher_anglecls_cost
        self.matcher_anglereg_cost = args.matcher_anglereg_cost

    def cost(self, o, records, G, nactual, label_override=-1, want_giou=False, rotated=None):
        """Launches the pairwise kernel of one stage.  Returns (cost_t [B,G,P], giou_t or None).  ``rotated``: device scalar
        (PreparedTargets.rotated) selecting the polygon-clip footprint overlap, None = axis-aligned."""
        d, keep = self._cost_desc(o, records, G, nactual, label_override, want_giou, rotated)
        L.check(L.lib().vdetr_match_cost_f32(ctypes.byref(d), L.stream_ptr()), "match_cost")
        return keep["cost_t"], keep["giou_t"]

    def cost_batch(self, items, rotated=None):
        """items: [(stage outputs, records, G, nactual, label_override)] -> [cost_t]; ONE launch for all stages."""
        built = [self._cost_desc(*it, rotated=rotated) for it in items]
        arr = (L.MatchDesc * len(built))(*[d for d, _ in built])
        L.check(L.lib().vdetr_match_cost_batch_f32(arr, len(built), L.stream_ptr()), "match_cost")
        return [keep["cost_t"] for _, keep in built]

    def _cost_desc(self, o, records, G, nactual, label_override=-1, want_giou=False, rotated=None):
        cls = o["sem_cls_prob"].detach().contiguous()
        B, P, C = cls.shape
        A = o["angle_logits"].shape[-1]
        d = L.MatchDesc()
        d.B, d.P, d.G, d.C, d.A = B, P, G, C, A
        d.cls_kind = L.VDETR_CLS_SIGMOID if self.cls_loss.split("_")[0] == "focalloss" else L.VDETR_CLS_SOFTMAX
        d.label_override = label_override
        d.w_cls, d.w_objectness, d.w_center, d.w_giou = self.cost_class, self.cost_objectness, self.cost_center, self.cost_giou
        d.w_size, d.w_angle_cls, d.w_angle_reg = self.cost_size, self.matcher_anglecls_cost, self.matcher_anglereg_cost
        keep = {"cls": cls, "gt": records, "nactual": nactual, "rotated": rotated}
        for dst, src in (("objectness", "objectness_prob"), ("center_reg", "center_reg"), ("size_reg", "size_reg"),
                         ("pre_center", "pre_box_center_unnormalized"), ("pre_size", "pre_box_size_unnormalized"),
                         ("corners", "box_corners"), ("angle_logits", "angle_logits"),
                         ("angle_res_norm", "angle_residual_normalized")):
            t = o[src].detach()
            L.require_gpu(t, src)
            L.require_float(t, src)
            keep[dst] = t.contiguous()
        keep["cost_t"] = cls.new_empty((B, G, P))
        keep["giou_t"] = cls.new_empty((B, G, P)) if want_giou else None
        for k, t in keep.items():
            setattr(d, k, t.data_ptr() if t is not None else None)
        return d, keep

    @staticmethod
    def solve(problems, status=None):
        """problems: list of (cost_t [B,G,P], nactual [B][, row_repeat]).  One launch per 16 problems.  Returns
        [(inds, mask)].  ``row_repeat`` > 1 promises that the first nactual rows of cost_t are that many identical tiles.
        ``status``: optional zero-filled int32 [sum B, 2] (<= 16 problems) -> (invalid flag, row scans) per scene."""
        out = []
        for s in range(0, len(problems), L.VDETR_LSA_MAX_PROBLEMS):
            chunk = problems[s:s + L.VDETR_LSA_MAX_PROBLEMS]
            batch = L.LsaBatch()
            batch.nproblems = len(chunk)
            for k, (cost_t, nactual, *hint) in enumerate(chunk):
                B, G, P = cost_t.shape
                inds = torch.empty((B, P), dtype=torch.int64, device=cost_t.device)
                mask = torch.empty((B, P), dtype=torch.float32, device=cost_t.device)
                pr = batch.p[k]
                pr.cost_t, pr.nactual, pr.inds, pr.mask = cost_t.data_ptr(), nactual.data_ptr(), inds.data_ptr(), mask.data_ptr()
                pr.B, pr.P, pr.G = B, P, G
                pr.row_repeat = int(hint[0]) if hint else 0
                out.append((inds, mask))
            L.check(L.lib().vdetr_lsa_f64(ctypes.byref(batch), L.ptr(status), L.stream_ptr()), "lsa")
        return out

    @torch.no_grad()
    def forward(self, outputs, targets):
        """outputs: one stage's dictionary; targets: the reference's target dictionary (with ``nactual_gt``) or a
        (records, G, nactual) triple.  Returns per_prop_gt_inds / proposal_matched_mask (criterion.py:223-227; the
        ``assignments`` list of index pairs is only consumed by commented-out code there and is not produced)."""
        if isinstance(targets, dict):
            records = pack_ground_truth(targets)
            G = records.shape[1]
            nactual = targets["nactual_gt"] if "nactual_gt" in targets else targets["gt_box_present"].sum(1).long()
            nactual = nactual.to(torch.int64).contiguous()
        else:
            records, G, nactual = targets
        rotated = None
        if isinstance(targets, dict):
            rotated = (targets["gt_box_angles"] > 0).any().to(torch.float32).reshape(1)   # criterion.py:616, on the device
        cost_t, _ = self.cost(outputs, records, G, nactual, rotated=rotated)
        inds, mask = self.solve([(cost_t, nactual)])[0]
        return {"per_prop_gt_inds": inds, "proposal_matched_mask": mask}


class _CriterionFn(torch.autograd.Function):
    """All stages' losses as ONE autograd node: forward launches the kernels (which also produce the gradients of the
    weighted total w.r.t. every differentiable input into one flat buffer), backward scales that buffer."""

    @staticmethod
    def forward(ctx, crit, prep, stages, point, *diff):
        dev = diff[0].device
        ns = len(stages)
        B0 = diff[0].shape[0]
        # per stage: 8 loss slots + B 64-bit cardinality tickets (kept in the same zero-filled allocation)
        losses = torch.zeros((ns + 1, 8 + 2 * B0), dtype=torch.float32, device=dev)
        sizes = [t.numel() for t in diff]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for t, n in zip(diff, sizes):
            grads.append(flat[off:off + n].view(t.shape))
            off += n
        m = crit.matcher
        # 1) cost matrices of every stage, 2) one assignment launch, 3) losses + gradients per stage
        metas = []
        for o, repeated, override in stages:
            records, G, nactual, nb = prep.stage(repeated)
            metas.append((records, G, nactual, nb, override))
        costs = m.cost_batch([(o, mt[0], mt[1], mt[2], mt[4]) for (o, _, _), mt in zip(stages, metas)], rotated=prep.rotated)
        problems = [(c, mt[2], prep.repeat if rep else 0) for c, mt, (_, rep, _) in zip(costs, metas, stages)]
        matches = m.solve(problems)
        lib, st = L.lib(), L.stream_ptr()
        keep, loss_descs = [], []
        for si, ((o, repeated, override), (records, G, nactual, nb, _), (inds, mask)) in enumerate(zip(stages, metas, matches)):
            ins = [t.detach().contiguous() for t in diff[si * 6:si * 6 + 6]]
            g = grads[si * 6:si * 6 + 6]
            B, P, C = ins[0].shape
            d = L.SetLossDesc()
            d.B, d.P, d.G, d.C, d.A, d.label_override = B, P, G, C, ins[4].shape[-1], override
            d.focal_alpha = crit.focal_alpha
            d.cls_kind = L.VDETR_CLS_SIGMOID if crit.focal else L.VDETR_CLS_SOFTMAX
            d.w_no_object = crit.no_object_weight
            w = crit.loss_weight_dict
            d.w_cls, d.w_angle_cls, d.w_angle_reg = w["loss_sem_cls_weight"], w["loss_angle_cls_weight"], w["loss_angle_reg_weight"]
            d.w_center, d.w_size, d.w_giou = w["loss_center_weight"], w["loss_size_weight"], w["loss_giou_weight"]
            pre_c = o["pre_box_center_unnormalized"].detach().contiguous()
            pre_s = o["pre_box_size_unnormalized"].detach().contiguous()
            d.cls_logits, d.center_reg, d.size_reg, d.corners = (ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(),
                                                                 ins[3].data_ptr())
            d.angle_logits, d.angle_res_norm = ins[4].data_ptr(), ins[5].data_ptr()
            d.pre_center, d.pre_size = pre_c.data_ptr(), pre_s.data_ptr()
            d.gt, d.nactual, d.inds, d.mask, d.labels = records.data_ptr(), nactual.data_ptr(), inds.data_ptr(), mask.data_ptr(), None
            d.num_boxes = nb.data_ptr()
            d.rotated = prep.rotated.data_ptr()
            d.losses = losses[si].data_ptr()
            d.card_ws = losses[si].data_ptr() + 32
            (d.d_cls_logits, d.d_center_reg, d.d_size_reg, d.d_corners, d.d_angle_logits,
             d.d_angle_res_norm) = (t.data_ptr() for t in g)
            loss_descs.append(d)
            keep.append((ins, pre_c, pre_s))
        point_labels = None
        if point is not None:
            seed_xyz = point.detach().contiguous()
            logits = diff[-1].detach().contiguous()
            B, N, C = logits.shape
            records, G, nactual, nb = prep.stage(False)
            point_labels = torch.empty((B, N), dtype=torch.int64, device=dev)
            # seeds outside every box: label C (no class) for the focal loss, the trailing "no object" class for cross entropy
            inside = None if crit.focal else torch.zeros(1, dtype=torch.float32, device=dev)
            L.check(lib.vdetr_point_labels_f32(L.ptr(seed_xyz), L.ptr(records), L.ptr(nactual), B, N, G,
                                               C if crit.focal else C - 1, L.ptr(point_labels), L.ptr(inside), st), "point_labels")
            d = L.SetLossDesc()
            d.B, d.P, d.G, d.C, d.A, d.label_override = B, N, G, C, 1, -1
            d.focal_alpha = crit.focal_alpha
            d.cls_kind = L.VDETR_CLS_SIGMOID if crit.focal else L.VDETR_CLS_SOFTMAX
            d.w_no_object = crit.no_object_weight
            d.ce_rows_matched = inside.data_ptr() if inside is not None else None
            d.w_cls = crit.args.point_cls_loss_weight
            d.cls_logits, d.labels, d.nactual, d.num_boxes = logits.data_ptr(), point_labels.data_ptr(), nactual.data_ptr(), nb.data_ptr()
            d.losses, d.d_cls_logits = losses[ns].data_ptr(), grads[-1].data_ptr()
            d.card_ws = losses[ns].data_ptr() + 32
            loss_descs.append(d)
            keep.append((seed_xyz, logits, inside))
        # every stage's losses + gradients and the seed-point loss: ONE launch
        arr = (L.SetLossDesc * len(loss_descs))(*loss_descs)
        L.check(lib.vdetr_set_loss_batch_f32(arr, len(loss_descs), st), "set_loss")
        total = losses[:, 7].sum()
        ctx.flat, ctx.grads = flat, grads
        ctx.mark_non_differentiable(losses)
        ctx.matches, ctx.point_labels = matches, point_labels
        crit._last = (matches, point_labels)
        return total, losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        # a fresh product (one launch, as an in-place scale would be): the cached buffer stays intact, so a second backward
        # through this node (retain_graph, gradient-accumulation probes) sees the same gradients, and what is returned
        # aliases nothing a later step can mutate
        prod = ctx.flat * g_total
        base = ctx.flat.storage_offset()
        return (None, None, None, None,
                *[prod[g.storage_offset() - base:g.storage_offset() - base + g.numel()].view(g.shape) for g in ctx.grads])


class SetCriterion(nn.Module):
    """criterion.py:231-708."""

    def __init__(self, args, matcher, dataset_config, loss_weight_dict):
        super().__init__()
        self.args, self.dataset_config, self.matcher = args, dataset_config, matcher
        self.loss_weight_dict = dict(loss_weight_dict)
        self.is_bilable, self.repeat_num, self.iou_type = args.is_bilable, args.repeat_num, args.iou_type
        if self.iou_type != "giou":
            raise NotImplementedError("iou_type 'diou' / 'iou' need mmcv's rotated-IoU ops (criterion.py:21-22); only 'giou'")
        # class loss: "focalloss_<alpha>" (the default, main.py:127) or the weighted cross entropy of 3DETR ("celoss": the
        # logits then carry a trailing "no object" class weighted by loss_no_object_weight, criterion.py:240-246)
        self.focal = args.cls_loss.split("_")[0] == "focalloss"
        self.focal_alpha = float(args.cls_loss.split("_")[1]) if self.focal else 0.0
        self.no_object_weight = float(self.loss_weight_dict.pop("loss_no_object_weight", 0.0))  # criterion.py:240,245
        if not self.focal and self.is_bilable:
            raise ValueError("cls_loss='celoss' with is_bilable: the reference's class-weight vector does not fit the binary "
                             "first stage (criterion.py:242-246, 679-684)")
        self._last = None

    def prepare_targets(self, targets):
        """Everything that depends on the targets only (runs one kernel and, on several ranks, the all-reduce of the box
        count): call it before a captured region, pass the result as ``targets``."""
        return targets if isinstance(targets, PreparedTargets) else PreparedTargets(targets, self.repeat_num)

    def forward(self, outputs, targets):
        prep = self.prepare_targets(targets)
        aux = list(outputs.get("aux_outputs", []))
        # (stage dictionary, repeated ground truth?, label override); "outputs" first as in the reference (:669-673)
        stages = [(outputs["outputs"], True, -1)]
        for k, o in enumerate(aux):
            stages.append((o, False, 0) if (k == 0 and self.is_bilable) else (o, True, -1))
        diff = [o[k] for o, _, _ in stages for k in _DIFF]
        point = None
        if "enc_outputs" in outputs:
            assert "point_cls_logits" in outputs["enc_outputs"]
            point = outputs["seed_xyz"]
            diff.append(outputs["enc_outputs"]["point_cls_logits"])
        for t in diff:
            L.require_gpu(t, "criterion input")
            L.require_float(t, "criterion input")
        total, losses = _CriterionFn.apply(self, prep, stages, point, *diff)
        loss_dict = {}
        for si in range(len(stages)):
            suffix = "" if si == 0 else f"_{si - 1}"
            for t, name in enumerate(LOSS_NAMES):
                loss_dict[name + suffix] = losses[si, t]
        if point is not None:
            loss_dict["enc_point_cls_loss"] = losses[len(stages), 0]
        return total, loss_dict

    def last_assignments(self):
        """[(per_prop_gt_inds, proposal_matched_mask)] of the last forward, "outputs" first, and the seed-point labels."""
        return self._last


def build_criterion(args, dataset_config):
    """criterion.py:711-739"""
    matcher = Matcher(cls_loss=args.cls_loss, cost_class=args.matcher_cls_cost, cost_giou=args.matcher_giou_cost,
                      cost_center=args.matcher_center_cost, cost_objectness=args.matcher_objectness_cost,
                      cost_size=args.matcher_size_cost, args=args)
    loss_weight_dict = {
        "loss_giou_weight": args.loss_giou_weight, "loss_sem_cls_weight": args.loss_sem_cls_weight,
        "loss_no_object_weight": args.loss_no_object_weight, "loss_angle_cls_weight": args.loss_angle_cls_weight,
        "loss_angle_reg_weight": args.loss_angle_reg_weight, "loss_center_weight": args.loss_center_weight,
        "loss_size_weight": args.loss_size_weight}
    return SetCriterion(args, matcher, dataset_config, loss_weight_dict)


def default_criterion_args(**overrides):
    """The criterion-related defaults of the reference's argument parser (main.py:86-137)."""
    from argparse import Namespace
    a = dict(cls_loss="focalloss_0.25", is_bilable=True, repeat_num=5, iou_type="giou", point_cls_loss_weight=0.05,
             matcher_giou_cost=2.0, matcher_cls_cost=3.0, matcher_center_cost=1.0, matcher_objectness_cost=0.0,
             matcher_size_cost=0.5, matcher_anglecls_cost=0.0, matcher_anglereg_cost=0.0, loss_giou_weight=2.0,
             loss_sem_cls_weight=3.0, loss_no_object_weight=0.0, loss_angle_cls_weight=0.1, loss_angle_reg_weight=0.5,
             loss_center_weight=1.0, loss_size_weight=0.5)
    a.update(overrides)
    return Namespace(**a)
