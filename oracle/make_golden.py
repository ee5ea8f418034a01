"""Generates tests/golden/*.npz by importing the REFERENCE's Python (read-only, /root/reference) on CPU.

Run in the build container only:  python oracle/make_golden.py
The reference cannot travel to the GPU box, so its inputs / outputs / gradients are committed as small fixtures
(weights are NOT stored: both sides fill their modules with oracle/param_fill.py, keyed by parameter name).

Import recipe: `models/__init__.py` pulls MinkowskiEngine, and pc_util / scannet / the transformer
import plyfile, trimesh and mmcv at module import time; none is installed, none is used on this path.  They are
replaced by empty stub modules, and `models` is registered as a namespace package so its __init__ is skipped.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))
from oracle.param_fill import fill_module  # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True
    for name, attrs in {"mmcv": [], "mmcv.ops": ["points_in_boxes_all"], "mmcv.ops.furthest_point_sample": [],
                        "plyfile": ["PlyData", "PlyElement"], "trimesh": []}.items():
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    sys.path.insert(0, REF)
    import models.vdetr_transformer as T  # noqa
    import models.position_embedding as PE  # noqa
    from datasets.scannet import ScannetDatasetConfig  # noqa
    return T, PE, ScannetDatasetConfig


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KB")


def args_ns(**kw):
    a = dict(log_scale=512.0, rpe_quant="bilinear_4_10", angle_type="", rpe_dim=128, share_selfattn=False)
    a.update(kw)
    return Namespace(**a)


def scene(g, B, nQ, nK, edge_cases=False):
    """Random boxes + key cloud in a 8x6x3 m room offset by +1 m."""
    lo, ext = torch.tensor([1.0, 1.0, 1.0]), torch.tensor([8.0, 6.0, 3.0])
    xyz = lo + torch.rand((B, nK, 3), generator=g) * ext
    center = lo + torch.rand((B, nQ, 3), generator=g) * ext
    size = 0.2 + torch.rand((B, nQ, 3), generator=g) * 1.8
    if edge_cases:
        xyz[:, 0] = center[:, 0] + 0.5 * size[:, 0] * torch.tensor([1.0, 1.0, -1.0])  # delta == 0 for one vertex
        xyz[:, 1] = torch.tensor([30.0, -25.0, 12.0])   # |delta| > 8 m on every axis: all corners out of range
        xyz[:, 2] = center[:, 1] + torch.tensor([9.5, 0.0, -9.5])  # g slightly above 1: half of the corners padded
    return xyz, center, size


def cross_attention_cases(T, Cfg):
    cfg = Cfg()
    for name, B, nQ, nK, angle_type, grads in [("cross_attn_small", 2, 5, 7, "", "all"),
                                                ("cross_attn_rot", 2, 6, 9, "object_coords", "all"),
                                                ("cross_attn_mid", 1, 64, 512, "", "some")]:
        g = torch.Generator().manual_seed(hash(name) & 0xFFFF if False else sum(map(ord, name)))
        mod = T.GlobalShareCrossAttention(256, 4, attn_drop=0.1, proj_drop=0.1, args=args_ns(angle_type=angle_type))
        fill_module(mod)
        # cpb MLPs: larger weights so that the bias is O(1) and the table has structure
        with torch.no_grad():
            for m in mod.cpb_mlps:
                m[0].weight.mul_(2.0)
                m[2].weight.mul_(1.5)
        mod.eval()
        xyz, center, size = scene(g, B, nQ, nK, edge_cases=(name == "cross_attn_small"))
        angle = (torch.rand((B, nQ), generator=g) * 2 - 1) * 3.1 if angle_type else torch.zeros((B, nQ))
        corners = cfg.box_parametrization_to_corners(center, size, angle)
        ref_pts = T.convert_corners_camera2lidar(corners.clone())
        query = torch.randn((nQ, B, 256), generator=g).requires_grad_(True)
        key = torch.randn((nK, B, 256), generator=g).requires_grad_(True)
        wout = torch.randn((nQ, B, 256), generator=g)
        x, attn = mod(query, key, ref_pts, angle if angle_type else None, xyz)
        (x * wout).sum().backward()
        arrays = dict(query=np_(query), key=np_(key), reference_point=np_(ref_pts), reference_angle=np_(angle),
                      xyz=np_(xyz), wout=np_(wout), x=np_(x), attn=np_(attn), grad_query=np_(query.grad),
                      grad_key=np_(key.grad), angle_type=np.array(angle_type))
        for pname, p in mod.named_parameters():
            if grads == "all" or pname.startswith("cpb_mlps") or pname in ("q.weight", "k.weight", "v.bias"):
                arrays["grad_param:" + pname] = np_(p.grad)
        if name == "cross_attn_small":  # the bias alone, for the stand-alone RPE check
            tables = torch.stack([m(mod.relative_coords_table)[0] for m in mod.cpb_mlps])
            arrays["tables"] = np_(tables)
        save(name, **arrays)


def share_self_attention_case(T):
    g = torch.Generator().manual_seed(11)
    mod = T.ShareSelfAttention(256, 4, dropout=0.1)
    fill_module(mod)
    mod.eval()
    N, B = 9, 2
    tgt = torch.randn((N, B, 256), generator=g).requires_grad_(True)
    pos = torch.randn((N, B, 256), generator=g)
    wout = torch.randn((N, B, 256), generator=g)
    x, _ = mod(tgt + pos, tgt + pos, value=tgt)
    (x * wout).sum().backward()
    arrays = dict(tgt=np_(tgt), pos=np_(pos), wout=np_(wout), x=np_(x), grad_tgt=np_(tgt.grad))
    for pname, p in mod.named_parameters():
        arrays["grad_param:" + pname] = np_(p.grad)
    save("share_self_attn", **arrays)


def decoder_cases(T, Cfg):
    from models.helpers import GenericMLP  # noqa: F401  (import check)
    cfg = Cfg()
    for name, dec_nlayers, share in [("decoder_c1_l2", 2, False), ("decoder_c1_l3", 3, False),
                                     ("decoder_c1_l3_share", 3, True)]:
        a = args_ns(share_selfattn=share)
        first = T.FFNLayer(d_model=256, dim_feedforward=256, dropout=0.1)
        layer = T.GlobalDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.1, pos_for_key=False, args=a)
        dec = T.TransformerDecoder(first, layer, cfg, num_layers=dec_nlayers - 1, decoder_dim=256, mlp_dropout=0.3,
                                   mlp_norm="bn1d", mlp_act="relu", mlp_sep=True, pos_for_key=False, num_queries=64,
                                   cls_loss="focalloss_0.25", is_bilable=True, q_content="random",
                                   return_intermediate=True, args=a)
        fill_module(dec)
        with torch.no_grad():
            for l in dec.layers:
                for m in l.multihead_attn.cpb_mlps:
                    m[0].weight.mul_(2.0)
                    m[2].weight.mul_(1.5)
            for h in dec.mlp_heads:  # keep box regressions small: size = exp(reg) * prior
                for k in ("center_head", "size_head"):
                    h[k].layers[-1].weight.mul_(0.2)
        dec.eval()
        g = torch.Generator().manual_seed(100 + dec_nlayers)
        B, nK = 1, 512
        xyz, _, _ = scene(g, B, 1, nK)
        dims = [xyz.min(1)[0], xyz.max(1)[0]]
        feats = torch.randn((nK, B, 256), generator=g).requires_grad_(True)
        size_un = torch.tensor(cfg.mean_size_arr, dtype=torch.float32)[torch.randint(0, 18, (B, nK), generator=g)]
        scene_size = dims[1] - dims[0]
        enc = {"center_normalized": (xyz - dims[0][:, None]) / scene_size[:, None],
               "size_normalized": size_un / scene_size[:, None]}
        out, _ = dec(None, feats, xyz, xyz, dims, query_pos=xyz, enc_box_predictions=enc, enc_box_features=feats)
        stages = out["aux_outputs"] + [out["outputs"]]
        loss = 0
        arrays = dict(feats=np_(feats), xyz=np_(xyz), dims_min=np_(dims[0]), dims_max=np_(dims[1]),
                      center_normalized=np_(enc["center_normalized"]), size_normalized=np_(enc["size_normalized"]),
                      nstages=np.array(len(stages)))
        for s, st in enumerate(stages):
            for k in ("sem_cls_logits", "center_unnormalized", "size_unnormalized", "box_corners", "center_normalized",
                      "size_normalized", "angle_continuous", "objectness_prob"):
                arrays[f"s{s}:{k}"] = np_(st[k])
            w = torch.randn(st["sem_cls_logits"].shape, generator=g)
            loss = loss + (st["sem_cls_logits"] * w).sum() + st["center_normalized"].sum() + st["size_normalized"].sum()
            arrays[f"s{s}:w"] = np_(w)
        loss.backward()
        arrays["loss"] = np_(loss)
        arrays["grad_feats"] = np_(feats.grad)
        keep = ["query_embed.weight", "layers.0.multihead_attn.q.weight", "layers.0.multihead_attn.k.weight",
                "layers.0.multihead_attn.v.weight", "layers.0.multihead_attn.proj.bias", "first_layer.linear1.weight",
                "layers.0.linear2.weight", "norm.weight", "mlp_heads.1.center_head.layers.8.weight",
                "query_pos_projection.0.position_embedding_head.0.weight"]
        keep += ["layers.0.self_attn.k.weight", "layers.0.self_attn.q.bias"] if share else \
            ["layers.0.self_attn.in_proj_weight", "layers.0.self_attn.out_proj.weight"]
        for pname, p in dec.named_parameters():
            if pname in keep or "cpb_mlps" in pname:
                arrays["grad_param:" + pname] = np_(p.grad)
        arrays["param_names"] = np.array(sorted(n for n, _ in dec.named_parameters()))
        arrays["buffer_names"] = np.array(sorted(n for n, _ in dec.named_buffers()))
        save(name, **arrays)


def misc_cases(T, PE, Cfg):
    g = torch.Generator().manual_seed(5)
    cfg = Cfg()
    center = torch.randn((3, 11, 3), generator=g) * 2
    size = 0.1 + torch.rand((3, 11, 3), generator=g) * 2
    angle = (torch.rand((3, 11), generator=g) * 2 - 1) * 3.1
    corners = cfg.box_parametrization_to_corners(center, size, angle)
    corners0 = cfg.box_parametrization_to_corners(center, size, torch.zeros_like(angle))
    save("box_corners", center=np_(center), size=np_(size), angle=np_(angle), corners=np_(corners),
         corners_zero_angle=np_(corners0), lidar=np_(T.convert_corners_camera2lidar(corners.clone())))
    xyz = torch.rand((2, 13, 3), generator=g) * 5 + 1
    rng = [xyz.min(1)[0] - 0.1, xyz.max(1)[0] + 0.1]
    four = fill_module(PE.PositionEmbeddingCoordsSine(d_pos=256, pos_type="fourier", normalize=True))
    sine = PE.PositionEmbeddingCoordsSine(pos_type="sine", normalize=True)
    save("pos_embed", xyz=np_(xyz), rmin=np_(rng[0]), rmax=np_(rng[1]), fourier=np_(four(xyz, input_range=rng)),
         fourier_64=np_(four(xyz, num_channels=64, input_range=rng)),
         sine_256=np_(sine(xyz, num_channels=256, input_range=rng)),
         sine_100=np_(sine(xyz, num_channels=100, input_range=rng)))


def import_reference_criterion():
    """criterion.py imports three mmcv ops at module level (criterion.py:20-22).  mmcv is absent: points_in_boxes_all is
    replaced by the restatement in oracle/criterion_oracle.py (so that one function stays unpinned), the rotated-IoU ops
    are never reached with iou_type='giou'."""
    from oracle import criterion_oracle as CO
    ops = sys.modules["mmcv.ops"]
    ops.points_in_boxes_all = CO.points_in_boxes_all
    ops.diff_iou_rotated_3d = None
    m = types.ModuleType("mmcv.ops.diff_iou_rotated")
    m.box2corners = m.oriented_box_intersection_2d = None
    sys.modules["mmcv.ops.diff_iou_rotated"] = m
    import criterion as C  # noqa  (the reference's /root/reference/criterion.py)
    return C


def synthetic_stage(g, cfg, B, P, C, nbin=1, rotated=False, near=None, softmax=False):
    """One stage's box-prediction dictionary with the keys/shapes of vdetr_transformer.py:319-333, from random heads."""
    lo, ext = torch.tensor([1.0, 1.0, 1.0]), torch.tensor([8.0, 6.0, 3.0])
    pre_c = lo + torch.rand((B, P, 3), generator=g) * ext
    pre_s = 0.3 + torch.rand((B, P, 3), generator=g) * 1.5
    if near is not None:  # some priors next to given centres, so that boxes really overlap
        k = near.shape[1]
        pre_c[:, :2 * k] = near.repeat(1, 2, 1) + 0.3 * torch.randn((B, 2 * k, 3), generator=g)
    center_reg = (torch.randn((B, P, 3), generator=g) * 0.3).requires_grad_(True)
    size_reg = (torch.randn((B, P, 3), generator=g) * 0.3).requires_grad_(True)
    logits = (torch.randn((B, P, C), generator=g) * 2 - 2).requires_grad_(True)
    angle_logits = torch.randn((B, P, nbin), generator=g).requires_grad_(True)
    angle_res = (torch.randn((B, P, nbin), generator=g) * 0.5).requires_grad_(True)
    center = center_reg * pre_s + pre_c
    size = torch.exp(size_reg) * pre_s
    # a leaf: the fixture holds d loss / d corners itself (the chain into centre/size is the box decode's business)
    angle = (torch.rand((B, P), generator=g) - 0.5) * 2.0 if rotated else torch.zeros((B, P))
    corners = cfg.box_parametrization_to_corners(center.detach(), size.detach(), angle).requires_grad_(True)
    prob = torch.softmax(logits, -1)[..., :-1].detach() if softmax else logits   # compute_objectness_and_cls_prob (:74-86)
    return {"sem_cls_logits": logits, "sem_cls_prob": prob, "center_unnormalized": center, "size_unnormalized": size,
            "center_normalized": center, "size_normalized": size,
            "angle_logits": angle_logits, "angle_residual_normalized": angle_res,
            "angle_continuous": torch.zeros((B, P)), "objectness_prob": torch.rand((B, P), generator=g),
            "box_corners": corners, "pre_box_center_unnormalized": pre_c, "center_reg": center_reg,
            "pre_box_size_unnormalized": pre_s, "size_reg": size_reg}


def synthetic_targets(g, cfg, B, G, counts, nclass, rotated=False):
    lo, ext = torch.tensor([1.0, 1.0, 1.0]), torch.tensor([8.0, 6.0, 3.0])
    centers = lo + torch.rand((B, G, 3), generator=g) * ext
    sizes = 0.3 + torch.rand((B, G, 3), generator=g) * 1.7
    present = torch.zeros((B, G))
    for b, n in enumerate(counts):
        present[b, :n] = 1
    if counts[0] >= 3:       # a hole in the list: the compaction of repeat_ground_truth is exercised
        present[0, 1] = 0
    centers, sizes = centers * present[..., None], sizes * present[..., None]
    angles = (torch.rand((B, G), generator=g) * 1.2 + 0.05) * present if rotated else torch.zeros((B, G))
    corners = cfg.box_parametrization_to_corners(centers, sizes, angles) * present[..., None, None]
    labels = (torch.randint(0, nclass, (B, G), generator=g) * present.long())
    return {"gt_box_corners": corners, "gt_box_centers": centers, "gt_box_centers_normalized": centers / 10.0,
            "gt_box_sem_cls_label": labels, "gt_box_present": present, "gt_box_sizes": sizes,
            "gt_box_sizes_normalized": sizes / 10.0, "gt_box_angles": angles,
            "gt_angle_class_label": torch.zeros((B, G), dtype=torch.int64),
            "gt_angle_residual_label": torch.zeros((B, G)), "scan_idx": torch.arange(B)}


def criterion_cases(Cfg):
    C = import_reference_criterion()
    cfg = Cfg()
    base = dict(cls_loss="focalloss_0.25", is_bilable=True, repeat_num=5, iou_type="giou", point_cls_loss_weight=0.05,
                matcher_giou_cost=2.0, matcher_cls_cost=3.0, matcher_center_cost=1.0, matcher_objectness_cost=0.0,
                matcher_size_cost=0.5, matcher_anglecls_cost=0.0, matcher_anglereg_cost=0.0, loss_giou_weight=2.0,
                loss_sem_cls_weight=3.0, loss_no_object_weight=0.0, loss_angle_cls_weight=0.1, loss_angle_reg_weight=0.5,
                loss_center_weight=1.0, loss_size_weight=0.5)
    # name, B, tokens of the first stage, queries, later stages, gt slots, gt per scene, repeat_num
    for name, B, N0, P, S, G, counts, rep in [("criterion_small", 2, 96, 48, 2, 8, (5, 3), 5),
                                              ("criterion_wide", 2, 64, 16, 1, 8, (7, 0), 5),   # 5*7 gt > 16 queries; empty scene
                                              ("criterion_norepeat", 1, 80, 40, 1, 8, (6,), 1),
                                              ("criterion_empty", 1, 32, 16, 1, 8, (0,), 5),
                                              # rotated ground truth: the footprint overlap becomes a polygon clip
                                              ("criterion_rotated", 1, 48, 24, 1, 8, (5,), 5),
                                              # cls_loss="celoss": softmax cost, weighted cross entropy, no binary first stage
                                              ("criterion_celoss", 2, 64, 32, 1, 8, (4, 6), 5)]:
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        ce = name == "criterion_celoss"
        a = Namespace(**{**base, "repeat_num": rep, **({"cls_loss": "celoss", "is_bilable": False, "loss_no_object_weight": 0.25}
                                                         if ce else {})})
        crit = C.build_criterion(a, cfg)
        rot = name == "criterion_rotated"
        targets = synthetic_targets(g, cfg, B, G, counts, cfg.num_semcls, rotated=rot)
        near = targets["gt_box_centers"][:, :counts[0]] if rot else None
        nc = cfg.num_semcls + (1 if ce else 0)
        stages = [synthetic_stage(g, cfg, B, N0, nc if ce else 1, rotated=rot, near=near, softmax=ce)] + [
            synthetic_stage(g, cfg, B, P, nc, rotated=rot, near=near, softmax=ce) for _ in range(S + 1)]
        seed_xyz = torch.tensor([1.0, 1.0, 1.0]) + torch.rand((B, N0, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0])
        seed_xyz[:, :G] = targets["gt_box_centers"]                 # some seeds certainly inside a box
        point_logits = (torch.randn((B, N0, nc), generator=g) - 1).requires_grad_(True)
        outputs = {"outputs": stages[-1], "aux_outputs": stages[:-1], "seed_inds": torch.zeros((B, N0), dtype=torch.int64),
                   "seed_xyz": seed_xyz, "enc_outputs": {"point_cls_logits": point_logits}}
        # the matcher's results are internal to the reference: record them through the matcher module
        records = []
        hook = crit.matcher.register_forward_hook(lambda m, i, o: records.append(o))
        loss, loss_dict = crit(outputs, {k: v.clone() for k, v in targets.items()})
        hook.remove()
        loss.backward()
        arrays = {"loss": np_(loss), "B": np.array(B), "N0": np.array(N0), "P": np.array(P), "S": np.array(S),
                  "repeat_num": np.array(rep), "celoss": np.array(int(ce)), "seed_xyz": np_(seed_xyz), "point_cls_logits": np_(point_logits),
                  "grad:point_cls_logits": np_(point_logits.grad)}
        for k, v in targets.items():
            arrays["target:" + k] = np_(v)
        for k, v in loss_dict.items():
            arrays["loss:" + k] = np_(torch.as_tensor(v))
        order = [len(stages) - 1] + list(range(len(stages) - 1))       # the reference matches "outputs" first
        for si, rec in zip(order, records):
            arrays[f"match{si}:inds"] = np_(rec["per_prop_gt_inds"])
            arrays[f"match{si}:mask"] = np_(rec["proposal_matched_mask"])
        for si, st in enumerate(stages):
            for k in ("sem_cls_logits", "center_reg", "size_reg", "angle_logits", "angle_residual_normalized", "box_corners",
                      "pre_box_center_unnormalized", "pre_box_size_unnormalized", "objectness_prob"):
                arrays[f"stage{si}:{k}"] = np_(st[k])
                if st[k].grad is not None:
                    arrays[f"grad{si}:{k}"] = np_(st[k].grad)
        save(name, **arrays)


def nms_cases():
    """utils/nms.py is plain numpy: the reference's own picks on random scenes of overlapping boxes."""
    from utils.nms import nms_2d_faster, nms_3d_faster, nms_3d_faster_samecls  # noqa  (the reference's /root/reference/utils/nms.py)
    rng = np.random.default_rng(7)
    arrays = {}
    for ci, (K, ncls) in enumerate([(300, 18), (64, 3), (1024, 18), (5, 1)]):
        center = rng.uniform([1, 1, 1], [9, 7, 4], (K, 3)) if K != 300 else rng.uniform([1, 1, 1], [4, 3, 2], (K, 3))
        size = rng.uniform(0.3, 2.0, (K, 3))
        corners = np.stack([center + 0.5 * size * np.array(sg) for sg in
                            [(1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1)]],
                           1).astype(np.float32)
        score = rng.random(K).astype(np.float32)
        cls = rng.integers(0, ncls, K)
        boxes = np.zeros((K, 8))                                   # float64, as ap_calculator.py:192 builds it
        boxes[:, 0:3], boxes[:, 3:6], boxes[:, 6], boxes[:, 7] = corners.min(1), corners.max(1), score, cls
        arrays[f"c{ci}:corners"], arrays[f"c{ci}:score"], arrays[f"c{ci}:cls"] = corners, score, cls.astype(np.int32)
        arrays[f"c{ci}:pick_samecls"] = np.array(nms_3d_faster_samecls(boxes, 0.25), dtype=np.int64)
        arrays[f"c{ci}:pick_any"] = np.array(nms_3d_faster(boxes[:, :7], 0.25), dtype=np.int64)
        arrays[f"c{ci}:pick_samecls_old"] = np.array(nms_3d_faster_samecls(boxes, 0.5, True), dtype=np.int64)
        boxes2d = np.zeros((K, 5))                                 # ap_calculator.py:119-135: x and z extents
        boxes2d[:, 0], boxes2d[:, 1] = corners[:, :, 0].min(1), corners[:, :, 2].min(1)
        boxes2d[:, 2], boxes2d[:, 3], boxes2d[:, 4] = corners[:, :, 0].max(1), corners[:, :, 2].max(1), score
        arrays[f"c{ci}:pick_2d"] = np.array(nms_2d_faster(boxes2d, 0.25), dtype=np.int64)
    arrays["ncases"] = np.array(4)
    save("nms3d", **arrays)


AP_VARIANTS = {  # name -> get_ap_config_dict overrides (ap_calculator.py:285-321)
    "default": {},
    "any_class": dict(cls_nms=False),
    "nms2d": dict(use_3d_nms=False),
    "old_type": dict(use_old_type_nms=True, nms_iou=0.5),
    "cls_conf": dict(per_class_proposal=False, use_cls_confidence_only=True),
    "obj_conf": dict(per_class_proposal=False, conf_thresh=0.3),
    "angle": dict(angle_nms=True, angle_conf=True),
    "no_nms": dict(no_nms=True),
    "keep_empty": dict(remove_empty_box=False, conf_thresh=0.05),
    "strict_points": dict(empty_pt_thre=400),   # no box holds 400 points in scene 1: the most confident box is kept
}


def ap_inputs(seed=11, B=2, K=96, N=3000, C=18):
    rng = np.random.default_rng(seed)
    lo, hi = np.array([1.0, 1.0, 1.0]), np.array([6.0, 5.0, 3.0])
    points = rng.uniform(lo - 0.5, hi + 0.5, (B, N, 3)).astype(np.float32)
    center = rng.uniform(lo, hi, (B, K, 3)).astype(np.float32)
    size = rng.uniform(0.2, 1.6, (B, K, 3)).astype(np.float32)
    size[1, :, :] *= 0.5                                            # scene 1: small boxes, some of them empty
    yaw = np.zeros((B, K), np.float32)
    yaw[1] = rng.uniform(-1.5, 1.5, K)                              # scene 1: rotated boxes
    sg = np.array([(1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1)], np.float32)
    local = 0.5 * size[:, :, None, :] * sg[None, None]
    ca, sa = np.cos(yaw)[..., None], np.sin(yaw)[..., None]
    rot = np.stack([local[..., 0] * ca - local[..., 1] * sa, local[..., 0] * sa + local[..., 1] * ca, local[..., 2]], -1)
    corners = (center[:, :, None, :] + rot).astype(np.float32)
    sem = rng.dirichlet(np.ones(C) * 0.3, (B, K)).astype(np.float32)
    obj = rng.random((B, K)).astype(np.float32)
    ang = rng.random((B, K)).astype(np.float32)
    csa = np.concatenate([center, size, yaw[..., None]], -1).astype(np.float32)
    return dict(corners=corners, sem=sem, obj=obj, ang=ang, points=points, csa=csa)


def ap_cases():
    """utils/ap_calculator.py parse_predictions, imported with the module stubs of import_reference (+ the CUDA extension
    module names); mmcv's points_in_boxes_all is the restatement, as for the criterion."""
    from oracle import criterion_oracle as CO
    _, _, Cfg = import_reference()
    ops = sys.modules["mmcv.ops"]
    ops.nms3d = ops.nms3d_normal = None
    ops.points_in_boxes_all = CO.points_in_boxes_all
    for name in ("pointnet2", "pointnet2._ext"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import utils.ap_calculator as AP  # noqa  (the reference's /root/reference/utils/ap_calculator.py)
    x = ap_inputs()
    arrays = {"in:" + k: v for k, v in x.items()}
    for name, over in AP_VARIANTS.items():
        cfg = AP.get_ap_config_dict(dataset_config=Cfg(), **over)
        t = {k: torch.from_numpy(v.copy()) for k, v in x.items()}
        res = AP.parse_predictions(t["corners"], t["sem"], t["obj"], t["ang"], t["points"], cfg, t["csa"])
        arrays[f"{name}:count"] = np.array([len(r) for r in res], np.int64)
        flat = [d for r in res for d in r]
        arrays[f"{name}:cls"] = np.array([d[0] for d in flat], np.int64)
        arrays[f"{name}:corners"] = np.stack([d[1] for d in flat]).astype(np.float32) if flat else np.zeros((0, 8, 3), np.float32)
        arrays[f"{name}:score"] = np.array([d[2] for d in flat], np.float32)
        print(name, arrays[f"{name}:count"])
    save("parse_predictions", **arrays)


def eval_inputs(get_3d_box, seed=5, nimg=14, ncls=5):
    """A small validation set: jittered copies of the ground truth (IoU around both thresholds), duplicates, false positives,
    a class without ground truth (4), an image without ground truth (5), an image without detections (6).  Scores are
    distinct: the reference ranks equal scores in the order numpy's (unstable) sort happens to leave them."""
    rng = np.random.default_rng(seed)
    gt_all, pred_all = {}, {}
    for img in range(nimg):
        gts = []
        for _ in range(0 if img == 5 else rng.integers(1, 7)):
            size, yaw = rng.uniform(0.4, 2.0, 3), rng.uniform(-3.1, 3.1) * (img % 2)
            center = rng.uniform([0, 0, 0], [6, 2, 5])
            gts.append((int(rng.integers(0, 4)), get_3d_box(size, yaw, center).astype(np.float32), size, yaw, center))
        gt_all[f"scene{img}"] = [(c, b) for c, b, *_ in gts]
        if img == 6:
            continue
        dets = []
        for c, _, size, yaw, center in gts:
            for _ in range(rng.integers(0, 4)):
                jit = rng.choice([0.02, 0.15, 0.4])
                b = get_3d_box(size * (1 + rng.normal(0, jit, 3)).clip(0.3, 2), yaw + rng.normal(0, jit), center + rng.normal(0, jit, 3) * size)
                dets.append((c if rng.random() < 0.9 else int(rng.integers(0, ncls)), b.astype(np.float32), np.float32(rng.random())))
        for _ in range(rng.integers(2, 8)):
            b = get_3d_box(rng.uniform(0.4, 2.0, 3), rng.uniform(-3, 3), rng.uniform([0, 0, 0], [6, 2, 5]))
            dets.append((int(rng.integers(0, ncls)), b.astype(np.float32), np.float32(rng.random())))
        pred_all[f"scene{img}"] = dets
    return pred_all, gt_all


def eval_cases():
    """utils/eval_det.py eval_det_multiprocessing (the form APCalculator.compute_metrics calls) on the synthetic set."""
    import_reference()
    from utils.box_util import get_3d_box  # noqa  (reference)
    from utils.eval_det import eval_det_multiprocessing, get_iou_obb  # noqa  (reference)
    pred_all, gt_all = eval_inputs(get_3d_box)
    arrays = {}
    pf = [(i, c, b, s) for i, dets in pred_all.items() for c, b, s in dets]
    gf = [(i, c, b) for i, boxes in gt_all.items() for c, b in boxes]
    arrays["pred_img"] = np.array([int(i[5:]) for i, *_ in pf]); arrays["pred_cls"] = np.array([c for _, c, _, _ in pf])
    arrays["pred_box"] = np.stack([b for _, _, b, _ in pf]); arrays["pred_score"] = np.array([s for *_, s in pf], np.float32)
    arrays["gt_img"] = np.array([int(i[5:]) for i, *_ in gf]); arrays["gt_cls"] = np.array([c for _, c, _ in gf])
    arrays["gt_box"] = np.stack([b for *_, b in gf])
    arrays["pred_imgs"] = np.array(sorted(int(i[5:]) for i in pred_all)); arrays["gt_imgs"] = np.array(sorted(int(i[5:]) for i in gt_all))
    for thr in (0.25, 0.5):
        rec, prec, ap = eval_det_multiprocessing(pred_all, gt_all, ovthresh=thr, get_iou_func=get_iou_obb)
        for c in ap:
            arrays[f"t{thr}:c{c}:ap"] = np.float64(ap[c])
            arrays[f"t{thr}:c{c}:rec"] = np.asarray(rec[c], np.float64)
            arrays[f"t{thr}:c{c}:prec"] = np.asarray(prec[c], np.float64)
        print(thr, {c: round(float(a), 4) for c, a in ap.items()})
    save("eval_det", **arrays)


def apcalc_inputs(get_3d_box, seed=21, nbatch=3, B=2, K=48, G=6, N=2000, C=18):
    """Batches of decoder outputs / targets with the keys APCalculator.step_meter reads (ap_calculator.py:378-398): predictions
    are jittered copies of the ground truth plus random boxes; boxes in the upright camera frame (corners) and the depth
    frame (centre / size / angle rows for the point test), points scattered around the box centres."""
    rng = np.random.default_rng(seed)
    batches = []
    for _ in range(nbatch):
        out = {k: [] for k in ("box_corners", "sem_cls_prob", "objectness_prob", "angle_prob", "center_unnormalized",
                               "size_unnormalized", "angle_continuous")}
        tgt = {k: [] for k in ("point_clouds", "gt_box_corners", "gt_box_sem_cls_label", "gt_box_present")}
        for _b in range(B):
            ng = int(rng.integers(1, G + 1))
            g_size, g_yaw = rng.uniform(0.4, 1.8, (G, 3)), rng.uniform(-1.5, 1.5, G)
            g_ctr, g_cls = rng.uniform([0, -1, 0], [6, 1, 5], (G, 3)), rng.integers(0, C, G)
            present = (np.arange(G) < ng).astype(np.int64)
            tgt["gt_box_corners"].append(np.stack([get_3d_box(g_size[j], g_yaw[j], g_ctr[j]) for j in range(G)]).astype(np.float32))
            tgt["gt_box_sem_cls_label"].append(g_cls), tgt["gt_box_present"].append(present)
            src = rng.integers(-1, ng, K)                             # -1: a random box
            size = np.where(src[:, None] >= 0, g_size[src] * (1 + rng.normal(0, 0.08, (K, 3))), rng.uniform(0.4, 1.8, (K, 3)))
            yaw = np.where(src >= 0, g_yaw[src] + rng.normal(0, 0.05, K), rng.uniform(-1.5, 1.5, K))
            ctr = np.where(src[:, None] >= 0, g_ctr[src] + rng.normal(0, 0.08, (K, 3)), rng.uniform([0, -1, 0], [6, 1, 5], (K, 3)))
            out["box_corners"].append(np.stack([get_3d_box(size[j], yaw[j], ctr[j]) for j in range(K)]).astype(np.float32))
            sem = rng.dirichlet(np.ones(C) * 0.2, K)
            hit = src >= 0
            sem[hit] = 0.2 * sem[hit]
            sem[hit, g_cls[src[hit]]] += 0.8
            out["sem_cls_prob"].append(sem.astype(np.float32))
            out["objectness_prob"].append(rng.random(K).astype(np.float32)), out["angle_prob"].append(rng.random(K).astype(np.float32))
            depth_ctr = np.stack([ctr[:, 0], ctr[:, 2], -ctr[:, 1]], 1)     # camera (x, y, z) -> depth (x, z, -y)
            out["center_unnormalized"].append(depth_ctr.astype(np.float32))
            out["size_unnormalized"].append(size[:, [0, 2, 1]].astype(np.float32)), out["angle_continuous"].append(yaw.astype(np.float32))
            pts = depth_ctr[rng.integers(0, K // 2, N)] + rng.normal(0, 0.25, (N, 3))   # the second half of the boxes sees few points
            tgt["point_clouds"].append(pts.astype(np.float32))
        batches.append(({k: np.stack(v) for k, v in out.items()}, {k: np.stack(v) for k, v in tgt.items()}))
    return batches


def apcalc_cases():
    """utils/ap_calculator.py APCalculator (step_meter over 3 batches, compute_metrics, metrics_to_str) with its default
    evaluation settings (get_ap_config_dict; remove_empty_box on), imported as in ap_cases."""
    from oracle import criterion_oracle as CO
    _, _, Cfg = import_reference()
    ops = sys.modules["mmcv.ops"]
    ops.nms3d = ops.nms3d_normal = None
    ops.points_in_boxes_all = CO.points_in_boxes_all
    for name in ("pointnet2", "pointnet2._ext"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import utils.ap_calculator as AP  # noqa  (reference)
    from utils.box_util import get_3d_box  # noqa  (reference)
    cfg = Cfg()
    batches = apcalc_inputs(get_3d_box)
    calc = AP.APCalculator(dataset_config=cfg, ap_iou_thresh=[0.25, 0.5], class2type_map=cfg.class2type, exact_eval=True,
                           ap_config_dict=AP.get_ap_config_dict(dataset_config=cfg, remove_empty_box=True))
    arrays = {}
    for bi, (out, tgt) in enumerate(batches):
        calc.step_meter({"outputs": {k: torch.from_numpy(v.copy()) for k, v in out.items()}},
                        {k: torch.from_numpy(v.copy()) for k, v in tgt.items()})
        arrays.update({f"b{bi}:out:{k}": v for k, v in out.items()})
        arrays.update({f"b{bi}:tgt:{k}": v for k, v in tgt.items()})
    ret = calc.compute_metrics()
    for thr, d in ret.items():
        arrays[f"t{thr}:keys"] = np.array(list(d.keys()))
        arrays[f"t{thr}:values"] = np.array([float(v) for v in d.values()], np.float64)
    arrays["text"] = np.array(calc.metrics_to_str(ret))
    arrays["nbatch"] = np.array(len(batches))
    arrays["class_ids"] = np.array(list(cfg.class2type.keys()))
    arrays["class_names"] = np.array(list(cfg.class2type.values()))
    print(calc.metrics_to_str(ret, per_class=False))
    save("ap_calculator", **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    T, PE, Cfg = import_reference()
    cross_attention_cases(T, Cfg)
    share_self_attention_case(T)
    decoder_cases(T, Cfg)
    misc_cases(T, PE, Cfg)
    criterion_cases(Cfg)
    nms_cases()
    ap_cases()
    eval_cases()
    apcalc_cases()


if __name__ == "__main__":
    main()
