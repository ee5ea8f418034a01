"""CPU tests (no GPU): the criterion oracle against the reference's own criterion.py (fixtures written by
oracle/make_golden.py) and the assignment restatement against scipy (the reference's dependency, criterion.py:19)."""
import os

import numpy as np
import pytest
import torch

from oracle import criterion_oracle as CO
from oracle.lsa_oracle import linear_sum_assignment as lsa_restated

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["criterion_small", "criterion_wide", "criterion_norepeat", "criterion_empty", "criterion_rotated", "criterion_celoss"]
DIFF = ("sem_cls_logits", "center_reg", "size_reg", "angle_logits", "angle_residual_normalized", "box_corners")


def load_case(name, device="cpu"):
    """-> (outputs dict as the model returns it, targets, raw npz)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    nst = int(z["S"]) + 2
    stages = []
    for si in range(nst):
        st = {}
        for k in DIFF + ("pre_box_center_unnormalized", "pre_box_size_unnormalized", "objectness_prob"):
            t = torch.from_numpy(z[f"stage{si}:{k}"]).to(device)
            st[k] = t.requires_grad_(True) if k in DIFF else t
        ce = "celoss" in z.files and int(z["celoss"])  # cls_loss="celoss": the matcher sees softmax[..., :-1] (:80-86)
        st["sem_cls_prob"] = torch.softmax(st["sem_cls_logits"].detach(), -1)[..., :-1] if ce else st["sem_cls_logits"]
        stages.append(st)
    targets = {k[len("target:"):]: torch.from_numpy(z[k]).to(device) for k in z.files if k.startswith("target:")}
    point_logits = torch.from_numpy(z["point_cls_logits"]).to(device).requires_grad_(True)
    outputs = {"outputs": stages[-1], "aux_outputs": stages[:-1], "seed_xyz": torch.from_numpy(z["seed_xyz"]).to(device),
               "seed_inds": torch.zeros(z["seed_xyz"].shape[:2], dtype=torch.int64, device=device),
               "enc_outputs": {"point_cls_logits": point_logits}}
    return outputs, targets, z


def check_against_golden(z, outputs, loss, loss_dict, matches, rtol=1e-4, atol=1e-5):
    """matches: {stage index: (inds, mask)} in the fixture's stage numbering (last = "outputs")."""
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=rtol, atol=atol)
    for k in z.files:
        if k.startswith("loss:"):
            np.testing.assert_allclose(float(loss_dict[k[5:]].detach()), float(z[k]), rtol=rtol, atol=atol, err_msg=k)
    nst = int(z["S"]) + 2
    for si in range(nst):
        inds, mask = matches[si]
        m = z[f"match{si}:mask"]
        np.testing.assert_array_equal(mask.cpu().numpy(), m, err_msg=f"matched mask of stage {si}")
        # identical columns (repeated ground truth) make the replica index of a match arbitrary only if the costs
        # differ in the last bit; the golden inputs are the same bits, so the indices must agree exactly
        np.testing.assert_array_equal(inds.cpu().numpy() * (m > 0), z[f"match{si}:inds"] * (m > 0), err_msg=f"stage {si}")
    stages = outputs["aux_outputs"] + [outputs["outputs"]]
    for si, st in enumerate(stages):
        for k in DIFF:
            key = f"grad{si}:{k}"
            if key in z.files:
                g = st[k].grad
                g = torch.zeros_like(st[k]) if g is None else g
                np.testing.assert_allclose(g.cpu().numpy(), z[key], rtol=1e-3, atol=1e-6, err_msg=key)
    g = outputs["enc_outputs"]["point_cls_logits"].grad
    np.testing.assert_allclose(g.cpu().numpy(), z["grad:point_cls_logits"], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_criterion(name):
    outputs, targets, z = load_case(name)
    ce = "celoss" in z.files and int(z["celoss"])
    loss, loss_dict, assigns = CO.set_criterion(outputs, targets, repeat_num=int(z["repeat_num"]),
                                                **(dict(focal_alpha=None, is_bilable=False,
                                                        weights={"loss_no_object_weight": 0.25}) if ce else {}))
    loss.backward()
    nst = int(z["S"]) + 2
    matches = {nst - 1: assigns["outputs"], **{k: assigns[k] for k in range(nst - 1)}}
    check_against_golden(z, outputs, loss, loss_dict, matches)


def test_assignment_restatement_matches_scipy():
    from scipy.optimize import linear_sum_assignment as sp
    rng = np.random.default_rng(0)
    mats = [np.zeros((0, 4)), np.zeros((3, 0)), np.ones((6, 6)), np.ones((9, 4)), np.ones((4, 9))]
    for shape in [(1, 1), (5, 7), (7, 5), (20, 20), (64, 300), (300, 64)]:
        mats.append(rng.random(shape).astype(np.float32))
        mats.append(rng.integers(0, 3, shape).astype(np.float32))           # tie-heavy
        base = rng.random((shape[0], max(shape[1] // 5, 1))).astype(np.float32)
        mats.append(np.tile(base, (1, 5)))                                   # repeated ground truth: duplicated columns
        mats.append(np.tile(base, (1, 5)).T.copy())
    for c in mats:
        a, b = sp(c)
        ra, rb = lsa_restated(c)
        np.testing.assert_array_equal(a, ra)
        np.testing.assert_array_equal(b, rb)


def test_points_in_boxes_known_answers():
    # unit cube resting on z=0 at the origin, no yaw; strict in x/y, inclusive in z (see oracle header)
    box = torch.tensor([[[0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0]]])
    pts = torch.tensor([[[0.0, 0.0, 0.5], [0.5, 0.0, 0.5], [0.49, -0.49, 1.0], [0.0, 0.0, 1.01], [0.0, 0.0, -0.01]]])
    assert CO.points_in_boxes_all(pts, box)[0, :, 0].tolist() == [1, 0, 1, 0, 0]
    # 2 x 1 box turned by 90 degrees: long side along y
    box = torch.tensor([[[0.0, 0.0, 0.0, 2.0, 1.0, 1.0, float(np.pi / 2)]]])
    pts = torch.tensor([[[0.0, 0.9, 0.5], [0.9, 0.0, 0.5]]])
    assert CO.points_in_boxes_all(pts, box)[0, :, 0].tolist() == [1, 0]
