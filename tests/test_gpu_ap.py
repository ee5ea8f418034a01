"""GPU: parse_predictions on the device (empty-box test, NMS, confidence test) vs the reference's own detection lists
(fixture) and the box-point counts vs the oracle; counts and kept sets are integers: bit-exact."""
import types

import numpy as np
import pytest
import torch

from oracle import ap_oracle as AO
from test_oracle_ap import NUM_SEMCLS, check_lists, config, load

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name", ["default", "any_class", "nms2d", "old_type", "cls_conf", "obj_conf", "angle", "no_nms",
                                  "keep_empty", "strict_points"])
def test_parse_predictions_matches_reference_lists(name):
    from vdetr_amd.ap_calculator import parse_predictions
    x, want = load()
    t = {k: torch.from_numpy(v).to(DEV) for k, v in x.items()}
    cfg = dict(config(name), dataset_config=types.SimpleNamespace(num_semcls=NUM_SEMCLS))
    csa = t["csa"].clone()
    res = parse_predictions(t["corners"], t["sem"], t["obj"], t["ang"], t["points"], cfg, csa)
    check_lists(res, want[name])
    assert torch.equal(csa, t["csa"])  # the reference shifts predicted_boxes_CSA in place and back; here it is left alone


@pytest.mark.parametrize("B,N,K", [(1, 1, 1), (2, 1023, 5), (1, 1025, 300), (4, 40000, 256), (3, 5000, 1024)])
def test_box_point_counts_vs_oracle(B, N, K):
    from vdetr_amd.ap_calculator import box_point_counts
    rng = np.random.default_rng(N + K)
    pts = rng.uniform(0, 4, (B, N, 3)).astype(np.float32)
    boxes = np.concatenate([rng.uniform(0, 4, (B, K, 3)), rng.uniform(0.05, 2.5, (B, K, 3)), rng.uniform(-3.2, 3.2, (B, K, 1))],
                           -1).astype(np.float32)
    boxes[:, ::2, 6] = 0.0                                  # axis-aligned boxes (ScanNet): cos / sin are exact
    pts[:, ::7, 0] = boxes[:, :1, 0] + boxes[:, :1, 3] / 2  # points exactly on a face of box 0: strict inequality
    got = box_point_counts(torch.from_numpy(pts).to(DEV), torch.from_numpy(boxes).to(DEV)).cpu().numpy()
    for b in range(B):
        want = AO.box_point_counts(pts[b], boxes[b])
        aligned = np.arange(K) % 2 == 0
        assert np.array_equal(got[b, aligned], want[aligned])
        # rotated boxes: device cosf / sinf vs numpy may differ in the last bit; a point within that of a face may flip
        assert np.abs(got[b, ~aligned] - want[~aligned]).max(initial=0) <= 1
    assert got.sum() > 0 or N == 1


def test_cpu_tensors_raise():
    from vdetr_amd.ap_calculator import box_point_counts
    with pytest.raises((RuntimeError, ValueError, TypeError)):
        box_point_counts(torch.zeros(1, 4, 3), torch.zeros(1, 2, 7))


def test_ap_calculator_matches_reference_metrics():
    """step_meter over 3 batches -> compute_metrics / metrics_to_str vs the reference's own APCalculator (fixture)."""
    import os
    from vdetr_amd.ap_calculator import APCalculator, get_ap_config_dict
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ap_calculator.npz"))
    class2type = {int(i): str(n) for i, n in zip(z["class_ids"], z["class_names"])}
    cfg = types.SimpleNamespace(num_semcls=NUM_SEMCLS)
    calc = APCalculator(dataset_config=cfg, ap_iou_thresh=[0.25, 0.5], class2type_map=class2type, exact_eval=True,
                        ap_config_dict=get_ap_config_dict(dataset_config=cfg, remove_empty_box=True))
    for bi in range(int(z["nbatch"])):
        out = {k.split(":")[2]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"b{bi}:out:")}
        tgt = {k.split(":")[2]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"b{bi}:tgt:")}
        calc.step_meter({"outputs": out}, tgt)
    ret = calc.compute_metrics()
    for thr in (0.25, 0.5):
        assert list(ret[thr].keys()) == [str(k) for k in z[f"t{thr}:keys"]]
        got = np.array([float(v) for v in ret[thr].values()])
        assert np.allclose(got, z[f"t{thr}:values"], rtol=1e-7, atol=1e-12), thr
    assert calc.metrics_to_str(ret) == str(z["text"])
    assert str(calc) == str(z["text"])
    # the list interface gives the same numbers
    from vdetr_amd.ap_calculator import parse_predictions
    calc2 = APCalculator(dataset_config=cfg, ap_iou_thresh=[0.25, 0.5], class2type_map=class2type, exact_eval=True,
                         ap_config_dict=calc.ap_config_dict)
    for bi in range(int(z["nbatch"])):
        out = {k.split(":")[2]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"b{bi}:out:")}
        tgt = {k.split(":")[2]: z[k] for k in z.files if k.startswith(f"b{bi}:tgt:")}
        csa = torch.cat((out["center_unnormalized"], out["size_unnormalized"], out["angle_continuous"].unsqueeze(-1)), -1)
        preds = parse_predictions(out["box_corners"], out["sem_cls_prob"], out["objectness_prob"], out["angle_prob"],
                                  torch.from_numpy(tgt["point_clouds"]).to(DEV), calc.ap_config_dict, csa)
        calc2.accumulate(preds, calc2.make_gt_list(tgt["gt_box_corners"], tgt["gt_box_sem_cls_label"], tgt["gt_box_present"]))
    assert calc2.metrics_to_dict(calc2.compute_metrics()) == calc.metrics_to_dict(ret)
