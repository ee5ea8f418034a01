"""CPU: the parse_predictions oracle against the detection lists of the reference's own utils/ap_calculator.py (fixture
from oracle/make_golden.py:ap_cases, every NMS / scoring variant)."""
import os

import numpy as np
import pytest

from oracle import ap_oracle as AO

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "parse_predictions.npz")
NUM_SEMCLS = 18  # ScannetDatasetConfig.num_semcls (datasets/scannet.py:40)


def load():
    z = np.load(GOLDEN)
    x = {k[3:]: z[k] for k in z.files if k.startswith("in:")}
    names = sorted({k.split(":")[0] for k in z.files if not k.startswith("in:")})
    return x, {n: {f: z[f"{n}:{f}"] for f in ("count", "cls", "corners", "score")} for n in names}


def config(name):
    from oracle.make_golden import AP_VARIANTS
    cfg = dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25, use_old_type_nms=False, cls_nms=True, per_class_proposal=True,
               use_cls_confidence_only=False, conf_thresh=0.0, no_nms=False, empty_pt_thre=5, rotated_nms=False, angle_nms=False,
               angle_conf=False)
    cfg.update(AP_VARIANTS[name])
    return cfg


def check_lists(result, want, exact_score=True):
    assert [len(r) for r in result] == want["count"].tolist()
    flat = [d for r in result for d in r]
    assert [int(d[0]) for d in flat] == want["cls"].tolist()
    if flat:
        assert np.array_equal(np.stack([np.asarray(d[1]) for d in flat]), want["corners"])
        got = np.array([d[2] for d in flat], np.float32)
        assert np.array_equal(got, want["score"]) if exact_score else np.allclose(got, want["score"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", ["default", "any_class", "nms2d", "old_type", "cls_conf", "obj_conf", "angle", "no_nms",
                                  "keep_empty", "strict_points"])
def test_oracle_matches_reference_lists(name):
    x, want = load()
    res = AO.parse_predictions(x["corners"], x["sem"], x["obj"], x["ang"], x["points"], config(name), x["csa"], NUM_SEMCLS)
    check_lists(res, want[name])
