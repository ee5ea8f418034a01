"""CPU: the AP oracle against the reference's own eval_det_multiprocessing (fixture from oracle/make_golden.py:eval_cases)."""
import os

import numpy as np

from oracle import eval_oracle as EO

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "eval_det.npz")


def load():
    z = np.load(GOLDEN)
    pred_all = {f"scene{i}": [] for i in z["pred_imgs"]}
    gt_all = {f"scene{i}": [] for i in z["gt_imgs"]}
    for i, c, b, s in zip(z["pred_img"], z["pred_cls"], z["pred_box"], z["pred_score"]):
        pred_all[f"scene{i}"].append((int(c), b, s))
    for i, c, b in zip(z["gt_img"], z["gt_cls"], z["gt_box"]):
        gt_all[f"scene{i}"].append((int(c), b))
    want = {}
    for k in z.files:
        if k.startswith("t"):
            thr, c, what = k.split(":")
            want.setdefault(float(thr[1:]), {}).setdefault(int(c[1:]), {})[what] = z[k]
    return pred_all, gt_all, want


def check(result, want, tol=0.0):
    rec, prec, ap = result
    assert set(ap) == set(want)
    for c, w in want.items():
        assert np.allclose(ap[c], w["ap"], rtol=tol, atol=tol), (c, ap[c], w["ap"])
        assert np.allclose(np.asarray(rec[c], np.float64), w["rec"], rtol=tol, atol=tol), c
        assert np.allclose(np.asarray(prec[c], np.float64), w["prec"], rtol=tol, atol=tol), c


def test_oracle_matches_reference_ap():
    pred_all, gt_all, want = load()
    for thr, w in want.items():
        check(EO.eval_det(pred_all, gt_all, thr), w)
    assert any(w["ap"] > 0 for w in want[0.25].values())
