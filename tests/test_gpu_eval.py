"""GPU: detection AP with the box matching on the device vs the reference's own eval_det_multiprocessing (fixture) and vs
the oracle on a larger random set; true-positive flags are decisions: identical."""
import numpy as np
import pytest
import torch

from oracle import eval_oracle as EO
from test_oracle_eval import check, load

pytestmark = pytest.mark.gpu


def box(size, yaw, center):
    """8 corners in the reference's order (upright camera frame: corner 0-3 on top, footprint 3,2,1,0 counter-clockwise)"""
    l, w, h = size
    x = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    y = np.array([h, h, h, h, -h, -h, -h, -h]) / 2
    z = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    c, s = np.cos(yaw), np.sin(yaw)
    return (np.stack([c * x + s * z, y, -s * x + c * z], 1) + center).astype(np.float32)


def test_ap_matches_reference_fixture():
    from vdetr_amd.eval_det import eval_det, eval_det_multiprocessing
    pred_all, gt_all, want = load()
    for thr, w in want.items():
        check(eval_det_multiprocessing(pred_all, gt_all, ovthresh=thr), w, tol=1e-12)
        check(eval_det(pred_all, gt_all, ovthresh=thr), w, tol=1e-12)


def random_set(seed, nimg, ncls, ngt, npred):
    rng = np.random.default_rng(seed)
    pred_all, gt_all = {}, {}
    for img in range(nimg):
        gts = [(int(rng.integers(0, ncls)), rng.uniform(0.3, 2.0, 3), rng.uniform(-3, 3) * (img % 3 != 0), rng.uniform(0, 5, 3))
               for _ in range(rng.integers(0, ngt + 1))]
        gt_all[img] = [(c, box(s, a, p)) for c, s, a, p in gts]
        dets = []
        for _ in range(rng.integers(1, npred + 1)):
            if gts and rng.random() < 0.7:
                c, s, a, p = gts[rng.integers(0, len(gts))]
                j = rng.choice([0.0, 0.05, 0.2])
                dets.append((c, box(s * (1 + rng.normal(0, j, 3)).clip(0.5, 1.5), a + rng.normal(0, j), p + rng.normal(0, j, 3)),
                             np.float32(rng.integers(0, 20) / 20)))          # repeated scores: the stable rank decides
            else:
                dets.append((int(rng.integers(0, ncls)), box(rng.uniform(0.3, 2, 3), rng.uniform(-3, 3), rng.uniform(0, 5, 3)),
                             np.float32(rng.random())))
        pred_all[img] = dets
    return pred_all, gt_all


@pytest.mark.parametrize("seed,nimg,ncls,ngt,npred", [(0, 1, 1, 1, 1), (1, 12, 4, 8, 40), (2, 30, 18, 12, 60)])
def test_ap_matches_oracle_on_random_sets(seed, nimg, ncls, ngt, npred):
    from vdetr_amd.eval_det import eval_det_multiprocessing
    pred_all, gt_all = random_set(seed, nimg, ncls, ngt, npred)
    for thr in (0.25, 0.5):
        rec, prec, ap = eval_det_multiprocessing(pred_all, gt_all, ovthresh=thr)
        wrec, wprec, wap = EO.eval_det(pred_all, gt_all, thr)
        assert set(ap) == set(wap)
        for c in wap:
            assert np.array_equal(np.asarray(rec[c]), np.asarray(wrec[c])), (thr, c)   # same true-positive flags
            assert np.allclose(np.asarray(prec[c]), np.asarray(wprec[c]), rtol=1e-14, atol=0)
            assert np.isclose(ap[c], wap[c], rtol=1e-12, atol=1e-15)


def test_iou_values_vs_oracle():
    from vdetr_amd import _lib as L
    rng = np.random.default_rng(9)
    P, G = 200, 7
    gt = np.stack([box(rng.uniform(0.5, 2, 3), rng.uniform(-3, 3), rng.uniform(0, 2, 3)) for _ in range(G)])
    gt[1] = gt[0]                                                           # duplicate box: the first one wins the tie
    pred = np.stack([box(rng.uniform(0.5, 2, 3), rng.uniform(-3, 3) * (i % 2), rng.uniform(0, 2, 3)) for i in range(P)])
    pred[:G] = gt                                                           # exact copies (rotated ones are ill-conditioned)
    dev = "cuda"
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
    ov, jm = torch.empty(P, dtype=torch.float64, device=dev), torch.empty(P, dtype=torch.int32, device=dev)
    args = [t(pred, np.float32), t(np.zeros(P), np.int32), t(np.zeros(P), np.int32), t(gt, np.float32), t(np.zeros(G), np.int32),
            t(np.array([0, G]), np.int32)]
    L.check(L.lib().vdetr_box3d_iou_max_f64(L.ptr(args[0]), L.ptr(args[1]), L.ptr(args[2]), P, L.ptr(args[3]), L.ptr(args[4]),
                                            L.ptr(args[5]), L.ptr(ov), L.ptr(jm), L.stream_ptr()), "iou")
    ov, jm = ov.cpu().numpy(), jm.cpu().numpy()
    for d in range(P):
        with np.errstate(all="ignore"):
            ious = np.array([EO.box3d_iou(pred[d].astype(np.float64), gt[j].astype(np.float64)) for j in range(G)])
        assert np.isclose(ov[d], ious.max(), rtol=1e-9, atol=1e-12), d
        assert np.isclose(ious[jm[d]], ious.max(), rtol=1e-9, atol=1e-12), d
    assert jm[1] == 0
