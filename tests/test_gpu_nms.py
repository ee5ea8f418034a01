"""GPU: device NMS through the C-ABI vs the reference's picks (fixture) and the oracle; kept sets must be identical."""
import numpy as np
import pytest
import torch

from oracle import nms_oracle as NO
from test_oracle_nms import cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_batched_nms_matches_reference_picks():
    from vdetr_amd.nms import batched_nms_3d, nms_3d_faster, nms_3d_faster_samecls
    for ci, c in cases():
        corners, score, cls = (torch.from_numpy(c[k]).to(DEV) for k in ("corners", "score", "cls"))
        K = score.shape[0]
        for key, kw in (("pick_samecls", dict(classes=cls[None])), ("pick_any", dict()),
                        ("pick_samecls_old", dict(classes=cls[None], iou_threshold=0.5, old_type=True))):
            keep = batched_nms_3d(corners[None], score[None], **kw)[0].cpu().numpy()
            want = np.zeros(K, bool)
            want[c[key]] = True
            assert np.array_equal(keep, want), (ci, key)
        rows = torch.from_numpy(NO.extents_with_score(c["corners"], c["score"], c["cls"])).to(DEV)
        assert nms_3d_faster_samecls(rows, 0.25).cpu().tolist() == c["pick_samecls"].tolist()     # same order too
        assert nms_3d_faster(rows[:, :7], 0.25).cpu().tolist() == c["pick_any"].tolist()


def test_batch_and_valid_mask_against_oracle():
    from vdetr_amd.nms import batched_nms_3d
    rng = np.random.default_rng(3)
    B, K = 3, 257
    center = rng.uniform(1, 4, (B, K, 1, 3))
    half = rng.uniform(0.2, 1.0, (B, K, 1, 3))
    sg = np.array([(1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1)])[None, None]
    corners = (center + half * sg).astype(np.float32)
    score = rng.random((B, K)).astype(np.float32)
    cls = rng.integers(0, 4, (B, K)).astype(np.int32)
    valid = rng.random((B, K)) > 0.3
    keep = batched_nms_3d(torch.from_numpy(corners).to(DEV), torch.from_numpy(score).to(DEV), torch.from_numpy(cls).to(DEV),
                          torch.from_numpy(valid).to(DEV), 0.25).cpu().numpy()
    for b in range(B):
        idx = np.nonzero(valid[b])[0]                                   # ap_calculator.py:209-219: NMS on the non-empty boxes
        rows = NO.extents_with_score(corners[b, idx], score[b, idx], cls[b, idx])
        want = np.zeros(K, bool)
        want[idx[NO.nms_3d(rows, 0.25, same_class=True)]] = True
        assert np.array_equal(keep[b], want), b


@pytest.mark.parametrize("K,classes,room", [(1, 1, 1.0), (63, 2, 1.5), (64, 1, 1.0), (65, 3, 2.0), (300, 2, 2.0), (1000, 1, 3.0),
                                            (1024, 18, 4.0), (1500, 3, 3.0), (4096, 6, 5.0)])
def test_sizes_dense_overlap(K, classes, room):
    """Word counts that are not powers of two, the last partial word, the relation kept in LDS (K <= 1024) and in the
    workspace (K > 1024), and rooms small enough that most boxes are suppressed (long suppression chains)."""
    from vdetr_amd.nms import batched_nms_3d
    rng = np.random.default_rng(K)
    center = rng.uniform(0, room, (K, 1, 3))
    half = rng.uniform(0.2, 0.6, (K, 1, 3))
    sg = np.array([(1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1)])[None]
    corners = (center + half * sg).astype(np.float32)
    score = rng.integers(0, 50, K).astype(np.float32) / 50          # repeated scores: the stable order decides
    cls = rng.integers(0, classes, K).astype(np.int32)
    keep = batched_nms_3d(torch.from_numpy(corners).to(DEV)[None], torch.from_numpy(score).to(DEV)[None],
                          torch.from_numpy(cls).to(DEV)[None], None, 0.25)[0].cpu().numpy()
    want = np.zeros(K, bool)
    want[NO.nms_3d(NO.extents_with_score(corners, score, cls), 0.25, same_class=True, stable=True)] = True
    assert 0 < want.sum() < max(K, 2)
    assert np.array_equal(keep, want)


def test_limits_raise():
    from vdetr_amd.nms import batched_nms_3d
    with pytest.raises((RuntimeError, ValueError)):
        batched_nms_3d(torch.zeros(1, 4097, 8, 3, device=DEV), torch.zeros(1, 4097, device=DEV))
    assert batched_nms_3d(torch.zeros(0, 5, 8, 3, device=DEV), torch.zeros(0, 5, device=DEV)).shape == (0, 5)
