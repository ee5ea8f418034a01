"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's greedy 3-D NMS
(utils/nms.py:78-118 nms_3d_faster, :121-162 nms_3d_faster_samecls) and of the way parse_predictions feeds it
(utils/ap_calculator.py:165-220: min / max extents of the 8 corners in a float64 array).

PINNED: tests/golden/nms3d.npz holds inputs and picks produced by the reference's own utils/nms.py imported in the build
container (oracle/make_golden.py); tests/test_oracle_nms.py checks this file against them.
"""
import numpy as np


def nms_3d(boxes, overlap_threshold, same_class=False, old_type=False, stable=False):
    """boxes [K, 7 or 8] = (x1,y1,z1,x2,y2,z2,score[,cls]).  Returns the picked indices, best score first.  ``stable``: visit
    exactly equal scores in the order of a stable arg-sort (the reference's default arg-sort leaves that order to numpy's
    sort implementation; the device path defines it)."""
    boxes = np.asarray(boxes)
    lo, hi, score = boxes[:, 0:3], boxes[:, 3:6], boxes[:, 6]
    vol = (hi[:, 0] - lo[:, 0]) * (hi[:, 1] - lo[:, 1]) * (hi[:, 2] - lo[:, 2])
    remaining = np.argsort(score, kind="stable" if stable else None)   # ascending; the best candidate is taken from the end (nms.py:89,133)
    picked = []
    while remaining.size:
        i = remaining[-1]
        picked.append(int(i))
        rest = remaining[:-1]
        ext = np.maximum(0, np.minimum(hi[i], hi[rest]) - np.maximum(lo[i], lo[rest]))
        inter = ext[:, 0] * ext[:, 1] * ext[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            o = inter / vol[rest] if old_type else inter / (vol[i] + vol[rest] - inter)
        if same_class:
            o = o * (boxes[i, 7] == boxes[rest, 7])
        remaining = rest[~(o > overlap_threshold)]
    return picked


def nms_2d(boxes, overlap_threshold, old_type=False, stable=False):
    """utils/nms.py:43-76 nms_2d_faster: rows (x1, y1, x2, y2, score); the 3-D loop with the third extent fixed to [0, 1]
    (areas and intersections times exactly 1.0)."""
    boxes = np.asarray(boxes, dtype=np.float64)
    K = boxes.shape[0]
    rows = np.concatenate([boxes[:, 0:2], np.zeros((K, 1)), boxes[:, 2:4], np.ones((K, 1)), boxes[:, 4:5]], axis=1)
    return nms_3d(rows, overlap_threshold, old_type=old_type, stable=stable)


def extents_with_score(corners, score, cls=None):
    """ap_calculator.py:168-214: float64 rows (min xyz, max xyz, score[, cls]) of one scene's corners [K,8,3]."""
    cols = [corners.min(1), corners.max(1), score[:, None]] + ([cls[:, None]] if cls is not None else [])
    return np.concatenate([np.asarray(c, dtype=np.float64) for c in cols], axis=1)
