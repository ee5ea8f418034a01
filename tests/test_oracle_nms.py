"""CPU: the NMS oracle against the picks of the reference's own utils/nms.py (fixture from oracle/make_golden.py)."""
import os

import numpy as np

from oracle import nms_oracle as NO

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "nms3d.npz")


def cases():
    z = np.load(GOLDEN)
    for ci in range(int(z["ncases"])):
        yield ci, {k.split(":", 1)[1]: z[k] for k in z.files if k.startswith(f"c{ci}:")}


def test_oracle_matches_reference_picks():
    for ci, c in cases():
        rows = NO.extents_with_score(c["corners"], c["score"], c["cls"])
        assert NO.nms_3d(rows, 0.25, same_class=True) == c["pick_samecls"].tolist(), ci
        assert NO.nms_3d(rows[:, :7], 0.25) == c["pick_any"].tolist(), ci
        assert NO.nms_3d(rows, 0.5, same_class=True, old_type=True) == c["pick_samecls_old"].tolist(), ci
        assert 0 < len(c["pick_samecls"]) <= len(c["score"])
        co = c["corners"]
        rows2d = np.stack([co[:, :, 0].min(1), co[:, :, 2].min(1), co[:, :, 0].max(1), co[:, :, 2].max(1), c["score"]], 1)
        assert NO.nms_2d(rows2d, 0.25) == c["pick_2d"].tolist(), ci
