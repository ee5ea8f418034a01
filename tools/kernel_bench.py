#!/usr/bin/env python3
"""Times the hand-written kernels one by one with HIP events (same code as bench.py's roofline leg) + FPS:
   python tools/kernel_bench.py [c2]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def time_fps(npts, m, reps=5):
    from vdetr_amd import pointnet2_utils as PU
    xyz, _ = bench.make_scene(npts, 0, "cuda")
    x = xyz[None].contiguous()
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PU.furthest_point_sample(x, m)
        e1.record()
        e1.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    return {"kernel": "fps", "n": int(x.shape[1]), "m": m, "ms": sum(ts) / len(ts), "us_per_round": sum(ts) / len(ts) * 1e3 / (m - 1)}


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    f, b = bench.kernel_rooflines(cfg, torch.device("cuda"))
    print(json.dumps({"variant": os.environ.get("VDETR_BWD_VARIANT", "default"), "fwd_us": f["launch_us"], "bwd_us": b["launch_us"]}))
    print(json.dumps(time_fps(bench.CONFIGS[cfg][0], 4096)))
