#!/usr/bin/env python3
"""Times parse_predictions on the device (8 scenes x 256 boxes x 40,000 points, 18 classes) next to the numpy oracle:
python tools/parse_bench.py"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ap_oracle as AO  # noqa: E402
from oracle.make_golden import ap_inputs  # noqa: E402
from vdetr_amd.ap_calculator import box_point_counts, get_ap_config_dict, parse_predictions, prediction_masks  # noqa: E402

B, K, N, C = 8, 256, 40000, 18
x = ap_inputs(seed=3, B=B, K=K, N=N, C=C)
t = {k: torch.from_numpy(v).cuda() for k, v in x.items()}
cfg = get_ap_config_dict(dataset_config=types.SimpleNamespace(num_semcls=C))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


args = (t["corners"], t["sem"], t["obj"], t["ang"], t["points"], cfg, t["csa"])
us_count = timeit(lambda: box_point_counts(t["points"], t["csa"]))
us_mask = timeit(lambda: prediction_masks(*args))
t0 = time.perf_counter()
res = parse_predictions(*args)
torch.cuda.synchronize()
ms_full = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
want = AO.parse_predictions(x["corners"], x["sem"], x["obj"], x["ang"], x["points"], cfg, x["csa"], C)
ms_cpu = (time.perf_counter() - t0) * 1e3
same = [len(r) for r in res] == [len(r) for r in want] and all(
    a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2] for ra, rb in zip(res, want) for a, b in zip(ra, rb))
tests = B * K * N
print(f"parse_predictions {B} scenes x {K} boxes x {N} points: box_point_count {us_count:.1f} us "
      f"({tests / us_count / 1e3:.1f} G point-box tests/s, {B * N * 12 / us_count / 1e3:.2f} GB/s of point reads), "
      f"masks on the device {us_mask:.1f} us, with the host lists {ms_full:.1f} ms; numpy oracle {ms_cpu:.0f} ms; "
      f"detections {sum(len(r) for r in res)}, identical {same}")
