#!/usr/bin/env python3
"""Times the detection AP (device box matching) on a validation-sized synthetic set next to the numpy/qhull oracle on a
sample of it: python tools/eval_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import eval_oracle as EO  # noqa: E402
from test_gpu_eval import random_set  # noqa: E402
from vdetr_amd import eval_det as ED  # noqa: E402

NIMG, NCLS, NGT, NPRED = 312, 18, 30, 4608        # ScanNet val: 312 scenes, 256 kept boxes x 18 class scores each
pred_all, gt_all = random_set(0, NIMG, NCLS, NGT, NPRED)
npred = sum(len(v) for v in pred_all.values())
ngt = sum(len(v) for v in gt_all.values())
ED.eval_det_multiprocessing(dict(list(pred_all.items())[:2]), gt_all, 0.25)  # warm-up (library load)
torch.cuda.synchronize()
t0 = time.perf_counter()
rec, prec, ap = ED.eval_det_multiprocessing(pred_all, gt_all, 0.25)
t_all = time.perf_counter() - t0
# the device part alone
flat = [(i, c, b, s) for i, d in pred_all.items() for c, b, s in d]
gflat = sorted([(i, c, b) for i, d in gt_all.items() for c, b in d], key=lambda r: r[0])
args = (np.stack([f[2] for f in flat]), np.array([f[0] for f in flat]), np.array([f[1] for f in flat]), np.array([f[3] for f in flat]),
        np.stack([g[2] for g in gflat]), np.array([g[0] for g in gflat]), np.array([g[1] for g in gflat]), NIMG, 0.25)
ED.match_detections(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
ED.match_detections(*args)
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
sample = dict(list(pred_all.items())[:3])
t0 = time.perf_counter()
with np.errstate(all="ignore"):
    EO.eval_det(sample, {k: gt_all[k] for k in sample}, 0.25)
t_cpu = (time.perf_counter() - t0) * NIMG / 3
print(f"eval_det {NIMG} scenes, {npred} detections, {ngt} ground-truth boxes, {NCLS} classes: whole call {t_all:.2f} s "
      f"(upload + kernel + ranking + download {t_dev * 1e3:.0f} ms; the rest is flattening the Python dictionaries), "
      f"numpy/qhull oracle ~{t_cpu:.0f} s (extrapolated from 3 scenes, one core), mAP {np.mean([float(v) for v in ap.values()]):.4f}")
