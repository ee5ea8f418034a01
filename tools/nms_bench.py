#!/usr/bin/env python3
"""Times the device NMS (1024 boxes per scene, 18 classes) next to the numpy oracle: python tools/nms_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nms_oracle as NO  # noqa: E402
from vdetr_amd.nms import batched_nms_3d  # noqa: E402

rng = np.random.default_rng(0)
B, K = 4, 1024
center = rng.uniform([1, 1, 1], [9, 7, 4], (B, K, 1, 3))
half = rng.uniform(0.15, 1.0, (B, K, 1, 3))
sg = np.array([(1, 1, 1), (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, -1, 1)])[None, None]
corners = (center + half * sg).astype(np.float32)
score = rng.random((B, K)).astype(np.float32)
cls = rng.integers(0, 18, (B, K)).astype(np.int32)
c, s, k = (torch.from_numpy(a).cuda() for a in (corners, score, cls))
for _ in range(3):
    keep = batched_nms_3d(c, s, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    keep = batched_nms_3d(c, s, k)
e1.record()
torch.cuda.synchronize()
gpu_us = e0.elapsed_time(e1) / 20 * 1e3
t0 = time.perf_counter()
want = [NO.nms_3d(NO.extents_with_score(corners[b], score[b], cls[b]), 0.25, same_class=True) for b in range(B)]
cpu_ms = (time.perf_counter() - t0) * 1e3
ok = all(set(np.nonzero(keep[b].cpu().numpy())[0]) == set(want[b]) for b in range(B))
print(f"nms3d {B} scenes x {K} boxes: device {gpu_us:.1f} us (sort + kernel), numpy oracle {cpu_ms:.1f} ms, kept {int(keep.sum())}, identical {ok}")
# components
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
print(f"  stable sort alone: {timeit(lambda: torch.sort(s, dim=1, stable=True)):.1f} us")
import ctypes
from vdetr_amd import _lib as L
order = torch.sort(s, dim=1, stable=True)[1].contiguous()
keep8 = torch.empty((B, K), dtype=torch.uint8, device="cuda")
nb = L.lib().vdetr_nms3d_workspace_bytes(B, K)
ws = L.workspace(nb, s.device)
print(f"  kernel alone: {timeit(lambda: L.lib().vdetr_nms3d_f32(L.ptr(c), L.ptr(s), L.ptr(k), None, L.ptr(order), B, K, 0.25, 0, L.ptr(keep8), L.ptr(ws), nb, L.stream_ptr())):.1f} us")
