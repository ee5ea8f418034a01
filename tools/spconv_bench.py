#!/usr/bin/env python3
"""Per-layer timing of the fused pair-list kernels on the maps of a synthetic 40k-point room scan, next to a plain library
GEMM of the same number of rows:  python tools/spconv_bench.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timeit(fn, reps=10):
    ts = []
    for i in range(reps + 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.mean(ts))


def main():
    dev = torch.device("cuda")
    from vdetr_amd import minkowski as ME
    from vdetr_amd import sparse_ops as S
    cloud = bench.make_room_cloud(40000, 0, dev)
    coords, feats = ME.batch_sparse_collate([(cloud / 0.01, cloud)])
    cm = ME.CoordinateManager(dev)
    cm.insert_points(coords)
    keys = {1: cm.keys[1]}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = cm.strided(keys[ts // 2], ts // 2, ts)
    for ts, c in ((4, 64), (8, 128), (16, 256), (32, 512)):
        nbr, inv, plan = cm.kernel_map(keys[ts], keys[ts], ts, ts, 3, False)
        n = keys[ts].shape[0]
        x = torch.randn((n, c), device=dev)
        w = torch.randn((27, c, c), device=dev) / 40
        dy = torch.randn((n, c), device=dev)
        flops = 2.0 * plan.P * c * c
        t_f = timeit(lambda: S.pairs_gemm(x, plan.pin, w, plan, False))
        t_d = timeit(lambda: S.pairs_gemm(dy, plan.pout, w, plan, True))
        t_w = timeit(lambda: S.pairs_wgrad(x, dy, plan, c, c))
        y = S.pairs_gemm(x, plan.pin, w, plan, False)
        t_g = timeit(lambda: S.gather_sum(y, plan.slot, flat=True))
        a = torch.randn((plan.P, c), device=dev)
        t_lib = timeit(lambda: a @ w[0])
        print(json.dumps({"stride": ts, "channels": c, "sites": n, "pairs": plan.P, "GFLOP": flops / 1e9,
                          "fwd_us": t_f * 1e6, "fwd_TF": flops / t_f / 1e12, "dgrad_us": t_d * 1e6, "dgrad_TF": flops / t_d / 1e12,
                          "wgrad_us": t_w * 1e6, "wgrad_TF": flops / t_w / 1e12, "gather_sum_us": t_g * 1e6,
                          "library_gemm_same_rows_us": t_lib * 1e6, "library_TF": flops / t_lib / 1e12}))


if __name__ == "__main__":
    main()
