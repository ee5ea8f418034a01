#!/usr/bin/env python3
"""The cross-attention backward's four matrix products at C2 size: the library (as the step calls them) next to the fused
pair-list GEMM kernel used as a plain dense GEMM (identity row map).   python tools/skinny_gemm_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def timeit(fn, reps=20):
    ts = []
    for i in range(reps + 5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if i >= 5:
            ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    from vdetr_amd import sparse_ops as S
    try:
        from vdetr_amd.runtime import enable_gemm_tuning
        enable_gemm_tuning(0)
    except Exception:
        pass
    dev = torch.device("cuda")
    HQ, NK, D = 4096, 4096, 64
    dO = torch.randn(1, HQ, D, device=dev)
    V = torch.randn(1, NK, D, device=dev)
    P = torch.randn(1, HQ, NK, device=dev)
    print("library dP = dO V^T        [4096x64]x[64x4096]  ", round(timeit(lambda: torch.bmm(dO, V.transpose(1, 2))), 1), "us")
    print("library dV = P^T dO        [4096x4096]x[4096x64]", round(timeit(lambda: torch.bmm(P.transpose(1, 2), dO)), 1), "us")
    print("library dQ = dS K          [4096x4096]x[4096x64]", round(timeit(lambda: torch.bmm(P, V)), 1), "us")

    class Plan:
        pass
    plan = Plan()
    plan.P = HQ
    tiles = [(0, s, 128) for s in range(0, HQ, 128)]
    plan.tiles = torch.tensor(tiles, dtype=torch.int32, device=dev)
    plan.ntiles = len(tiles)
    ident = torch.arange(HQ, dtype=torch.int32, device=dev)
    # dP[hq][k] = sum_d dO[hq][d] V[k][d]: transposed mode, W = V as [1, cin = NK, cout = D]
    w = V.view(1, NK, D)
    y = S.pairs_gemm(dO[0], ident, w, plan, True)
    ref = torch.bmm(dO, V.transpose(1, 2))[0]
    print("pair-list kernel as dense dP: max err", float((y - ref).abs().max()), " ", round(timeit(lambda: S.pairs_gemm(dO[0], ident, w, plan, True)), 1), "us")
    # dQ[hq][d] = sum_k dS[hq][k] K[k][d]: plain mode, W = K as [1, cin = NK, cout = D]
    y2 = S.pairs_gemm(P[0], ident, w, plan, False)
    ref2 = torch.bmm(P, V)[0]
    print("pair-list kernel as dense dQ: rel err", float((y2 - ref2).abs().max() / ref2.abs().max()), " ", round(timeit(lambda: S.pairs_gemm(P[0], ident, w, plan, False)), 1), "us")


if __name__ == "__main__":
    main()
