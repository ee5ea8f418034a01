#!/usr/bin/env python3
"""column sums of the shapes the step's bias gradients have: helpers.colsum_batched against torch's sum (HIP events)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vdetr_amd.helpers import colsum_batched

from vdetr_amd import attention as A


def t(fn, n=20):
    """device time per call: n calls captured in one graph, replayed"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        A.begin_step("cuda")
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            A.begin_step("cuda")
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1000

for shape in [(64, 1024, 256), (2, 4096, 256), (1, 4096, 1024), (1, 4096, 256), (1, 1024, 256), (40, 1024, 256), (8, 1024, 1280), (1, 4096, 1280)]:
    x = torch.randn(shape, device="cuda")
    print(shape, f"colsum_batched {t(lambda: colsum_batched(x)):7.1f} us   torch.sum {t(lambda: x.sum(1)):7.1f} us")
