#!/usr/bin/env python3
"""cProfile of the HOST side of one steady-state step with the sparse backbone (static geometry: no loader thread):
which Python functions the ~1,200 eager launches spend their enqueue time in.   python tools/backbone_host_profile.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    os.environ["VDETR_BENCH_GEOMETRY"] = "static"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    try:
        from vdetr_amd.runtime import enable_gemm_tuning
        enable_gemm_tuning(0)
    except Exception:
        pass
    tr = bench.BackboneTrainer("c2", dev)
    tr.capture()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
