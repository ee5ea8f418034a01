#!/usr/bin/env python3
"""Exercises the N > 1 code path of bench.py on ONE GPU with a single-rank RCCL communicator: process-group init, the
watchdog thread next to hipGraph capture, the split graphs (fwd+bwd | all-reduce | update) and the eager RCCL launches
between two graph replays.  python tools/rccl_path_check.py [c1|c2]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
from vdetr_amd.dist import broadcast_parameters  # noqa: E402
from vdetr_amd.runtime import enable_gemm_tuning  # noqa: E402
enable_gemm_tuning(0)
model = bench.build_model(cfg, device)
broadcast_parameters(model)
inputs = bench.make_inputs(cfg, device, 0)
for use_graph in (True, False):
    tr = bench.Trainer(model, inputs, 2, use_graph, overlap=True)   # world = 2: the multi-GPU control flow
    if use_graph:
        tr.capture()
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    loss = float(tr.loss.item())
    assert loss == loss, "non-finite loss"
    print(f"{'graph' if use_graph else 'eager (hooked buckets)'} path with RCCL all-reduce (1 rank): {dt:.2f} ms/step, loss {loss:.4f}")
dist.destroy_process_group()
