#!/usr/bin/env python3
"""torch.profiler view of one eager train step (which ATen ops / call sites launch the small kernels)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda")
model = bench.build_model("c2", dev)
inputs = bench.make_inputs("c2", dev, 0)
tr = bench.Trainer(model, inputs, 1, use_graph=False, overlap=False, fps_prefetch=False)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=50, max_src_column_width=90))
