#!/usr/bin/env python3
"""The cut C5 model of tests/test_gpu_model.py's full-config case (2 scenes, 3 RPE layers, rotated boxes) on the device under the
table gradient's switches: side stream / in line, box kernel (device decides) / general kernel only, dynamic distribution on / off,
weight gradients parked / in line.  Prints, per cpb MLP parameter group of every layer, the relative difference to the first run."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as TM  # noqa: E402
from vdetr_amd import attention as A  # noqa: E402
from vdetr_amd import runtime  # noqa: E402

bs, nl = int(os.environ.get("DIAG_BS", "2")), int(os.environ.get("DIAG_NL", "4"))
model = TM._make_model(nq=1024, npre=4096, nl=nl, angle_type="object_coords").train()
TM._zero_dropout(model)
inp_cpu = TM._inputs(20000, 3, "cpu", bs)
gpu = copy.deepcopy(model).to("cuda")


def run(async_mode, bwd_kernel, dynamic, park):
    A.set_async_table_grad(async_mode)
    A.BWD_KERNEL = bwd_kernel
    A.DYNAMIC_BWD = dynamic
    gpu.zero_grad(set_to_none=True)
    inp = {k: ([t.detach().to("cuda").requires_grad_(t.requires_grad) for t in v] if isinstance(v, list) else v.to("cuda"))
           for k, v in inp_cpu.items()}
    runtime.defer_weight_grads(park)
    try:
        out = gpu(inp)
        TM._loss(out).backward()
        if park:
            runtime.flush_weight_grads()
    finally:
        runtime.defer_weight_grads(False)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in gpu.named_parameters() if p.grad is not None}


ref = None
for mode in [("0", 1, True, False), ("0", 0, True, False), ("0", 0, False, False), ("1", 0, True, False), ("1", 0, True, True), ("1", 1, True, True), ("0", 1, True, False)]:
    g = run(*mode)
    if ref is None:
        ref = g
        print("reference: async, bwd_kernel, dynamic, park =", mode)
        continue
    worst = {}
    for n, t in g.items():
        key = ".".join(n.split(".")[:3]) + (".cpb" if "cpb_mlps" in n else ".other")
        r = ref[n]
        rel = float((t - r).norm() / r.norm().clamp_min(1e-30))
        worst[key] = max(worst.get(key, 0.0), rel)
    print(mode, " ".join(f"{k}={v:.1e}" for k, v in sorted(worst.items()) if "cpb" in k or v > 1e-2))
