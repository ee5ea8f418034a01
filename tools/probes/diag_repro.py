#!/usr/bin/env python3
"""Is the full-config training step's feature gradient the same from run to run (same process, same switches)?"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as TM  # noqa: E402
from vdetr_amd import attention as A  # noqa: E402
from vdetr_amd import rowblock as RB  # noqa: E402
from vdetr_amd import runtime  # noqa: E402
from vdetr_amd import vdetr_transformer as T  # noqa: E402

model = TM._make_model(nq=1024, npre=4096, nl=9).train()
TM._zero_dropout(model)
inp_cpu = TM._inputs(40000, 3, "cpu", 1)
gpu = copy.deepcopy(model).to("cuda")
inp = {k: ([t.detach().to("cuda").requires_grad_(t.requires_grad) for t in v] if isinstance(v, list) else v.to("cuda")) for k, v in inp_cpu.items()}
keys = ("sem_cls_logits", "center_unnormalized", "size_unnormalized")


def run(rowblock, fwd_kernel):
    T._ROWBLOCK, A.FWD_KERNEL = rowblock, fwd_kernel
    A.set_async_table_grad("0")
    gpu.zero_grad(set_to_none=True)
    for f in inp["backbone_features"]:
        f.grad = None
    out = gpu(inp)
    TM._loss(out).backward()
    torch.cuda.synchronize()
    st = [torch.cat([s[k].detach().flatten() for k in keys]).cpu() for s in out["aux_outputs"] + [out["outputs"]]]
    pg = {n: p.grad.detach().cpu().clone() for n, p in gpu.named_parameters() if p.grad is not None}
    return inp["backbone_features"][0].grad.detach().cpu().double().numpy(), st, pg


runs = [run(False, 1) for _ in range(4)]
for i in range(1, 4):
    g0, g1 = runs[0][0], runs[i][0]
    fwd = [float((a - b).abs().max()) for a, b in zip(runs[0][1], runs[i][1])]
    allrel = sorted(((float((runs[0][2][n] - runs[i][2][n]).abs().max() / (runs[0][2][n].abs().max() + 1e-30)), n) for n in runs[0][2]
                     if not n.endswith(".k.bias")), reverse=True)
    worst = allrel[:6] + [allrel[len(allrel) // 2]]
    print(f"run 0 vs run {i}: forward stages max |diff| {['%.1e' % x for x in fwd]}; feature grad max |diff| {np.abs(g1 - g0).max():.3e}; "
          f"param grads worst rel {[(n, '%.1e' % e) for e, n in worst]}", flush=True)
g1, g2 = runs[1][0], runs[2][0]
print(f"run 1 vs run 2: feature grad max |diff| {np.abs(g1 - g2).max():.3e}")
d = np.abs(runs[1][0] - runs[0][0])
rows = np.argsort(-d.max(-1))[:6]
for r in rows:
    ch = np.argsort(-d[r])[:5]
    print(f"row {r}: max diff {d[r].max():.3e} of row max {np.abs(runs[0][0][r]).max():.3e}; channels off >1e-4: {(d[r] > 1e-4).sum()}; top channels {ch.tolist()} diffs {['%.2e' % d[r][c] for c in ch]}")
