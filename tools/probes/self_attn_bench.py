#!/usr/bin/env python3
"""The query self-attention's forward alone (per-head kind, 1024 x 1024 x 4 heads, dropout 0.1, scores stored): the lean kernel
(attn_fwd_self.hip) against attn_fwd.hip's body, HIP events over 200 calls each, interleaved."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vdetr_amd import attention as A

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn((1, 1024, 256), generator=g).to(dev).requires_grad_(True) for _ in range(3))
rng = A.begin_step(dev)
def run(body, reps=200):
    A.SELF_FWD_BODY = body
    for _ in range(5):
        A.fused_attention(q, k, v, num_heads=4, scale=0.125, shared_kv=False, dropout_p=0.1, rng_state=rng, salt=3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        A.fused_attention(q, k, v, num_heads=4, scale=0.125, shared_kv=False, dropout_p=0.1, rng_state=rng, salt=3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rnd in range(3):
    print(json.dumps({"round": rnd, "lean_us": run(False), "body_us": run(True)}))
with torch.no_grad():  # no scores stored (the kernels' STORE = false instantiations)
    for rnd in range(2):
        print(json.dumps({"no_scores_round": rnd, "lean_us": run(False), "body_us": run(True)}))
