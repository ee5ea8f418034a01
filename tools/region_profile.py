#!/usr/bin/env python3
"""Kernel launches and device time per module region of one eager train step (forward only is attributed to
modules; backward shows up under the autograd node names).  python tools/region_profile.py [depth]"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile, record_function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda")
model = bench.build_model("c2", dev)
inputs = bench.make_inputs("c2", dev, 0)
tr = bench.Trainer(model, inputs, 1, use_graph=False, overlap=False, fps_prefetch=False)

stack = {}
for name, mod in model.named_modules():
    if not name or name.count(".") >= depth:
        continue
    def pre(m, a, _n=name):
        rf = record_function("MOD:" + _n)
        rf.__enter__()
        stack.setdefault(id(m), []).append(rf)
    def post(m, a, o):
        stack[id(m)].pop().__exit__(None, None, None)
    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post)

for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step()
    torch.cuda.synchronize()

def walk(ev):
    n, t = len(ev.kernels), sum(k.duration for k in ev.kernels)
    for c in ev.cpu_children:
        cn, ct = walk(c)
        n += cn
        t += ct
    return n, t

agg = defaultdict(lambda: [0, 0, 0.0])
tops = [e for e in prof.events() if e.cpu_parent is None]
def visit(ev, inside_mod):
    label = None
    if ev.name.startswith("MOD:"):
        label = ev.name
    elif not inside_mod and (ev.name.startswith("autograd::engine::evaluate_function") or ev.name.startswith("Optimizer")):
        label = "BWD:" + ev.name.split(": ")[-1]
    if label is not None:
        n, t = walk(ev)
        a = agg[label]
        a[0] += 1; a[1] += n; a[2] += t
    if label is None or label.startswith("MOD:"):
        for c in ev.cpu_children:
            visit(c, inside_mod or label is not None)
for e in tops:
    visit(e, False)
# collapse numbered layers: layers.3.xxx -> layers.*.xxx
coll = defaultdict(lambda: [0, 0, 0.0])
import re
for k, (c, n, t) in agg.items():
    kk = re.sub(r"\.\d+(\.|$)", r".*\1", k)
    a = coll[kk]; a[0] += c; a[1] += n; a[2] += t
print(f"{'region':70s} {'calls':>6s} {'kernels':>8s} {'dev_us':>10s}")
for k, (c, n, t) in sorted(coll.items(), key=lambda kv: -kv[1][2]):
    print(f"{k[:70]:70s} {c:6d} {n:8d} {t:10.1f}")
tot_n, tot_t = 0, 0.0
for e in tops:
    n, t = walk(e); tot_n += n; tot_t += t
print(f"TOTAL kernels {tot_n}, device {tot_t:.1f} us")
