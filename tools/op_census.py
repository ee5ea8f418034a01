#!/usr/bin/env python3
"""Which ATen operators (by name and input shapes) launch the small kernels of one eager C2 step?

    python tools/op_census.py [--config c2] [--min-count 2]

Runs the bench Trainer eagerly under torch.profiler (record_shapes) for one step and prints, per (operator, shapes), the
number of calls and the device time — the census the launch-count work in DESIGN.md §4 is planned from."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--min-count", type=int, default=1)
    ap.add_argument("--top", type=int, default=80)
    ap.add_argument("--stacks", action="store_true", help="group forward operators by Python call site instead")
    ap.add_argument("--parents", action="store_true", help="group by the enclosing autograd node / operator chain instead")
    ap.add_argument("--timeline", default=None, help="write every device kernel of the step in launch order to this file")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build_model(a.config, device)
    inputs = bench.make_inputs(a.config, device, 0)
    tr = bench.Trainer(model, inputs, 1, False, overlap=True)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=a.stacks) as prof:
        tr.step()
        torch.cuda.synchronize()
    if a.timeline:
        items = []
        for ev in prof.events():
            if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
                continue
            if ev.cpu_children and any(c.kernels for c in ev.cpu_children):
                continue
            chain, q = [], ev.cpu_parent
            while q is not None and len(chain) < 3:
                chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
                q = q.cpu_parent
            for k in ev.kernels:
                items.append((ev.time_range.start, k.duration, k.name[:70], ev.name, str(ev.input_shapes)[:70], " < ".join(chain)[:90]))
        items.sort()
        os.makedirs(os.path.dirname(os.path.abspath(a.timeline)), exist_ok=True)
        with open(a.timeline, "w") as f:
            for i, (t, dur, kn, op, shp, ch) in enumerate(items):
                f.write(f"{i:4d} {dur:8.1f} us  {kn:70s} | {op:28s} {shp:70s} | {ch}\n")
        print(f"# wrote {len(items)} kernels to {a.timeline}")
    rows = collections.defaultdict(lambda: [0, 0.0, 0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        kern = [k for k in ev.kernels]
        if not kern:
            continue
        if ev.cpu_children and any(c.kernels for c in ev.cpu_children):
            continue  # count the innermost operator that owns the kernels
        key = (ev.name, str(ev.input_shapes))
        if a.stacks:
            site = next((s for s in ev.stack if "vdetr_amd" in s or "v-detr_amd" in s or "bench.py" in s), "<autograd>")
            key = (ev.name, site)
        if a.parents:
            chain, q = [], ev.cpu_parent
            while q is not None and len(chain) < 4:
                chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
                q = q.cpu_parent
            key = (ev.name + " " + str(ev.input_shapes)[:60], " < ".join(chain))
        r = rows[key]
        r[0] += 1
        r[1] += sum(k.duration for k in kern)
        r[2] += len(kern)
    tot_calls = sum(r[0] for r in rows.values())
    tot_kern = sum(r[2] for r in rows.values())
    print(f"# {tot_calls} kernel-owning ATen calls, {tot_kern} kernels in one eager step")
    for (name, shapes), (n, us, nk) in sorted(rows.items(), key=lambda kv: -kv[1][0])[: a.top]:
        if n >= a.min_count:
            print(f"{n:5d} calls {nk:5d} kernels {us:9.1f} us  {name:28s} {shapes[:150]}")


if __name__ == "__main__":
    main()
