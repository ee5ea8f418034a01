#!/usr/bin/env python3
"""Times the device criterion alone (C2 shapes, outputs of the randomly initialised model) and reports the solver's
row-scan counts:  python tools/criterion_bench.py [--boxes 24]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=24)
    ap.add_argument("--config", default="c2")
    a = ap.parse_args()
    from vdetr_amd.criterion import Matcher, build_criterion, default_criterion_args
    dev = torch.device("cuda", 0)
    model = bench.build_model(a.config, dev)
    inputs = bench.make_inputs(a.config, dev, 0)
    targets = bench.make_targets(a.config, dev, 0, boxes_per_scene=a.boxes)
    crit = build_criterion(default_criterion_args(), model.dataset_config)
    out = model(inputs)
    prep = crit.prepare_targets(targets)

    def run():
        loss, _ = crit(out, prep)
        loss.backward(retain_graph=True)
        return loss

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"criterion fwd+bwd (eager launches, decoder backward included): {e0.elapsed_time(e1) / 10:.3f} ms")
    # solver alone + scan counts
    stages = [(out["outputs"], True, -1)] + [((o, False, 0) if k == 0 else (o, True, -1)) for k, o in enumerate(out["aux_outputs"])]
    problems = []
    for o, rep, ov in stages:
        records, G, nactual, _ = prep.stage(rep)
        problems.append((crit.matcher.cost(o, records, G, nactual, label_override=ov)[0], nactual, prep.repeat if rep else 0))
    status = torch.zeros((sum(p[0].shape[0] for p in problems), 2), dtype=torch.int32, device=dev)
    for _ in range(2):
        Matcher.solve(problems, status)
    torch.cuda.synchronize()
    e0.record()
    Matcher.solve(problems, status)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = status.cpu().tolist()
    print(f"assignment launch: {ms * 1e3:.1f} us; per problem (invalid, scans): {st}")
    print(f"slowest problem: {max(s[1] for s in st)} scans -> {ms * 1e3 / max(max(s[1] for s in st), 1):.2f} us per scan")


if __name__ == "__main__":
    main()
