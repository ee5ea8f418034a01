#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv) into the per-kernel table that is
committed under profiles/:  python tools/rocprof_summary.py <results.db|kernel_trace.csv> [steps [warmup]] > profiles/xxx.txt
With `warmup`, the first `warmup` of the `steps + warmup` traced steps are dropped (library auto-tuning runs every
candidate solver once during the first steps); step boundaries are the optimizer's kernels."""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(n, (e - s) / 1e3, s, e) for n, s, e in cur.execute("select name, start, end from kernels")]


def rows_from_csv(path):
    out = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            out.append((r["Kernel_Name"], (e - s) / 1e3, s, e))
    return out


def _is_optimizer(name):
    """the step boundary: the optimizer's launch (csrc/optim.hip, or torch's multi-tensor AdamW with VDETR_OWN_ADAMW=0)"""
    return "adamw_clip_kernel" in name or "FusedOptimizerTensorListMetadata" in name


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    if warmup:
        marks = sorted(e for n, _, _, e in rows if _is_optimizer(n))
        per_step = len(marks) // (steps + warmup)
        t0 = marks[warmup * per_step - 1]
        rows = [r for r in rows if r[2] > t0]
        print(f"# first {warmup} steps dropped (auto-tuning), {steps} steady-state steps summarised")
    if "--timeline" in sys.argv:  # dispatch sequence of the last traced step, in start order, with gaps
        marks = sorted(e for n, _, _, e in rows if _is_optimizer(n))
        per_step = len(marks) // max(steps, 1)
        t0 = marks[-per_step - 1] if len(marks) > per_step else 0
        seq = sorted((r for r in rows if r[2] > t0), key=lambda r: r[2])
        prev = None
        for n, d, st, en in seq:
            gap = (st - prev) / 1e3 if prev is not None else 0.0
            print(f"{(st - seq[0][2]) / 1e3:10.1f} {d:8.2f} {gap:7.2f}  {n[:150]}")
            prev = en
        return
    rows = [(n, d) for n, d, _, _ in rows]
    agg = defaultdict(list)
    for n, d in rows:
        agg[n].append(d)
    tot = sum(d for _, d in rows)
    print(f"# source: {path}")
    print(f"# kernels: {len(rows)} dispatches, {tot / 1e3:.2f} ms total GPU kernel time" +
          (f", {steps} steps -> {tot / 1e3 / steps:.2f} ms/step, {len(rows) // steps} dispatches/step" if steps else ""))
    print(f"{'%':>6} {'calls':>7} {'total_us':>11} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
    for n, ds in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
        print(f"{sum(ds) / tot * 100:6.2f} {len(ds):7d} {sum(ds):11.1f} {sum(ds) / len(ds):10.2f} {min(ds):9.2f} {max(ds):9.2f}  {n[:260]}")


if __name__ == "__main__":
    main()
