#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (csv) of the last N ms: per queue the busy time and the top kernels, and for a named
kernel its durations over time — to see WHICH concurrent work stretches the main stream's kernels.
    python tools/trace_overlap.py <kernel_trace.csv> [window_ms] [kernel substring]
    python tools/trace_overlap.py --allreduce <kernel_trace.csv>
        per training step (delimited by the fused AdamW launch): does every RCCL all-reduce kernel START before the last kernel
        of the backbone's backward pass has ended (i.e. inside the backward window), as `bench.py --force-dist` (1-rank
        communicator) / `--with-backbone-dist` promise?  """
import csv
import sys
from collections import defaultdict


def allreduce_window(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    adam = [r for r in rows if "adam" in r["Kernel_Name"].lower()]
    print(f"# {len(rows)} kernels, {len(adam)} optimizer launches")
    ok = bad = 0
    for a0, a1 in zip(adam[:-1], adam[1:]):
        step = [r for r in rows if a0["e"] <= r["s"] < a1["s"]]
        nccl = [r for r in step if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
        bwd = [r for r in step if "sp_" in r["Kernel_Name"] and ("wgrad" in r["Kernel_Name"] or "bwd" in r["Kernel_Name"] or "gather_sum" in r["Kernel_Name"])]
        if not nccl or not bwd:
            continue
        last = max(r["e"] for r in bwd)
        first = min(r["s"] for r in bwd)
        inside = [r for r in nccl if r["s"] < last]
        ok += len(inside)
        bad += len(nccl) - len(inside)
        print(f"step: backbone backward {(last - first) / 1e6:.2f} ms; {len(nccl)} all-reduce kernels, {len(inside)} started inside it; "
              f"starts at {[round((r['s'] - first) / 1e6, 2) for r in nccl]} ms after its first kernel, durations {[round((r['e'] - r['s']) / 1e3) for r in nccl]} us")
    print(f"# all-reduce kernels inside the backward window: {ok}, after it: {bad}")



if len(sys.argv) > 2 and sys.argv[1] == "--allreduce":
    allreduce_window(sys.argv[2])
    sys.exit(0)


rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
pat = sys.argv[3] if len(sys.argv) > 3 else "attn_bwd_box4"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
end = max(r["e"] for r in rows)
rows = [r for r in rows if r["s"] >= end - win * 1e6]
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "?")].append(r)
print(f"# last {win} ms: {len(rows)} kernels on {len(byq)} queues")
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs) / 1e6
    top = defaultdict(float)
    for r in rs:
        top[r["Kernel_Name"][:60]] += (r["e"] - r["s"]) / 1e6
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.2f} ms; top:", [(k, round(v, 2)) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:4]])
sel = [r for r in rows if pat in r["Kernel_Name"]]
print(f"# '{pat}': ", [round((r["e"] - r["s"]) / 1e3) for r in sel][:64], "us")
# what runs concurrently with the selected kernel's slowest instance
if sel:
    worst = max(sel, key=lambda r: r["e"] - r["s"])
    conc = [r for r in rows if r is not worst and r["s"] < worst["e"] and r["e"] > worst["s"]]
    print("# concurrent with the slowest instance:", [(r.get("Queue_Id"), r["Kernel_Name"][:50], round((r["e"] - r["s"]) / 1e3)) for r in conc][:12])
