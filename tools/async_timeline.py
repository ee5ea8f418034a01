#!/usr/bin/env python3
"""Timeline of one captured step from a rocprofv3 kernel trace: every kernel between the ends of the last two steps (the optimizer's launch), in
start order, with its queue, start offset, duration and how much of it ran while a table-gradient kernel was running.
    python tools/async_timeline.py <kernel_trace.csv> [table kernel substring] [--all]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "attn_bwd_box4"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "adamw_clip_kernel" in r["Kernel_Name"] or "FusedOptimizerTensorListMetadata" in r["Kernel_Name"]]  # a step's last launch
a0, a1 = adam[-2], adam[-1]
step = [r for r in rows if a0["s"] < r["s"] <= a1["s"] and "fps_rows" not in r["Kernel_Name"]]
t0 = step[0]["s"]
tab = [r for r in step if pat in r["Kernel_Name"]]
print(f"# step of {(a1['s'] - a0['s']) / 1e6:.3f} ms, {len(step)} kernels, {len(tab)} table kernels, sum of kernel time {sum(r['e'] - r['s'] for r in step) / 1e6:.3f} ms")


def overlap(r):
    return sum(max(0, min(r["e"], t["e"]) - max(r["s"], t["s"])) for t in tab if t is not r)


tot_ov = 0
for r in step:
    ov = overlap(r) if pat not in r["Kernel_Name"] else 0
    tot_ov += ov
    if "--all" in sys.argv or pat in r["Kernel_Name"]:
        print(f"q{r.get('Queue_Id', '?'):>3} {(r['s'] - t0) / 1e3:9.1f} us  {(r['e'] - r['s']) / 1e3:8.1f} us  under-table {ov / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
print(f"# main-chain kernel time that ran under a table kernel: {tot_ov / 1e6:.3f} ms; table kernels: {sum(t['e'] - t['s'] for t in tab) / 1e6:.3f} ms")
# per kernel name: mean duration under a table kernel vs alone
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in step:
    if pat in r["Kernel_Name"]:
        continue
    d = r["e"] - r["s"]
    under = overlap(r) > 0.5 * d
    a = acc[r["Kernel_Name"][:70]]
    a[0 if under else 2] += 1
    a[1 if under else 3] += d / 1e3
print("# kernel: n under table, mean us under | n alone, mean us alone")
for k, a in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:40]:
    print(f"{a[0]:4d} {a[1] / max(a[0], 1):8.1f} | {a[2]:4d} {a[3] / max(a[2], 1):8.1f}  {k}")
