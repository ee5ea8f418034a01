# the default (captured) bench command under rocprofv3 --kernel-trace --stats: per-kernel durations INSIDE the replayed step
# (the last 20 replays: the timed loop), next to the roofline leg's live figures in the JSON line
mkdir -p gpurun_out/r5n; export TMPDIR=/tmp; cd /tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/rp_graph -o graph -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg > gpurun_out/r5n/bench_graph_traced.json 2> gpurun_out/r5n/bench_graph_traced.err
db=$(find /tmp/rp_graph -name '*.db' | head -1)
python3 - "$db" > gpurun_out/r5n/graph_kernel_summary.txt <<'PY'
import sqlite3, sys
from collections import defaultdict
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(n, (e - s) / 1e3, s, e) for n, s, e in cur.execute("select name, start, end from kernels")]
rows.sort(key=lambda r: r[2])
# the timed loop = the last 20 optimizer launches of the captured step before the roofline leg's stand-alone launches: find the
# longest run of optimizer kernels at a regular spacing
marks = [r[3] for r in rows if "FusedOptimizerTensorListMetadata" in r[0] or "multi_tensor_apply" in r[0]]
gaps = [(marks[i + 1] - marks[i]) / 1e6 for i in range(len(marks) - 1)]
best, cur_run, run0 = (0, 0), 0, 0
for i, g in enumerate(gaps):
    if 5.0 < g < 14.0:
        if cur_run == 0:
            run0 = i
        cur_run += 1
        if cur_run > best[0]:
            best = (cur_run, run0)
    else:
        cur_run = 0
n, i0 = best
last = min(20, n)
t0, t1 = marks[i0 + n - last], marks[i0 + n]
sel = [r for r in rows if t0 < r[2] <= t1]
print(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg")
print(f"# the last {last} replays of the captured step in the trace: {(t1 - t0) / 1e6 / last:.3f} ms per step between optimizer launches (under the profiler), {len(sel) / last:.0f} kernels per step")
agg = defaultdict(list)
for name, us, _, _ in sel:
    agg[name].append(us)
tot = sum(sum(v) for v in agg.values())
print(f"{'%':>6} {'calls/step':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel")
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{100 * sum(v) / tot:6.2f} {len(v) / last:10.1f} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f}  {name[:150]}")
PY
head -30 gpurun_out/r5n/graph_kernel_summary.txt | cut -c1-170
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5n/bench_graph_traced.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 3), d["roofline"]["kernel"] if "kernel" in d["roofline"] else "", d["roofline"].get("launch_us"), d["roofline_secondary"].get("launch_us"))
PY
