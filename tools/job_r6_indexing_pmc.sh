#!/bin/bash
# round 6: HBM bytes of the indexing kernels from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md's
# recipe: FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported), next to their device times -> achieved HBM GB/s per kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  VDETR_ROOFLINE_STEP_GRID_ONLY=1 timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rpi_$c -o pmc -- python3 tools/kernel_bench.py c2 --indexing > /tmp/pmci_$c.log 2>&1
done
VDETR_ROOFLINE_STEP_GRID_ONLY=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/rpi_t -o t -- python3 tools/kernel_bench.py c2 --indexing > /tmp/pmci_t.log 2>&1
python3 - > gpurun_out/r06_indexing_hbm.txt <<'PY'
import csv, glob
from collections import defaultdict
def counter(c):
    f = glob.glob(f"/tmp/rpi_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
f = glob.glob("/tmp/rpi_t/**/*kernel_trace.csv", recursive=True)[0]
dur = defaultdict(list)
for r in csv.DictReader(open(f)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# indexing kernels at the shapes of tools/kernel_bench.py --indexing (c2 scene: 39,642 points): HBM bytes per launch from separate")
print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (KB as reported; FETCH doubled: gfx950 tallies 128-B requests at 64 B),")
print("# device time from a --kernel-trace pass; GB/s = (2 x FETCH + WRITE) / time; frac = of 8 TB/s")
print(f"{'kernel':60s} {'us':>8s} {'fetch MB':>9s} {'write MB':>9s} {'GB/s':>8s} {'frac':>6s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if not any(s in k for s in ("ball_query", "group_points", "gather_points", "three_", "scatter_rows", "fps_rows")):
        continue
    t = sorted(dur[k])[len(dur[k]) // 2]
    fb, wb = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    print(f"{k[:60]:60s} {t:8.2f} {fb/1e6:9.2f} {wb/1e6:9.2f} {(fb+wb)/t/1e3:8.0f} {(fb+wb)/t/1e3/8000:6.3f}")
PY
cat gpurun_out/r06_indexing_hbm.txt
