# C5 (4 scenes x 20k points, rotated boxes): the side-branch table kernel's CU count (tuned on C2: 190)
mkdir -p gpurun_out/r5k
F="--config c5 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
for g in 190 160 128 224 96; do
  VDETR_BWD_ASYNC_GRID=$g python bench.py $F > gpurun_out/r5k/bench_c5_g$g.json 2> gpurun_out/r5k/bench_c5_g$g.err
  python - "$g" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5k/bench_c5_g{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("grid", sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3))
PY
done
