# the step's launch-shape knobs once more, with the sampling branch forked in front of decoder layer 2
mkdir -p gpurun_out/r5m
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
run() { tag=$1; shift; env "$@" python bench.py $F > gpurun_out/r5m/bench_$tag.json 2> gpurun_out/r5m/bench_$tag.err; python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5m/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3), d["config"]["fps_fork_layer"])
PY
}
run default A=1
run grid192 VDETR_BWD_ASYNC_GRID=192
run grid188 VDETR_BWD_ASYNC_GRID=188
run kv1 VDETR_BWD_KV_ONE_WG=1
run grid192_kv1 VDETR_BWD_ASYNC_GRID=192 VDETR_BWD_KV_ONE_WG=1
