mkdir -p gpurun_out/r5o
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
python bench.py $F > gpurun_out/r5o/bench_c2.json 2> gpurun_out/r5o/bench_c2.err; echo "c2 exit $?"
timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 3 $F > gpurun_out/r5o/bench_gloo_2ranks.json 2> gpurun_out/r5o/bench_gloo_2ranks.log; echo "gloo exit $?"
python -m pytest tests/test_gpu_attention.py -x -q -m gpu -k "rounded or async or side" 2>&1 | tail -2
for f in c2 gloo_2ranks; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5o/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3), d["n_gpus"], d["config"]["fallback_level"], d["config"]["fps_fork_layer"])
PY
done
