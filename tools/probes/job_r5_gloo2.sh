# two ranks on the one GPU over gloo: the N > 1 path of bench.py end to end (fallback chain, sampling fork choice with two ranks)
mkdir -p gpurun_out/r5i
s=$(date +%s)
timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-backbone-leg > gpurun_out/r5i/bench_gloo_2ranks.json 2> gpurun_out/r5i/bench_gloo_2ranks.log
echo "exit $? after $(( $(date +%s) - s )) s"
grep "\[bench\]" gpurun_out/r5i/bench_gloo_2ranks.log | cut -c1-400
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5i/bench_gloo_2ranks.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["fallback_level"], d["config"].get("fps_fork_layer"), d["config"]["grad_allreduce"])
PY
