mkdir -p gpurun_out/r5l
s=$(date +%s)
python bench.py > gpurun_out/r5l/bench_n1.json 2> gpurun_out/r5l/bench_n1.err
echo "exit $? after $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5l/bench_n1.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), round(d["ms_per_step"], 3), d["config"]["fps_fork_layer"], d["roofline"]["frac"], d["cpu_baseline"]["value"], (d.get("with_backbone") or {}).get("ms_per_step"), (d.get("criterion") or {}).get("ms_per_step"))
PY
grep "\[bench\]" gpurun_out/r5l/bench_n1.err | cut -c1-300
