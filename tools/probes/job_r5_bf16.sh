# the rounded-operand bf16 configuration: parity tests, then C4 bf16 / f32 side by side
mkdir -p gpurun_out/r5h; export TMPDIR=/tmp
python -m pytest tests/test_gpu_attention.py -x -q -m gpu -k "rounded or bf16 or kv_images" 2>&1 | tail -8
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
for i in 1 2; do
python bench.py --config c4 --dtype bf16 $F > gpurun_out/r5h/bench_c4_bf16_$i.json 2> gpurun_out/r5h/bench_c4_bf16_$i.err
python bench.py --config c4 --dtype f32 $F > gpurun_out/r5h/bench_c4_f32_$i.json 2> gpurun_out/r5h/bench_c4_f32_$i.err
done
python bench.py --dtype bf16 $F > gpurun_out/r5h/bench_c2_bf16.json 2> gpurun_out/r5h/bench_c2_bf16.err
for f in c4_bf16_1 c4_f32_1 c4_bf16_2 c4_f32_2 c2_bf16; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5h/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3), d["dtype"], d["config"].get("fps_fork_layer"), d["loss"])
PY
done
tail -3 gpurun_out/r5h/*.err | tail -20
