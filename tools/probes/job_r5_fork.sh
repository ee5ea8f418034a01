# the FPS fork-layer choice of bench.py on the four configurations (C2 three times)
mkdir -p gpurun_out/r5e
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
for i in 1 2 3; do python bench.py $F > gpurun_out/r5e/bench_c2_$i.json 2> gpurun_out/r5e/bench_c2_$i.err; done
python bench.py --config c1 $F > gpurun_out/r5e/bench_c1.json 2> gpurun_out/r5e/bench_c1.err
python bench.py --config c4 --dtype bf16 $F > gpurun_out/r5e/bench_c4_bf16.json 2> gpurun_out/r5e/bench_c4_bf16.err
python bench.py --config c4 --dtype f32 $F > gpurun_out/r5e/bench_c4_f32.json 2> gpurun_out/r5e/bench_c4_f32.err
python bench.py --config c5 $F > gpurun_out/r5e/bench_c5.json 2> gpurun_out/r5e/bench_c5.err
for f in c2_1 c2_2 c2_3 c1 c4_bf16 c4_f32 c5; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5e/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["dtype"], d["config"].get("fps_lookahead"), d["config"].get("fps_fork_layer"))
PY
grep "\[bench\] sampling" gpurun_out/r5e/bench_$f.err
done
