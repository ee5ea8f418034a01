mkdir -p gpurun_out/r5c
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg"
VDETR_PMC_TRAFFIC=profiles/r05_pmc_traffic.json python bench.py --steps 20 --warmup 3 > gpurun_out/r5c/bench_n1.json 2> gpurun_out/r5c/bench_n1.err
for c in c1 c4 c5; do python bench.py --config $c $F > gpurun_out/r5c/bench_$c.json 2> gpurun_out/r5c/bench_$c.err; done
python bench.py --config c4 --dtype f32 $F > gpurun_out/r5c/bench_c4_f32.json 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline > gpurun_out/r5c/bench_torchrun_1rank.json 2> gpurun_out/r5c/bench_torchrun_1rank.err
for f in n1 c1 c4 c4_f32 c5 torchrun_1rank; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5c/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3), d["dtype"], (d.get("with_backbone") or {}).get("ms_per_step"), (d.get("criterion") or {}).get("ms_per_step"))
PY
done
