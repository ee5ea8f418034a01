# final lines of round 5 after the sampling fork choice: full GPU suite, smoke, the default bench three times, the other configs
mkdir -p gpurun_out/r5f; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5f/pytest_gpu.txt; cat gpurun_out/r5f/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r5f/smoke.txt
VDETR_PMC_TRAFFIC=profiles/r05_pmc_traffic.json python bench.py > gpurun_out/r5f/bench_n1.json 2> gpurun_out/r5f/bench_n1.err
F="--no-cpu-baseline --no-criterion-leg --no-backbone-leg"
python bench.py $F > gpurun_out/r5f/bench_n1_b.json 2> gpurun_out/r5f/bench_n1_b.err
python bench.py $F > gpurun_out/r5f/bench_n1_c.json 2> gpurun_out/r5f/bench_n1_c.err
for c in c1 c4 c5; do python bench.py --config $c $F > gpurun_out/r5f/bench_$c.json 2> gpurun_out/r5f/bench_$c.err; done
python bench.py --config c4 --dtype f32 $F > gpurun_out/r5f/bench_c4_f32.json 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline > gpurun_out/r5f/bench_torchrun_1rank.json 2> gpurun_out/r5f/bench_torchrun_1rank.err
for f in n1 n1_b n1_c c1 c4 c4_f32 c5 torchrun_1rank; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5f/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), round(d["ms_per_step"], 3), d["dtype"], d["config"].get("fps_lookahead"), d["config"].get("fps_fork_layer"), (d.get("with_backbone") or {}).get("ms_per_step"), (d.get("criterion") or {}).get("ms_per_step"), (d.get("roofline") or {}).get("frac"))
PY
done
grep -h "\[bench\]" gpurun_out/r5f/*.err | head -20
