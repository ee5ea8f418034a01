#!/bin/bash
# the tail's two changes, interleaved (C2): pointer-array batched weight gradients on the side branch, the main stream's flush shared with it
F="--steps 30 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" VDETR_BENCH_FPS_AT_LAYER=1 timeout 200 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2 3; do
  run VDETR_PTR_BATCH=0 VDETR_FLUSH_SPLIT=0
  run VDETR_PTR_BATCH=1 VDETR_FLUSH_SPLIT=0
  run VDETR_PTR_BATCH=1 VDETR_FLUSH_SPLIT=1
  run VDETR_PTR_BATCH=0 VDETR_FLUSH_SPLIT=1
done
