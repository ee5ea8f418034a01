#!/bin/bash
F="--steps 40 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2 3; do
  run VDETR_HEADS_FUSED=0 VDETR_POS_FUSED=0
  run VDETR_HEADS_FUSED=0
  run VDETR_POS_FUSED=0
  run VDETR_HEADS_FUSED=0 VDETR_POS_FUSED=0 VDETR_DEFER_STAGE0=0
done
