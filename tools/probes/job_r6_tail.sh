#!/bin/bash
# the tail of the step (after the last decoder layer's backward): which stream runs which parked gradients
F="--steps 40 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" timeout 120 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2 3; do
  run A=0
  run VDETR_FLUSH_SIDE_POS=0
  run VDETR_FLUSH_SIDE_POS=0 VDETR_FLUSH_SIDE_LN=0
  run VDETR_FLUSH_SIDE_POS=0 VDETR_FLUSH_SIDE_LN=0 VDETR_HEADS_SIDE=0
done
