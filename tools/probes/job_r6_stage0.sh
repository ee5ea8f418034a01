#!/bin/bash
# the first stage's heads' weight gradients: in line on the main chain (default) or on the side branch at once (C2, fixed fork layer)
F="--steps 30 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" VDETR_BENCH_FPS_AT_LAYER=1 timeout 200 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2 3; do
  run VDETR_STAGE0_WG=inline
  run VDETR_STAGE0_WG=side
done
