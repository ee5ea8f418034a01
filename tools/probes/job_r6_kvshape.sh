#!/bin/bash
# the key-side passes' workgroup shape next to the table kernel, re-measured on the round's final step (C2)
F="--steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" timeout 200 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2; do
  run VDETR_X=0
  run VDETR_BWD_ASYNC_KV_WAVES=4
  run VDETR_BWD_KV_ONE_WG=1
  run VDETR_BWD_KV_ONE_WG=2
done
