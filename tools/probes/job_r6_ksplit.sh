#!/bin/bash
# needs the library built with -DVDETR_AB_SWITCHES: key splits of the persistent forward next to the sampling kernel
F="--steps 40 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" timeout 120 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2; do
  run VDETR_FWD_KSPLIT=4
  run VDETR_FWD_KSPLIT=8
  run VDETR_FWD_KSPLIT=16
  run VDETR_FWD_KSPLIT=2
done
