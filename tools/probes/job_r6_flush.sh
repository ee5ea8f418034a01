#!/bin/bash
F="--steps 30 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2; do
  run A=0
  run VDETR_FLUSH_PER_LAYER=0
done
python3 tools/probes/step_timeline.py 2>&1 | grep -v "self-attention" | tail -64
