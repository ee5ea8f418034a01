#!/bin/bash
# needs the library built with -DVDETR_AB_SWITCHES: what the key-side pass's operand-packing launches cost the step (skipped: garbage operands, timing only)
F="--steps 30 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" timeout 200 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['config'].get('fps_fork_layer'))"; }
for rep in 1 2; do
  run VDETR_KV_SKIP_PACK=0
  run VDETR_KV_SKIP_PACK=1
done
