#!/bin/bash
out=gpurun_out/r6_c5hang; mkdir -p $out
for c in c5 c5 c5 c1 c4; do
  VDETR_BENCH_WATCHDOG=70 timeout -s KILL 100 python3 bench.py --config $c --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline 2> $out/verify.err | tail -1 | python3 -c "import sys,json
l=sys.stdin.read().strip()
print('$c', json.loads(l)['ms_per_step'] if l else 'NO OUTPUT')"
done
