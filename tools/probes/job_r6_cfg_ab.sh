#!/bin/bash
# fused heads on / off on the other configurations; every run bounded (an unbounded one cost 30 GPU-minutes once)
F="--no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { c=$1; shift; env "$@" timeout 100 python3 bench.py --config $c $F 2>gpurun_out/ab_err.txt | python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('$c $*', round(d['ms_per_step'],3), round(d['value'],1), d['config'].get('fps_fork_layer'))
except Exception as e:
    print('$c $* FAILED', e); print(open('gpurun_out/ab_err.txt').read()[-600:])
"; }
for cfg in c5 c4 c1; do
  run $cfg A=0
  run $cfg VDETR_HEADS_FUSED=0
  run $cfg A=0
  run $cfg VDETR_HEADS_FUSED=0
done
