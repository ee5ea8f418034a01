#!/bin/bash
# round 6 (probe build, -DVDETR_AB_SWITCHES): the next scene's sampling kernel as a 4-wave tenant instead of 16 waves
F="--no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg --steps 40 --warmup 5"
for i in 1 2; do
  for wv in 16 4 8; do
    VDETR_FPS_WAVES=$wv timeout 300 python bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps waves $wv', d['ms_per_step'], d['value'], d.get('side_stream',{}).get('fps_fork_layer'))"
  done
done
