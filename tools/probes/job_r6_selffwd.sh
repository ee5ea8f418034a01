#!/bin/bash
# round 6: the lean self-attention forward in the captured C2 step, A/B against attn_fwd.hip's body (interleaved, two runs each)
mkdir -p gpurun_out
F="--no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg --steps 40 --warmup 5"
for i in 1 2; do
  for v in lean body; do
    if [ $v = body ]; then export VDETR_SELF_FWD=body; else unset VDETR_SELF_FWD; fi
    timeout 300 python bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
