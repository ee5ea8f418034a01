#!/bin/bash
F="--no-cpu-baseline --no-criterion-leg --no-roofline --no-backbone-leg --steps 30 --warmup 5"
python -m pytest tests/test_gpu_model.py -q -x -k "deferred or captured or side_stream" 2>&1 | tail -2
for c in c2 c5; do for v in 1 0 1 0; do
  echo -n "$c side_pos=$v "; VDETR_FLUSH_SIDE_POS=$v python bench.py $F --config $c 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
