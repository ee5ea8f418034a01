#!/bin/bash
# round 6: the fused heads / position MLP against the launches they replace: step time both ways, per-kernel times of the eager step
out=gpurun_out/r6_heads; mkdir -p $out
export TMPDIR=/tmp
B="--steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
python3 bench.py $B > $out/bench_fused.json 2> $out/bench_fused.err
VDETR_HEADS_FUSED=0 VDETR_POS_FUSED=0 python3 bench.py $B > $out/bench_unfused.json 2> $out/bench_unfused.err
VDETR_POS_FUSED=0 python3 bench.py $B > $out/bench_heads_only.json 2> $out/bench_heads_only.err
rocprofv3 --kernel-trace --stats -d /tmp/rp_eager -o eager -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_eager.log 2>&1
db=$(find /tmp/rp_eager -name '*.db' | head -1); csv=$(find /tmp/rp_eager -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py ${db:-$csv} 5 3 > $out/eager_kernel_summary.txt 2>&1
for f in fused unfused heads_only; do python3 - $out/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])
PY
done
grep -E "heads_l|pos_mlp|box_decode_fwd|bn_act_fwd|rb_transpose" $out/eager_kernel_summary.txt
