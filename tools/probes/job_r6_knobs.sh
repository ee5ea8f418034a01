#!/bin/bash
# round 6: the exact-f32 leg (arith.exact_f32 / max_rel_vs_exact), and dQ by attn_bwd_dq.hip re-measured
out=gpurun_out/r6_knobs; mkdir -p $out
F="--steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline"
python3 bench.py $F > $out/bench_exact_leg.json 2> $out/bench_exact_leg.err
F="$F --no-exact-leg"
for dq in 0 1 2; do VDETR_BWD_DQ=$dq python3 bench.py $F > $out/bench_dq$dq.json 2> $out/bench_dq$dq.err; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_knobs/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d["arith"].get("exact_f32"), d["arith"].get("max_rel_vs_exact"), d.get("exact_f32_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
