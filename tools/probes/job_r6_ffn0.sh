#!/bin/bash
out=gpurun_out/r6_ffn0; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rowblock.py tests/test_capi.py tests/test_gpu_model.py -x -q -k "first_layer or capi or captured or model_forward or deferred or full_config" > $out/tests.log 2>&1; tail -3 $out/tests.log
F="--steps 30 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline"
run() { env "$@" VDETR_BENCH_FPS_AT_LAYER=1 timeout 200 python3 bench.py $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3), d['loss'])"; }
for rep in 1 2 3; do
  run VDETR_FFN0_FUSED=1
  run VDETR_FFN0_FUSED=0
done
