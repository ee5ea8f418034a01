#!/bin/bash
# parked weight gradients through rocBLAS' pointer-array batched GEMM (no torch.stack copies): parity, then the C2 step with and without
out=gpurun_out/r6_ptrbatch; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_rowblock.py tests/test_gpu_heads.py -x -q -k "not full_config" > $out/tests.log 2>&1; tail -5 $out/tests.log
for i in 1 2; do
  for v in 1 0; do
    VDETR_PTR_BATCH=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/b_${v}_${i}.json 2> $out/b_${v}_${i}.err
    echo "ptr_batch=$v run $i: $(tail -1 $out/b_${v}_${i}.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("fps_fork_layer"), d["loss"])')"
  done
done
