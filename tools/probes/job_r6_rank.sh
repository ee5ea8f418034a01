#!/bin/bash
# the two orders by counting ranks on many workgroups: parity, kernel durations in the eager step, the C2 step
out=gpurun_out/r6_rank; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "morton or proposal_order" > $out/tests.log 2>&1; tail -5 $out/tests.log
bash tools/probes/job_r6_eager_prof.sh > /dev/null 2>&1
grep -n "rank_kernel\|order_kernel\|dispatches" gpurun_out/r6_eager/eager_kernel_summary.txt | cut -c1-160
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/b_$i.json 2> $out/b_$i.err
  echo "run $i: $(tail -1 $out/b_$i.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"].get("fps_fork_layer"), d["loss"])')"
done
