#!/bin/bash
# eager kernel summary of the C2 step (which kernels run, how long): gpurun_out/r6_eager/
out=$PWD/gpurun_out/r6_eager; mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/rp_eager
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_eager -o eager -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_eager.log 2>&1
db=$(find /tmp/rp_eager -name '*.db' | head -1); csv=$(find /tmp/rp_eager -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py ${db:-$csv} 5 3 > $out/eager_kernel_summary.txt 2>&1
head -60 $out/eager_kernel_summary.txt | cut -c1-150
