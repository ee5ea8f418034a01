#!/bin/bash
# round 6: per-kernel durations of the indexing bench (rocprofv3), for the LDS-tiled ball query
mkdir -p gpurun_out
R=$PWD; export TMPDIR=/tmp
VDETR_ROOFLINE_STEP_GRID_ONLY=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/rp_idx -o idx -- python3 tools/kernel_bench.py c2 --indexing > /tmp/idx.log 2>&1
tail -3 /tmp/idx.log
db=$(find /tmp/rp_idx -name '*.db' | head -1)
[ -n "$db" ] && python3 tools/rocprof_summary.py $db > gpurun_out/r06_kernel_bench_indexing_rocprof2.txt 2>&1
grep -i "ball\|group\|gather\|three" gpurun_out/r06_kernel_bench_indexing_rocprof2.txt | cut -c1-160
