#!/bin/bash
# round 6, final tree: the eager step's per-kernel summary and the captured step's queue timeline (the first two steps of tools/profile_round.sh)
out=$PWD/gpurun_out/prof_r06f; mkdir -p $out; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/rp_eager -o eager -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_eager.log 2>&1
db=$(find /tmp/rp_eager -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db 5 3 > $out/eager_kernel_summary.txt 2>&1
rm -rf /tmp/rp_graph
VDETR_BENCH_NORMAL_EXIT=1 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_graph -o graph -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_graph_traced.log 2>&1
gcsv=$(find /tmp/rp_graph -name '*kernel_trace.csv' | head -1)
[ -n "$gcsv" ] && python3 tools/async_timeline.py $gcsv attn_bwd_box4 > $out/graph_timeline.txt 2>&1
head -12 $out/eager_kernel_summary.txt | cut -c1-150
