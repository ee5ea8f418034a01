#!/bin/bash
# the captured step under the kernel trace: every kernel with its queue and start offset (tools/async_timeline.py --all)
out=gpurun_out/r6_timeline; mkdir -p $out
export TMPDIR=/tmp
rm -rf /tmp/rp_graph
VDETR_BENCH_NORMAL_EXIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_graph -o graph -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_graph_traced.log 2>&1
gcsv=$(find /tmp/rp_graph -name '*kernel_trace.csv' | head -1)
python3 tools/async_timeline.py $gcsv attn_bwd_box4 --all > $out/graph_timeline_all.txt 2>&1
tail -3 $out/bench_graph_traced.log | cut -c1-300
