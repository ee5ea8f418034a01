mkdir -p gpurun_out/r5a; export TMPDIR=/tmp
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k full_config 2>&1 | tail -70 > gpurun_out/r5a/fullcfg.txt; tail -3 gpurun_out/r5a/fullcfg.txt
rocprofv3 --kernel-trace --stats -d /tmp/rp_eager -o eager -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-backbone-leg > gpurun_out/r5a/bench_eager5.log 2>&1
db=$(find /tmp/rp_eager -name '*.db' | head -1); csv=$(find /tmp/rp_eager -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py ${db:-$csv} 5 3 > gpurun_out/r5a/eager_kernel_summary5.txt 2>&1
head -45 gpurun_out/r5a/eager_kernel_summary5.txt | cut -c1-160
python3 tools/op_census.py > gpurun_out/r5a/op_census5.txt 2>&1; tail -5 gpurun_out/r5a/op_census5.txt
