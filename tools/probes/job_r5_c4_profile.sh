# C4 eager step under rocprofv3, bf16 and f32 operand storage side by side (which launches the bf16 configuration adds / shortens)
mkdir -p gpurun_out/r5g; export TMPDIR=/tmp
F="--steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-backbone-leg"
for dt in bf16 f32; do
  rocprofv3 --kernel-trace --stats -d /tmp/rp_$dt -o c4 -- python3 bench.py --config c4 --dtype $dt $F > gpurun_out/r5g/bench_c4_$dt.log 2>&1
  db=$(find /tmp/rp_$dt -name '*.db' | head -1); csv=$(find /tmp/rp_$dt -name '*kernel_trace.csv' | head -1)
  python3 tools/rocprof_summary.py ${db:-$csv} 5 3 > gpurun_out/r5g/c4_${dt}_kernel_summary.txt 2>&1
  head -60 gpurun_out/r5g/c4_${dt}_kernel_summary.txt | cut -c1-150
done
