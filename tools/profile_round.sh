#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the eager step, the hot kernels' micro-bench, and the HBM-traffic
# PMC passes.  Raw rocprofv3 output stays in /tmp (too large to copy back); summaries go to gpurun_out/prof_$1/.
tag=${1:-x}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/rp_eager -o eager -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_eager.log 2>&1
db=$(find /tmp/rp_eager -name '*.db' | head -1); csv=$(find /tmp/rp_eager -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py ${db:-$csv} 5 3 > $out/eager_kernel_summary.txt 2>&1
python3 tools/kernel_bench.py c2 > $out/kernel_bench_c2.txt 2>&1
# the CAPTURED step under the kernel trace: which hardware queue the chain and the table-gradient branch run on.
# NOTE: the tracer delays the cross-queue start of the side branch (the traced step is ~1.5 ms longer than the untraced one and
# shows the table kernels back to back behind the chain); the untraced sweep (profiles/*_async_table_sweep.txt) is the timing evidence.
rm -rf /tmp/rp_graph
VDETR_BENCH_NORMAL_EXIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_graph -o graph -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_graph_traced.log 2>&1
gcsv=$(find /tmp/rp_graph -name '*kernel_trace.csv' | head -1)
[ -n "$gcsv" ] && python3 tools/async_timeline.py $gcsv attn_bwd_box4 > $out/graph_timeline.txt 2>&1
# the step with the device criterion as its loss: kernel trace + the solver's scan counts
rocprofv3 --kernel-trace --stats -d /tmp/rp_crit -o crit -- python3 bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-backbone-leg --loss criterion > $out/bench_criterion_eager.log 2>&1 < /dev/null
cdb=$(find /tmp/rp_crit -name '*.db' | head -1)
[ -n "$cdb" ] && python3 tools/rocprof_summary.py $cdb 5 3 > $out/criterion_eager_kernel_summary.txt 2>&1
python3 tools/criterion_bench.py > $out/criterion_bench.txt 2>&1 < /dev/null
# the same micro-benchmark under the kernel trace: the per-kernel averages bench.py's `roofline` objects must agree with
VDETR_ROOFLINE_STEP_GRID_ONLY=1 rocprofv3 --kernel-trace --stats -d /tmp/rp_kb -o kb -- python3 tools/kernel_bench.py c2 > /dev/null 2>&1
kdb=$(find /tmp/rp_kb -name '*.db' | head -1)
[ -n "$kdb" ] && python3 tools/rocprof_summary.py $kdb > $out/kernel_bench_rocprof.txt 2>&1
python3 tools/kernel_bench.py c2 --indexing > $out/kernel_bench_indexing.txt 2>&1
# one cross-attention layer's backward: library-GEMM path vs the fused key-side pass, and the pieces alone (HIP events)
python3 tools/bwd_layer_bench.py > $out/bwd_layer_bench.txt 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  VDETR_ROOFLINE_STEP_GRID_ONLY=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp_$c -o pmc -- python3 tools/kernel_bench.py c2 > $out/pmc_$c.log 2>&1
  f=$(find /tmp/rp_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" $c > $out/pmc_$c.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print(f"# {sys.argv[2]} per launch (KB as reported by rocprofv3), mean over launches")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{sum(v)/len(v):14.1f} {len(v):5d}  {k[:200]}")
PY
done
# SQ counters of the hot kernels (what bounds them): two passes of 8 counters
pass=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  VDETR_ROOFLINE_STEP_GRID_ONLY=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rp_sq$pass -o pmc -- python3 tools/kernel_bench.py c2 > $out/pmc_sq$pass.log 2>&1
  f=$(find /tmp/rp_sq$pass -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" > $out/pmc_sq_pass$pass.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "vdetr" in r["Kernel_Name"]:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k[:150])
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
PY
  pass=$((pass+1))
done
# sparse-convolution backbone: per-kernel device time of a backbone step, per-layer kernel rates, geometry
python3 tools/backbone_bench.py 40000 --torch-profile > $out/backbone_profile.txt 2>&1 < /dev/null
python3 tools/backbone_bench.py 40000 > $out/backbone_bench.txt 2>&1 < /dev/null
python3 tools/spconv_bench.py > $out/spconv_bench.txt 2>&1 < /dev/null
python3 tools/backbone_step_probe.py --timeline > $out/backbone_step_probe.txt 2>&1 < /dev/null
python3 tools/op_census.py --top 60 > $out/op_census.txt 2>&1 < /dev/null
[ -x tools/probes/bin/mfma_f32_rate ] && timeout 120 tools/probes/bin/mfma_f32_rate > $out/mfma_f32_rate.txt 2>&1
[ -x tools/probes/bin/mfma_patterns ] && timeout 120 tools/probes/bin/mfma_patterns > $out/mfma_patterns.txt 2>&1
bash tools/pmc_spconv.sh $tag > /dev/null 2>&1; cat gpurun_out/pmc_sp_$tag/pmc_sq_pass*.txt > $out/spconv_pmc_sq.txt 2>/dev/null
# per-launch HBM bytes for bench.py's `roofline.traffic`, from THIS run's PMC passes (copy to profiles/rNN_pmc_traffic.json)
python3 tools/pmc_traffic.py $out > $out/pmc_traffic.json 2>/dev/null
python3 tools/fps_variants.py > $out/fps_variants.txt 2>&1 < /dev/null
[ -x tools/probes/bin/lat_probe ] && timeout 120 tools/probes/bin/lat_probe > $out/lat_probe.txt 2>&1
# the launcher path the driver uses for N > 1, with one rank (RCCL communicator, gradient all-reduce captured in the graph)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg --no-roofline > $out/bench_torchrun_1rank.json 2> $out/bench_torchrun_1rank.err; echo "exit code $?" >> $out/bench_torchrun_1rank.err
for c in c1 c4 c5; do timeout 600 python3 bench.py --config $c --no-cpu-baseline --no-criterion-leg --no-exact-leg --no-backbone-leg > $out/bench_$c.json 2> $out/bench_$c.err; done
VDETR_PMC_TRAFFIC=$out/pmc_traffic.json timeout 900 python3 bench.py --steps 20 --warmup 3 > $out/bench_n1.json 2> $out/bench_n1.err
tail -1 $out/bench_n1.json | cut -c1-400
