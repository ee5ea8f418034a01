#!/bin/bash
# SQ counters of one forward kernel of the 3DV-RPE attention (tools/fwd_ab.py --only pipe|grid), two passes of 8 counters.
# usage (on the GPU box): bash tools/pmc_fwd.sh <tag> [pipe|grid] [c2|c5]
tag=${1:-x}; which=${2:-pipe}; cfg=${3:-c2}
out=$PWD/gpurun_out/pmc_fwd_$tag
mkdir -p $out
export TMPDIR=/tmp
cd $PWD
pass=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/rp_f$pass
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rp_f$pass -o pmc -- python3 tools/fwd_ab.py $cfg --only $which --reps 8 > $out/pass$pass.log 2>&1
  f=$(find /tmp/rp_f$pass -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" > $out/${which}_${cfg}_sq_pass$pass.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_fwd" in r["Kernel_Name"]:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k[:150])
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
PY
  pass=$((pass+1))
done
rm -rf /tmp/rp_fk
rocprofv3 --kernel-trace --stats -d /tmp/rp_fk -o kt -- python3 tools/fwd_ab.py $cfg --only $which --reps 20 > $out/kt.log 2>&1
db=$(find /tmp/rp_fk -name '*.db' | head -1); csv=$(find /tmp/rp_fk -name '*kernel_trace.csv' | head -1)
python3 tools/rocprof_summary.py ${db:-$csv} > $out/${which}_${cfg}_kernel_trace.txt 2>&1
