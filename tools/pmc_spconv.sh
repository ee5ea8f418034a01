#!/bin/bash
# SQ / TA counter passes over tools/spconv_bench.py (fused pair-list convolution kernels).  usage: bash tools/pmc_spconv.sh <tag>
tag=${1:-x}; shift
out=$PWD/gpurun_out/pmc_sp_$tag
mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
pass=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/rp_sp$pass
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rp_sp$pass -o pmc -- python3 tools/spconv_bench.py > $out/pmc_sq$pass.log 2>&1
  f=$(find /tmp/rp_sp$pass -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" > $out/pmc_sq_pass$pass.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "vdetr::sp_pairs" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:90], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
PY
  pass=$((pass+1))
done
cat $out/pmc_sq_pass*.txt
