#!/bin/bash
# SQ counter passes over tools/kernel_bench.py (hot attention kernels).  usage: bash tools/pmc_hot.sh <tag> [env assignments...]
tag=${1:-x}; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
pass=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/rp_sq$pass
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rp_sq$pass -o pmc -- python3 tools/kernel_bench.py c2 > $out/pmc_sq$pass.log 2>&1
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
cat $out/pmc_sq_pass1.txt $out/pmc_sq_pass2.txt | grep -A9 "attn_bwd"
