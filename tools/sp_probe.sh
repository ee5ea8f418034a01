#!/bin/bash
# pair-GEMM kernel with parts of its loop removed (VDETR_SP_PROBE, csrc/sparse_conv.hip): where does the time go?
for p in ${PROBES:-0 1 2 3}; do
  export VDETR_EXTRA_HIPCC_FLAGS="-DVDETR_SP_PROBE=$p -DVDETR_SP_PADTEST"
  python3 -c "from vdetr_amd import build; build.build(force=True)" > /dev/null 2>&1 || python3 v-detr_amd/build.py --force > /dev/null
  echo "== probe $p LDS pad ${VDETR_SP_LDS_PAD:-0}"
  python3 tools/spconv_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: round(v, 1) if isinstance(v, float) else v for k, v in d.items() if k in ('channels', 'pairs', 'fwd_us', 'fwd_TF', 'dgrad_us', 'dgrad_TF')})
"
done
unset VDETR_EXTRA_HIPCC_FLAGS
python3 v-detr_amd/build.py --force > /dev/null
