#!/bin/bash
# fused pair-list kernels at 2 / 3 / 4 waves per SIMD (VDETR_SP_WAVES, csrc/sparse_conv.hip)
for w in ${WAVES:-2 3 4}; do
  export VDETR_EXTRA_HIPCC_FLAGS="-DVDETR_SP_WAVES=$w"
  python3 -c "from vdetr_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== waves per SIMD <= $w"
  python3 tools/spconv_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: round(v, 1) if isinstance(v, float) else v for k, v in d.items() if k in ('channels', 'pairs', 'fwd_us', 'fwd_TF', 'dgrad_us', 'dgrad_TF', 'wgrad_us', 'wgrad_TF', 'library_TF')})
"
done
unset VDETR_EXTRA_HIPCC_FLAGS
python3 v-detr_amd/build.py --force > /dev/null
