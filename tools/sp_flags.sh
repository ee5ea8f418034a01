#!/bin/bash
# spconv_bench under several build flag sets: bash tools/sp_flags.sh "<flags 1>" "<flags 2>" ...
for f in "$@"; do
  export VDETR_EXTRA_HIPCC_FLAGS="$f"
  python3 -c "from vdetr_amd import build; build.build(force=True)" 2>&1 | grep -i "error" | head -5
  echo "== flags: $f"
  python3 tools/spconv_bench.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: round(v, 1) if isinstance(v, float) else v for k, v in d.items() if k in ('channels', 'fwd_us', 'fwd_TF', 'dgrad_us', 'dgrad_TF', 'wgrad_us', 'wgrad_TF')})
"
done
unset VDETR_EXTRA_HIPCC_FLAGS
python3 v-detr_amd/build.py --force > /dev/null
