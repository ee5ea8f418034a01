#!/bin/bash
# Per-phase shader cycles of attn_bwd_box3_kernel (csrc/attn_bwd_box3.hip built with -DVDETR_B3_PROF: s_memtime marks around
# the five phases of a tile, printed by workgroup 2 at the end of the launch); C2-size launch of tools/kernel_bench.py.
export VDETR_EXTRA_HIPCC_FLAGS="-DVDETR_B3_PROF ${B3_FLAGS:-}"
python3 v-detr_amd/build.py --force > /dev/null 2>&1 || { echo build failed; exit 1; }
VDETR_BWD_BOX=4 python3 tools/kernel_bench.py c2 2>&1 | grep -E "box3 prof|bwd_us" | tail -8
unset VDETR_EXTRA_HIPCC_FLAGS
python3 v-detr_amd/build.py --force > /dev/null 2>&1
