python -m pytest tests/test_gpu_attention.py -x -q -k "fused_attention_forward_backward or key_side or box_backward" 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('nt on:', d['ms_per_step'])"; done
export VDETR_EXTRA_HIPCC_FLAGS="-DVDETR_STREAM_NT=0"; python3 v-detr_amd/build.py --force > /dev/null 2>&1
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('nt off:', d['ms_per_step'])"; done
