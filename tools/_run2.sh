python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('c5', d['ms_per_step'], d['value']); print(d['roofline']['kernel'], d['roofline']['launch_us']); print(d['roofline_secondary']['kernel'], d['roofline_secondary']['launch_us'])"
VDETR_BWD_BOX=2 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-criterion-leg --no-backbone-leg --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('c5 general kernel', d['ms_per_step'], d['value'])"
