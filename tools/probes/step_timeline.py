#!/usr/bin/env python3
"""The captured C2 step's real timeline: timestamp kernels between its phases on both streams (runtime.ts_mark), read after a
replay — no tracer in the way (rocprofv3 delays the cross-queue start of the side branch by ~1.5 ms).
    VDETR_TS_PROBE=1 python tools/probes/step_timeline.py [bench.py arguments]"""
import contextlib
import io
import os
import sys

os.environ["VDETR_TS_PROBE"] = "1"
os.environ.setdefault("VDETR_FPS_DEPTH", "1")  # (one trainer, one capture: the marks of the step that is replayed)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from vdetr_amd import runtime  # noqa: E402

orig_capture = bench.Trainer.capture


def capture(self):
    runtime.ts_reset()
    # (the warm-up steps inside capture() add marks too: only those of the captured pass are kept, see below)
    r = orig_capture(self)
    return r


bench.Trainer.capture = capture
orig_fwd_bwd = bench.Trainer._fwd_bwd


def fwd_bwd(self):
    import torch
    if torch.cuda.is_current_stream_capturing():
        runtime.ts_reset()
    return orig_fwd_bwd(self)


bench.Trainer._fwd_bwd = fwd_bwd
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-criterion-leg", "--no-exact-leg",
            "--no-backbone-leg"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
import json  # noqa: E402
line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
print(f"# ms_per_step {json.loads(line)['ms_per_step']:.3f} (with the timestamp kernels in the graph)")
prev = {}
for lab, us in runtime.ts_read():
    lane = "side" if lab.startswith("side") else "main"
    print(f"{us:9.1f} us  (+{us - prev.get(lane, 0.0):7.1f})  {lab}")
    prev[lane] = us
