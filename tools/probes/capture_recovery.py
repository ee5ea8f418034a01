#!/usr/bin/env python3
"""After a stream capture that fails with a forked, un-joined side stream: which streams stay in capture state, and does
runtime.end_stray_captures() bring the process back (a pageable host-to-device copy works again)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vdetr_amd import runtime  # noqa: E402

dev = torch.device("cuda")
s, side = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.ones(1024, device=dev)
torch.cuda.synchronize()
os.environ["VDETR_DEBUG_CAPTURE"] = "1"
print("status before any capture: origin", runtime.capture_status(s), "side", runtime.capture_status(side))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        y = x * 2
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            z = y + 1          # the side stream joins the capture ...
        t = torch.tensor([1.0]).to(dev)   # ... and a pageable copy invalidates it before the join
        torch.cuda.current_stream().wait_stream(side)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e).splitlines()[0][:100])
print("status after the failed capture: origin", runtime.capture_status(s), "side", runtime.capture_status(side))
try:
    print("copy:", torch.tensor([3.0]).to(dev).item())
except Exception as e:
    print("copy fails:", str(e).splitlines()[0][:100])
try:
    s.wait_stream(side)
    g.capture_end()
    print("second capture_end succeeded")
except Exception as e:
    print("second capture_end:", type(e).__name__, str(e).splitlines()[0][:120])
print("status now: origin", runtime.capture_status(s), "side", runtime.capture_status(side))
try:
    print("copy:", torch.tensor([3.0]).to(dev).item())
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s, capture_error_mode="thread_local"):
        y = x * 3
    g2.replay()
    torch.cuda.synchronize()
    print("a new capture works:", float(y[0]))
except Exception as e:
    print("still broken:", str(e).splitlines()[0][:100])

# second kind of failure: the capture is invalidated from inside (an operation that is not permitted while capturing)
g3 = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g3, stream=s, capture_error_mode="thread_local"):
        y = x * 2
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            z = y + 1
        v = float(y[0].item())   # a device-to-host copy with a synchronise
except Exception as e:
    print("capture failed:", type(e).__name__, str(e).splitlines()[0][:100])
print("status after the invalidated capture: origin", runtime.capture_status(s), "side", runtime.capture_status(side))
try:
    s.wait_stream(side)
    g3.capture_end()
    print("second capture_end succeeded")
except Exception as e:
    print("second capture_end:", type(e).__name__, str(e).splitlines()[0][:120])
print("status now: origin", runtime.capture_status(s), "side", runtime.capture_status(side))
try:
    print("copy:", torch.tensor([5.0]).to(dev).item())
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, stream=s, capture_error_mode="thread_local"):
        y = x * 4
    g4.replay()
    torch.cuda.synchronize()
    print("a new capture works:", float(y[0]))
except Exception as e:
    print("still broken:", str(e).splitlines()[0][:100])
