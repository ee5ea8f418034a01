#!/usr/bin/env python3
"""Where does the captured step's time go between its main chain and the table-gradient branch?  Three runs of bench.py's step
(separate processes, selected by argv[1]):
  normal    the step as benched
  notable   the RPE-table gradient launches left out (NOT a valid step: the lower bound the main chain alone would reach)
  inline    the table gradient on the main stream, all 256 CUs (VDETR_BWD_ASYNC_TABLE=0)
  grid=N    the side-stream launch on N workgroups
  noflush   the decoder layers' parked weight gradients dropped (not a valid step)
  nofps     the next scene's sampling launch left out (not a valid step)
  nomlp     the RPE tables' own backward left out (not a valid step)
    for m in normal notable inline; do python tools/probes/step_bounds.py $m | tail -1; done"""
import json
import os
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "normal"
if mode == "inline":
    os.environ["VDETR_BWD_ASYNC_TABLE"] = "0"
if mode.startswith("grid="):
    os.environ["VDETR_BWD_ASYNC_GRID"] = mode[5:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import vdetr_amd.attention as A  # noqa: E402
if os.environ.get("VDETR_PROBE_LIB"):  # a variant build of the library (experiments)
    import vdetr_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["VDETR_PROBE_LIB"])

if mode == "noflush":  # the decoder layers' parked weight gradients dropped (NOT a valid step): what the flush on the side branch costs
    from vdetr_amd.helpers import DeferredParamGrads as _D
    _orig = _D.flush.__func__

    def _drop(cls, select=None, collect=None, keepalive=None):
        if select is not None:
            cls.pending = [it for it in cls.pending if not select(it)]
            return
        return _orig(cls, select, collect, keepalive)
    _D.flush = classmethod(_drop)
if mode == "nofps":  # the next scene's sampling launch left out (NOT a valid step): what its CU and its 4.7 ms cost the step
    from vdetr_amd import model_vdetr as _M
    _real = _M.ModelVDETR.sample_indices
    _cache = {}

    def _cached(self, inputs):
        if "i" not in _cache:
            _cache["i"] = _real(self, inputs)
        return _cache["i"]
    _M.ModelVDETR.sample_indices = _cached
if mode == "nomlp":  # the RPE tables' own backward (three batched GEMMs at the end of the side branch) left out (not valid)
    A.DeferredTableGrads.begin_flush = classmethod(lambda cls: None)
if mode == "notable":
    A._launch_table_async = lambda lib, d, q, ds, table, aux, vertices, xyz, mask, fork, dtable, also=(): dtable
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-criterion-leg", "--no-backbone-leg"] + sys.argv[2:]  # (e.g. --config c5)
import io
import contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print(json.dumps({"mode": mode, "ms_per_step": d["ms_per_step"], "scenes_per_s": d["value"]}))
