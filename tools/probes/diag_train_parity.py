#!/usr/bin/env python3
"""Train-mode forward of the small model (tests/test_gpu_model.py sizes) on the GPU under switch combinations against the CPU
oracle: which switch moves which stage."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as TM  # noqa: E402
from vdetr_amd import runtime  # noqa: E402
from vdetr_amd import vdetr_transformer as T  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
nq, npre, nl, npts = (int(x) for x in (argv[:4] + ["64", "512", "3", "4000"][len(argv):]))
model = TM._make_model(nq=nq, npre=npre, nl=nl).train("--eval" not in sys.argv)
TM._zero_dropout(model)
inp_cpu = TM._inputs(npts, 3, "cpu", 1)
gpu0 = copy.deepcopy(model)
keys = ("sem_cls_logits", "center_unnormalized", "size_unnormalized")


def stages(out):
    return [{k: s[k].detach().cpu().clone() for k in keys} for s in out["aux_outputs"] + [out["outputs"]]]


res = {}
combos = [("all on", True, True, True), ("no rowblock", False, True, True), ("no deferred heads", True, False, True),
          ("no parked weight grads", True, True, False), ("all off", False, False, False)]
if "--few" in sys.argv:
    combos = [combos[0], combos[-1]]
for name, rb, dh, dw in combos:
    T._ROWBLOCK, T._DEFER_HEADS = rb, dh
    gpu = copy.deepcopy(gpu0).to("cuda")
    inp = {k: ([t.detach().to("cuda").requires_grad_(t.requires_grad) for t in v] if isinstance(v, list) else v.to("cuda")) for k, v in inp_cpu.items()}
    runtime.defer_weight_grads(dw)
    try:
        out = gpu(inp)
        TM._loss(out).backward()
        if dw:
            runtime.flush_weight_grads()
    finally:
        runtime.defer_weight_grads(False)
    res[name] = stages(out)

# CPU oracle
import vdetr_amd.attention as A  # noqa: E402
import vdetr_amd.pointnet2_utils as PU  # noqa: E402
from conftest import _OracleExt  # noqa: E402
from oracle.attention_oracle import fused_attention_reference  # noqa: E402
import vdetr_amd.box_decode as BD  # noqa: E402
from oracle.box_oracle import decode_boxes_reference  # noqa: E402
import vdetr_amd.add_ln as ALN  # noqa: E402
from oracle import add_ln_oracle  # noqa: E402
A.fused_attention, A.begin_step, A.current_rng = fused_attention_reference, (lambda d: None), (lambda d: None)
PU._ext = _OracleExt()
BD.decode_boxes = decode_boxes_reference
ALN.layer_norm, ALN.add_dropout_layer_norm = add_ln_oracle.layer_norm, add_ln_oracle.add_dropout_layer_norm
out_cpu = model(inp_cpu)
ref = stages(out_cpu)
for name, st in res.items():
    d = [max(float((a[k] - b[k]).abs().max()) for k in keys) for a, b in zip(st, ref)]
    nbad = [int(((a["sem_cls_logits"] - b["sem_cls_logits"]).abs().amax(-1) > 1e-3).sum()) for a, b in zip(st, ref)]
    print(f"{name:24s} vs CPU oracle: max |diff| per stage {['%.2e' % x for x in d]} rows off {nbad}")
