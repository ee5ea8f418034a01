#!/usr/bin/env python3
"""Where do the teacher-forced parameter-gradient differences of tests/test_gpu_teacher_forced.py come from?  Per stage: the
gradient that reaches the stage's features (d loss / d normed) on the two sides, and the number of hidden units of the stage's
heads whose ReLU gate differs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_teacher_forced as TF  # noqa: E402
import vdetr_amd.vdetr_transformer as T  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
grads = {"s": {}, "t": {}}
side = {"now": "t"}
orig_rec, orig_ref = T.TransformerDecoder._stage_recorded, T.TransformerDecoder.get_proposal_box_predictions_refine


def rec(self, idx, dims, box_features, *a, **k):
    if box_features.requires_grad:
        box_features.register_hook(lambda g, idx=idx: grads[side["now"]].__setitem__(idx, g.detach().double().cpu()))
    return orig_rec(self, idx, dims, box_features, *a, **k)


def ref(self, idx, query_xyz, dims, box_features, **k):
    if box_features.requires_grad:
        box_features.register_hook(lambda g, idx=idx: grads[side["now"]].__setitem__(idx, g.detach().double().cpu()))
    return orig_ref(self, idx, query_xyz, dims, box_features, **k)


T.TransformerDecoder._stage_recorded = rec
T.TransformerDecoder.get_proposal_box_predictions_refine = ref
orig_teacher = TF._torch_backend


def backend(m, store):
    side["now"] = "t"
    return orig_teacher(m, store)


TF._torch_backend = backend
import vdetr_amd.runtime as R  # noqa: E402
orig_defer = R.defer_weight_grads


def defer(flag=True):
    if flag:
        side["now"] = "s"
    return orig_defer(flag)


R.defer_weight_grads = defer
student, teacher, out_s, out_t, fs, ft, store = TF._run(cfg)
for idx in sorted(grads["t"]):
    a, b = grads["s"].get(idx), grads["t"][idx]
    if a is None:
        print(f"stage {idx}: no student gradient recorded")
        continue
    rows = ((a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-30))
    print(f"stage {idx}: d loss / d features rel Frobenius {float((a - b).norm() / b.norm()):.2e}; rows more than 1e-2 off: "
          f"{int((rows > 1e-2).sum())} of {rows.numel()}, worst row {float(rows.max()):.2e}")
