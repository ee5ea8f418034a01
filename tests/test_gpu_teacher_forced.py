"""Whole-step parity of the FULL configurations, teacher-forced: BASELINE config 5 (4 scenes of 20k points, rotated boxes,
4096 keys, 1024 queries, 8 RPE layers, 9 head stages) and config 2, every decoder stage held to the north_star's 1e-3 ON ITS OWN
INPUTS, with the gradients that reach every stage, the encoder features and every parameter compared as well (see the test for
what a single ReLU gate does to them).

Why teacher forcing: the decoder feeds each stage's boxes and features into the next layer, and a ReLU network amplifies fp32
rounding along that chain — two DEVICE runs of the same step are 7e-4 apart at stage 7 (profiles/r05_diag_repro.txt), so the
free-running comparison (test_gpu_model.py::test_full_config_training_step_vs_cpu_oracle) has to allow for it and its CPU side
takes 9.5 minutes for the full config 5.  Here the TEACHER is the same model as plain torch in fp64 — the oracle's attention
(oracle/attention_oracle.py, RPE through eight F.grid_sample passes as the reference composes it, vdetr_transformer.py:710-731),
box decode and residual blocks, torch's own BatchNorm / Linear / sort — run on the GPU through ATen's kernels, none of this
library's: it finishes in seconds and touches no entry point of libvdetr_hip.so.  It records the input of every decoder layer
(residual stream, box corners, angles, box centre / size for the position MLP) and the proposal order; the STUDENT is the product
model (fused launches, side-stream table gradient, parked weight gradients) whose layers get those inputs.  Both sides cut the
graph at the same points (the layer inputs are constants), so the parameter gradients are comparable term by term.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _torch_backend(m, store):
    """every native entry point of the host modules -> plain torch / the oracle, for any device and dtype (the teacher's side)"""
    import torch.nn.functional as F
    import vdetr_amd.add_ln as ALN
    import vdetr_amd.attention as A
    import vdetr_amd.bn_act as BNA
    import vdetr_amd.box_decode as BD
    import vdetr_amd.helpers as H
    import vdetr_amd.vdetr_transformer as T
    from oracle import add_ln_oracle
    from oracle.attention_oracle import fused_attention_reference
    from oracle.box_oracle import decode_boxes_reference
    def attention(q, k, v, *, table=None, **kw):
        # The tables enter as CONSTANTS on this side: F.grid_sample's backward scatters 8 x B x nQ x nK contributions into 4,000
        # table cells with fp64 atomics — 15 s per layer and scene on this GPU (tools/probes/tf_oracle_probe.py), 8 minutes for
        # config 5 — so the RPE-table MLPs' gradients are not part of this comparison; the table gradient itself is held against
        # the fp64 oracle at this very layer size, boxes / rotated boxes / general vertices, in
        # test_gpu_attention.py::test_full_size_forward_backward_vs_oracle.
        return fused_attention_reference(q, k, v, table=table.detach() if table is not None else None, rpe_impl="grid_sample", **kw)
    m.setattr(A, "fused_attention", attention)
    m.setattr(A, "begin_step", lambda device: None)
    m.setattr(A, "current_rng", lambda device: None)
    m.setattr(BD, "decode_boxes", decode_boxes_reference)
    m.setattr(ALN, "layer_norm", add_ln_oracle.layer_norm)
    m.setattr(ALN, "add_dropout_layer_norm", add_ln_oracle.add_dropout_layer_norm)
    m.setattr(T, "_DEFER_HEADS", False)
    m.setattr(T.TransformerDecoder, "_bn_relu_drop",
              lambda self, x, bns, p, key: F.dropout(F.relu(self._bn_group(x, bns, self.training)), p, self.training))

    def pos_mlp(self, xyz):  # Conv1d -> BatchNorm1d -> ReLU -> Conv1d as torch modules; its input is what the student will be fed
        store["qref"].append(xyz.detach())
        return BNA.run_sequential(self.position_embedding_head, xyz.transpose(1, 2).contiguous())
    m.setattr(H.PositionEmbeddingLearned, "forward", pos_mlp)


def _run(cfg):
    import bench
    import vdetr_amd.helpers as H
    import vdetr_amd.vdetr_transformer as T
    from test_gpu_model import _inputs, _loss, _make_model, _zero_dropout
    from vdetr_amd import runtime
    npts, bs, npre, nq, nl, angle_type, _ = bench.CONFIGS[cfg]
    model = _make_model(nq=nq, npre=npre, nl=nl, angle_type=angle_type).train()
    _zero_dropout(model)
    student = copy.deepcopy(model).to(DEV)
    teacher = copy.deepcopy(model).double().to(DEV)
    teacher.decoder.sort_keys = False  # (attention does not depend on the key order; the Morton launch is this library's)
    inp = _inputs(npts, 3, DEV, bs)
    with torch.no_grad():  # the sampled tokens once, through the product (FPS / gather are bit-exact: test_gpu_pointnet2.py)
        enc_xyz, enc_feat, enc_inds = student.run_encoder(inp)
    feats_s = enc_feat.detach().clone().requires_grad_(True)
    feats_t = enc_feat.detach().double().requires_grad_(True)
    dims = {k: inp[k] for k in ("point_cloud_dims_min", "point_cloud_dims_max")}
    inp_s = dict(dims, enc_xyz=enc_xyz, enc_features=feats_s, enc_inds=enc_inds)
    inp_t = dict({k: v.double() for k, v in dims.items()}, enc_xyz=enc_xyz.double(), enc_features=feats_t, enc_inds=enc_inds)
    for mdl in (student, teacher):
        for i, l in enumerate(mdl.decoder.layers):
            l._tf_index = i
    store = {"tgt": {}, "ref": {}, "ang": {}, "qref": [], "order": None, "pe_calls": 0, "dfeat": {"t": {}, "s": {}}}
    layer_forward = T.GlobalDecoderLayer.forward
    stage_recorded, stage_plain = T.TransformerDecoder._stage_recorded, T.TransformerDecoder.get_proposal_box_predictions_refine

    def watch(side):  # the gradient that reaches a stage's features, on either path through the heads
        def keep(idx, feats):
            if feats.requires_grad:
                feats.register_hook(lambda g: store["dfeat"][side].__setitem__(idx, g.detach().double().cpu()))

        def rec(self, idx, dims_, box_features, *a, **k):
            keep(idx, box_features)
            return stage_recorded(self, idx, dims_, box_features, *a, **k)

        def plain(self, idx, query_xyz, dims_, box_features, **k):
            keep(idx, box_features)
            return stage_plain(self, idx, query_xyz, dims_, box_features, **k)
        return rec, plain
    rank = T._proposal_order
    pos_forward = H.PositionEmbeddingLearned.forward

    # ---- the teacher: records the input of every layer, and is cut there ---------------------------------------------------------
    with pytest.MonkeyPatch.context() as m:
        _torch_backend(m, store)

        def record(self, tgt, memory, reference_point, reference_angle, *a, **k):
            i = self._tf_index
            tgt = tgt.detach()
            self.pre_normed = None  # (norm1 of this input arrives attached to the previous layer's graph: recomputed from the cut input)
            store["tgt"][i], store["ref"][i] = tgt, reference_point.detach()
            store["ang"][i] = reference_angle.detach() if reference_angle is not None else None
            return layer_forward(self, tgt, memory, reference_point, reference_angle, *a, **k)
        m.setattr(T.GlobalDecoderLayer, "forward", record)

        def record_order(objectness, n):
            store["order"] = rank(objectness, n)
            return store["order"]
        m.setattr(T, "_proposal_order", record_order)
        rec, plain = watch("t")
        m.setattr(T.TransformerDecoder, "_stage_recorded", rec)
        m.setattr(T.TransformerDecoder, "get_proposal_box_predictions_refine", plain)
        out_t = teacher(inp_t)
        _loss(out_t).backward()
    assert len(store["tgt"]) == nl - 1 and len(store["qref"]) == nl - 1

    # ---- the student: the product, its layers fed the teacher's inputs -------------------------------------------------------------
    with pytest.MonkeyPatch.context() as m:
        def force(self, tgt, memory, reference_point, reference_angle, *a, **k):
            i = self._tf_index
            self.pre_normed = None  # (norm1 of the previous layer's OWN output came with it: recomputed from the forced input)
            ang = store["ang"][i]
            return layer_forward(self, store["tgt"][i].float(), memory, store["ref"][i].float().contiguous(),
                                 ang.float() if ang is not None else None, *a, **k)
        m.setattr(T.GlobalDecoderLayer, "forward", force)

        def forced_pos(self, xyz):
            k = store["pe_calls"]
            store["pe_calls"] += 1
            return pos_forward(self, store["qref"][k].float().contiguous())
        m.setattr(H.PositionEmbeddingLearned, "forward", forced_pos)

        def pinned_order(objectness, n):  # (stages >= 1 decode relative to the stage-0 proposals: the same token at every rank)
            mine, ref = rank(objectness, n), store["order"]
            store["differ"] = int((mine != ref).sum())
            assert store["differ"] <= 4 * objectness.shape[0], f"the two sides rank {store['differ']} proposals differently"
            return ref
        m.setattr(T, "_proposal_order", pinned_order)
        rec, plain = watch("s")
        m.setattr(T.TransformerDecoder, "_stage_recorded", rec)
        m.setattr(T.TransformerDecoder, "get_proposal_box_predictions_refine", plain)
        runtime.defer_weight_grads(True)
        try:
            out_s = student(inp_s)
            _loss(out_s).backward()
            runtime.flush_weight_grads()
        finally:
            runtime.defer_weight_grads(False)
    torch.cuda.synchronize()
    assert store["pe_calls"] == nl - 1
    return student, teacher, out_s, out_t, feats_s, feats_t, store


@pytest.mark.parametrize("cfg", ["c5", "c2"])
def test_full_config_teacher_forced_vs_fp64_torch(cfg):
    student, teacher, out_s, out_t, feats_s, feats_t, store = _run(cfg)
    stages_s = out_s["aux_outputs"] + [out_s["outputs"]]
    stages_t = out_t["aux_outputs"] + [out_t["outputs"]]
    assert len(stages_s) == len(stages_t) == 9
    keys = ("sem_cls_logits", "center_unnormalized", "size_unnormalized", "box_corners", "angle_continuous")
    worst = (0.0, "")
    for s_, (a, b) in enumerate(zip(stages_s, stages_t)):
        for k in keys:
            ref = b[k].detach().cpu().numpy()
            # 1e-3 relative (BASELINE.json north_star); entries near zero at 1e-4 of the tensor's largest (free-running: 2e-4 .. 4e-4)
            atol = 1e-4 * max(1.0, float(np.abs(ref).max()))
            use = np.abs(a[k].detach().cpu().double().numpy() - ref) / (1e-3 * np.abs(ref) + atol)
            worst = max(worst, (float(use.max()), f"stage {s_} {k}"))
            assert_close(a[k], ref, 1e-3, atol, f"stage {s_} {k} ({store.get('differ', 0)} ranks differed before pinning)")
    print(f"[teacher-forced {cfg}] stages: largest error / tolerance {worst[0]:.3f} ({worst[1]})")
    # ---- gradients.  What is left without the chain's amplification are single ReLU gates of the heads that sit within fp32
    # rounding of zero and open on one side only.  The loss is a plain sum and the heads normalise with batch statistics, so the
    # gradient that reaches a stage's features is small everywhere (BatchNorm's backward removes its mean over the tokens) and ONE
    # flipped gate moves its token's row by up to ~30 % and, through the statistics' terms, the stage's whole gradient by up to
    # ~1 % — and with it every parameter gradient of that layer (tools/probes/diag_teacher_forced.py: config 2, stage 7: one row of
    # 1024 is 27 % off, Frobenius 7.9e-3, the layer's parameters 1.5e-2; stage 6: no row, 3e-5).  Checked accordingly: per stage at
    # most 1 % of the rows more than 1e-2 off and the whole within 2e-2; every parameter within 3e-2, the medians as below
    # (the free-running comparison has to allow 5e-2 and a median of 1e-2).
    dfeat = store["dfeat"]
    assert sorted(dfeat["t"]) == sorted(dfeat["s"]) == list(range(9))
    clean = set()  # stages whose gradient shows no flipped gate
    for idx in range(9):
        a, b = dfeat["s"][idx], dfeat["t"][idx]
        rows = (a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-30)
        off, fro = int((rows > 1e-2).sum()), float((a - b).norm() / b.norm())
        print(f"[teacher-forced {cfg}] stage {idx}: d loss / d features {fro:.2e} (Frobenius), {off} of {rows.numel()} rows more than 1e-2 off")
        assert off <= 0.01 * rows.numel() and fro <= 2e-2, (idx, off, fro)
        if fro <= 1e-3:
            clean.add(idx)
    g, c = feats_s.grad.detach().cpu().double(), feats_t.grad.detach().cpu()
    fro = float((g - c).norm() / c.norm())
    print(f"[teacher-forced {cfg}] d loss / d encoder features: relative Frobenius error {fro:.2e}")
    assert fro <= 5e-3, fro
    tp = dict(teacher.named_parameters())
    bad, rels = [], []
    for n, ps in student.named_parameters():
        pt = tp[n]
        if "cpb_mlps" in n:  # (constants on the teacher's side, see _torch_backend)
            assert ps.grad is not None and torch.isfinite(ps.grad).all() and float(ps.grad.abs().max()) > 0.0, n
            continue
        if pt.grad is None:
            assert ps.grad is None or float(ps.grad.abs().max()) == 0.0, n
            continue
        if ps.grad is None:
            sib = tp.get(n[:-4] + "weight") if n.endswith(".bias") else None
            assert sib is not None and sib.grad is not None and float(pt.grad.abs().max()) <= 1e-6 * float(sib.grad.abs().max()), \
                f"{n}: no gradient on the device"
            continue
        g, c = ps.grad.detach().cpu().double().flatten(), pt.grad.detach().cpu().flatten()
        scale = float(c.norm())
        sib = tp.get(n[:-4] + "weight") if n.endswith(".bias") else None
        if sib is not None and sib.grad is not None:  # (a bias whose gradient is zero in exact arithmetic: on its weight's scale)
            w = sib.grad.detach().cpu()
            scale = max(scale, 1e-2 * float(w.norm()) * (c.numel() / w.numel()) ** 0.5)
        rel = float((g - c).norm()) / max(scale, 1e-30)
        rels.append((rel, n))
        if rel > 3e-2:
            bad.append((n, rel))
    rels.sort()
    above = sum(1 for r, _ in rels if r > 5e-3)
    print(f"[teacher-forced {cfg}] parameter gradients: median rel {rels[len(rels) // 2][0]:.2e}, {above} of {len(rels)} above 5e-3, largest "
          + ", ".join(f"{n} {r:.2e}" for r, n in rels[-3:]))
    assert not bad, "parameter gradients off: " + ", ".join(f"{n}: {e:.2e}" for n, e in bad[:8])
    # How many stages carry a flipped gate is a draw: any change in the order of an fp32 sum upstream redraws it.  Config 5 with the
    # general attention body in the query self-attention: stages 2-7 (Frobenius 1e-3 .. 8e-3), median 4.1e-4, 96 of 551 parameters
    # above 5e-3; with the lean kernel (attn_fwd_self.hip, equal to the body within 1e-5: test_gpu_attention.py): stages 1-8, median
    # 1.5e-3, 109 above 5e-3 — a decoder layer's ~60 parameters move together with its stage.  So the median is held to 1e-3 where it
    # says something about the arithmetic — over the layers whose stage shows NO flip (decoder.layers.i <-> stage i + 1) — and the
    # whole set to a lower quartile of 1e-3 and a median of 3e-3 (a systematic error would lift every layer).
    import re
    def stage_of(name):
        m_ = re.match(r"decoder\.layers\.(\d+)\.", name)
        return int(m_.group(1)) + 1 if m_ else None
    quiet = sorted(r for r, n in rels if stage_of(n) in clean)
    q25, med = rels[len(rels) // 4][0], rels[len(rels) // 2][0]
    print(f"[teacher-forced {cfg}] parameter gradients: lower quartile {q25:.2e}; stages without a flipped gate {sorted(clean)}: "
          + (f"median of their layers' {len(quiet)} parameters {quiet[len(quiet) // 2]:.2e}" if quiet else "no decoder layer among them"))
    assert q25 <= 1e-3 and med <= 3e-3, (q25, med)
    if quiet:
        assert quiet[len(quiet) // 2] <= 1e-3, quiet[len(quiet) // 2]
