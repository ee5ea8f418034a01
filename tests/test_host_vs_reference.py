"""Host logic (module wiring, projections, box decode, top-k, state-dict layout) + the CPU oracle, checked
against vectors produced by the REFERENCE's own Python (tests/golden/, made by oracle/make_golden.py).

Runs without a GPU: the two native entry points are routed to the oracle by the `cpu_oracle_backend` fixture.
Tolerances: the reference ran in fp32 on CPU; so does this.  1e-4 relative / 2e-5 absolute (observed ~1e-6).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import (assert_close, build_cross_attention, build_decoder, build_share_self_attention,
                     run_cross_attention_case, run_decoder_case, t)

RTOL, ATOL = 1e-4, 2e-5


@pytest.mark.parametrize("case", ["cross_attn_small", "cross_attn_rot", "cross_attn_mid"])
def test_cross_attention_module(case, cpu_oracle_backend):
    g = load_golden(case)
    mod = build_cross_attention(str(g["angle_type"]))
    res = run_cross_attention_case(g, mod, "cpu")
    assert_close(res["x"], g["x"], RTOL, ATOL, "x")
    assert_close(res["attn"], g["attn"], RTOL, 1e-7, "attn")
    for k in g.files:
        if k.startswith("grad_"):
            assert_close(res[k], g[k], 5e-4, max(1e-4 * np.abs(g[k]).max(), 2e-6), k)


def test_rpe_bias_oracle_edge_cases():
    """|delta| > 8 m (all corners padded), delta == 0, g slightly > 1: oracle bias == reference grid_sample path.
    The reference's attn = softmax(qk + rpe): here the bias alone is rebuilt from the stored tables."""
    import torch.nn.functional as F
    from oracle.attention_oracle import rpe_bias_reference
    g = load_golden("cross_attn_small")
    tables, ref_pts, xyz = t(g["tables"]), t(g["reference_point"]), t(g["xyz"])
    mine = rpe_bias_reference(tables, ref_pts, xyz)
    # independent statement through F.grid_sample, as the reference composes it (vdetr_transformer.py:722-731)
    B, nQ, nK = ref_pts.shape[0], ref_pts.shape[1], xyz.shape[1]
    rpe = 0
    for i in range(8):
        d = ref_pts[:, :, None, i, :] - xyz[:, None, :, :]
        d = torch.sign(d) * torch.log2(torch.abs(d) * 512.0 + 1.0) / np.log2(8) / 4.0
        tab = tables[i][None].permute(0, 4, 1, 2, 3)
        rpe = rpe + F.grid_sample(tab, d.view(1, 1, 1, -1, 3), mode="bilinear", align_corners=False) \
            .squeeze().view(-1, B, nQ, nK).permute(1, 0, 2, 3)
    assert_close(mine, rpe.numpy(), 1e-5, 1e-6 * float(rpe.abs().max()), "rpe")
    from oracle.attention_oracle import rpe_bias_grid_sample
    assert_close(rpe_bias_grid_sample(tables, ref_pts, xyz), rpe.numpy(), 1e-6, 1e-7, "rpe (timed variant)")
    cs = torch.stack((torch.cos(t(g["reference_angle"]) + 0.7), torch.sin(t(g["reference_angle"]) + 0.7)), -1)
    assert_close(rpe_bias_grid_sample(tables, ref_pts, xyz, cos_sin=cs),
                 rpe_bias_reference(tables, ref_pts, xyz, cos_sin=cs).numpy(), 1e-5, 1e-5, "rpe rotated")
    assert float(mine[:, :, :, 1].abs().max()) == 0.0  # key 1 is > 8 m away on every axis: zero padding everywhere


def test_share_self_attention_module(cpu_oracle_backend):
    g = load_golden("share_self_attn")
    mod = build_share_self_attention()
    tgt, pos = t(g["tgt"], grad=True), t(g["pos"])
    x, _ = mod(tgt + pos, tgt + pos, value=tgt)
    (x * t(g["wout"])).sum().backward()
    assert_close(x, g["x"], RTOL, ATOL, "x")
    assert_close(tgt.grad, g["grad_tgt"], 5e-4, 1e-6, "grad_tgt")
    for pname, p in mod.named_parameters():
        assert_close(p.grad, g["grad_param:" + pname], 5e-4, max(1e-4 * np.abs(g["grad_param:" + pname]).max(), 2e-6), pname)


@pytest.mark.parametrize("case,nl,share", [("decoder_c1_l2", 2, False), ("decoder_c1_l3", 3, False),
                                           ("decoder_c1_l3_share", 3, True)])
def test_decoder(case, nl, share, cpu_oracle_backend):
    """BASELINE config 1 shape (512 tokens, 64 queries, 1-2 RPE layers): outputs of every stage + gradients."""
    g = load_golden(case)
    dec = build_decoder(nl, share)
    assert sorted(n for n, _ in dec.named_parameters()) == list(g["param_names"])  # checkpoint key layout
    assert sorted(n for n, _ in dec.named_buffers()) == list(g["buffer_names"])
    stages, loss, gfeats = run_decoder_case(g, dec, "cpu")
    assert len(stages) == int(g["nstages"])
    for s, st in enumerate(stages):
        for k in ("sem_cls_logits", "center_unnormalized", "size_unnormalized", "box_corners", "center_normalized",
                  "size_normalized", "angle_continuous", "objectness_prob"):
            assert_close(st[k], g[f"s{s}:{k}"], 1e-3, 1e-4, f"stage {s} {k}")
    assert_close(loss, g["loss"], 1e-4, 1e-3, "loss")
    assert_close(gfeats, g["grad_feats"], 2e-3, 1e-4 * np.abs(g["grad_feats"]).max(), "grad_feats")
    params = dict(dec.named_parameters())
    for k in g.files:
        if k.startswith("grad_param:"):
            assert_close(params[k[11:]].grad, g[k], 2e-3, max(2e-4 * np.abs(g[k]).max(), 2e-6), k)


def test_box_corners_and_lidar():
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.vdetr_transformer import convert_corners_camera2lidar
    g = load_golden("box_corners")
    cfg = ScannetDatasetConfig()
    c = cfg.box_parametrization_to_corners(t(g["center"]), t(g["size"]), t(g["angle"]))
    assert_close(c, g["corners"], 1e-5, 1e-6, "corners")
    c0 = cfg.box_parametrization_to_corners(t(g["center"]), t(g["size"]), torch.zeros_like(t(g["angle"])))
    assert_close(c0, g["corners_zero_angle"], 1e-6, 1e-7, "corners0")
    # angle=None: the one-launch form ModelVDETR.forward uses for the (yaw 0) encoder proposals == the general form on zeros
    assert torch.equal(cfg.box_parametrization_to_corners(t(g["center"]), t(g["size"]), None), c0)
    assert_close(convert_corners_camera2lidar(c), g["lidar"], 1e-5, 1e-6, "lidar")


def test_position_embedding():
    from oracle.param_fill import fill_module
    from vdetr_amd.position_embedding import PositionEmbeddingCoordsSine
    g = load_golden("pos_embed")
    xyz, rng = t(g["xyz"]), [t(g["rmin"]), t(g["rmax"])]
    four = fill_module(PositionEmbeddingCoordsSine(d_pos=256, pos_type="fourier", normalize=True))
    sine = PositionEmbeddingCoordsSine(pos_type="sine", normalize=True)
    assert_close(four(xyz, input_range=rng), g["fourier"], 1e-4, 1e-4, "fourier")
    assert_close(four(xyz, num_channels=64, input_range=rng), g["fourier_64"], 1e-4, 1e-4, "fourier64")
    assert_close(sine(xyz, num_channels=256, input_range=rng), g["sine_256"], 1e-4, 1e-4, "sine256")
    assert_close(sine(xyz, num_channels=100, input_range=rng), g["sine_100"], 1e-4, 1e-4, "sine100")
    assert xyz.equal(t(g["xyz"]))  # the input is not modified


def test_decoder_plain_composition_paths(cpu_oracle_backend, monkeypatch):
    """The decoder also has to work where the fused entry points do not apply (other norm types / widths, per-layer K/V):
    with them switched off it takes the reference's plain composition and still reproduces the golden outputs."""
    import vdetr_amd.add_ln as ALN
    import vdetr_amd.vdetr_transformer as VT
    g = load_golden("decoder_c1_l3")
    dec = build_decoder(3, False)
    monkeypatch.setattr(ALN, "supported", lambda *mods: False)             # no fused residual + LayerNorm launches
    monkeypatch.setattr(VT.GlobalShareCrossAttention, "precompute", staticmethod(lambda mods, key: [None] * len(mods)))
    stages, loss, gfeats = run_decoder_case(g, dec, "cpu")
    for s, st in enumerate(stages):
        for k in ("sem_cls_logits", "center_unnormalized", "size_unnormalized", "box_corners"):
            assert_close(st[k], g[f"s{s}:{k}"], 1e-3, 1e-4, f"stage {s} {k}")
    assert_close(loss, g["loss"], 1e-4, 1e-3, "loss")
    assert_close(gfeats, g["grad_feats"], 2e-3, 1e-4 * np.abs(g["grad_feats"]).max(), "grad_feats")


def test_parked_table_gradients_reach_the_mlps_at_the_flush():
    """attention.DeferredTableGrads (host logic of the side-stream table gradient): tables cut from their graph, the layers'
    gradients parked by `_ParkTableGrad`, the tables' own backward run by flush() — the same parameter gradients as the
    plain graph, nothing before the flush, a layer without a gradient counts as zero."""
    from vdetr_amd import attention as A
    torch.manual_seed(0)
    w1, w2 = torch.randn(5, 3, requires_grad=True), torch.randn(7, 5, requires_grad=True)
    coords, wl = torch.randn(3, 11, 3), torch.randn(3, 11, 7)

    def tables():
        return torch.relu(coords @ w1.t()) @ w2.t()  # [3 layers, 11, 7]

    ((tables()[0] * wl[0]).sum() + 2.0 * (tables()[2] * wl[2]).sum()).backward()
    ref = (w1.grad.clone(), w2.grad.clone())
    w1.grad = w2.grad = None
    parts = A.DeferredTableGrads.park(tables())
    assert len(parts) == 3 and all(p.requires_grad for p in parts)
    ((parts[0] * wl[0]).sum() + 2.0 * (parts[2] * wl[2]).sum()).backward()  # layer 1 unused
    assert w1.grad is None and w2.grad is None and len(A.DeferredTableGrads.pending) == 1
    A.DeferredTableGrads.flush()
    assert not A.DeferredTableGrads.pending
    assert torch.allclose(w1.grad, ref[0], rtol=1e-6, atol=1e-6) and torch.allclose(w2.grad, ref[1], rtol=1e-6, atol=1e-6)
    # nothing parked, nothing to do; a pass whose layers received no gradient at all leaves the parameters untouched
    A.DeferredTableGrads.flush()
    A.DeferredTableGrads.park(tables())
    A.DeferredTableGrads.flush()
    assert torch.allclose(w1.grad, ref[0], rtol=1e-6, atol=1e-6)
    # tables parked WITH their MLP's operands (computed without a graph): begin_flush writes the backward out as three GEMMs
    n8, T3, hd, H = 6, 50, 16, 4
    c1 = torch.cat((torch.randn(T3, 3), torch.ones(T3, 1)), 1)
    v1, c_1, v2 = (torch.randn(n8, hd, 3, requires_grad=True), torch.randn(n8, hd, requires_grad=True),
                   torch.randn(n8, H, hd, requires_grad=True))
    wt = torch.randn(3, 2, T3, H)

    def mlp(grad):
        with torch.enable_grad() if grad else torch.no_grad():
            hid = torch.relu(torch.bmm(c1.unsqueeze(0).expand(n8, -1, -1), torch.cat((v1, c_1.unsqueeze(-1)), 2).transpose(1, 2)))
            return torch.bmm(hid, v2.transpose(1, 2)).view(3, 2, T3, H), hid

    (mlp(True)[0] * wt).sum().backward()
    ref2 = [p.grad.clone() for p in (v1, c_1, v2)]
    v1.grad = c_1.grad = v2.grad = None
    tabs, hid = mlp(False)
    parts = A.DeferredTableGrads.park(tabs, mlp=(c1, v1, c_1, v2, hid))
    sum((parts[i] * wt[i]).sum() for i in range(3)).backward()
    assert v1.grad is None
    A.DeferredTableGrads.begin_flush()
    assert not A.DeferredTableGrads.pending and len(A.DeferredTableGrads._begun) == 1
    A.DeferredTableGrads.flush()
    for p, r in zip((v1, c_1, v2), ref2):
        assert torch.allclose(p.grad, r, rtol=1e-5, atol=1e-5)
    # modes: "0" never, "1" always, "auto" from 2^21 query-key pairs on
    prev = A.set_async_table_grad("auto")
    try:
        assert A._async_wanted(1, 1024, 4096) and not A._async_wanted(1, 64, 4096)
        A.set_async_table_grad("1")
        assert A._async_wanted(1, 1, 1)
        A.set_async_table_grad("0")
        assert not A._async_wanted(4, 1024, 4096) and not A.ASYNC_TABLE_GRAD
    finally:
        A.set_async_table_grad(prev)


def test_parked_weight_gradients_can_be_collected_instead_of_delivered():
    """helpers.DeferredParamGrads.flush(collect=, keepalive=, select=): what the side branch uses (attention.
    flush_layer_params_on_side) — the (parameter, gradient) pairs of the selected items come back instead of being written, the
    operands are handed to the caller to keep alive, the other items stay parked, and delivering the pairs afterwards
    (attention.SideResults) gives the gradients of a plain flush."""
    from vdetr_amd import attention as A
    from vdetr_amd.helpers import DeferredParamGrads as D
    torch.manual_seed(1)
    ws = [torch.randn(6, 4, requires_grad=True) for _ in range(3)] + [torch.randn(5, 4, requires_grad=True)]
    bs = [torch.randn(6, requires_grad=True) for _ in range(3)] + [torch.randn(5, requires_grad=True)]
    gs = [torch.randn(10, 6) for _ in range(3)] + [torch.randn(7, 5)]
    xs = [torch.randn(10, 4) for _ in range(3)] + [torch.randn(7, 4)]
    ref_w = [g.t() @ x for g, x in zip(gs, xs)]
    ref_b = [g.sum(0) for g in gs]
    assert not D.pending
    D.pending.extend((w, b, g, x) for w, b, g, x in zip(ws, bs, gs, xs))
    pairs, keep = [], []
    D.flush(select=lambda it: it[2].shape[0] == 10, collect=pairs, keepalive=keep)
    assert len(D.pending) == 1 and D.pending[0][2].shape[0] == 7, "the 7-row item stays parked"
    assert len(keep) == 3 and all(w.grad is None for w in ws), "nothing delivered yet"
    assert len(pairs) == 6
    A.SideResults.pending.append((torch.device("cpu"), pairs, keep))
    A.SideResults.flush()
    D.flush()
    assert not D.pending and not A.SideResults.pending
    for w, b, rw, rb in zip(ws, bs, ref_w, ref_b):
        assert torch.allclose(w.grad, rw, rtol=1e-5, atol=1e-5) and torch.allclose(b.grad, rb, rtol=1e-5, atol=1e-5)
