"""csrc/heads.hip through the C-ABI: the five box heads of a stage as three launches and the learned query-position MLP as one,
against the one-launch-per-op path (library GEMMs + bn_act.hip) they replace — same dropout streams, same records for the
batched backward — and against torch's own modules in fp64.

Reference: models/vdetr_transformer.py:244-285 (five GenericMLPs per stage), models/helpers.py:74-141 (GenericMLP), :17-33
(PositionEmbeddingLearned).  Tolerance: 1e-3 relative is the north_star's bound for boxes / logits; two fp32 evaluation orders of
the same chain agree far better, which is what is asserted (2e-5 of the tensor's scale).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(nq, npre, nl=3, angle_type="", seed=0):
    from test_gpu_model import _make_model
    from vdetr_amd.dist import FlatParams
    model = _make_model(nq=nq, npre=npre, nl=nl, angle_type=angle_type, seed=seed).to(DEV).train()
    params = [p for p in model.parameters() if p.requires_grad]
    flat = FlatParams(params, groups=model.flat_param_groups())  # the heads' parameters adjacent: the batched / fused layout
    return model, flat


def _close(a, b, tol, what):
    scale = float(b.abs().max()) + 1e-20
    err = float((a.double() - b.double()).abs().max())
    assert err <= tol * scale, f"{what}: max |diff| {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,N,p", [(1, 1024, 0.3), (1, 4096, 0.3), (2, 128, 0.3), (4, 96, 0.0), (1, 32, 0.3)])
def test_fused_heads_equal_the_batched_path(B, N, p):
    """vdetr_heads_fwd_f32 vs the launches it replaces (mm / bmm + vdetr_bn_act_fwd_f32), on the same features, parameters, salts
    and generator state: hidden activations (zero patterns of the dropout masks identical), saved statistics, outputs, running
    statistics, num_batches_tracked."""
    from vdetr_amd import attention as A
    from vdetr_amd import heads as HD
    model, _flat = _model(64, 512)
    dec = model.decoder
    heads = dec.mlp_heads[1]
    names = dec._HEAD_NAMES
    for n in names:
        for i in (3, 7):
            heads[n].layers[i].p = p
    g = torch.Generator().manual_seed(N + B)
    seq = (torch.randn((N, B, 256), generator=g) * 1.5 + 0.2).to(DEV)
    with torch.no_grad():  # BatchNorm parameters and statistics that are not the initial 1 / 0
        for n in names:
            for i in (1, 5):
                bn = heads[n].layers[i]
                bn.weight.copy_(torch.rand(256, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(256, generator=g) * 0.3)
    for i in (1, 5):  # (the first training-mode call lays a group's running statistics out adjacently)
        dec._bn_group(seq.permute(1, 2, 0).repeat(1, 5, 1), [heads[n].layers[i] for n in names], True)
    state = {k: v.clone() for k, v in heads.state_dict().items()}
    res = {}
    keep = HD.FUSED
    try:
        for fused in (True, False):
            HD.FUSED = fused
            with torch.no_grad():
                for k, v in heads.state_dict().items():
                    v.copy_(state[k])
            A.reset_rng()
            torch.manual_seed(0)
            A.begin_step(DEV)
            feats = seq.permute(1, 2, 0)
            assert dec._heads_recordable(heads, feats)
            y, chans, rec = dec._run_heads_recorded(heads, feats, None, seq)
            torch.cuda.synchronize()
            res[fused] = (y.clone(), {k: rec[k].clone() for k in ("h1", "h2")}, [t.clone() for t in rec["bn1"][:5]],
                          [t.clone() for t in rec["bn2"][:5]], {k: v.clone() for k, v in heads.state_dict().items()})
            assert rec["bn1"][6][2] == p and (rec["bn1"][5] is None) == (p == 0)
    finally:
        HD.FUSED = keep
    (yf, hf, b1f, b2f, sf), (yu, hu, b1u, b2u, su) = res[True], res[False]
    assert HD.heads_usable(seq, [heads[n].layers for n in names], 18)
    _close(b1f[0], b1u[0], 2e-5, "pre1")
    _close(b1f[3], b1u[3], 2e-5, "save_mean1")
    _close(b1f[4], b1u[4], 2e-5, "save_invstd1")
    # the dropout masks are the same streams: an element is zero on one side only where relu(.) sits within rounding of zero
    for k in ("h1", "h2"):
        differ = ((hf[k] == 0) != (hu[k] == 0))
        assert int(differ.sum()) <= 1e-4 * hf[k].numel(), (k, int(differ.sum()))
        assert float(torch.where(differ, (hf[k] - hu[k]).abs(), torch.zeros_like(hf[k])).max()) <= 1e-4
        _close(hf[k], hu[k], 5e-5, k)
        if p > 0:
            assert 0.25 < float((hf[k] == 0).float().mean()) < 0.9   # relu and a 0.3 dropout both leave their zeros
    _close(b2f[0], b2u[0], 5e-5, "pre2")
    _close(b2f[3], b2u[3], 5e-5, "save_mean2")
    _close(b2f[4], b2u[4], 5e-5, "save_invstd2")
    _close(yf, yu, 5e-5, "y")
    for k in sf:
        if "num_batches_tracked" in k:
            assert torch.equal(sf[k], su[k]) and int(sf[k]) == int(state[k]) + 1, k
        elif "running" in k:
            _close(sf[k], su[k], 2e-5, k)


def test_fused_heads_vs_torch_modules_fp64():
    """the three launches against the five GenericMLPs as torch modules in fp64 (dropout 0: the modules draw other masks)"""
    import copy
    from vdetr_amd import attention as A
    model, _flat = _model(64, 512)
    dec = model.decoder
    heads = dec.mlp_heads[2]
    names = dec._HEAD_NAMES
    for n in names:
        for i in (3, 7):
            heads[n].layers[i].p = 0.0
    g = torch.Generator().manual_seed(7)
    B, N = 2, 256
    seq = torch.randn((N, B, 256), generator=g).to(DEV)
    feats = seq.permute(1, 2, 0)
    dec._bn_group(feats.repeat(1, 5, 1), [heads[n].layers[1] for n in names], True)  # (lays the running statistics out adjacently)
    dec._bn_group(feats.repeat(1, 5, 1), [heads[n].layers[5] for n in names], True)
    ref = copy.deepcopy(heads).double().train()
    A.reset_rng()
    y, chans, rec = dec._run_heads_recorded(heads, feats, None, seq)
    want = {n: ref[n](feats.double()) for n in names}
    for gi, n in enumerate(names):
        _close(y[:, gi, :chans[gi]], want[n], 2e-5, n)
        for i in (1, 5):
            _close(heads[n].layers[i].running_mean, ref[n].layers[i].running_mean, 2e-5, f"{n} running_mean {i}")
            _close(heads[n].layers[i].running_var, ref[n].layers[i].running_var, 2e-5, f"{n} running_var {i}")


@pytest.mark.parametrize("B,N", [(1, 1024), (2, 128), (4, 1024), (1, 16)])
def test_fused_position_mlp_equals_the_module(B, N):
    """vdetr_pos_mlp_fwd_f32 (batch statistics from the coordinates' fp64 mean and covariance) vs Conv1d -> BatchNorm1d -> ReLU ->
    Conv1d in fp64: output, running statistics, and — through helpers.DeferredPosEmbedGrads — every parameter gradient."""
    import copy
    from vdetr_amd import heads as HD
    from vdetr_amd.helpers import PositionEmbeddingLearned
    from vdetr_amd.runtime import defer_weight_grads, flush_weight_grads
    torch.manual_seed(3)
    mod = PositionEmbeddingLearned(6, 256).to(DEV).train()
    with torch.no_grad():
        mod.position_embedding_head[1].weight.uniform_(0.5, 1.5)
        mod.position_embedding_head[1].bias.normal_(0, 0.3)
    ref = copy.deepcopy(mod).double().cpu()  # (on the CPU: the module's plain torch composition)
    g = torch.Generator().manual_seed(N)
    # box centres a few metres from the origin with a small spread along z (the case a one-pass variance would lose digits on), sizes
    xyz = torch.cat((torch.rand((B, N, 3), generator=g) * torch.tensor([8.0, 6.0, 0.05]) + torch.tensor([1.0, 1.0, 1.5]),
                     torch.rand((B, N, 3), generator=g) * 2 + 0.05), -1).to(DEV)
    wout = torch.randn((B, 256, N), generator=g).to(DEV)
    assert HD.pos_mlp_usable(mod, xyz)
    defer_weight_grads(True)
    try:
        out = mod(xyz)
        (out * wout).sum().backward()
        flush_weight_grads()
    finally:
        defer_weight_grads(False)
    want = ref(xyz.double().cpu())
    (want * wout.double().cpu()).sum().backward()
    _close(out.cpu(), want.detach(), 2e-5, "out")
    bn, rbn = mod.position_embedding_head[1], ref.position_embedding_head[1]
    _close(bn.running_mean.cpu(), rbn.running_mean, 2e-5, "running_mean")
    _close(bn.running_var.cpu(), rbn.running_var, 2e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        if n == "position_embedding_head.0.bias":  # cancels under batch statistics: exactly zero here, rounding noise in the module
            assert p.grad is None or float(p.grad.abs().max()) <= 1e-3 * float(mod.position_embedding_head[0].weight.grad.abs().max())
            continue
        _close(p.grad.cpu(), q.grad, 2e-4, n)


def test_whole_step_with_fused_heads_equals_the_unfused_step(monkeypatch):
    """a training step of the model (4 stages, dropout on) with the heads / position MLPs fused and unfused: losses, every
    gradient, the BatchNorm buffers"""
    from helpers import fixed_salts
    from test_gpu_model import _inputs, _loss
    fixed_salts(monkeypatch)  # (the same dropout streams whatever ran before in this process)
    from vdetr_amd import attention as A
    from vdetr_amd import heads as HD
    from vdetr_amd.runtime import defer_weight_grads, flush_weight_grads
    model, _flat = _model(64, 512, nl=4)
    inp = _inputs(3000, 5, DEV, 2)
    defer_weight_grads(True)
    keep = HD.FUSED, HD.FUSED_POS
    try:
        _loss(model(inp)).backward()
        flush_weight_grads()
        state = {k: v.clone() for k, v in model.state_dict().items()}
        res = {}
        for fused in (True, False):
            HD.FUSED = HD.FUSED_POS = fused
            A.reset_rng()
            torch.manual_seed(0)
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(state[k])
            model.zero_grad(set_to_none=True)
            for f in inp["backbone_features"]:
                f.grad = None
            out = model(inp)
            loss = _loss(out) + sum((o["box_corners"] ** 2).sum() + o["angle_logits"].sum() for o in out["aux_outputs"])
            loss.backward()
            flush_weight_grads()
            torch.cuda.synchronize()
            res[fused] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                          [f.grad.clone() for f in inp["backbone_features"]],
                          {n: b.clone() for n, b in model.named_buffers() if "running" in n or "num_batches" in n})
    finally:
        HD.FUSED, HD.FUSED_POS = keep
        defer_weight_grads(False)
    (l1, g1, f1, b1), (l0, g0, f0, b0) = res[True], res[False]
    assert abs(l1 - l0) <= 2e-5 * abs(l0)
    assert g1.keys() == g0.keys()
    # Two evaluation orders of the same fp32 chain: ~1e-5 apart, EXCEPT where a ReLU gate of the heads sits within rounding of zero
    # and opens on one side only — that moves the gradients of its stage's parameters by up to ~1 % (tests/test_gpu_teacher_forced.py
    # measures and explains it; seen here: one stage-0 weight 2.2e-3 off, depending on which GEMM kernels the process tuned before).
    # Hence: every parameter within 2e-2 (relative Frobenius), the median within 1e-4, the feature gradient within 5e-3.
    rels = []
    for n in g0:
        scale = float(g0[n].norm())
        if n.endswith(".bias") and n[:-4] + "weight" in g0:   # a bias in front of batch statistics: zero up to summation noise
            w = g0[n[:-4] + "weight"]
            scale = max(scale, 1e-2 * float(w.norm()) * (g0[n].numel() / w.numel()) ** 0.5)
        rel = float((g1[n] - g0[n]).norm()) / (scale + 1e-30)
        rels.append(rel)
        assert rel <= 2e-2, (n, rel)
    rels.sort()
    assert rels[len(rels) // 2] <= 1e-4, rels[len(rels) // 2]
    for a, b in zip(f1, f0):
        assert float((a - b).norm() / b.norm()) <= 5e-3
    for n in b0:
        assert torch.allclose(b1[n].float(), b0[n].float(), rtol=2e-5, atol=1e-6), n


def test_shapes_outside_the_fused_launches_take_the_batched_path():
    from vdetr_amd import heads as HD
    model, _flat = _model(64, 512)
    dec = model.decoder
    L5 = [dec.mlp_heads[1][n].layers for n in dec._HEAD_NAMES]
    assert HD.heads_usable(torch.zeros((64, 1, 256), device=DEV), L5, 18)
    assert not HD.heads_usable(torch.zeros((40, 1, 256), device=DEV), L5, 18)          # tokens not a multiple of 32
    assert not HD.heads_usable(torch.zeros((64, 1, 256), device=DEV)[:, :, :], L5, 40)  # a slab higher than 32 rows
    assert not HD.heads_usable(torch.zeros((64, 2, 256), device=DEV).transpose(0, 1), L5, 18)  # not dense sequence-first
    with pytest.raises(RuntimeError, match="CPU|GPU|cuda"):
        HD.refresh([torch.zeros(256, 256)])
