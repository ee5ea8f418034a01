"""The decoder layer's fused glue launches (csrc/rowblock.hip, v-detr_amd/rowblock.py) against the composition of separate
launches they replace (helpers.linear, add_ln.add_dropout_layer_norm, bn_act.relu_dropout): same dropout streams, so values
and gradients must agree to fp32 rounding of a different summation order."""
import numpy as np
import pytest
import torch

from helpers import args_ns

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(a, b, what, rtol=2e-4, frac=2e-5, floor=0.0):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    np.testing.assert_allclose(a, b, rtol=rtol, atol=max(frac * max(np.abs(b).max(), 1e-6), floor), err_msg=what)


def _floor(name, names, tensors):
    """the key projection's bias has a zero gradient in exact arithmetic (every row of dS sums to zero): what is compared there is
    summation noise, whose scale is the weight gradient's (tests/helpers.py: grad_atol)"""
    if name.endswith(".k.bias"):
        w = tensors[names.index(name[:-5] + ".weight")]
        return 1e-4 * float(w.abs().max())
    return 0.0


def _layer(seed=0):
    from oracle.param_fill import fill_module
    from vdetr_amd.vdetr_transformer import GlobalDecoderLayer
    torch.manual_seed(seed)
    layer = GlobalDecoderLayer(d_model=256, nhead=4, dim_feedforward=256, dropout=0.1, pos_for_key=False, args=args_ns())
    fill_module(layer)
    with torch.no_grad():
        for m in layer.multihead_attn.cpb_mlps:
            m[0].weight.mul_(2.0)
            m[2].weight.mul_(1.5)
        for ln in (layer.norm1, layer.norm2, layer.norm3):
            ln.weight.add_(0.2 * torch.randn(256))
            ln.bias.add_(0.1 * torch.randn(256))
    return layer.to(DEV)


def _scene(B, nQ, nK, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand((B, nK, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0]) + 1.0
    centre = xyz[:, torch.randint(0, nK, (nQ,), generator=g)]
    half = 0.1 + torch.rand((B, nQ, 1, 3), generator=g)
    signs = torch.tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]], dtype=torch.float32)
    verts = centre[:, :, None, :] + half * signs
    return xyz.to(DEV), verts.contiguous().to(DEV)


@pytest.mark.parametrize("fused_bwd", [True, False])
@pytest.mark.parametrize("B,nQ,nK,train", [(1, 64, 256, True), (1, 37, 100, True), (2, 40, 130, True), (1, 64, 256, False), (3, 17, 64, True),
                                           (1, 1024, 512, True)])
def test_fused_layer_equals_separate_launches(monkeypatch, B, nQ, nK, train, fused_bwd):
    """GlobalDecoderLayer.forward_pre through rowblock.py (three launches between the attention cores, and three for their
    backward; fused_bwd=False: the separate backward launches behind the fused forward) and through one launch per op: output,
    the two output norms, and the gradients of every input and parameter."""
    from vdetr_amd import attention as A
    from vdetr_amd import rowblock as RB
    from vdetr_amd import vdetr_transformer as T
    monkeypatch.setattr(RB, "FUSED_BWD", fused_bwd)
    layer = _layer(3)
    layer.train(train)
    out_norm, next_norm = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        out_norm.weight.add_(0.1 * torch.randn(256, device=DEV))
        next_norm.bias.add_(0.1 * torch.randn(256, device=DEV))
    g = torch.Generator().manual_seed(B * 100 + nQ)
    tgt0 = torch.randn((nQ, B, 256), generator=g).to(DEV)
    mem0 = torch.randn((nK, B, 256), generator=g).to(DEV)
    pos0 = (0.5 * torch.randn((nQ, B, 256), generator=g)).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 5)
    wts = [torch.randn((nQ, B, 256), generator=g).to(DEV) for _ in range(3)]
    params = list(layer.parameters()) + list(out_norm.parameters()) + list(next_norm.parameters())
    names = [n for n, _ in layer.named_parameters()] + ["out_norm.weight", "out_norm.bias", "next_norm.weight", "next_norm.bias"]

    def run(fused):
        monkeypatch.setattr(T, "_ROWBLOCK", fused)
        A.reset_rng()
        torch.manual_seed(0)
        for p in params:
            p.grad = None
        tgt, mem, pos = (t.clone().requires_grad_(True) for t in (tgt0, mem0, pos0))
        layer.post_norms = (out_norm, next_norm)
        layer.pre_normed = None
        out, _ = layer(tgt, mem, verts, None, xyz, None, query_pos=pos)
        o1, o2 = layer.post_normed
        layer.post_norms = layer.post_normed = None
        ((out * wts[0]).sum() + (o1 * wts[1]).sum() + (o2 * wts[2]).sum()).backward()
        return [out, o1, o2, tgt.grad, mem.grad, pos.grad] + [p.grad for p in params]

    ref = run(False)
    got = run(True)
    again = run(True)
    all_names = ["out", "norm(out)", "next norm1(out)", "d tgt", "d memory", "d query_pos"] + names
    for n, a, b, c in zip(all_names, got, ref, again):
        if b is None:
            assert a is None, n
            continue
        _close(a, b, n, floor=_floor(n, all_names, ref))
        if "cpb_mlps" in n:  # behind the table gradient, whose workgroups draw their queries dynamically: partial sums regroup
            _close(a, c, n + " (second run)", rtol=1e-5, frac=1e-6)
        else:
            assert torch.equal(a, c), f"{n}: not reproducible"


@pytest.mark.parametrize("B,nQ,nK", [(1, 256, 1024), (2, 130, 1100), (1, 1024, 512), (1, 16, 4096), (1, 128, 2048), (1, 64, 256), (1, 64, 200)])
def test_key_split_merge_inside_rb_ffn_equals_the_merge_launch(monkeypatch, B, nQ, nK):
    """The cross-attention forward that leaves the merge of its key-split partials to rb_ffn (vdetr_attn_fwd_parts_f32 ->
    vdetr_rb_ffn_parts_f32: 4, 2, 16 and 8 chunks, and a size that is not split at all) against the forward that ends with the merge
    launch: the same arithmetic in the same order, so the layer's outputs, the attention's saved output / lse (through every
    gradient) are bit-identical."""
    from vdetr_amd import attention as A
    from vdetr_amd import vdetr_transformer as T
    monkeypatch.setattr(T, "_ROWBLOCK", True)
    layer = _layer(5).train()
    out_norm, next_norm = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    g = torch.Generator().manual_seed(B * 1000 + nQ)
    tgt0 = torch.randn((nQ, B, 256), generator=g).to(DEV)
    mem0 = torch.randn((nK, B, 256), generator=g).to(DEV)
    pos0 = (0.5 * torch.randn((nQ, B, 256), generator=g)).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 9)
    wts = [torch.randn((nQ, B, 256), generator=g).to(DEV) for _ in range(3)]
    params = list(layer.parameters()) + list(out_norm.parameters()) + list(next_norm.parameters())
    names = [n for n, _ in layer.named_parameters()] + ["out_norm.weight", "out_norm.bias", "next_norm.weight", "next_norm.bias"]
    taken = []
    real_take = A.take_pending_parts

    def spy(out):
        rec = real_take(out)
        taken.append(None if rec is None else int(rec[0].ksplit))
        return rec
    monkeypatch.setattr(A, "take_pending_parts", spy)

    def run(defer):
        monkeypatch.setattr(A, "DEFER_COMBINE", defer)
        A.reset_rng()
        for p in params:
            p.grad = None
        tgt, mem, pos = (t.clone().requires_grad_(True) for t in (tgt0, mem0, pos0))
        layer.post_norms = (out_norm, next_norm)
        layer.pre_normed = None
        out, _ = layer(tgt, mem, verts, None, xyz, None, query_pos=pos)
        o1, o2 = layer.post_normed
        layer.post_norms = layer.post_normed = None
        ((out * wts[0]).sum() + (o1 * wts[1]).sum() + (o2 * wts[2]).sum()).backward()
        return [out, o1, o2, tgt.grad, mem.grad, pos.grad] + [p.grad for p in params]

    ref = run(False)
    assert taken == [None]
    got = run(True)
    wgs, ntiles = B * ((nQ + 3) // 4), (nK + 15) // 16
    assert taken[1] == {1024: 4, 1100: 4, 512: 4, 4096: 16, 2048: 8, 256: 2, 200: None}[nK], (taken, wgs, ntiles)
    assert not A._pending_parts
    for n, a, b in zip(["out", "norm(out)", "next norm1(out)", "d tgt", "d memory", "d query_pos"] + names, got, ref):
        if "cpb_mlps" in n:  # (behind the table gradient's dynamically grouped partial sums)
            _close(a, b, n, rtol=1e-5, frac=1e-6)
        else:
            assert torch.equal(a, b), f"{n}: max |diff| {float((a - b).abs().max())}"


@pytest.mark.parametrize("B,nQ,nK", [(1, 64, 256), (1, 1024, 512), (2, 48, 130), (3, 16, 64)])
def test_position_mlp_inside_rb_qkv_equals_its_own_launch(monkeypatch, B, nQ, nK):
    """The layer whose learned query position is computed by its q / k / v launch (vdetr_rb_qkv_pos_f32, heads.lazy_pos) against
    the layer behind the position MLP's own launch: outputs, the MLP's saved tensors and running statistics, every gradient
    (the MLP's through the flush).  The product's summation order differs between the two kernels: fp32 rounding apart."""
    from vdetr_amd import attention as A
    from vdetr_amd import heads as HD
    from vdetr_amd import vdetr_transformer as T
    from vdetr_amd.helpers import PositionEmbeddingLearned
    from vdetr_amd.runtime import defer_weight_grads, flush_weight_grads
    monkeypatch.setattr(T, "_ROWBLOCK", True)
    layer = _layer(7).train()
    torch.manual_seed(11)
    posm = PositionEmbeddingLearned(6, 256).to(DEV).train()
    with torch.no_grad():
        posm.position_embedding_head[1].weight.uniform_(0.5, 1.5)
        posm.position_embedding_head[1].bias.normal_(0, 0.3)
    out_norm = torch.nn.LayerNorm(256).to(DEV)
    g = torch.Generator().manual_seed(B * 77 + nQ)
    tgt0 = torch.randn((nQ, B, 256), generator=g).to(DEV)
    mem0 = torch.randn((nK, B, 256), generator=g).to(DEV)
    boxes = torch.cat((torch.rand((B, nQ, 3), generator=g) * torch.tensor([8.0, 6.0, 0.5]) + 1.0, torch.rand((B, nQ, 3), generator=g) + 0.1), -1).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 12)
    wts = [torch.randn((nQ, B, 256), generator=g).to(DEV) for _ in range(2)]
    params = list(layer.parameters()) + list(out_norm.parameters()) + list(posm.parameters())
    names = [n for n, _ in layer.named_parameters()] + ["out_norm.weight", "out_norm.bias"] + ["pos." + n for n, _ in posm.named_parameters()]
    bn = posm.position_embedding_head[1]
    stats0 = (bn.running_mean.clone(), bn.running_var.clone())
    launches = []
    real = HD.take_pending_pos
    monkeypatch.setattr(HD, "take_pending_pos", lambda pos: launches.append(real(pos)) or launches[-1])

    def run(lazy):
        A.reset_rng()
        with torch.no_grad():
            bn.running_mean.copy_(stats0[0]); bn.running_var.copy_(stats0[1]); bn.num_batches_tracked.zero_()
        for p in params:
            p.grad = None
        tgt, mem = (t.clone().requires_grad_(True) for t in (tgt0, mem0))
        defer_weight_grads(True)
        try:
            prev = HD.lazy_pos(lazy)
            try:
                pos = posm(boxes).permute(2, 0, 1)
            finally:
                HD.lazy_pos(prev)
            assert bool(HD._pending_pos) == lazy
            layer.post_norms = (out_norm,)
            layer.pre_normed = None
            out, _ = layer(tgt, mem, verts, None, xyz, None, query_pos=pos)
            (o1,) = layer.post_normed
            layer.post_norms = layer.post_normed = None
            assert not HD._pending_pos
            ((out * wts[0]).sum() + (o1 * wts[1]).sum()).backward()
            flush_weight_grads()
        finally:
            defer_weight_grads(False)
        return [out, o1, pos.detach().clone(), bn.running_mean.clone(), bn.running_var.clone(), tgt.grad, mem.grad] + [p.grad for p in params]

    ref = run(False)
    got = run(True)
    assert [r is not None for r in launches] == [False, True], launches
    assert int(bn.num_batches_tracked) == 1
    all_names = ["out", "norm(out)", "query_pos", "running_mean", "running_var", "d tgt", "d memory"] + names
    for n, a, b in zip(all_names, got, ref):
        if b is None:
            assert a is None or n.endswith("head.0.bias"), n
            continue
        if n.endswith("position_embedding_head.0.bias"):  # cancels under batch statistics: rounding noise on both sides
            continue
        _close(a, b, n, floor=_floor(n, all_names, ref))


def test_a_position_mlp_left_to_rb_qkv_is_launched_by_a_layer_that_takes_another_path(monkeypatch):
    """heads.lazy_pos(True) and then a layer that does NOT run through rowblock.py: the layer launches the position MLP itself
    (heads.materialize_pos) — same values as without the switch; a pending one that nothing consumed is an error at the next forward"""
    from vdetr_amd import attention as A
    from vdetr_amd import heads as HD
    from vdetr_amd import vdetr_transformer as T
    from vdetr_amd.helpers import PositionEmbeddingLearned
    from vdetr_amd.runtime import defer_weight_grads
    monkeypatch.setattr(T, "_ROWBLOCK", False)
    layer = _layer(8).train()
    torch.manual_seed(12)
    posm = PositionEmbeddingLearned(6, 256).to(DEV).train()
    B, nQ, nK = 1, 48, 128
    g = torch.Generator().manual_seed(5)
    tgt0, mem0 = (torch.randn(s, generator=g).to(DEV) for s in ((nQ, B, 256), (nK, B, 256)))
    boxes = (torch.rand((B, nQ, 6), generator=g) + 0.2).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 13)
    outs = []
    defer_weight_grads(True)
    try:
        for lazy in (False, True):
            A.reset_rng()
            torch.manual_seed(0)
            prev = HD.lazy_pos(lazy)
            pos = posm(boxes).permute(2, 0, 1)
            HD.lazy_pos(prev)
            out, _ = layer(tgt0.clone().requires_grad_(True), mem0, verts, None, xyz, None, query_pos=pos)
            assert not HD._pending_pos
            outs.append((out.detach().clone(), pos.detach().clone()))
        assert torch.equal(outs[0][1], outs[1][1]), "query_pos"
        assert torch.equal(outs[0][0], outs[1][0]), "layer output"
        HD.lazy_pos(True)
        posm(boxes)
        HD.lazy_pos(False)
        with pytest.raises(RuntimeError, match="never ran"):
            HD.no_pending_pos("test")
    finally:
        defer_weight_grads(False)
        from vdetr_amd.helpers import DeferredPosEmbedGrads
        DeferredPosEmbedGrads.pending.clear()


@pytest.mark.parametrize("nQ,nK,train", [(64, 256, True), (1024, 512, True), (96, 200, False)])
def test_key_side_pass_without_its_packing_launch(monkeypatch, nQ, nK, train):
    """The layer's backward with the operands of both key-side passes left behind by rb_ffn_bwd / rb_proj_q_bwd (+ the step's one prep
    launch) against the backward whose passes pack their own operands: every gradient; and that the packed path is the one that ran.
    (delta is summed in another order: fp32 rounding apart.)"""
    from vdetr_amd import attention as A
    from vdetr_amd import vdetr_transformer as T
    monkeypatch.setattr(T, "_ROWBLOCK", True)
    B = 1
    layer = _layer(9).train(train)
    out_norm, next_norm = torch.nn.LayerNorm(256).to(DEV), torch.nn.LayerNorm(256).to(DEV)
    g = torch.Generator().manual_seed(nQ)
    tgt0 = torch.randn((nQ, B, 256), generator=g).to(DEV)
    mem0 = torch.randn((nK, B, 256), generator=g).to(DEV)
    pos0 = (0.5 * torch.randn((nQ, B, 256), generator=g)).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 15)
    wts = [torch.randn((nQ, B, 256), generator=g).to(DEV) for _ in range(3)]
    params = list(layer.parameters()) + list(out_norm.parameters()) + list(next_norm.parameters())
    names = [n for n, _ in layer.named_parameters()] + ["out_norm.weight", "out_norm.bias", "next_norm.weight", "next_norm.bias"]
    emitted = []

    def run(prepack):
        monkeypatch.setattr(A, "KV_PREPACK", prepack)
        A.reset_rng()
        A._kv_recs.clear()
        A._kv_by_out.clear()
        for p in params:
            p.grad = None
        tgt, mem, pos = (t.clone().requires_grad_(True) for t in (tgt0, mem0, pos0))
        layer.post_norms = (out_norm, next_norm)
        layer.pre_normed = None
        out, _ = layer(tgt, mem, verts, None, xyz, None, query_pos=pos)
        o1, o2 = layer.post_normed
        layer.post_norms = layer.post_normed = None
        recs = [r for lst in A._kv_recs.values() for r in lst]
        ((out * wts[0]).sum() + (o1 * wts[1]).sum() + (o2 * wts[2]).sum()).backward()
        emitted.append([(r.kind, r.prepared, r.emitted is not None) for r in recs])
        return [tgt.grad, mem.grad, pos.grad] + [p.grad for p in params]

    ref = run(False)
    got = run(True)
    assert emitted[0] == [] and len(emitted[1]) == 2 and all(p and e for _, p, e in emitted[1]), emitted
    all_names = ["d tgt", "d memory", "d query_pos"] + names
    for n, a, b in zip(all_names, got, ref):
        _close(a, b, n, rtol=2e-5, frac=2e-6, floor=_floor(n, all_names, ref))


def test_a_deferred_merge_that_nobody_takes_is_an_error():
    """fused_attention(defer_combine=True) whose output does not reach rowblock.ffn: the next step (and the next deferred forward)
    refuse to go on instead of letting somebody read an unwritten tensor"""
    from vdetr_amd import attention as A
    B, nQ, nK = 1, 256, 1024
    g = torch.Generator().manual_seed(1)
    q = torch.randn((B, nQ, 256), generator=g).to(DEV)
    k, v = (torch.randn((B, nK, 64), generator=g).to(DEV) for _ in range(2))
    table = (0.1 * torch.randn((8, 10, 10, 10, 4), generator=g)).to(DEV)
    xyz, verts = _scene(B, nQ, nK, 2)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, table=table, rpe=A.RPEConfig(), vertices=verts, xyz=xyz)
    ref = A.fused_attention(q, k, v, **kw)
    out = A.fused_attention(q, k, v, defer_combine=True, **kw)
    assert A._pending_parts, "the forward at this size splits the keys"
    with pytest.raises(RuntimeError, match="never ran"):
        A.begin_step(q.device)
    assert not A._pending_parts
    del out, ref


def test_fused_layer_parks_weight_gradients(monkeypatch):
    """with runtime.defer_weight_grads() the fused launches park their weight / bias / LayerNorm gradients like the separate
    ones: after the flush every parameter holds the same gradient as without parking"""
    from vdetr_amd import attention as A
    from vdetr_amd import runtime
    from vdetr_amd import vdetr_transformer as T
    monkeypatch.setattr(T, "_ROWBLOCK", True)
    layer = _layer(4).train()
    out_norm = torch.nn.LayerNorm(256).to(DEV)
    B, nQ, nK = 1, 48, 128
    g = torch.Generator().manual_seed(7)
    tgt0, mem0, pos0 = (torch.randn(s, generator=g).to(DEV) for s in ((nQ, B, 256), (nK, B, 256), (nQ, B, 256)))
    xyz, verts = _scene(B, nQ, nK, 6)
    params = list(layer.parameters()) + list(out_norm.parameters())

    def run(defer):
        runtime.defer_weight_grads(defer)
        try:
            A.reset_rng()
            for p in params:
                p.grad = None
            tgt = tgt0.clone().requires_grad_(True)
            layer.post_norms = (out_norm,)
            out, _ = layer(tgt, mem0, verts, None, xyz, None, query_pos=pos0)
            (o1,) = layer.post_normed
            layer.post_norms = layer.post_normed = None
            (out.sum() + (o1 * o1).sum()).backward()
            if defer:
                runtime.flush_weight_grads()
            return [tgt.grad] + [p.grad.clone() if p.grad is not None else None for p in params]
        finally:
            runtime.defer_weight_grads(False)

    ref, got = run(False), run(True)
    all_names = ["d tgt"] + [n for n, _ in layer.named_parameters()] + ["out_norm.weight", "out_norm.bias"]
    for n, a, b in zip(all_names, got, ref):
        if b is None:
            assert a is None, n
        else:
            _close(a, b, n, floor=_floor(n, all_names, ref))


@pytest.mark.parametrize("n,B,train", [(4096, 1, True), (512, 2, True), (100, 3, True), (64, 1, False)])
def test_first_layer_ffn_as_one_launch_equals_its_composition(monkeypatch, n, B, train):
    """FFNLayer.forward_pre + the decoder's norm on its output through vdetr_rb_ffn0_f32 / _bwd_f32 (one launch each) against the five /
    seven launches they replace: output, normed output, the input's and every parameter's gradient (same dropout streams)."""
    from vdetr_amd import attention as A
    from vdetr_amd import rowblock as RB
    from vdetr_amd.vdetr_transformer import FFNLayer
    from oracle.param_fill import fill_module
    torch.manual_seed(3)
    layer = FFNLayer(256, dim_feedforward=256, dropout=0.1)
    fill_module(layer)
    layer = layer.to(DEV).train(train)
    post = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        post.weight.add_(0.2 * torch.randn(256, device=DEV)); post.bias.add_(0.1 * torch.randn(256, device=DEV))
        layer.norm.weight.add_(0.2 * torch.randn(256, device=DEV)); layer.norm.bias.add_(0.1 * torch.randn(256, device=DEV))
    g = torch.Generator().manual_seed(n + B)
    x0 = torch.randn((n, B, 256), generator=g).to(DEV)
    wts = [torch.randn((n, B, 256), generator=g).to(DEV) for _ in range(2)]
    params = list(layer.parameters()) + list(post.parameters())
    names = [k for k, _ in layer.named_parameters()] + ["post.weight", "post.bias"]

    def run(fused):
        monkeypatch.setattr(RB, "FUSED_FFN0", fused)
        A.reset_rng()
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        layer.post_norm = post
        out = layer(x)
        normed = layer.post_normed
        layer.post_norm = layer.post_normed = None
        ((out * wts[0]).sum() + (normed * wts[1]).sum()).backward()
        return [out, normed, x.grad] + [p.grad for p in params]

    ref = run(False)
    got = run(True)
    for k, a, b in zip(["out", "norm(out)", "d x"] + names, got, ref):
        _close(a, b, k)
    assert torch.equal(run(True)[0], got[0])
