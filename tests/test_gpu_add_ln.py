"""HIP residual-add + dropout + LayerNorm (vdetr_add_ln_{fwd,bwd}_f32) vs the torch oracle (oracle/add_ln_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mods(C, seed, n=2):
    torch.manual_seed(seed)
    out = []
    for _ in range(n):
        ln = torch.nn.LayerNorm(C)
        with torch.no_grad():
            ln.weight.copy_(1 + 0.3 * torch.randn(C))
            ln.bias.copy_(0.2 * torch.randn(C))
        out.append(ln)
    return out


def _compare(got, ref, names, rtol=1e-4):
    for n, a, b in zip(names, got, ref):
        b = b.detach().cpu().numpy()
        np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=rtol, atol=1e-5 * max(np.abs(b).max(), 1e-3), err_msg=n)


@pytest.mark.parametrize("rows,C,dual", [(1024, 256, True), (77, 256, False), (4096, 256, False), (33, 512, True)])
def test_layer_norm_matches_oracle(rows, C, dual):
    from oracle import add_ln_oracle as O
    from vdetr_amd import add_ln as ALN
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, 1, C, generator=g) * 2 + 0.5
    w = [torch.randn(rows, 1, C, generator=g) for _ in range(2)]
    res = {}
    for name, mod, dev in (("hip", ALN, DEV), ("ref", O, "cpu")):
        lns = [m.to(dev) for m in _mods(C, 3)]
        xx = x.to(dev).requires_grad_(True)
        outs = mod.layer_norm(xx, lns[0], lns[1] if dual else None)
        outs = outs if dual else (outs,)
        sum((o * ww.to(dev)).sum() for o, ww in zip(outs, w)).backward()
        res[name] = list(outs) + [xx.grad] + [p.grad for m in lns[:2 if dual else 1] for p in (m.weight, m.bias)]
    names = ["out", "out2"][:len(res["hip"]) - (5 if dual else 3)] + ["dx", "dgamma", "dbeta", "dgamma2", "dbeta2"]
    _compare(res["hip"], res["ref"], names)


@pytest.mark.parametrize("rows,C,p,dual", [(1024, 256, 0.0, True), (1024, 256, 0.1, True), (100, 256, 0.3, False),
                                           (64, 768, 0.1, False)])
def test_add_dropout_layer_norm_matches_oracle(rows, C, p, dual):
    from oracle import add_ln_oracle as O
    from vdetr_amd import add_ln as ALN
    from vdetr_amd import attention as A
    g = torch.Generator().manual_seed(rows + C + int(p * 100))
    x, r = torch.randn(rows, 1, C, generator=g), torch.randn(rows, 1, C, generator=g)
    w = [torch.randn(rows, 1, C, generator=g) for _ in range(3)]
    drop = torch.nn.Dropout(p).train()
    A.reset_rng()
    rng = A.begin_step(torch.device(DEV))
    salt = 1234
    keep = None
    if p > 0:  # the kernel's own keep-mask: x = 0, r = 1 -> y = keep / (1 - p)
        ln0 = _mods(C, 1, 1)[0].to(DEV)
        y0 = ALN.add_dropout_layer_norm(torch.zeros(rows, 1, C, device=DEV), torch.ones(rows, 1, C, device=DEV), drop, ln0,
                                        salt=salt)[0]
        keep = (y0.detach() > 0).float().cpu()
        frac = float(keep.mean())
        assert abs(frac - (1 - p)) < 0.02, frac
        np.testing.assert_allclose(y0.detach().cpu().numpy(), (keep / (1 - p)).numpy(), rtol=1e-4)
    res = {}
    for name, dev in (("hip", DEV), ("ref", "cpu")):
        lns = [m.to(dev) for m in _mods(C, 5)]
        xx, rr = x.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
        if name == "hip":
            outs = ALN.add_dropout_layer_norm(xx, rr, drop, lns[0], lns[1] if dual else None, salt=salt)
        else:
            outs = O.add_dropout_layer_norm(xx, rr, drop, lns[0], lns[1] if dual else None, keep=keep)
        sum((o * ww.to(dev)).sum() for o, ww in zip(outs, w)).backward()  # y, out (, out2) all used
        res[name] = list(outs) + [xx.grad, rr.grad] + [q.grad for m in lns[:2 if dual else 1] for q in (m.weight, m.bias)]
    names = (["y", "out", "out2"] if dual else ["y", "out"]) + ["dx", "dr", "dgamma", "dbeta", "dgamma2", "dbeta2"]
    _compare(res["hip"], res["ref"], names)


def test_add_ln_partial_gradients_and_errors():
    from vdetr_amd import add_ln as ALN
    lns = [m.to(DEV) for m in _mods(256, 9)]
    x = torch.randn(50, 256, device=DEV, requires_grad=True)
    r = torch.randn(50, 256, device=DEV, requires_grad=True)
    y, o1, o2 = ALN.add_dropout_layer_norm(x, r, None, lns[0], lns[1])
    o2.sum().backward()  # only the second output used: NULL d_y / d_out
    assert lns[0].weight.grad.abs().max() == 0 and lns[1].bias.grad.abs().max() > 0
    assert torch.equal(x.grad, r.grad)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ALN.layer_norm(torch.randn(4, 256), torch.nn.LayerNorm(256))
    assert not ALN.supported(torch.nn.LayerNorm(100)) and not ALN.supported(torch.nn.BatchNorm1d(256))
