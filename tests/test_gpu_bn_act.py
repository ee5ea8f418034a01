"""HIP BatchNorm1d + ReLU + Dropout (vdetr_bn_act_{fwd,bwd}_f32) vs the torch oracle (oracle/bn_act_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,C,N,p,relu", [(1, 1280, 1024, 0.3, True), (1, 288, 1024, 0.0, True), (3, 70, 333, 0.1, True),
                                          (4, 96, 1024, 0.3, True), (2, 40, 256, 0.1, True),
                                          (2, 64, 4096, 0.0, False)])
def test_bn_act_training_matches_oracle(B, C, N, p, relu):
    from oracle.bn_act_oracle import bn_act as ref_fn
    from vdetr_amd import attention as A
    from vdetr_amd.bn_act import bn_act
    g = torch.Generator().manual_seed(B * 7 + C + N)
    x = torch.randn(B, C, N, generator=g) * 2 + 0.3
    w, b = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    wout = torch.randn(B, C, N, generator=g)
    A.reset_rng()
    A.begin_step(torch.device(DEV))
    salt = 77
    keep = None
    if p > 0:  # the kernel's own keep-mask: relu(bn(x)) > 0 everywhere for x = ramp with gamma=1, beta=10
        ones_w, big_b = torch.ones(C, device=DEV), torch.full((C,), 10.0, device=DEV)
        y0 = bn_act(x.to(DEV), ones_w, big_b, None, None, True, 1e-5, 0.1, relu=True, dropout_p=p, salt=salt)
        keep = (y0 > 0).float().cpu()
        assert abs(float(keep.mean()) - (1 - p)) < 0.02
    res = {}
    for name, dev in (("hip", DEV), ("ref", "cpu")):
        xx = x.to(dev).requires_grad_(True)
        ww, bb = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        rm, rv = rm0.clone().to(dev), rv0.clone().to(dev)
        if name == "hip":
            y = bn_act(xx, ww, bb, rm, rv, True, 1e-5, 0.1, relu=relu, dropout_p=p, salt=salt)
        else:
            y = ref_fn(xx, ww, bb, rm, rv, True, 1e-5, 0.1, relu=relu, dropout_p=p, keep=keep)
        (y * wout.to(dev)).sum().backward()
        res[name] = [y.detach(), xx.grad, ww.grad, bb.grad, rm, rv]
    for n, a, r in zip(["y", "dx", "dgamma", "dbeta", "running_mean", "running_var"], res["hip"], res["ref"]):
        r = r.cpu().numpy()
        np.testing.assert_allclose(a.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(np.abs(r).max(), 1e-3), err_msg=n)


def test_bn_act_eval_and_errors():
    from vdetr_amd.bn_act import bn_act
    C = 64
    x = torch.randn(2, C, 100, device=DEV)
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rm, rv = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    y = bn_act(x, w, b, rm.clone(), rv.clone(), False, 1e-5, 0.1, relu=True)
    ref = torch.relu(torch.nn.functional.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5))
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        bn_act(x.cpu(), w.cpu(), b.cpu(), rm.cpu(), rv.cpu(), True, 1e-5, 0.1)


def test_bn_act_pre_bias_and_counters():
    """A convolution bias in front of a batch-statistics BatchNorm is left out of x: same output, the running mean
    accounts for it; num_batches_tracked counters are incremented by the launch."""
    from oracle.bn_act_oracle import bn_act as ref_fn
    from vdetr_amd.bn_act import bn_act
    B, C, N = 1, 288, 1024
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, N, generator=g)
    w, b, cb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.randn(C, generator=g) * 3
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    cnt = [torch.tensor(5, dtype=torch.int64, device=DEV), torch.tensor(0, dtype=torch.int64, device=DEV)]
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    y = bn_act(x.to(DEV), w.to(DEV), b.to(DEV), rm, rv, True, 1e-5, 0.1, relu=True, pre_bias=cb.to(DEV), counters=cnt)
    rm_r, rv_r = rm0.clone(), rv0.clone()
    y_r = ref_fn(x, w, b, rm_r, rv_r, True, 1e-5, 0.1, relu=True, pre_bias=cb)
    torch.testing.assert_close(y.cpu(), y_r, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm.cpu(), rm_r, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), rv_r, rtol=1e-4, atol=1e-5)
    assert [int(c) for c in cnt] == [6, 1]


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_relu_dropout_matches_oracle(p):
    from oracle.bn_act_oracle import relu_dropout as ref_fn
    from vdetr_amd import attention as A
    from vdetr_amd.bn_act import relu_dropout
    A.reset_rng()
    A.begin_step(torch.device(DEV))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1024, 1, 256, generator=g)
    wout = torch.randn(1024, 1, 256, generator=g)
    drop = torch.nn.Dropout(p).train()
    keep = None
    if p > 0:
        y0 = relu_dropout(torch.ones(1024, 1, 256, device=DEV), drop, salt=9)
        keep = (y0 > 0).float().cpu()
        assert abs(float(keep.mean()) - (1 - p)) < 0.02
    res = []
    for fn, dev, kw in ((relu_dropout, DEV, {}), (ref_fn, "cpu", {"keep": keep})):
        xx = x.to(dev).requires_grad_(True)
        y = fn(xx, drop, salt=9, **kw)
        (y * wout.to(dev)).sum().backward()
        res.append((y.detach().cpu(), xx.grad.cpu()))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
