"""clip_grad_norm_ + AdamW on the flat parameter buffer as one launch (csrc/optim.hip, v-detr_amd/optim.py) against
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the separate parameters (engine.py:105-107, optimizer.py:6-26)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _models(seed=0):
    torch.manual_seed(seed)
    def make():
        return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.ReLU(), torch.nn.Linear(64, 129), torch.nn.LayerNorm(129),
                                   torch.nn.Linear(129, 5, bias=False)).to(DEV)
    a = make()
    b = make()
    b.load_state_dict(a.state_dict())
    return a, b


@pytest.mark.parametrize("max_norm,from_pack", [(0.1, True), (0.1, False), (1e4, True), (None, True)])
def test_clip_adamw_equals_torch(max_norm, from_pack):
    """six steps; max_norm 0.1 clips every step, 1e4 never, None skips the norm; from_pack False: the norm's partial sums from a
    launch of their own over the flat gradient (the N > 1 order: pack, all-reduce, norm)"""
    from vdetr_amd.dist import FlatParams
    from vdetr_amd.optim import ClipAdamW
    ref, own = _models(1)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=7e-4, weight_decay=0.1)
    flat = FlatParams(list(own.parameters()))
    opt = ClipAdamW(flat, lr=7e-4, weight_decay=0.1, max_norm=max_norm, norm_from_pack=from_pack)
    g = torch.Generator().manual_seed(3)
    for it in range(6):
        x = torch.randn((16, 37), generator=g).to(DEV) * (10.0 if it % 2 else 0.1)
        for m, o in ((ref, opt_ref), (own, opt)):
            for p in m.parameters():
                p.grad = None
            (m(x) ** 2).sum().backward()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm) if max_norm is not None else None
        opt_ref.step()
        flat.pack_grads()
        opt.step()
        if max_norm is not None:
            np.testing.assert_allclose(float(opt.grad_norm), float(norm_ref), rtol=2e-6)
        for (n, a), b in zip(ref.named_parameters(), own.parameters()):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-7, err_msg=f"step {it}: {n}")
    st = opt.state[flat.param]
    assert float(st["step"]) == 6.0 and int(opt._ticket[0]) == 0
    # torch's state keys: FlatParams' per-parameter optimizer state keeps working
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"}


def test_clip_adamw_in_a_captured_graph_counts_its_steps():
    """the step count is device-resident and written back by the last workgroup: five replays of ONE captured launch are steps
    2 .. 6 of the same sequence that six eager launches produce"""
    from vdetr_amd.dist import FlatParams
    from vdetr_amd.optim import ClipAdamW
    a, b = _models(2)
    flats = [FlatParams(list(m.parameters())) for m in (a, b)]
    opts = [ClipAdamW(f, lr=1e-3, weight_decay=0.05, max_norm=0.5, norm_from_pack=False) for f in flats]
    gsrc = torch.randn(flats[0].grad.numel(), device=DEV)
    for f in flats:
        f.grad.copy_(gsrc)
    for _ in range(6):
        opts[0].step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opts[1].step()  # step 1, eager (allocates the partials)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            opts[1].step()
    torch.cuda.current_stream().wait_stream(s)
    # (capturing does not execute: the graph's five replays are steps 2 .. 6)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    assert float(opts[1].state[flats[1].param]["step"]) == 6.0
    assert torch.equal(flats[0].data, flats[1].data)


def test_adamw_entry_point_tail_and_errors():
    """the C entry point on a buffer whose length is not a multiple of four, without clipping; argument errors come back as codes"""
    from vdetr_amd import _lib as L
    lib = L.lib()
    n = 1003
    g = torch.Generator().manual_seed(5)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p0.clone().to(DEV))
    pr.grad = gr.clone().to(DEV)
    ref = torch.optim.AdamW([pr], lr=1e-2, weight_decay=0.2, betas=(0.8, 0.95), eps=1e-6)
    ref.step()
    ref.step()
    buf = torch.zeros(4 * 1004, device=DEV).view(4, 1004)[:, :n]  # rows at a 16-B aligned pitch
    p, grad, m, v = buf[0], buf[1], buf[2], buf[3]
    p.copy_(p0.to(DEV)); grad.copy_(gr.to(DEV))
    step = torch.zeros((), device=DEV)
    ticket = torch.zeros(4, dtype=torch.int32, device=DEV)
    d = L.AdamWDesc()
    d.param, d.grad, d.exp_avg, d.exp_avg_sq = p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr()
    d.n, d.step, d.ticket = n, step.data_ptr(), ticket.data_ptr()
    d.lr, d.beta1, d.beta2, d.eps, d.weight_decay = 1e-2, 0.8, 0.95, 1e-6, 0.2
    for _ in range(2):
        assert lib.vdetr_adamw_clip_f32(ctypes.byref(d), L.stream_ptr()) == 0
    np.testing.assert_allclose(p.cpu().numpy(), pr.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    assert float(step) == 2.0
    d.beta1 = 1.0
    assert lib.vdetr_adamw_clip_f32(ctypes.byref(d), L.stream_ptr()) != 0 and b"betas" in lib.vdetr_last_error()
    d.beta1, d.param = 0.9, p.data_ptr() + 4
    assert lib.vdetr_adamw_clip_f32(ctypes.byref(d), L.stream_ptr()) != 0 and b"aligned" in lib.vdetr_last_error()
    assert lib.vdetr_sumsq_f32(grad.data_ptr(), n, m.data_ptr(), 7, L.stream_ptr()) != 0


def test_pack_partials_are_the_gradient_norm():
    from vdetr_amd.dist import FlatParams
    m, _ = _models(4)
    flat = FlatParams(list(m.parameters()))
    flat.want_sumsq = True
    (m(torch.randn((8, 37), device=DEV)) ** 2).sum().backward()
    flat.pack_grads()
    want = torch.linalg.vector_norm(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double())
    np.testing.assert_allclose(float(flat.sumsq.double().sum().sqrt()), float(want), rtol=1e-6)
    np.testing.assert_allclose(float(torch.linalg.vector_norm(flat.grad.double())), float(want), rtol=1e-6)


def test_clip_adamw_equals_the_fused_torch_step_it_replaces():
    """bench.py's previous update (FlatParams.clip_scale as the grad_scale of torch's fused, capturable AdamW on the flat parameter)
    and ClipAdamW from the same gradients, ten steps"""
    from vdetr_amd.dist import FlatParams
    from vdetr_amd.optim import ClipAdamW
    a, b = _models(6)
    fa, fb = FlatParams(list(a.parameters())), FlatParams(list(b.parameters()))
    opt_a = torch.optim.AdamW([fa.param], lr=7e-4, weight_decay=0.1, capturable=True, fused=True)
    opt_b = ClipAdamW(fb, lr=7e-4, weight_decay=0.1, max_norm=0.1)
    g = torch.Generator().manual_seed(8)
    for it in range(10):
        x = torch.randn((16, 37), generator=g).to(DEV)
        for m in (a, b):
            for p in m.parameters():
                p.grad = None
            (m(x) ** 2).sum().backward()
        fa.pack_grads()
        fb.pack_grads()
        opt_a.grad_scale = fa.clip_scale(0.1)[0]
        opt_a.step()
        opt_b.step()
        np.testing.assert_allclose(fb.data.cpu().numpy(), fa.data.cpu().numpy(), rtol=2e-5, atol=2e-7, err_msg=f"step {it}")
