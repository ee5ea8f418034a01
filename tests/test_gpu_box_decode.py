"""HIP box decode (vdetr_box_decode_{fwd,bwd}_f32) vs the torch oracle (oracle/box_oracle.py, pinned by the decoder
golden vectors): every key of the stage dictionary, and the gradients of every differentiable key."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIFF_KEYS = ["center_normalized", "center_unnormalized", "size_normalized", "size_unnormalized", "angle_residual",
             "angle_continuous", "angle_prob", "box_corners", "box_corners_axis_align", "center_reg", "size_reg",
             "sem_cls_logits", "angle_logits", "angle_residual_normalized"]


def _inputs(B, N, A, C1, seed, device):
    g = torch.Generator().manual_seed(seed)
    raw = {"center_head": torch.randn(B, 3, N, generator=g) * 0.3, "size_head": torch.randn(B, 3, N, generator=g) * 0.4,
           "angle_cls_head": torch.randn(B, A, N, generator=g) * 2, "angle_residual_head": torch.randn(B, A, N, generator=g),
           "sem_cls_head": torch.randn(B, C1, N, generator=g) * 2}
    pre_c = torch.rand(B, N, 3, generator=g)
    pre_s = torch.rand(B, N, 3, generator=g) * 0.3 + 0.02
    dmin = -torch.rand(B, 3, generator=g) * 4 - 1
    dmax = torch.rand(B, 3, generator=g) * 4 + 1
    mv = lambda t: t.to(device)
    return {k: mv(v) for k, v in raw.items()}, mv(pre_c), mv(pre_s), [mv(dmin), mv(dmax)]


@pytest.mark.parametrize("B,N,A,C1,cls_loss", [(1, 1024, 1, 19, "celoss"), (2, 300, 12, 11, "celoss"),
                                               (3, 77, 12, 10, "focalloss_0.25"), (1, 4096, 1, 19, "celoss")])
def test_box_decode_matches_oracle(B, N, A, C1, cls_loss):
    from oracle.box_oracle import decode_boxes_reference
    from vdetr_amd.box_decode import decode_boxes
    dev = torch.device("cuda")
    raw, pre_c, pre_s, dims = _inputs(B, N, A, C1, 7 + N, dev)
    nbin = A
    res, grads = {}, {}
    for name, fn, where in (("hip", decode_boxes, dev), ("ref", decode_boxes_reference, torch.device("cpu"))):
        r = {k: v.detach().to(where).requires_grad_(True) for k, v in raw.items()}
        out = fn(r, pre_c.to(where), pre_s.to(where), [d.to(where) for d in dims], nbin, cls_loss)
        g = torch.Generator().manual_seed(99)
        loss = 0
        for k in DIFF_KEYS:
            loss = loss + (out[k] * torch.randn(out[k].shape, generator=g).to(where)).sum()
        loss.backward()
        res[name] = {k: v.detach().cpu().numpy() for k, v in out.items()}
        grads[name] = {k: v.grad.detach().cpu().numpy() for k, v in r.items()}
    assert set(res["hip"]) == set(res["ref"])
    for k, ref in res["ref"].items():
        got = res["hip"][k]
        assert got.shape == ref.shape, k
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5, err_msg=k)
    for k, ref in grads["ref"].items():
        tol = 1e-4 * max(np.abs(ref).max(), 1e-3)
        np.testing.assert_allclose(grads["hip"][k], ref, rtol=1e-3, atol=tol, err_msg="d" + k)


def test_box_decode_unused_outputs_and_cpu_error():
    from vdetr_amd.box_decode import decode_boxes
    dev = torch.device("cuda")
    raw, pre_c, pre_s, dims = _inputs(1, 64, 1, 19, 3, dev)
    r = {k: v.requires_grad_(True) for k, v in raw.items()}
    out = decode_boxes(r, pre_c, pre_s, dims, 1)
    out["center_normalized"].sum().backward()  # every other output unused: NULL gradient pointers
    assert torch.isfinite(r["center_head"].grad).all() and r["size_head"].grad.abs().max() == 0
    with pytest.raises(RuntimeError, match="CPU not supported"):
        decode_boxes({k: v.cpu() for k, v in raw.items()}, pre_c.cpu(), pre_s.cpu(), [d.cpu() for d in dims], 1)


@pytest.mark.parametrize("B,N,A,C1,cls_loss", [(1, 1024, 1, 19, "celoss"), (2, 200, 12, 11, "celoss"), (2, 64, 12, 10, "focalloss_0.25")])
def test_box_decode_joint_slabs_match_oracle(B, N, A, C1, cls_loss):
    """The five head outputs as slabs of one [B,5,rows,N] tensor: same dictionary, and the backward delivers the whole
    gradient of the joint tensor (zeros in the slab padding) in one launch."""
    from oracle.box_oracle import decode_boxes_reference
    from vdetr_amd.box_decode import decode_boxes_joint
    dev = torch.device("cuda")
    chans = (C1, 3, 3, A, A)
    rows = max(chans) + 2
    g = torch.Generator().manual_seed(N + A)
    y0 = torch.randn(B, 5, rows, N, generator=g)
    _, pre_c, pre_s, dims = _inputs(B, N, A, C1, 3, dev)
    names = ("sem_cls_head", "center_head", "size_head", "angle_cls_head", "angle_residual_head")
    res, grads = {}, {}
    for name, where in (("hip", dev), ("ref", torch.device("cpu"))):
        y = y0.to(where).requires_grad_(True)
        if name == "hip":
            out = decode_boxes_joint(y, chans, pre_c, pre_s, dims, A, cls_loss)
        else:
            raw = {n: y[:, i, :chans[i]] for i, n in enumerate(names)}
            out = decode_boxes_reference(raw, pre_c.cpu(), pre_s.cpu(), [d.cpu() for d in dims], A, cls_loss)
        gg = torch.Generator().manual_seed(5)
        loss = 0
        for k in DIFF_KEYS:
            loss = loss + (out[k] * torch.randn(out[k].shape, generator=gg).to(where)).sum()
        loss.backward()
        res[name] = {k: v.detach().cpu().numpy() for k, v in out.items()}
        grads[name] = y.grad.detach().cpu().numpy()
    for k, ref in res["ref"].items():
        np.testing.assert_allclose(res["hip"][k], ref, rtol=1e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(grads["hip"], grads["ref"], rtol=1e-3, atol=1e-4 * np.abs(grads["ref"]).max())
    for i in range(5):
        assert np.abs(grads["hip"][:, i, chans[i]:]).max() == 0


DEV = "cuda"


@pytest.mark.parametrize("B,N,ncls", [(1, 4096, 18), (3, 1000, 10), (2, 1, 18)])
def test_anchor_boxes_launch_equals_the_tensor_expressions(B, N, ncls):
    """vdetr_anchor_boxes_f32 (models/model_vdetr.py:348-362 as one launch) against the expressions it replaces: class by the arg max
    of the sigmoid (saturated logits tie as there), anchor sizes, convert_unnorm2norm of centre and size, yaw-0 corners."""
    from vdetr_amd import box_decode as BD
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.model_vdetr import convert_unnorm2norm
    g = torch.Generator().manual_seed(B * 100 + N)
    logits = (torch.randn((B, N, ncls), generator=g) * 3).to(DEV)   # (|x| < 16: no probability rounds to 1 by accident)
    logits[:, ::7, 3:6] = 40.0   # saturated: sigmoid == 1 for several classes, the first of them wins
    xyz = (torch.rand((B, N, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0]) + 1).to(DEV)
    dims = [xyz.min(1)[0] - 0.1, xyz.max(1)[0] + 0.2]
    anchors = (torch.rand((ncls, 3), generator=g) + 0.3).to(DEV)
    size, cn, sn, corners = BD.anchor_boxes(logits, xyz, dims, anchors)
    cls = logits.sigmoid().max(dim=-1)[1]
    want_size = anchors[cls]
    assert torch.equal(size, want_size)
    assert torch.allclose(cn, convert_unnorm2norm(xyz, dims), rtol=1e-6, atol=1e-7)
    assert torch.allclose(sn, convert_unnorm2norm(want_size, dims, with_offset=False), rtol=1e-6, atol=1e-7)
    want = ScannetDatasetConfig().box_parametrization_to_corners(xyz, want_size, None)
    assert torch.allclose(corners, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,N,nq,camera", [(1, 4096, 1024, False), (2, 300, 64, True), (3, 50, 50, False)])
def test_gather_proposals_launch_equals_the_gathers(B, N, nq, camera):
    """vdetr_gather_proposals_f32 (models/vdetr_transformer.py:364-398) against torch.gather + convert_corners_camera2lidar + cat"""
    from vdetr_amd import box_decode as BD
    from vdetr_amd.vdetr_transformer import convert_corners_camera2lidar
    g = torch.Generator().manual_seed(N + nq)
    pred = {"box_corners": torch.randn((B, N, 8, 3), generator=g).to(DEV), "center_unnormalized": torch.randn((B, N, 3), generator=g).to(DEV),
            "size_unnormalized": torch.rand((B, N, 3), generator=g).to(DEV), "angle_continuous": torch.randn((B, N), generator=g).to(DEV),
            "center_normalized": torch.rand((B, N, 3), generator=g).to(DEV), "size_normalized": torch.rand((B, N, 3), generator=g).to(DEV)}
    if not camera:
        pred["_reference_point_lidar"] = convert_corners_camera2lidar(pred["box_corners"]).contiguous()
    topk = torch.stack([torch.randperm(N, generator=g)[:nq] for _ in range(B)]).to(DEV)
    assert BD.proposals_fusable(topk, pred)
    ref, center, size, ang, cn, sn, qref = BD.gather_proposals(topk, pred)

    def take(t):
        index = topk.view(topk.shape + (1,) * (t.dim() - 2)).expand(topk.shape + t.shape[2:])
        return torch.gather(t, 1, index)
    assert torch.equal(ref, convert_corners_camera2lidar(take(pred["box_corners"])))
    assert torch.equal(center, take(pred["center_unnormalized"])) and torch.equal(size, take(pred["size_unnormalized"]))
    assert torch.equal(ang, take(pred["angle_continuous"]))
    assert torch.equal(cn, take(pred["center_normalized"])) and torch.equal(sn, take(pred["size_normalized"]))
    assert torch.equal(qref, torch.cat((center, size), -1))


@pytest.mark.parametrize("n,hidden,T", [(64, 128, 10), (8, 128, 10), (3, 48, 7), (1, 256, 4)])
def test_cpb_tables_launch_equals_the_mlps(n, hidden, T):
    """vdetr_cpb_tables_f32 (the RPE tables of n cpb MLPs in one launch, csrc/cpb_tables.hip) against Linear -> ReLU -> Linear in fp64:
    the hidden activations (what the tables' backward reads) and the tables"""
    from vdetr_amd import _lib as L
    g = torch.Generator().manual_seed(n * 1000 + hidden)
    lin = torch.linspace(-4.0, 4.0, T)
    coords = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), dim=-1).reshape(-1, 3).contiguous().to(DEV)
    w1 = torch.randn((n, hidden, 3), generator=g).to(DEV)
    b1 = torch.randn((n, hidden), generator=g).to(DEV)
    w2 = (torch.randn((n, 4, hidden), generator=g) / hidden ** 0.5).to(DEV)
    P = coords.shape[0]
    hid = torch.full((n, P, hidden), float("nan"), device=DEV)
    tab = torch.full((n, P, 4), float("nan"), device=DEV)
    L.check(L.lib().vdetr_cpb_tables_f32(L.ptr(coords), L.ptr(w1), L.ptr(b1), L.ptr(w2), n, P, hidden, 4, L.ptr(hid), L.ptr(tab), L.stream_ptr()),
            "cpb_tables")
    want_h = torch.relu(torch.einsum("pc,nhc->nph", coords.double(), w1.double()) + b1.double()[:, None, :])
    want_t = torch.einsum("nph,nkh->npk", want_h, w2.double())
    np.testing.assert_allclose(hid.cpu().numpy(), want_h.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tab.cpu().numpy(), want_t.cpu().numpy(), rtol=1e-5, atol=2e-5 * float(want_t.abs().max()))
    lib = L.lib()
    assert lib.vdetr_cpb_tables_f32(L.ptr(coords), L.ptr(w1), L.ptr(b1), L.ptr(w2), n, P, 100, 4, L.ptr(hid), L.ptr(tab), L.stream_ptr()) != 0
    assert b"hidden" in lib.vdetr_last_error()
