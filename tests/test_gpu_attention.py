"""Fused attention kernels (HIP, through the C-ABI) vs the CPU oracle and the reference's golden vectors.

Tolerance (BASELINE.json north_star): box/logit tensors within 1e-3 relative fp32.  Forward outputs are checked at
1e-4 relative (+1e-5 abs), gradients at 1e-3 relative to the tensor's max (fp32 sums in a different order).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import (assert_close, grad_atol, build_cross_attention, build_decoder, build_share_self_attention,
                     run_cross_attention_case, run_decoder_case, t)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_mfma_layout_selftest():
    """A[16,64] x B[16,64]^T through v_mfma_f32_16x16x4_f32 with the operand mapping the kernels use; asymmetric
    operands so a transposed result cannot pass."""
    import ctypes
    from vdetr_amd import _lib as L
    a = torch.randn(16, 64, device=DEV)
    b = torch.randn(16, 64, device=DEV) + torch.arange(16, device=DEV)[:, None]
    c = torch.zeros(16, 16, device=DEV)
    L.check(L.lib().vdetr_selftest_mfma_f32(L.ptr(a), L.ptr(b), L.ptr(c), L.stream_ptr()), "selftest")
    assert_close(c, (a.double() @ b.double().T).cpu().numpy(), 1e-5, 1e-4, "mfma")


def _scene(B, nQ, nK, seed, rot=False):
    g = torch.Generator().manual_seed(seed)
    lo, ext = torch.tensor([1.0, 1.0, 1.0]), torch.tensor([8.0, 6.0, 3.0])
    xyz = lo + torch.rand((B, nK, 3), generator=g) * ext
    center = lo + torch.rand((B, nQ, 3), generator=g) * ext
    half = 0.1 + torch.rand((B, nQ, 3), generator=g)
    signs = torch.tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]],
                         dtype=torch.float32)
    verts = center[:, :, None, :] + half[:, :, None, :] * signs
    tables = torch.randn((8, 10, 10, 10, 4), generator=g)
    cs = None
    if rot:
        ang = (torch.rand((B, nQ), generator=g) * 2 - 1) * 3.1
        cs = torch.stack((torch.cos(ang), torch.sin(ang)), -1)
    return xyz, verts, tables, cs


@pytest.mark.parametrize("B,nQ,nK,rot", [(2, 5, 7, False), (1, 33, 100, True), (1, 64, 1000, False)])
def test_rpe_bias_kernel(B, nQ, nK, rot):
    from oracle.attention_oracle import rpe_bias_reference
    from vdetr_amd import attention as A
    xyz, verts, tables, cs = _scene(B, nQ, nK, 3, rot)
    xyz[:, 1] = torch.tensor([30.0, -25.0, 12.0])  # all-padding key
    xyz[:, 0] = verts[:, 0, 0]                     # delta == 0
    ref = rpe_bias_reference(tables.double(), verts.double(), xyz.double(), cos_sin=None if cs is None else cs.double())
    got = A.rpe_bias(tables.to(DEV), verts.to(DEV).contiguous(), xyz.to(DEV), A.RPEConfig(), None if cs is None else cs.to(DEV))
    assert_close(got, ref.numpy(), 1e-4, 2e-5 * float(ref.abs().max()), "rpe")
    assert float(got[:, :, :, 1].abs().max()) == 0.0


CASES = [  # B, nQ, nK, shared, rpe, rot, mask
    (2, 5, 7, True, True, False, None),
    (1, 64, 512, True, True, True, None),
    (1, 37, 301, True, True, False, "bool"),
    (2, 16, 96, True, True, False, "float"),
    (1, 4, 4096, True, True, False, None),      # few queries -> key-split path
    (1, 48, 1024, True, True, False, None),     # axis-aligned boxes, every wave of a backward workgroup busy
    (1, 24, 640, True, True, "jitter", None),   # arbitrary vertices without a rotation operand: the general kernels
    (2, 9, 9, True, False, False, None),        # ShareSelfAttention core
    (1, 50, 50, False, False, False, None),     # nn.MultiheadAttention core
    (2, 130, 77, False, False, False, "float"),
]


@pytest.mark.parametrize("B,nQ,nK,shared,rpe,rot,mask", CASES)
def test_fused_attention_forward_backward(B, nQ, nK, shared, rpe, rot, mask):
    from oracle.attention_oracle import fused_attention_reference
    from vdetr_amd import attention as A
    H = 4
    g = torch.Generator().manual_seed(B * 1000 + nQ + nK)
    xyz, verts, tables, cs = _scene(B, nQ, nK, 7, rot is True)
    if rot == "jitter":  # one query that is not a box is enough to send the whole launch down the general path
        verts[:, nQ // 2] += 0.05 * torch.randn(verts[:, nQ // 2].shape, generator=g)
    q = torch.randn((B, nQ, 256), generator=g)
    kd = 64 if shared else 256
    k = torch.randn((B, nK, kd), generator=g)
    v = torch.randn((B, nK, kd), generator=g)
    wout = torch.randn((B, nQ, 256), generator=g)
    m = None
    if mask == "bool":
        m = torch.rand((B, nQ, nK), generator=g) < 0.2
    elif mask == "float":
        m = torch.randn((B, nQ, nK), generator=g)
    cfg = A.RPEConfig()
    kw = dict(num_heads=H, scale=0.125, shared_kv=shared)

    def run(fn, dev, dtype):
        args = [x.to(dev, dtype).requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(dev, dtype).requires_grad_(True) if rpe else None
        extra = dict(table=tb, rpe=cfg, vertices=verts.to(dev, dtype).contiguous(), xyz=xyz.to(dev, dtype),
                     cos_sin=None if cs is None else cs.to(dev, dtype)) if rpe else {}
        mm = None if m is None else (m.to(dev) if m.dtype == torch.bool else m.to(dev, dtype))
        out = fn(*args, attn_mask=mm, **kw, **extra)
        (out * wout.to(dev, dtype)).sum().backward()
        return [out] + [a.grad for a in args] + ([tb.grad] if rpe else [])

    ref = run(fused_attention_reference, "cpu", torch.float64)
    got = run(A.fused_attention, DEV, torch.float32)
    names = ["out", "dq", "dk", "dv", "dtable"]
    for name, r, o in zip(names, ref, got):
        scale = float(r.abs().max())
        if name == "out":
            assert_close(o, r.detach().numpy(), 1e-4, 1e-5 * max(scale, 1.0), name)
        else:
            assert_close(o, r.detach().numpy(), 1e-3, 1e-4 * scale + 1e-7, name)


def _oracle_in_chunks(q, k, v, tables, verts, xyz, wout, kw, cs=None, chunk=32, device="cpu"):
    """fused_attention_reference in fp64, `chunk` queries at a time (softmax rows are independent; dk, dv and dtable are sums over
    the chunks) -> [out, dq, dk, dv, dtable] as CPU tensors.  device "cpu": as always.  device = the GPU: the SAME torch code,
    executed by ATen's fp64 kernels (none of this library's) — 6 s instead of 50-90 s for a 1024 x 4096 layer.  One full-size case
    of every test keeps the CPU evaluation, and test_oracle_on_the_gpu_equals_the_oracle_on_the_cpu holds the two together."""
    from oracle.attention_oracle import fused_attention_reference
    dd = lambda t_: t_.double().to(device)  # noqa: E731
    rk, rv, rtb = dd(k).requires_grad_(True), dd(v).requires_grad_(True), dd(tables).requires_grad_(True)
    rxyz, rw = dd(xyz), dd(wout)
    routs, rdq = [], []
    nQ = q.shape[1]
    for c in range(0, nQ, chunk):
        sl = slice(c, c + chunk)
        rq = dd(q[:, sl]).requires_grad_(True)
        o = fused_attention_reference(rq, rk, rv, table=rtb, vertices=dd(verts[:, sl]).contiguous(), xyz=rxyz,
                                      cos_sin=None if cs is None else dd(cs[:, sl]), **kw)
        (o * rw[:, sl]).sum().backward()
        routs.append(o.detach().cpu())
        rdq.append(rq.grad.cpu())
    return [torch.cat(routs, 1), torch.cat(rdq, 1), rk.grad.cpu(), rv.grad.cpu(), rtb.grad.cpu()]


def test_oracle_on_the_gpu_equals_the_oracle_on_the_cpu():
    """the fp64 oracle through ATen's GPU kernels against the same code on the CPU (a rotated, general-vertex case): 1e-10"""
    from vdetr_amd import attention as A
    B, nQ, nK = 2, 48, 640
    g = torch.Generator().manual_seed(3)
    xyz, verts, tables, cs = _scene(B, nQ, nK, 4, True)
    q, k, v = (torch.randn(s_, generator=g) for s_ in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, rpe=A.RPEConfig())
    a = _oracle_in_chunks(q, k, v, tables, verts, xyz, wout, kw, cs, 16, "cpu")
    b = _oracle_in_chunks(q, k, v, tables, verts, xyz, wout, kw, cs, 48, DEV)
    for name, x, y in zip(["out", "dq", "dk", "dv", "dtable"], a, b):
        assert float((x - y).abs().max()) <= 1e-10 * max(1.0, float(x.abs().max())), name


@pytest.mark.parametrize("boxes", [True, False, "rotated", "rotated_boxes"])
def test_full_size_forward_backward_vs_oracle(boxes):
    """The launch bench.py times (B=1, nQ=1024, nK=4096, H=4: BASELINE config 2's layer size) against the fp64 oracle:
    out, dq, dk, dv and the RPE-table gradient within 1e-3 relative.  boxes=True: axis-aligned box vertices, i.e.
    the box kernel (attn_bwd_box4.hip) with its dynamic query distribution and every wave of every workgroup busy; boxes=False: a few
    perturbed vertices send the same launch down the general kernel; "rotated": the (cos, sin) operand of angle_type
    "object_coords" (vdetr_transformer.py:712-720, BASELINE config 5) at the full size on arbitrary vertices (general kernels);
    "rotated_boxes": the same operand with the corners of rotated boxes (forward: one rotation per pair + the box body,
    backward: attn_bwd_box4.hip).  The oracle is evaluated in chunks of queries (_oracle_in_chunks)."""
    from vdetr_amd import attention as A
    B, nQ, nK, H = 1, 1024, 4096, 4
    g = torch.Generator().manual_seed(21)
    xyz, verts, tables, cs = _scene(B, nQ, nK, 5, boxes == "rotated")  # "rotated": BASELINE config 5's operand (object_coords)
    if boxes == "rotated_boxes":  # corners of ROTATED boxes, as box_decode writes them: the box bodies of forward and backward
        xyz, verts, tables, cs = _rotated_boxes(B, nQ, nK, 5)
    if not boxes:
        verts[:, ::97] += 0.05 * torch.randn(verts[:, ::97].shape, generator=g)
    q, k, v = (torch.randn(s, generator=g) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g)
    cfg = A.RPEConfig()
    kw = dict(num_heads=H, scale=0.125, shared_kv=True, rpe=cfg)
    # device
    dq, dk, dv = (x.to(DEV).requires_grad_(True) for x in (q, k, v))
    dtb = tables.to(DEV).requires_grad_(True)
    out = A.fused_attention(dq, dk, dv, table=dtb, vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV),
                            cos_sin=None if cs is None else cs.to(DEV), **kw)
    (out * wout.to(DEV)).sum().backward()
    # oracle: the axis-aligned-box case on the CPU, 32 queries at a time (52 s); the other three through ATen's fp64 kernels on the GPU
    ref = _oracle_in_chunks(q, k, v, tables, verts, xyz, wout, kw, cs, 32 if boxes is True else 256, "cpu" if boxes is True else DEV)
    got = [out.detach(), dq.grad, dk.grad, dv.grad, dtb.grad]
    for name, r, o in zip(["out", "dq", "dk", "dv", "dtable"], ref, got):
        scale = float(r.abs().max())
        if name == "out":
            assert_close(o, r.numpy(), 1e-4, 1e-5 * max(scale, 1.0), name)
        elif name == "dtable":
            # 33.5 M (query, key, vertex) contributions (exact fp32 products in the box kernel, 2-term split-bf16 in the general
            # one) summed in int32 fixed point whose scale comes from a worst-case BOUND of a bin's sum (all of a query's
            # attention mass in one bin); real bins hold ~1e-3 of that, so a group's sum keeps ~10 bits.  The
            # rounding is unbiased and absolute: ~2e-4 of the largest entry whatever the entry's size, ~4e-4 relative L2.
            assert_close(o, r.numpy(), 1e-3, 3e-4 * scale, name)
            assert float((o.cpu().double() - r).norm() / r.norm()) < 1e-3, "dtable, relative L2 error"
        else:
            assert_close(o, r.numpy(), 1e-3, 1e-4 * scale + 1e-7, name)


@pytest.mark.parametrize("case", ["rpe_boxes", "rpe_general", "plain_dropout_mask", "ragged", "per_head", "per_head_ragged",
                                  "per_head_one_wg", "per_head_ragged_one_wg"])
def test_fused_key_side_backward_equals_gemm_path(monkeypatch, case):
    """attn_bwd_kv.hip (dO V^T, softmax backward, dV, dK in one pass; dS handed to the table kernels) against the library
    GEMM path on the same launch: dq, dk, dv within 2e-5 of the largest entry (split-bf16 products, 2^-16 each, against
    fp32 GEMMs), the table gradient within the fixed-point resolution, and the fused path bit-identical run to run (its
    reduction tree is fixed and two commutative adds meet in memory)."""
    from vdetr_amd import attention as A
    g = torch.Generator().manual_seed(31)
    per_head = case.startswith("per_head")
    if case.endswith("_one_wg"):  # vdetr_attn_desc.kv_halves = 1: the shape the step takes while a table-gradient kernel is live
        monkeypatch.setattr(A, "_side_keep", [None])
        case = case[:-len("_one_wg")]
    if case == "ragged":
        B, nQ, nK = 2, 37, 301  # partial row tiles, partial key tiles, idle row slots
    elif case == "per_head":
        B, nQ, nK = 1, 1024, 1024  # the decoder's query self-attention
    elif case == "per_head_ragged":
        B, nQ, nK = 3, 77, 45
    else:
        B, nQ, nK = 1, 256, 1536
    xyz, verts, tables, _ = _scene(B, nQ, nK, 5)
    if case == "rpe_general":
        verts[:, ::7] += 0.05 * torch.randn(verts[:, ::7].shape, generator=g)
    kvc = 256 if per_head else 64
    q, k, v = (torch.randn(s, generator=g).to(DEV) for s in ((B, nQ, 256), (B, nK, kvc), (B, nK, kvc)))
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)
    kw = dict(num_heads=4, scale=0.125, shared_kv=not per_head)
    rpe = case not in ("plain_dropout_mask", "per_head", "per_head_ragged")
    if rpe:
        kw.update(rpe=A.RPEConfig(), vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))
    else:
        kw.update(dropout_p=0.1, rng_state=A.new_rng_state(DEV, 5), attn_mask=(torch.rand((B, nQ, nK), generator=g) < 0.1).to(DEV))

    def run():
        args = [x.clone().requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(DEV).requires_grad_(True) if rpe else None
        (A.fused_attention(*args, table=tb, **kw) * wout).sum().backward()
        return [a.grad for a in args] + ([tb.grad] if rpe else [])

    monkeypatch.setattr(A, "FUSED_KV_BWD", False)
    ref = run()
    monkeypatch.setattr(A, "FUSED_KV_BWD", True)
    got = run()
    again = run()
    for name, r, o, o2 in zip(["dq", "dk", "dv", "dtable"], ref, got, again):
        scale = float(r.abs().max())
        tol = 3e-4 if name == "dtable" else 2e-5
        err = float((o - r).abs().max())
        assert err <= tol * scale, f"{case} {name}: max |diff| {err:.3e} vs scale {scale:.3e}"
        assert torch.equal(o, o2), f"{case} {name}: not reproducible"


def test_box_backward_kernel_equals_general_kernel(monkeypatch, fused=True):
    """The axis-aligned-box backward kernel (attn_bwd_box4.hip) against the general one on the same launch, at a size
    where every wave of every workgroup is busy (the size at which a packed-math code-generation problem once showed):
    P~ and dS bit-identical (same element-wise code), table gradient within the fixed-point resolution, three times.
    fused: the kernels read the dS that attn_bwd_kv.hip wrote instead of forming it (dq, dk, dv are then that pass's)."""
    from vdetr_amd import attention as A
    monkeypatch.setattr(A, "FUSED_KV_BWD", fused)
    B, nQ, nK = 1, 192, 2048
    g = torch.Generator().manual_seed(11)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 5)
    q, k, v = (torch.randn(s, generator=g).to(DEV) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, rpe=A.RPEConfig(), vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))

    def run():
        args = [x.clone().requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(DEV).requires_grad_(True)
        (A.fused_attention(*args, table=tb, **kw) * wout).sum().backward()
        return [a.grad for a in args] + [tb.grad]

    monkeypatch.setattr(A, "BWD_KERNEL", 1)  # vdetr_attn_desc.bwd_kernel: the general kernel alone
    ref = run()
    monkeypatch.setattr(A, "BWD_KERNEL", 0)
    for rep in range(3):
        got = run()
        for name, r, o in zip(("dq", "dk", "dv"), ref, got):
            assert torch.equal(r, o), f"{name} differs between the box and the general kernel (rep {rep})"
        scale = float(ref[3].abs().max())
        assert float((got[3] - ref[3]).abs().max()) <= 3e-4 * scale, f"dtable rep {rep}"


@pytest.mark.parametrize("B,nQ,nK", [(1, 8, 64), (1, 5, 3), (2, 33, 700), (1, 300, 1024), (1, 70, 1500), (3, 40, 2049), (1, 520, 4096)])
def test_sorted_box_backward_kernel_shapes(monkeypatch, B, nQ, nK):
    """attn_bwd_box4.hip (64-key chunks sorted inside a wave) at sizes that leave chunks, waves and
    quads partly filled — fewer keys than a wave, a ragged last tile, one key in the last tile, several scenes, more queries
    than two rounds of the grid — against the general kernel on the same dS: table gradient within the fixed-point resolution,
    and bit-identical run to run (integer sums)."""
    from vdetr_amd import attention as A
    monkeypatch.setattr(A, "FUSED_KV_BWD", True)
    g = torch.Generator().manual_seed(B * 7 + nQ + nK)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 5)
    q, k, v = (torch.randn(s, generator=g).to(DEV) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, rpe=A.RPEConfig(), vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))

    def run():
        args = [x.clone().requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(DEV).requires_grad_(True)
        (A.fused_attention(*args, table=tb, **kw) * wout).sum().backward()
        return tb.grad

    monkeypatch.setattr(A, "BWD_KERNEL", 1)
    ref = run()
    monkeypatch.setattr(A, "BWD_KERNEL", 0)
    got, again = run(), run()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-4 * scale, "dtable"
    assert float((got - ref).norm() / ref.norm()) < 1e-3, "dtable, relative L2"
    assert torch.equal(got, again), "not reproducible"


@pytest.mark.parametrize("grid,tol", [(192, 3e-4), (190, 3e-4), (64, 3e-4), (16, 1e-3)])
def test_table_gradient_on_fewer_workgroups(monkeypatch, grid, tol):
    """vdetr_attn_desc.table_grid: the table-gradient launches on fewer (persistent) workgroups than CUs — what a caller
    does who runs them on a side stream next to the main chain — give the default grid's gradient to the rounding of the
    histogram's fixed-point scale (which follows the queries per workgroup: measured 0 / 1.5e-4 / 3.5e-4 / 2.4e-3 of the largest
    entry at 192 / 64 / 16 / 2 workgroups), are reproducible per count, and the default comes back with 0."""
    from vdetr_amd import _lib as L
    from vdetr_amd import attention as A
    monkeypatch.setattr(A, "FUSED_KV_BWD", True)
    B, nQ, nK = 1, 300, 1024
    g = torch.Generator().manual_seed(grid)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 9)
    q, k, v = (torch.randn(s, generator=g).to(DEV) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, rpe=A.RPEConfig(), vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))

    def run():
        args = [x.clone().requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(DEV).requires_grad_(True)
        (A.fused_attention(*args, table=tb, **kw) * wout).sum().backward()
        return tb.grad, args[0].grad

    monkeypatch.setattr(A, "ASYNC_TABLE_GRAD", False)  # in line: the launch then takes A.TABLE_GRID
    ref, dq_ref = run()
    monkeypatch.setattr(A, "TABLE_GRID", grid)
    got, dq = run()
    again, _ = run()
    monkeypatch.setattr(A, "TABLE_GRID", 0)
    back, _ = run()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= tol * scale and float((got - ref).norm() / ref.norm()) < 1e-3
    assert torch.equal(got, again), "not reproducible"
    assert torch.equal(dq, dq_ref), "the other gradients do not depend on the grid"
    assert torch.equal(back, ref), "default grid not restored"
    for bad in (3, 258):  # odd, more than the CUs: refused per call
        monkeypatch.setattr(A, "TABLE_GRID", bad)
        with pytest.raises(RuntimeError, match="table_grid"):
            run()
    monkeypatch.setattr(A, "TABLE_GRID", 0)


@pytest.mark.parametrize("kind,B,nQ,nK,kstride", [("shared", 1, 1024, 4096, 1024), ("shared", 2, 33, 700, 64), ("shared", 1, 5, 3, 64),
                                                 ("shared", 3, 40, 2049, 128), ("perhead", 1, 1024, 1024, 256), ("perhead", 2, 50, 77, 256),
                                                 ("perhead", 1, 17, 130, 256), ("shared", 1, 64, 63, 64)])
def test_dq_row_owner_kernel_equals_matmul(kind, B, nQ, nK, kstride):
    """vdetr_attn_bwd_dq_f32 (attn_bwd_dq.hip): dQ = scale * dS K with exact fp32 products against the fp64 matmul, for both
    kinds, keys that do not fill a 64-key chunk or a 16-byte load, rows that do not fill a tile, several scenes and K as a
    column block of a wider projection (k_row_stride)."""
    import ctypes
    from vdetr_amd import _lib as L
    from vdetr_amd import attention as A
    H = 4
    g = torch.Generator().manual_seed(nQ * 7 + nK)
    shared = kind == "shared"
    ds = torch.randn((B, nQ, H, nK) if shared else (B, H, nQ, nK), generator=g).to(DEV)
    wide = torch.randn((B, nK, kstride), generator=g).to(DEV)
    k = wide[:, :, :64] if shared else wide[:, :, :256]
    d = A._desc(L.VDETR_ATTN_SHARED_KV if shared else L.VDETR_ATTN_PER_HEAD, B, H, nQ, nK, 0.125, None, None, None, None, None, None,
                0.0, None, 0, kstride, 0)
    dq = torch.full((B, nQ, H * 64), float("nan"), device=DEV)
    L.check(L.lib().vdetr_attn_bwd_dq_f32(ctypes.byref(d), L.ptr(ds), L.ptr(k), L.ptr(dq), L.stream_ptr()), "attn_bwd_dq")
    if shared:
        ref = 0.125 * torch.einsum("bqhk,bkd->bqhd", ds.double(), k.double()).reshape(B, nQ, H * 64)
    else:
        ref = 0.125 * torch.einsum("bhqk,bkhd->bqhd", ds.double(), k.double().reshape(B, nK, H, 64)).reshape(B, nQ, H * 64)
    assert torch.isfinite(dq).all()
    assert float((dq.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) * max(1.0, (nK / 64) ** 0.5)


def _rotated_boxes(B, nQ, nK, seed):
    """keys, the corners of ROTATED boxes (centre + R(angle)^T (+-half), as box_decode writes them), tables, (cos, sin)"""
    xyz, verts, tables, _ = _scene(B, nQ, nK, seed)
    g = torch.Generator().manual_seed(seed + 100)
    ang = (torch.rand((B, nQ), generator=g) * 2 - 1) * 3.1
    c, s_ = torch.cos(ang)[:, :, None], torch.sin(ang)[:, :, None]
    centre = verts.mean(2, keepdim=True)
    off = verts - centre
    # corners = centre + R^T off, so that rpe_rotate (x' = x c - y s, y' = x s + y c) turns them back onto the axes
    rot = torch.stack((off[..., 0] * c + off[..., 1] * s_, -off[..., 0] * s_ + off[..., 1] * c, off[..., 2]), -1)
    return xyz, (centre + rot).contiguous(), tables, torch.stack((c[..., 0], s_[..., 0]), -1).contiguous()


@pytest.mark.parametrize("B,nQ,nK", [(1, 40, 640), (2, 33, 700), (1, 300, 1024)])
def test_rotated_boxes_take_the_box_backward_kernel(monkeypatch, B, nQ, nK):
    """angle_type "object_coords" (BASELINE config 5): the corners of a rotated box are an axis-aligned box in the frame the
    look-up is turned into, so the sorted box kernel (attn_bwd_box4.hip) does their table gradient too.  Against the fp64 oracle
    (1e-3 of the scale, as every gradient), against the general kernel on the same launch (the box kernel looks up at
    R (P_0 - X) + edge instead of R (P_i - X): a few ulps of the coordinates), and: vertices that are NOT a rotated box keep
    the general kernel (the gate counts them)."""
    from oracle.attention_oracle import fused_attention_reference
    from vdetr_amd import attention as A
    monkeypatch.setattr(A, "FUSED_KV_BWD", True)
    g = torch.Generator().manual_seed(nQ + nK)
    xyz, verts, tables, cs = _rotated_boxes(B, nQ, nK, 7)
    q, k, v = (torch.randn(s, generator=g) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, rpe=A.RPEConfig())

    def run(vv):
        args = [x.to(DEV).requires_grad_(True) for x in (q, k, v)]
        tb = tables.to(DEV).requires_grad_(True)
        out = A.fused_attention(*args, table=tb, vertices=vv.to(DEV).contiguous(), xyz=xyz.to(DEV), cos_sin=cs.to(DEV), **kw)
        (out * wout.to(DEV)).sum().backward()
        return out.detach(), tb.grad

    monkeypatch.setattr(A, "BWD_KERNEL", 1)
    _, gen = run(verts)
    monkeypatch.setattr(A, "BWD_KERNEL", 0)
    out, box = run(verts)
    _, again = run(verts)
    rq, rk, rv = (x.double().requires_grad_(True) for x in (q, k, v))
    rtb = tables.double().requires_grad_(True)
    ro = fused_attention_reference(rq, rk, rv, table=rtb, vertices=verts.double(), xyz=xyz.double(), cos_sin=cs.double(), **kw)
    (ro * wout.double()).sum().backward()
    scale = float(rtb.grad.abs().max())
    assert_close(out, ro.detach().numpy(), 1e-4, 1e-5 * max(float(ro.abs().max()), 1.0), "out")
    assert_close(box, rtb.grad.numpy(), 1e-3, 3e-4 * scale, "dtable (box kernel) vs oracle")
    assert float((box - gen).abs().max()) <= 3e-4 * scale, "box kernel vs general kernel"
    assert not torch.equal(box, gen), "the box kernel did not run"   # (different arithmetic: equal bits would mean the same kernel)
    assert torch.equal(box, again), "not reproducible"
    bent = verts.clone()
    bent[:, nQ // 3, 5] += 0.01   # one corner of one query off its box: the whole launch goes down the general path
    monkeypatch.setattr(A, "BWD_KERNEL", 1)
    _, gen2 = run(bent)
    monkeypatch.setattr(A, "BWD_KERNEL", 0)
    _, box2 = run(bent)
    assert torch.equal(gen2, box2), "a non-box query must keep the general kernel"


def test_attention_probabilities_and_dropout_statistics():
    from oracle.attention_oracle import fused_attention_reference
    from vdetr_amd import attention as A
    B, nQ, nK, H = 1, 32, 256, 4
    xyz, verts, tables, _ = _scene(B, nQ, nK, 9)
    g = torch.Generator().manual_seed(1)
    q, k, v = torch.randn((B, nQ, 256), generator=g), torch.randn((B, nK, 64), generator=g), torch.randn((B, nK, 64), generator=g)
    cfg = A.RPEConfig()
    common = dict(num_heads=H, scale=0.125, shared_kv=True, rpe=cfg)
    dev = dict(table=tables.to(DEV), vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))
    _, probs = fused_attention_reference(q.double(), k.double(), v.double(), table=tables.double(), vertices=verts.double(),
                                         xyz=xyz.double(), return_probs=True, **common)
    got = A.attention_probabilities(q.to(DEV), k.to(DEV), **common, **dev)
    assert_close(got, probs.numpy(), 1e-4, 1e-7, "probs")
    # dropout: the kernels' keep mask has the right rate, differs per salt / step, and fwd+bwd are consistent with it
    p = 0.1
    rng = A.begin_step(DEV)
    keep = A.dropout_keep_mask(B, H, nQ, nK, True, p, rng, salt=5)
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    assert not torch.equal(keep, A.dropout_keep_mask(B, H, nQ, nK, True, p, rng, salt=6))
    assert not torch.equal(keep, A.dropout_keep_mask(B, H, nQ, nK, True, p, A.begin_step(DEV), salt=5))
    qd, kd, vd, td = (x.to(DEV).requires_grad_(True) for x in (q, k, v, tables))
    out = A.fused_attention(qd, kd, vd, **common, table=td, vertices=dev["vertices"], xyz=dev["xyz"], dropout_p=p,
                            rng_state=rng, salt=5)
    out.sum().backward()
    qr, kr, vr, tr = (x.double().requires_grad_(True) for x in (q, k, v, tables))
    ref = fused_attention_reference(qr, kr, vr, table=tr, vertices=verts.double(), xyz=xyz.double(), dropout_p=p,
                                    keep_mask=keep.cpu(), **common)
    ref.sum().backward()
    assert_close(out, ref.detach().numpy(), 1e-4, 1e-5, "dropout out")
    for name, a, b in [("dq", qd, qr), ("dk", kd, kr), ("dv", vd, vr), ("dtable", td, tr)]:
        assert_close(a.grad, b.grad.numpy(), 1e-3, 1e-4 * float(b.grad.abs().max()), "dropout " + name)


def test_multihead_self_attention_matches_torch():
    """MultiheadSelfAttention (HIP core) vs torch.nn.MultiheadAttention with the same parameters."""
    from vdetr_amd.vdetr_transformer import MultiheadSelfAttention
    torch.manual_seed(0)
    ref = torch.nn.MultiheadAttention(256, 4, dropout=0.1).eval()
    mine = MultiheadSelfAttention(256, 4, dropout=0.1).eval()
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    L_, B = 70, 2
    tgt, pos = torch.randn(L_, B, 256), torch.randn(L_, B, 256)
    a = tgt.clone().requires_grad_(True)
    out_ref = ref(a + pos, a + pos, value=a)[0]
    out_ref.square().sum().backward()
    b = tgt.clone().to(DEV).requires_grad_(True)
    qk = b + pos.to(DEV)
    out = mine(qk, qk, value=b)[0]
    out.square().sum().backward()
    assert_close(out, out_ref.detach().numpy(), 1e-4, 1e-5, "mha out")
    assert_close(b.grad, a.grad.numpy(), 1e-3, 1e-4 * float(a.grad.abs().max()), "mha dgrad")
    for (n1, p1), (n2, p2) in zip(sorted(ref.named_parameters()), sorted(mine.named_parameters())):
        assert n1 == n2
        assert_close(p2.grad, p1.grad.numpy(), 1e-3, 1e-4 * float(p1.grad.abs().max()) + 1e-7, n1)


@pytest.mark.parametrize("B,n,p", [(1, 1024, 0.0), (1, 1024, 0.1), (2, 1000, 0.0), (1, 4096, 0.0)])
def test_per_head_attention_at_the_models_size_vs_torch(B, n, p):
    """The query self-attention at BASELINE config 2's size (1024 queries, 4 heads: 256 workgroups = one per CU) and beyond takes
    the four-wave workgroups of attn_fwd.hip (several per CU: they all start next to a CU that is busy elsewhere); the smaller
    cases of this file take the eight-wave form.  Forward and gradients against softmax(q k^T / 8) v in fp64; with dropout the
    kernel's own keep mask (dropout_keep_mask) is applied to the reference."""
    from vdetr_amd import attention as A
    g = torch.Generator().manual_seed(n + B)
    q, k, v = (torch.randn((B, n, 256), generator=g).to(DEV).requires_grad_(True) for _ in range(3))
    wout = torch.randn((B, n, 256), generator=g).to(DEV)
    A.reset_rng()
    rng = A.begin_step(DEV) if p > 0 else None
    out = A.fused_attention(q, k, v, num_heads=4, scale=0.125, shared_kv=False, dropout_p=p, rng_state=rng, salt=5)
    (out * wout).sum().backward()
    qd, kd, vd = (t_.detach().double().view(B, n, 4, 64).transpose(1, 2).requires_grad_(True) for t_ in (q, k, v))
    prob = torch.softmax(qd @ kd.transpose(2, 3) * 0.125, dim=-1)
    if p > 0:
        keep = A.dropout_keep_mask(B, 4, n, n, False, p, rng, salt=5).double()
        assert 0.85 < float(keep.mean()) < 0.95
        prob = prob * keep / (1.0 - round(p * 65536) / 65536.0)
    ref = (prob @ vd).transpose(1, 2).reshape(B, n, 256)
    (ref * wout.double()).sum().backward()
    assert_close(out, ref.detach().cpu().numpy(), 1e-4, 1e-5, "out")
    for name, got, want in (("dq", q.grad, qd.grad), ("dk", k.grad, kd.grad), ("dv", v.grad, vd.grad)):
        want = want.transpose(1, 2).reshape(B, n, 256)
        assert_close(got, want.cpu().numpy(), 1e-3, 2e-5 * float(want.abs().max()), name)


@pytest.mark.parametrize("B,n,p", [(1, 1024, 0.1), (2, 256, 0.0), (1, 384, 0.3), (4, 1024, 0.1), (3, 128, 0.2)])
def test_lean_self_attention_kernel_equals_the_general_body(monkeypatch, B, n, p):
    """attn_fwd_self.hip (no mask, nQ % 16 == 0, nK % 128 == 0: two key tiles per step, hoisted dropout counters) against
    attn_fwd.hip's body on the same inputs and the same dropout state (vdetr_attn_desc.fwd_kernel = 4): the same keep mask, outputs
    and gradients equal up to the order of the key sums."""
    from vdetr_amd import attention as A
    g = torch.Generator().manual_seed(7 * n + B)
    base = [torch.randn((B, n, 256), generator=g).to(DEV) for _ in range(3)]
    wout = torch.randn((B, n, 256), generator=g).to(DEV)
    res = {}
    for body in (False, True):
        monkeypatch.setattr(A, "SELF_FWD_BODY", body)
        A.reset_rng()
        rng = A.begin_step(DEV) if p > 0 else None
        q, k, v = (t_.clone().requires_grad_(True) for t_ in base)
        out = A.fused_attention(q, k, v, num_heads=4, scale=0.125, shared_kv=False, dropout_p=p, rng_state=rng, salt=3)
        (out * wout).sum().backward()
        res[body] = (out.detach(), q.grad, k.grad, v.grad)
    for name, a, b in zip(("out", "dq", "dk", "dv"), res[False], res[True]):
        assert_close(a, b.cpu().numpy(), 1e-5, 1e-5 * float(b.abs().max()), name)  # (measured: 3e-6 of the largest entry)
    if p > 0:  # a dropped probability is an exact zero in both or in neither: the same mask (dv rows see every kept pair)
        assert float((res[False][0] - res[True][0]).abs().max()) < 1e-4 * float(res[True][0].abs().max())


@pytest.mark.parametrize("case", ["cross_attn_small", "cross_attn_rot", "cross_attn_mid"])
def test_cross_attention_module_vs_reference_vectors(case):
    g = load_golden(case)
    mod = build_cross_attention(str(g["angle_type"]), DEV)
    res = run_cross_attention_case(g, mod, DEV)
    assert_close(res["x"], g["x"], 1e-3, 1e-5, "x")
    assert_close(res["attn"], g["attn"], 1e-3, 1e-7, "attn")
    for k in g.files:
        if k.startswith("grad_"):
            assert_close(res[k], g[k], 1e-3, grad_atol(g, k, 2e-4), k)


def test_share_self_attention_module_vs_reference_vectors():
    g = load_golden("share_self_attn")
    mod = build_share_self_attention(DEV)
    tgt, pos = t(g["tgt"], DEV, grad=True), t(g["pos"], DEV)
    x, _ = mod(tgt + pos, tgt + pos, value=tgt)
    (x * t(g["wout"], DEV)).sum().backward()
    assert_close(x, g["x"], 1e-3, 1e-5, "x")
    # (2e-5 of the largest entry: the key-side contractions run on split-bf16 products, 2^-17 per product; measured 6e-6)
    assert_close(tgt.grad, g["grad_tgt"], 1e-3, 2e-5 * float(np.abs(g["grad_tgt"]).max()), "grad_tgt")
    for pname, p in mod.named_parameters():
        assert_close(p.grad, g["grad_param:" + pname], 1e-3, grad_atol(g, "grad_param:" + pname, 2e-4), pname)


@pytest.mark.parametrize("case,nl,share", [("decoder_c1_l2", 2, False), ("decoder_c1_l3", 3, False),
                                           ("decoder_c1_l3_share", 3, True)])
def test_decoder_vs_reference_vectors(case, nl, share):
    """BASELINE config 1: the whole decoder (HIP attention cores) against the reference's outputs + gradients."""
    g = load_golden(case)
    dec = build_decoder(nl, share, DEV)
    stages, loss, gfeats = run_decoder_case(g, dec, DEV)
    for s, st in enumerate(stages):
        for k in ("sem_cls_logits", "center_unnormalized", "size_unnormalized", "box_corners"):
            assert_close(st[k], g[f"s{s}:{k}"], 1e-3, 2e-4, f"stage {s} {k}")   # 1e-3 relative (north_star)
    assert_close(loss, g["loss"], 1e-3, 1e-2, "loss")
    # gradients: 1e-3 relative (north_star) + 1e-4 of the tensor's largest entry (the golden vectors are the reference's own
    # fp32 CPU run: its summation-order noise is of that size; measured need <= 6e-5, worst in the cpb_mlps of the last layer)
    assert_close(gfeats, g["grad_feats"], 1e-3, 1e-4 * np.abs(g["grad_feats"]).max(), "grad_feats")
    params = dict(dec.named_parameters())
    for k in g.files:
        if k.startswith("grad_param:"):
            assert_close(params[k[11:]].grad, g[k], 1e-3, grad_atol(g, k, 1e-4), k)


@pytest.mark.parametrize("B", [1, 2])
def test_row_strided_kv_equals_contiguous(B):
    """K / V as 64-wide column blocks of a wider projection output (k_row_stride / v_row_stride of the ABI): the output
    is bit-identical to the dense call, the gradients agree to rounding."""
    from vdetr_amd import attention as A
    nQ, nK, H = 37, 203, 4
    g = torch.Generator().manual_seed(5)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 11)
    q = torch.randn((B, nQ, 256), generator=g).to(DEV)
    kv = torch.randn((B, nK, 6, 64), generator=g).to(DEV)
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)
    res = []
    for strided in (False, True):
        qq = q.clone().requires_grad_(True)
        kvv = kv.clone().requires_grad_(True)
        parts = kvv.unbind(2)
        k, v = parts[1], parts[4]
        assert not k.is_contiguous()
        if not strided:
            k, v = k.contiguous(), v.contiguous()
        tb = tables.to(DEV).requires_grad_(True)
        out = A.fused_attention(qq, k, v, num_heads=H, scale=0.125, shared_kv=True, table=tb, rpe=A.RPEConfig(),
                                vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV))
        (out * wout).sum().backward()
        res.append((out.detach(), qq.grad, kvv.grad, tb.grad))
    for name, a, b in zip(("out", "dq", "dkv", "dtable"), *res):
        if name == "out":
            assert torch.equal(a, b), name  # same kernel, same arithmetic, different addresses
        else:  # library GEMMs pick other tilings for other strides; the table reduction sums its partials atomically
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()), msg=name)


@pytest.mark.parametrize("B,nQ,nK,boxes", [(1, 64, 512, True), (2, 37, 301, False), (1, 1024, 4096, True)])
def test_bf16_attention_core_vs_oracle(B, nQ, nK, boxes):
    """BASELINE config 4's arithmetic: q, k, v stored as bf16, QK^T / PV on the bf16 matrix instructions, everything else
    fp32.  Against the fp64 oracle evaluated on the SAME bf16-rounded operands the remaining error is the probabilities'
    rounding to bf16 in front of PV (2^-9 relative per term) and the fp32 accumulation: 1e-2 of the output scale forward, 2e-2
    of the gradients' scale backward (stated tolerance for the bf16 configuration; the fp32 path keeps 1e-3)."""
    from oracle.attention_oracle import fused_attention_reference
    from vdetr_amd import attention as A
    H = 4
    g = torch.Generator().manual_seed(nQ + nK)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 9)
    if not boxes:
        verts[:, ::5] += 0.05 * torch.randn(verts[:, ::5].shape, generator=g)
    q, k, v = (torch.randn(s, generator=g).bfloat16() for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g)
    kw = dict(num_heads=H, scale=0.125, shared_kv=True, rpe=A.RPEConfig())
    dq, dk, dv = (x.to(DEV).requires_grad_(True) for x in (q, k, v))
    dtb = tables.to(DEV).requires_grad_(True)
    out = A.fused_attention(dq, dk, dv, table=dtb, vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV), **kw)
    assert out.dtype == torch.float32 and dq.dtype == torch.bfloat16
    (out * wout.to(DEV)).sum().backward()
    assert dq.grad.dtype == torch.bfloat16
    big = nQ * nK >= (1 << 21)  # (the full-size case: the fp64 oracle through ATen's GPU kernels, see _oracle_in_chunks)
    ref = _oracle_in_chunks(q, k, v, tables, verts, xyz, wout, kw, None, 256 if big else 64, DEV if big else "cpu")
    got = [out.detach(), dq.grad, dk.grad, dv.grad, dtb.grad]
    for name, r, o in zip(["out", "dq", "dk", "dv", "dtable"], ref, got):
        scale = float(r.abs().max())
        tol = 1e-2 if name == "out" else 2e-2
        assert_close(o.float(), r.numpy(), tol, tol * scale, name)
        rel = float((o.float().cpu().double() - r).norm() / r.norm())
        assert rel < 5e-3, f"{name}: relative L2 error {rel:.2e}"


@pytest.mark.parametrize("B,nQ,nK,boxes", [(1, 64, 512, True), (2, 37, 301, False), (1, 1024, 4096, True)])
def test_rounded_operand_attention_vs_oracle(B, nQ, nK, boxes):
    """fused_attention(operand_bf16=True) — vdetr_attn_desc.fwd_kernel 3: f32 q / k / v, rounded to bf16 inside the kernels (the
    bf16 configuration without bf16 tensors).  Forward against the fp64 oracle on the ROUNDED operands at the bf16 path's stated
    1e-2; gradients (the rounding's derivative taken as the identity, the backward products on the unrounded f32 operands: 2^-9
    relative per operand next to the oracle's) at its 2e-2.  Images packed ahead with one part give the identical output."""
    from oracle.attention_oracle import fused_attention_reference
    from vdetr_amd import _lib as L
    from vdetr_amd import attention as A
    H = 4
    g = torch.Generator().manual_seed(nQ + nK + 1)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 9)
    if not boxes:
        verts[:, ::5] += 0.05 * torch.randn(verts[:, ::5].shape, generator=g)
    q = torch.randn((B, nQ, 256), generator=g)
    kv = torch.randn((B, nK, 128), generator=g)
    wout = torch.randn((B, nQ, 256), generator=g)
    kw = dict(num_heads=H, scale=0.125, shared_kv=True, rpe=A.RPEConfig())
    dq = q.to(DEV).requires_grad_(True)
    dkv = kv.to(DEV).requires_grad_(True)
    dtb = tables.to(DEV).requires_grad_(True)
    dev = dict(table=dtb, vertices=verts.to(DEV).contiguous(), xyz=xyz.to(DEV), operand_bf16=True, **kw)
    out = A.fused_attention(dq, dkv[..., :64], dkv[..., 64:], **dev)
    assert out.dtype == torch.float32
    (out * wout.to(DEV)).sum().backward()
    assert dq.grad.dtype == torch.float32
    img = A.pack_kv_images(dkv.detach(), 1, parts=1)
    assert img is not None and img[0].data.numel() == L.lib().vdetr_attn_kv_image_parts_bytes(B, nK, 1) < L.lib().vdetr_attn_kv_image_bytes(B, nK)
    with torch.no_grad():
        out_img = A.fused_attention(dq, dkv[..., :64], dkv[..., 64:], kv_img=img[0], **dev)
        out_f32 = A.fused_attention(dq, dkv[..., :64], dkv[..., 64:], **{**dev, "operand_bf16": False})
    assert torch.equal(out_img, out.detach())
    assert not torch.equal(out_f32, out.detach())  # (the rounded operands are really what the kernel multiplied)
    big = nQ * nK >= (1 << 21)
    r_ = _oracle_in_chunks(q.bfloat16(), kv[..., :64].bfloat16(), kv[..., 64:].bfloat16(), tables, verts, xyz, wout, kw, None,
                           256 if big else 64, DEV if big else "cpu")
    ref = [r_[0], r_[1], torch.cat((r_[2], r_[3]), -1), r_[4]]
    got = [out.detach(), dq.grad, dkv.grad, dtb.grad]
    for name, r, o in zip(["out", "dq", "dkv", "dtable"], ref, got):
        scale = float(r.abs().max())
        tol = 1e-2 if name == "out" else 2e-2
        assert_close(o.float(), r.numpy(), tol, tol * scale, name)
        rel = float((o.float().cpu().double() - r).norm() / r.norm())
        assert rel < 5e-3, f"{name}: relative L2 error {rel:.2e}"


def test_bf16_decoder_layer_close_to_fp32():
    """set_attention_dtype(bf16) on the cross attention module: the module's output stays within 1e-2 of the fp32 module's."""
    from vdetr_amd.vdetr_transformer import set_attention_dtype
    g = load_golden("cross_attn_mid")
    mod = build_cross_attention(str(g["angle_type"]), DEV)
    ref = {k: v.detach().clone() for k, v in run_cross_attention_case(g, mod, DEV).items() if v is not None}
    mod.zero_grad(set_to_none=True)
    assert set_attention_dtype(mod, torch.bfloat16) == 1
    got = run_cross_attention_case(g, mod, DEV)
    set_attention_dtype(mod, torch.float32)
    assert_close(got["x"], ref["x"].detach().cpu().numpy(), 2e-2, 2e-2 * float(ref["x"].abs().max()), "x")
    # (k.bias has a mathematically ZERO gradient — a constant added to every key shifts all scores of a row alike — so
    #  the absolute tolerance is tied to the largest parameter gradient, not to each tensor's own)
    gmax = max(float(ref[k].abs().max()) for k in ref if k.startswith("grad_param:"))
    for k in ref:
        if k.startswith("grad_"):
            r = ref[k].cpu().numpy()
            assert_close(got[k], r, 5e-2, 3e-2 * max(np.abs(r).max(), 0.05 * gmax) + 1e-7, k)


@pytest.mark.parametrize("B,nQ,nK,n", [(1, 64, 512, 3), (2, 37, 203, 2), (1, 1024, 4096, 8)])
def test_kv_images_packed_ahead_equal_the_forwards_own(B, nQ, nK, n):
    """vdetr_attn_pack_kv_f32 (the K / V operand images of n layers' forwards in one launch, from their joint projection
    [B, nK, n * 128]) + vdetr_attn_desc.kv_img: every layer's forward returns the bits it returns when it packs its own."""
    from vdetr_amd import attention as A
    g = torch.Generator().manual_seed(B * 1000 + nK)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 9)
    kv = torch.randn((B, nK, n * 128), generator=g).to(DEV)
    q = torch.randn((B, nQ, 256), generator=g).to(DEV)
    imgs = A.pack_kv_images(kv, n)
    assert imgs is not None and len(imgs) == n and all((im.parts, im.B, im.nK) == (3, B, nK) for im in imgs)
    kw = dict(num_heads=4, scale=0.125, shared_kv=True, table=tables.to(DEV), rpe=A.RPEConfig(), vertices=verts.to(DEV).contiguous(),
              xyz=xyz.to(DEV))
    parts = kv.view(B, nK, 2 * n, 64).unbind(2)
    for i in range(n):
        own = A.fused_attention(q, parts[2 * i], parts[2 * i + 1], **kw)
        ahead = A.fused_attention(q, parts[2 * i], parts[2 * i + 1], kv_img=imgs[i], **kw)
        assert torch.equal(own, ahead), f"layer {i}"
    # an image packed for another part count or other sizes is refused, not read
    one_part = A.pack_kv_images(kv, n, parts=1)
    with pytest.raises(RuntimeError, match="kv_img was packed for parts=1"):
        A.fused_attention(q, parts[0], parts[1], kv_img=one_part[0], **kw)
    A.fused_attention(q, parts[0], parts[1], kv_img=one_part[0], operand_bf16=True, **kw)
    if nK > 16:
        short = A.pack_kv_images(kv[:, :nK - 16].contiguous(), n)
        with pytest.raises(RuntimeError, match="kv_img was packed for"):
            A.fused_attention(q, parts[0], parts[1], kv_img=short[0], **kw)
    with pytest.raises(TypeError):
        A.fused_attention(q, parts[0], parts[1], kv_img=imgs[0].data, **kw)


def test_table_gradient_with_boxes_vouched_for():
    """fused_attention(vertices_are_boxes=True) -> vdetr_attn_desc.bwd_kernel = 2: the box kernel alone (the general kernel's launch
    is saved).  Same table gradient, bit for bit, as the default for boxes; NaN — not a silently wrong gradient — where a query is
    not a box after all."""
    from vdetr_amd import attention as A
    B, nQ, nK = 1, 96, 700
    g = torch.Generator().manual_seed(12)
    xyz, verts, tables, _ = _scene(B, nQ, nK, 21)
    q, k, v = (torch.randn(s, generator=g).to(DEV) for s in ((B, nQ, 256), (B, nK, 64), (B, nK, 64)))
    wout = torch.randn((B, nQ, 256), generator=g).to(DEV)

    def run(vv, vouch):
        tb = tables.to(DEV).requires_grad_(True)
        out = A.fused_attention(q, k, v, num_heads=4, scale=0.125, shared_kv=True, table=tb, rpe=A.RPEConfig(),
                                vertices=vv.to(DEV).contiguous(), xyz=xyz.to(DEV), vertices_are_boxes=vouch)
        (out * wout).sum().backward()
        return tb.grad

    ref, got = run(verts, False), run(verts, True)
    assert torch.isfinite(got).all() and torch.equal(ref, got)
    bent = verts.clone()
    bent[0, 5, 3, 0] += 0.01  # one vertex of one query off its box
    assert torch.isfinite(run(bent, False)).all()
    assert torch.isnan(run(bent, True)).all()
