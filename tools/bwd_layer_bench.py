#!/usr/bin/env python3
"""One decoder layer's cross-attention backward at the bench's size (B=1, nQ=1024, nK=4096, 4 heads), both ways:
the library-GEMM path (dO V^T, element-wise + table kernel, three contractions) and the fused key-side pass
(attn_bwd_kv.hip + table kernel on the given dS + the dQ GEMM).  HIP events around autograd's backward, and around the
C entry points alone.

    python tools/bwd_layer_bench.py [c2|c4]
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timeit(fn, reps=20, prep=None):
    ts = []
    for i in range(reps + 3):
        if prep is not None:
            prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    from vdetr_amd import _lib as L
    from vdetr_amd import attention as A
    from vdetr_amd.pc_util import morton_argsort
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    _, bs, nK, nQ, *_ = bench.CONFIGS[cfg]
    dev = torch.device("cuda")
    B, H = bs, 4
    g = torch.Generator().manual_seed(0)
    xyz, _ = bench.make_scene(40000, 0, dev)
    kxyz = xyz[torch.randperm(xyz.shape[0], generator=g)[:nK].to(dev)][None].repeat(B, 1, 1).contiguous()
    kxyz = torch.gather(kxyz, 1, morton_argsort(kxyz).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    center = kxyz[:, torch.randperm(nK, generator=g)[:nQ].to(dev)]
    half = (0.1 + torch.rand((B, nQ, 1, 3), generator=g)).to(dev)
    signs = torch.tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]],
                         dtype=torch.float32, device=dev)
    verts = (center[:, :, None, :] + half * signs).contiguous()
    q = torch.randn((B, nQ, 256), generator=g).to(dev).requires_grad_(True)
    k = torch.randn((B, nK, 64), generator=g).to(dev).requires_grad_(True)
    v = torch.randn((B, nK, 64), generator=g).to(dev).requires_grad_(True)
    table = torch.randn((8, 10, 10, 10, 4), generator=g).to(dev).requires_grad_(True)
    dout = torch.randn((B, nQ, 256), generator=g).to(dev) * 1e-3
    kw = dict(num_heads=H, scale=0.125, shared_kv=True, rpe=A.RPEConfig(), vertices=verts, xyz=kxyz, dropout_p=0.1)
    res = {}
    grads = {}
    rng_once = A.new_rng_state(dev, 1234)  # ONE dropout state for both paths: their gradients are then comparable
    for name, fused in (("gemm_path", False), ("fused_path", True)):
        A.FUSED_KV_BWD = fused
        A.begin_step(dev)
        out = A.fused_attention(q, k, v, table=table, rng_state=rng_once.clone(), **kw)

        def bwd():
            torch.autograd.grad(out, (q, k, v, table), dout, retain_graph=True)

        res[name + "_backward_us"] = timeit(bwd, prep=lambda: A.begin_step(dev))  # (a fresh pool of zeroed scratch per run)
        grads[name] = torch.autograd.grad(out, (q, k, v, table), dout, retain_graph=True)
    for i, nm in enumerate(("dq", "dk", "dv", "dtable")):
        a, b = grads["gemm_path"][i], grads["fused_path"][i]
        res["maxdiff_" + nm] = float((a - b).abs().max() / a.abs().max())

    # the fused pieces alone, through the C entry points
    lib = L.lib()
    st = L.stream_ptr()
    rng = A.begin_step(dev)
    d = A._desc(L.VDETR_ATTN_SHARED_KV, B, H, nQ, nK, 0.125, table.detach(), A.RPEConfig(), verts, kxyz, None, None, 0.1, rng, 1)
    o = torch.empty((B, nQ, 256), device=dev)
    lse = torch.empty((B, nQ, H), device=dev)
    scores = torch.empty((B, nQ, H, nK), device=dev)
    wsf = lib.vdetr_attn_fwd_workspace_bytes(ctypes.byref(d))
    ws = L.workspace(max(wsf, 1), dev)
    qd, kd, vd = q.detach(), k.detach(), v.detach()
    L.check(lib.vdetr_attn_fwd_f32(ctypes.byref(d), L.ptr(qd), L.ptr(kd), L.ptr(vd), L.ptr(o), L.ptr(lse), L.ptr(scores), L.ptr(ws), wsf, st), "fwd")
    delta = torch.zeros((B, nQ, H), device=dev)
    aux = torch.zeros(8, dtype=torch.int32, device=dev)
    ds = torch.empty_like(scores)
    dkv = torch.empty((2, B, nK, 64), device=dev)
    dtable = torch.zeros((8, 10, 10, 10, 4), device=dev)
    wkv = lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(d))
    ws_kv = torch.empty(wkv, dtype=torch.uint8, device=dev)
    wtb = lib.vdetr_attn_bwd_workspace_bytes(ctypes.byref(d))
    ws_tb = torch.empty(wtb, dtype=torch.uint8, device=dev)

    def prep():
        aux.zero_()
        d.bwd_aux = aux.data_ptr()
        L.check(lib.vdetr_attn_delta_f32(ctypes.byref(d), L.ptr(dout), L.ptr(o), L.ptr(vd), L.ptr(delta), st), "delta")

    def kv():
        L.check(lib.vdetr_attn_bwd_kv_f32(ctypes.byref(d), L.ptr(qd), L.ptr(vd), L.ptr(dout), L.ptr(scores), L.ptr(lse), L.ptr(delta),
                                          L.ptr(ds), L.ptr(dkv[0]), L.ptr(dkv[1]), L.ptr(ws_kv), wkv, st), "kv")

    def tb():
        L.check(lib.vdetr_attn_bwd_table_f32(ctypes.byref(d), L.ptr(ds), L.ptr(dtable), L.ptr(ws_tb), wtb, st), "table")

    dq = torch.empty((B, nQ * H, 64), device=dev)

    def dqg():
        torch.baddbmm(dq, ds.view(B, nQ * H, nK), kd, beta=0.0, alpha=0.125, out=dq)

    prep()
    res["kv_pass_us (pack + attn_bwd_kv_kernel)"] = timeit(kv)
    res["table_from_ds_us (box4 + reduce)"] = timeit(tb, prep=prep)
    res["dq_gemm_us"] = timeit(dqg)
    res["delta_us"] = timeit(lambda: L.check(lib.vdetr_attn_delta_f32(ctypes.byref(d), L.ptr(dout), L.ptr(o), L.ptr(vd), L.ptr(delta), st), "delta"),
                             prep=lambda: aux.zero_())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
