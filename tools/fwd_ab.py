#!/usr/bin/env python3
"""A/B of the two forward kernels of the 3DV-RPE attention (vdetr_attn_desc.fwd_kernel: 0 = persistent workgroups,
attn_fwd_pipe.hip; 1 = one workgroup per (query quad, key chunk), attn_fwd.hip): same inputs, outputs compared, launches timed
interleaved with HIP events in one process.
    python tools/fwd_ab.py [c2|c5] [--reps N] [--with-fps] [--cases]
--cases: small ragged shapes (nQ % 4 != 0, nK % 16 != 0, general vertices, no dropout / no stored scores) checked against the
grid kernel as well."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vdetr_amd import _lib as L  # noqa: E402
from vdetr_amd import attention as A  # noqa: E402

SIGNS = [[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]


def make_case(B, nQ, nK, rot, general, device, seed=0, npts=40000):
    from vdetr_amd.pc_util import morton_argsort
    g = torch.Generator().manual_seed(seed)
    xyz, _ = bench.make_scene(npts, 0, device)
    kxyz = xyz[torch.randperm(xyz.shape[0], generator=g)[:nK].to(device)][None].repeat(B, 1, 1).contiguous()
    kxyz = torch.gather(kxyz, 1, morton_argsort(kxyz).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    center = kxyz[:, torch.randint(0, nK, (nQ,), generator=g).to(device)]
    half = (0.1 + torch.rand((B, nQ, 1, 3), generator=g)).to(device)
    signs = torch.tensor(SIGNS, dtype=torch.float32, device=device)
    cos_sin = None
    off = half * signs
    if rot:
        ang = ((torch.rand((B, nQ), generator=g) * 2 - 1) * 3.1).to(device)
        c, sn = torch.cos(ang)[:, :, None], torch.sin(ang)[:, :, None]
        off = torch.stack((off[..., 0] * c + off[..., 1] * sn, -off[..., 0] * sn + off[..., 1] * c, off[..., 2]), -1)
        cos_sin = torch.stack((c[..., 0], sn[..., 0]), -1).contiguous()
    verts = (center[:, :, None, :] + off).contiguous()
    if general:  # every third query: eight unrelated vertices
        noise = torch.randn((B, nQ, 8, 3), generator=g).to(device) * 0.3
        pick = (torch.arange(nQ, device=device) % 3 == 0)[None, :, None, None]
        verts = torch.where(pick, verts + noise, verts).contiguous()
    q = torch.randn((B, nQ, 256), generator=g).to(device)
    k = torch.randn((B, nK, 64), generator=g).to(device)
    v = torch.randn((B, nK, 64), generator=g).to(device)
    table = torch.randn((8, 10, 10, 10, 4), generator=g).to(device)
    return dict(B=B, nQ=nQ, nK=nK, q=q, k=k, v=v, table=table, verts=verts, kxyz=kxyz, cos_sin=cos_sin)


class Runner:
    def __init__(self, case, kernel, dropout, store_scores, rng):
        c = case
        self.c = c
        dev = c["q"].device
        self.d = A._desc(L.VDETR_ATTN_SHARED_KV, c["B"], 4, c["nQ"], c["nK"], 0.125, c["table"], A.RPEConfig(), c["verts"], c["kxyz"],
                         c["cos_sin"], None, dropout, rng if dropout > 0 else None, 1)
        self.d.fwd_kernel = kernel
        self.out = torch.empty_like(c["q"])
        self.lse = torch.empty((c["B"], c["nQ"], 4), device=dev)
        self.scores = torch.empty((c["B"], c["nQ"], 4, c["nK"]), device=dev) if store_scores else None
        self.lib = L.lib()
        self.nws = self.lib.vdetr_attn_fwd_workspace_bytes(ctypes.byref(self.d))
        self.ws = L.workspace(self.nws, dev)

    def __call__(self):
        c = self.c
        L.check(self.lib.vdetr_attn_fwd_f32(ctypes.byref(self.d), L.ptr(c["q"]), L.ptr(c["k"]), L.ptr(c["v"]), L.ptr(self.out),
                                            L.ptr(self.lse), L.ptr(self.scores), L.ptr(self.ws), self.nws, L.stream_ptr()), "attn_fwd")


NAMES = {0: "split", 2: "pipe", 1: "grid"}


def compare(case, dropout, store_scores, rng, label):
    return all([compare_one(case, dropout, store_scores, rng, f"{label} [{NAMES[k]}]", k) for k in (0, 2)])


def compare_one(case, dropout, store_scores, rng, label, kernel):
    new, old = Runner(case, kernel, dropout, store_scores, rng), Runner(case, 1, dropout, store_scores, rng)
    new.out.fill_(float("nan")); new.lse.fill_(float("nan"))
    new(); old()
    torch.cuda.synchronize()
    res = {"case": label, "out_maxdiff": float((new.out - old.out).abs().max()), "out_scale": float(old.out.abs().max()),
           "lse_maxdiff": float((new.lse - old.lse).abs().max())}
    if store_scores:
        res["scores_maxdiff"] = float((new.scores - old.scores).abs().max())
        res["scores_scale"] = float(old.scores.abs().max())
    res["ok"] = bool(res["out_maxdiff"] <= 2e-5 * max(1.0, res["out_scale"]) and res["lse_maxdiff"] <= 5e-5 and
                     res.get("scores_maxdiff", 0.0) <= 2e-5 * max(1.0, res.get("scores_scale", 1.0)))
    # a second launch on the same counter word must behave the same (the launch leaves it zero)
    o1 = new.out.clone()
    new()
    torch.cuda.synchronize()
    res["relaunch_identical"] = bool(torch.equal(o1, new.out))
    print(json.dumps(res), flush=True)
    return res["ok"] and res["relaunch_identical"]


def time_pair(case, dropout, reps, rng):
    runners = {k: Runner(case, k, dropout, True, rng) for k in (0, 2, 1)}
    ts = {0: [], 1: [], 2: []}
    for i in range(reps + 3):
        for kern, r in runners.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r()
            e1.record()
            e1.synchronize()
            if i >= 3:
                ts[kern].append(e0.elapsed_time(e1) * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    return {"split_us": med(ts[0]), "split_min_us": min(ts[0]), "pipe_us": med(ts[2]), "pipe_min_us": min(ts[2]),
            "grid_us": med(ts[1]), "grid_min_us": min(ts[1])}


if __name__ == "__main__":
    if "--lib" in sys.argv:  # a variant build of the library (experiments): before the first call loads the default one
        L.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    dev = torch.device("cuda")
    cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c2"
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    rng = A.begin_step(dev)
    ok = True
    if "--only" in sys.argv:  # one kernel, `reps` launches: the process a rocprofv3 --pmc pass wraps
        kern = {"split": 0, "pipe": 2, "grid": 1}[sys.argv[sys.argv.index("--only") + 1]]
        _, bs, nK, nQ, *_ = bench.CONFIGS[cfg]
        r = Runner(make_case(bs, nQ, nK, bench.CONFIGS[cfg][5] == "object_coords", False, dev), kern, 0.1, True, rng)
        for _ in range(reps):
            r()
        torch.cuda.synchronize()
        sys.exit(0)
    if "--cases" in sys.argv:
        for (B, nQ, nK, rot, general, drop, store) in [(1, 64, 512, False, False, 0.1, True), (2, 37, 203, False, False, 0.0, True),
                                                       (2, 37, 203, True, False, 0.1, False), (1, 50, 1000, False, True, 0.0, True),
                                                       (1, 50, 1000, True, True, 0.1, True), (3, 5, 7, False, False, 0.0, True),
                                                       (1, 1024, 4096, False, False, 0.1, True), (4, 256, 1024, True, False, 0.1, True)]:
            case = make_case(B, nQ, nK, rot, general, dev, seed=nQ + nK)
            ok &= compare(case, drop, store, rng, f"B{B} nQ{nQ} nK{nK} rot{int(rot)} gen{int(general)} drop{drop} scores{int(store)}")
    _, bs, nK, nQ, *_ = bench.CONFIGS[cfg]
    rot = bench.CONFIGS[cfg][5] == "object_coords"
    case = make_case(bs, nQ, nK, rot, False, dev)
    ok &= compare(case, 0.1, True, rng, cfg)
    if "--with-fps" in sys.argv:  # one CU busy with the side-stream sampling, as in the training step
        from vdetr_amd import pointnet2_utils as PU
        pts = bench.make_scene(40000, 0, "cuda")[0][None].contiguous()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(12):
                PU.furthest_point_sample(pts, 4096)
    t = time_pair(case, 0.1, reps, rng)
    torch.cuda.synchronize()
    ok &= compare(case, 0.1, True, rng, cfg + " (after the timed launches)")
    t.update({"config": cfg, "with_fps": "--with-fps" in sys.argv, "all_ok": bool(ok)})
    print(json.dumps(t), flush=True)
    sys.exit(0 if ok else 1)
