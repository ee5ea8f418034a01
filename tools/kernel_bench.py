#!/usr/bin/env python3
"""Times the hand-written kernels one by one with HIP events (same code as bench.py's roofline leg) + FPS:
   python tools/kernel_bench.py [c2]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def time_fps(npts, m, reps=5):
    from vdetr_amd import pointnet2_utils as PU
    xyz, _ = bench.make_scene(npts, 0, "cuda")
    x = xyz[None].contiguous()
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PU.furthest_point_sample(x, m)
        e1.record()
        e1.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    return {"kernel": "fps", "n": int(x.shape[1]), "m": m, "ms": sum(ts) / len(ts), "us_per_round": sum(ts) / len(ts) * 1e3 / (m - 1)}


def _time(fn, reps=10):
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    return sum(ts) / len(ts)


def indexing_ops():
    """Stand-alone pointnet2 ops at the VoteNet-like shapes of SURVEY.md §8a (a11/a12); bytes = the algorithmic
    formulas of SURVEY.md §8d."""
    from vdetr_amd import pointnet2_utils as PU
    dev = "cuda"
    out = []
    xyz, _ = bench.make_scene(40000, 0, dev)
    xyz = xyz[None].contiguous()
    n = xyz.shape[1]
    g = torch.Generator().manual_seed(0)
    feats = torch.randn((1, 256, n), generator=g).to(dev)
    idx = PU.furthest_point_sample(xyz, 4096)
    t = _time(lambda: PU._ext.gather_points(feats, idx))
    out.append(("gather_points c=256 n=40k m=4096", t, 2 * 4 * 256 * 4096 + 4 * 4096))
    go = torch.randn((1, 256, 4096), generator=g).to(dev)
    t = _time(lambda: PU._ext.gather_points_grad(go, idx, n))
    out.append(("gather_points_grad c=256 n=40k m=4096 (rows accumulated in LDS, written once: no zero-fill, no global atomics)", t, 4 * 256 * n + 4 * 256 * 4096 + 4 * 4096))
    new_xyz = xyz[:, :2048].contiguous()
    t = _time(lambda: PU._ext.ball_query(new_xyz, xyz, 0.2, 64))
    out.append(("ball_query n=40k m=2048 r=0.2 nsample=64 (bytes = the 12nm scanned, out of LDS tiles shared by 8 queries; ~9 hits per query here: every scan runs to the end)", t, 12.0 * n * 2048))
    bidx = PU._ext.ball_query(new_xyz, xyz, 0.2, 64)
    f128 = feats[:, :128].contiguous()
    t = _time(lambda: PU._ext.group_points(f128, bidx))
    out.append(("group_points c=128 np=2048 ns=64", t, 2 * 4 * 128 * 2048 * 64 + 4 * 2048 * 64))
    gg = torch.randn((1, 128, 2048, 64), generator=g).to(dev)
    t = _time(lambda: PU._ext.group_points_grad(gg, bidx, n))
    out.append(("group_points_grad c=128 np=2048 ns=64 (rows accumulated in LDS, written once)", t, 4 * 128 * 2048 * 64 + 4 * 2048 * 64 + 4 * 128 * n))
    unk, kn = xyz[:, :2048].contiguous(), xyz[:, 5000:6024].contiguous()
    t = _time(lambda: PU._ext.three_nn(unk, kn))
    out.append(("three_nn n=2048 m=1024 (12nm scan bytes)", t, 12.0 * 2048 * 1024))
    d2, tidx = PU._ext.three_nn(unk, kn)
    w = torch.rand((1, 2048, 3), generator=g).to(dev)
    fk = torch.randn((1, 256, 1024), generator=g).to(dev)
    t = _time(lambda: PU._ext.three_interpolate(fk, tidx, w))
    out.append(("three_interpolate c=256 n=2048", t, 4 * 256 * 2048 * 4 + 2048 * 24))
    for name, t, nbytes in out:
        print(json.dumps({"kernel": name, "us": t * 1e6, "algorithmic_GB_s": nbytes / t / 1e9, "frac_of_8TBs": nbytes / t / 8e12}))


def lds_probe():
    from vdetr_amd import _lib as L
    sink = torch.zeros(4, device="cuda")
    iters = 2000
    for mode, name in [(0, "ds_add_f32 scattered"), (1, "ds_add_u32 scattered"), (2, "plain read-add-write"), (3, "ds_add_f32, 8 hot bins")]:
        t = _time(lambda: L.check(L.lib().vdetr_selftest_lds_atomics(mode, iters, L.ptr(sink), L.stream_ptr()), "probe"), reps=5)
        lane_ops = 256 * 512 * iters
        print(json.dumps({"lds_probe": name, "us": t * 1e6, "lane_ops_per_clk_per_CU": lane_ops / 256 / (t * 2.1e9)}))


def issue_probe():
    """Can a SIMD overlap MFMA with VALU / LDS issue?  (cycles per round per wave at an assumed 2.4 GHz)"""
    from vdetr_amd import _lib as L
    sink = torch.zeros(4, device="cuda")
    iters = 4000
    for mode, name in [(10, "16 mfma_f32_16x16x4"), (11, "128 v_fma_f32"), (12, "both, same wave"),
                       (13, "waves 0-3 mfma / waves 4-7 valu"), (14, "128 v_fma_f32 + 16 ds_read_b128"), (15, "16 mfma_f32_16x16x32_bf16"),
                       (16, "waves 0-3 bf16 mfma / waves 4-7 valu"), (17, "bf16 mfma + valu, same wave")]:
        t = _time(lambda: L.check(L.lib().vdetr_selftest_lds_atomics(mode, iters, L.ptr(sink), L.stream_ptr()), "probe"), reps=5)
        print(json.dumps({"issue_probe": name, "us": t * 1e6, "cycles_per_round_at_2.4GHz": t * 2.4e9 / iters}))


if __name__ == "__main__":
    if "--issue" in sys.argv:
        issue_probe()
        sys.exit(0)
    cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c2"
    if "--with-fps" in sys.argv:  # one CU busy with the side-stream FPS, as in the training step
        from vdetr_amd import pointnet2_utils as PU
        pts = bench.make_scene(40000, 0, "cuda")[0][None].contiguous()
        side = torch.cuda.Stream()
        _orig = bench.kernel_rooflines.__globals__["torch"].cuda.Event
        def keep_busy():
            with torch.cuda.stream(side):
                for _ in range(3):
                    PU.furthest_point_sample(pts, 4096)
        keep_busy()
    f, b = bench.kernel_rooflines(cfg, torch.device("cuda"))
    print(json.dumps({"variant": os.environ.get("VDETR_BWD_VARIANT", "default"), "fwd_us": f["launch_us"], "bwd_us": b["launch_us"]}))
    print(json.dumps(time_fps(bench.CONFIGS[cfg][0], 4096)))
    if "--lds" in sys.argv:
        lds_probe()
    if "--indexing" in sys.argv:
        indexing_ops()
        print(json.dumps(time_fps(80000, 4096)))
        print(json.dumps(time_fps(4000, 1024)))
