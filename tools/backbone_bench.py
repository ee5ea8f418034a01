#!/usr/bin/env python3
"""End-to-end step WITH the sparse-conv backbone on a synthetic indoor scene: voxel counts per tensor
stride, eager fwd+bwd time of the backbone alone and of the full model, the index kernels' GB/s.
   python tools/backbone_bench.py [npoints]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    npts = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40000
    dev = torch.device("cuda")
    from vdetr_amd import minkowski as ME
    from vdetr_amd import sparse_ops as S
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.model_vdetr import build_vdetr, default_args
    from vdetr_amd.runtime import enable_gemm_tuning
    if "--no-gemm-tuning" not in sys.argv:
        enable_gemm_tuning(0)
    torch.manual_seed(0)
    model = build_vdetr(default_args(), ScannetDatasetConfig(), "minkowski").to(dev).train()
    cloud = bench.make_room_cloud(npts, 0, dev)
    inputs = {"point_clouds": [cloud], "point_cloud_dims_min": cloud.min(0)[0][None], "point_cloud_dims_max": cloud.max(0)[0][None]}

    # geometry: voxel counts per stride
    coords, feats = ME.batch_sparse_collate([(cloud / model.voxel_size, cloud)])
    x = ME.SparseTensor(feats, coordinates=coords)
    stages = model.pre_encoder(x)
    counts = {1: int(x.keys.shape[0]), **{s.tensor_stride: int(s.keys.shape[0]) for s in stages}, 2: int(x.coordinate_manager.keys[2].shape[0])}
    nbr, _, plan4 = x.coordinate_manager.kernel_map(stages[0].keys, stages[0].keys, 4, 4, 3, False)
    print(json.dumps({"points": npts, "voxels_per_stride": dict(sorted(counts.items())),
                      "mean_neighbours_of_27_at_stride_4": float((nbr >= 0).float().sum(0).mean()),
                      "pairs_stride_4": None if plan4 is None else plan4.pairs, "dense_pairs_stride_4": 27 * nbr.shape[1]}))

    def timed(fn, reps=5, warm=3):
        ts = []
        for i in range(reps + warm):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return float(np.mean(ts)) * 1e3

    def backbone_step():
        model.zero_grad(set_to_none=True)
        scenes = model.backbone_forward(inputs)
        sum(f.square().mean() for _, f in scenes).backward()

    def full_step():
        model.zero_grad(set_to_none=True)
        out = model(inputs)
        loss = sum((o["sem_cls_logits"].sum() + o["center_normalized"].sum() + o["size_normalized"].sum())
                   for o in out["aux_outputs"] + [out["outputs"]])
        loss.backward()

    def decoder_only_step():
        model.zero_grad(set_to_none=True)
        with torch.no_grad():
            scenes = model.backbone_forward(inputs)
        inp = dict(inputs, backbone_xyz=[s[0] for s in scenes], backbone_features=[s[1].detach().requires_grad_(True) for s in scenes])
        saved, model.sparse_backbone = model.sparse_backbone, False
        pe, model.pre_encoder = model.pre_encoder, (lambda i: list(zip(i["backbone_xyz"], i["backbone_features"])))
        try:
            out = model(inp)
            loss = sum((o["sem_cls_logits"].sum() + o["center_normalized"].sum() + o["size_normalized"].sum())
                       for o in out["aux_outputs"] + [out["outputs"]])
            loss.backward()
        finally:
            model.sparse_backbone, model.pre_encoder = saved, pe

    t_geo = timed(lambda: model.prepare_geometry(inputs), reps=3, warm=1)
    inputs["geometry"] = model.prepare_geometry(inputs)  # as the FPS indices: built ahead of time, outside the step
    print(json.dumps({"geometry_ms": t_geo, "kernel_maps": len(inputs["geometry"].maps)}))
    for key, (nbr_, inv_, plan_, ik, ok) in inputs["geometry"].maps.items():
        if plan_ is not None:
            print(json.dumps({"map": f"stride {key[2]}->{key[3]} k{key[4]}{' T' if key[5] else ''}", "nin": plan_.nin, "nout": plan_.nout,
                              "pairs": plan_.pairs, "dense": plan_.K * plan_.nout}))
    if "--torch-profile" in sys.argv:  # per-kernel device time of 3 steady-state backbone steps
        for _ in range(4):
            backbone_step()
        torch.cuda.synchronize()
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                backbone_step()
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)
        tot = sum(r.device_time_total for r in rows)
        print(f"# 3 backbone steps: {tot / 3e3:.2f} ms device time per step, {sum(r.count for r in rows) // 3} launches per step")
        for r in rows[:32]:
            print(f"{r.device_time_total / tot * 100:6.2f}% {r.count // 3:5d}/step {r.device_time_total / 3e3:8.3f} ms/step {r.device_time_total / max(r.count, 1):8.1f} us  {r.key[:110]}")
        return
    if "--backbone-only" in sys.argv:  # for rocprofv3 --kernel-trace: 3 warm-up + 5 steps of the backbone alone
        print(json.dumps({"backbone_fwd_bwd_ms": timed(backbone_step)}))
        return
    res = {"backbone_fwd_bwd_ms": timed(backbone_step), "full_model_fwd_bwd_ms_eager": timed(full_step)}
    res["scenes_per_s_eager_with_backbone"] = 1e3 / res["full_model_fwd_bwd_ms_eager"]
    print(json.dumps(res))

    # index kernels alone, at the stride-4 level (64 channels, 27 offsets)
    f = torch.randn((nbr.shape[1], 64), device=dev)
    inv = S.inverse_map(nbr, nbr.shape[1])
    col = S.gather_cols(f, nbr)
    e = lambda: torch.cuda.Event(enable_timing=True)
    # compulsory bytes: every operand once (rows of f are re-read from L2, not HBM, so a fraction above
    # 1 of the HBM rate cannot appear); the look-up counts one 8-byte probe per (row, offset)
    for name, fn, nbytes in (("sp_gather_cols c=64 K=27", lambda: S.gather_cols(f, nbr), (col.numel() + f.numel() + nbr.numel()) * 4.0),
                             ("sp_gather_sum c=64 K=27", lambda: S.gather_sum(col, inv), (col.numel() + f.numel() + inv.numel()) * 4.0),
                             ("sp_kernel_map K=27", lambda: S.kernel_map(stages[0].keys, stages[0].keys, torch.zeros((27, 3), dtype=torch.int32, device=dev)),
                              nbr.numel() * 4.0 + nbr.numel() * 8.0 + stages[0].keys.numel() * 8.0)):
        ts = []
        for i in range(12):
            a, b = e(), e()
            a.record(); fn(); b.record(); b.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b) * 1e-3)
        t = float(np.mean(ts))
        print(json.dumps({"kernel": name, "us": t * 1e6, "compulsory_GB_s": nbytes / t / 1e9, "frac_of_8TBs": nbytes / t / 8e12}))


if __name__ == "__main__":
    main()
