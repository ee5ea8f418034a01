#!/usr/bin/env python3
"""Benchmark of the V-DETR hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

A "step" is one TRAINING step of the post-backbone path on one synthetic scene per GPU:
  FPS + gather (40k voxels -> 4096 tokens) -> projection -> decoder (FFN stage, top-1024 proposals, 8 x [self-attn,
  3DV-RPE cross-attn, FFN], 9 head stages) -> scalar loss -> backward -> (N>1) gradient all-reduce over RCCL ->
  gradient clipping + AdamW.
Workload = BASELINE.json configs[1] ("ScanNet 40k-point scene, full V-DETR config, bs=1, 1xMI355X"); N>1 is
configs[2] (one scene per GPU, weak scaling).  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

# (GPU_MAX_HW_QUEUES is left at HIP's default of 4, on purpose — see DESIGN.md §4.12 "Hardware queues": 16 queues cure the
# backbone leg's `inline` geometry mode but make the criterion leg's captured step 2.3x slower.)
import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (points, batch per GPU, preenc_npoints, nqueries, dec_nlayers, angle_type, description)
    "c1": (4000, 1, 4096, 64, 3, "", "C1: synthetic 4k-point scene, 64 queries, 2 RPE decoder layers"),
    "c2": (40000, 1, 4096, 1024, 9, "", "C2: synthetic 40k-point ScanNet-like scene, full V-DETR decoder config "
           "(4096 keys, 1024 queries, 8 RPE layers, 9 head stages), bs=1 per GPU"),
    "c4": (80000, 1, 4096, 1024, 9, "", "C4: synthetic 80k-point dense scene, 1024 queries (bf16 attention operands by default: --dtype)"),
    "c5": (20000, 4, 4096, 1024, 9, "object_coords", "C5: synthetic 20k-point rotated-box scenes, bs=4 per GPU"),
}


def make_scene(npoints, seed, device):
    """N points uniform in an 8x6x3 m room offset by +1 m, quantised to the 4 cm grid and de-duplicated (what the
    backbone's out.C * voxel_size looks like), + 256-channel random features (the backbone's out.F)."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([0, 0, 0], [8, 6, 3], (npoints, 3)) + 1.0
    vox = np.unique(np.round(pts / 0.04).astype(np.int64), axis=0)
    rng.shuffle(vox)
    xyz = torch.from_numpy((vox * 0.04).astype(np.float32)).to(device)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn((xyz.shape[0], 256), generator=g).to(device)
    return xyz, feats


def make_room_cloud(npoints, seed, device):
    """Raw input points of a synthetic indoor scene for the sparse backbone: an 8x6x3 m room (floor, four walls) and a few
    box-shaped pieces of furniture, points ON the surfaces with 5 mm noise, offset by +1 m (as make_scene)."""
    rng = np.random.default_rng(seed)
    surf = []  # (origin, edge u, edge v, area)
    L, W, H = 8.0, 6.0, 3.0
    planes = [((0, 0, 0), (L, 0, 0), (0, W, 0)), ((0, 0, 0), (L, 0, 0), (0, 0, H)), ((0, W, 0), (L, 0, 0), (0, 0, H)),
              ((0, 0, 0), (0, W, 0), (0, 0, H)), ((L, 0, 0), (0, W, 0), (0, 0, H))]
    for _ in range(12):  # furniture: top + four sides
        sx, sy, sz = rng.uniform(0.4, 2.0), rng.uniform(0.4, 1.5), rng.uniform(0.4, 1.8)
        ox, oy = rng.uniform(0.2, L - sx - 0.2), rng.uniform(0.2, W - sy - 0.2)
        planes += [((ox, oy, sz), (sx, 0, 0), (0, sy, 0)), ((ox, oy, 0), (sx, 0, 0), (0, 0, sz)), ((ox, oy + sy, 0), (sx, 0, 0), (0, 0, sz)),
                   ((ox, oy, 0), (0, sy, 0), (0, 0, sz)), ((ox + sx, oy, 0), (0, sy, 0), (0, 0, sz))]
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in planes])
    which = rng.choice(len(planes), size=npoints, p=areas / areas.sum())
    o = np.array([planes[i][0] for i in which]); u = np.array([planes[i][1] for i in which]); v = np.array([planes[i][2] for i in which])
    pts = o + rng.uniform(0, 1, (npoints, 1)) * u + rng.uniform(0, 1, (npoints, 1)) * v + rng.normal(0, 0.005, (npoints, 3))
    return torch.from_numpy((pts + 1.0).astype(np.float32)).to(device)


def build_model(cfg_name, device, seed=0):
    from vdetr_amd.dataset_config import RotatedBoxDatasetConfig, ScannetDatasetConfig
    from vdetr_amd.model_vdetr import build_vdetr, default_args
    npts, bs, npre, nq, nl, angle_type, _ = CONFIGS[cfg_name]
    torch.manual_seed(seed)
    args = default_args(dec_nlayers=nl, nqueries=nq, preenc_npoints=npre, angle_type=angle_type)
    ds = RotatedBoxDatasetConfig() if angle_type else ScannetDatasetConfig()
    model = build_vdetr(args, ds)
    with torch.no_grad():  # centre/size regressors are zero-initialised (vdetr_transformer.py:169-173): perturb them
        for h in model.decoder.mlp_heads:  # so boxes (and the RPE vertices) differ between stages
            for k in ("center_head", "size_head"):
                h[k].layers[-1].weight.add_(0.01 * torch.randn_like(h[k].layers[-1].weight))
    return model.to(device).train()


def make_inputs(cfg_name, device, rank):
    npts, bs, *_ = CONFIGS[cfg_name]
    xyzs, feats = [], []
    for i in range(bs):
        x, f = make_scene(npts, rank * 1000 + i, device)
        xyzs.append(x)
        feats.append(f.requires_grad_(True))  # gradient flows back into the (out-of-scope) backbone
    # scenes keep their own voxel counts (they differ after de-duplication): the batch is sampled by ONE variable-length
    # FPS launch and gathered row-wise from the point-major tables
    return {"backbone_xyz": xyzs, "backbone_features": feats,
            "point_cloud_dims_min": torch.stack([x.min(0)[0] for x in xyzs]),
            "point_cloud_dims_max": torch.stack([x.max(0)[0] for x in xyzs])}


def make_targets(cfg_name, device, rank, boxes_per_scene=24):
    """Synthetic ground truth with the keys / shapes of the reference loader (datasets/scannet.py:592-625): 64 box slots,
    `boxes_per_scene` axis-aligned boxes inside the room, random classes."""
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    npts, bs, *_ = CONFIGS[cfg_name]
    cfg = ScannetDatasetConfig()
    g = torch.Generator().manual_seed(7000 + rank)
    G = cfg.max_num_obj
    present = torch.zeros((bs, G))
    present[:, :boxes_per_scene] = 1
    centers = (torch.rand((bs, G, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0]) + 1.0) * present[..., None]
    sizes = (0.3 + torch.rand((bs, G, 3), generator=g) * 1.7) * present[..., None]
    corners = cfg.box_parametrization_to_corners(centers, sizes, torch.zeros((bs, G))) * present[..., None, None]
    t = {"gt_box_corners": corners, "gt_box_centers": centers, "gt_box_sizes": sizes, "gt_box_angles": torch.zeros((bs, G)),
         "gt_box_sem_cls_label": torch.randint(0, 10, (bs, G), generator=g) * present.long(), "gt_box_present": present,
         "gt_angle_class_label": torch.zeros((bs, G), dtype=torch.int64), "gt_angle_residual_label": torch.zeros((bs, G))}
    return {k: v.to(device) for k, v in t.items()}


class _SumAll(torch.autograd.Function):
    """sum of all elements of several tensors: one concatenation + one reduction forward, ONE fill backward (every tensor's
    gradient is a contiguous slice of the same buffer of ones).  27 `.sum()` calls added up in Python are 53 launches
    forward and 27 stride-0 gradients that each consumer first has to materialise."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.shapes = [t.shape for t in ts]
        return torch.cat([t.reshape(-1) for t in ts]).sum()

    @staticmethod
    def backward(ctx, g):
        sizes = [int(torch.Size(sh).numel()) for sh in ctx.shapes]
        flat = g.expand(sum(sizes)).contiguous()
        return tuple(part.view(sh) for part, sh in zip(flat.split(sizes), ctx.shapes))


def loss_fn(out):
    """synthetic scalar loss of SURVEY.md §8d: sum over the 9 stages of sem_cls_logits + centre + size"""
    if os.environ.get("VDETR_BENCH_LOSS_SUMS", "0") != "0":  # A/B: the term-by-term form
        return sum(o["sem_cls_logits"].sum() + o["center_normalized"].sum() + o["size_normalized"].sum()
                   for o in out["aux_outputs"] + [out["outputs"]])
    return _SumAll.apply(*[o[k] for o in out["aux_outputs"] + [out["outputs"]]
                           for k in ("sem_cls_logits", "center_normalized", "size_normalized")])


def fps_fork_layer(t1, t_fps, nlayers):
    """LATEST decoder layer of the forward in front of which the next scene's sampling branch may be forked, or -1 (the step's start),
    from the measured step (t1 ms, sampling forked at the start) and the sampling alone (t_fps ms).  The one-workgroup kernel holds a
    CU for 1.12 t_fps inside the step and should end 0.5 ms before the step does; layer k of the forward starts at
    (0.03 + 0.044 k) t1 (round 6's timeline); below 0.3 t1 the held CU is not worth a second capture.  The caller
    captures and times this layer AND the two in front of it and keeps what measures fastest: the bound only has to be generous
    (round 5's 1.2 ms sat exactly on the edge once the step had become 5 % shorter: half the runs fell back to the step's start, 2 % slower)."""
    if nlayers <= 0 or t1 <= 0 or t_fps < 0.3 * t1:
        return -1
    k = min(int(((t1 - 1.12 * t_fps - 0.5) / t1 - 0.03) / 0.044), nlayers - 1)
    return k if k >= 1 else -1


def agree_any_failed(store, world, failed, tag, timeout_s=600.0):
    """Did the capture fail on ANY rank?  Agreed through the process group's key-value store (CPU only): a process whose
    capture failed cannot issue device work any more — the HIP runtime leaves the capture's streams in capture state and
    refuses every later attempt to end it ("attempt to terminate a thread-local capture sequence from another thread";
    tools/probes/capture_recovery.py) — so neither a device all-reduce nor a retry in this process is possible.
    Every rank calls it once per `tag`; all get the same answer."""
    store.add(f"bench_{tag}_failed", 1 if failed else 0)
    store.add(f"bench_{tag}_seen", 1)
    deadline = time.time() + timeout_s
    while store.add(f"bench_{tag}_seen", 0) < world:
        if time.time() > deadline:
            raise RuntimeError(f"bench: the ranks did not all report their capture within {timeout_s:.0f} s")
        time.sleep(0.01)
    return store.add(f"bench_{tag}_failed", 0) > 0


def free_port(preferred):
    """`preferred` if nothing listens there, otherwise a port the OS picks"""
    import socket
    for port in (preferred, 0):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                sock.bind(("127.0.0.1", port))
                return sock.getsockname()[1]
            except OSError:
                continue
    raise RuntimeError("bench: no free rendezvous port for the child processes")


def allreduce_bucket_model(grad_bytes, world):
    """How many gradient buckets the captured decoder step should use, from a model instead of a 1-rank timing (where the
    collective is free).  Ring all-reduce over the node's xGMI mesh: every GPU sends and receives 2 (W-1)/W x S bytes, RCCL
    spreads its rings over the W-1 direct links (MI355X: 7 links x 153.6 GB/s bidirectional = 76.8 GB/s per direction and
    link; VDETR_XGMI_GBPS / VDETR_XGMI_EFF override), so t(S) = 2 (W-1)/W x S / (links x 76.8 GB/s x eff) + latency.  With k
    buckets all but the last all-reduce hide under the next bucket's batched weight-gradient GEMMs, but every extra phase
    splits those batches: +0.33 ms per extra phase (measured).  exposed(k) = t(S / k) + (k - 1) x 0.33 ms."""
    if world < 2:
        return {"buckets": 1, "bucket_mb": 64.0, "note": "1 rank: nothing to overlap"}
    per_dir = float(os.environ.get("VDETR_XGMI_GBPS", "76.8"))
    eff = float(os.environ.get("VDETR_XGMI_EFF", "0.7"))
    links = min(world - 1, 7)
    busbw = per_dir * links * eff * 1e9
    t = lambda nbytes: 2.0 * (world - 1) / world * nbytes / busbw * 1e3 + 0.03  # ms  # noqa: E731
    split_ms = float(os.environ.get("VDETR_SPLIT_COST_MS", "0.33"))
    exposed = {k: t(grad_bytes / k) + (k - 1) * split_ms for k in (1, 2, 3, 4)}
    k = min(exposed, key=exposed.get)
    return {"buckets": k, "bucket_mb": grad_bytes / k / (1 << 20) + 1.0, "bus_GBps_assumed": busbw / 1e9, "allreduce_ms_model": t(grad_bytes),
            "exposed_ms_model": {str(kk): round(v, 3) for kk, v in exposed.items()},
            "note": "bucket count = argmin of the modelled exposed time (allreduce_bucket_model in bench.py); VDETR_BUCKET_MB overrides"}


def quiesce_collectives(self):
    """Before a capture begins: nothing of the warm-up's collectives may still sit in ProcessGroupNCCL's watchdog list.
    The watchdog thread polls the completion events of outstanding collectives every 100 ms, and an event query from
    ANOTHER thread while this thread captures is an error under the default (global) capture mode — seen as an
    intermittent SIGABRT of the rank ("operation not permitted when stream is capturing", 1 run in 3).  After a device
    synchronise the warm-up's collectives are polled until each reports completion (Work.is_completed()), then ONE watchdog
    period lets that thread drop them.  The captures below also run in thread-local error mode, which permits such calls
    from other threads."""
    torch.cuda.synchronize()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        deadline = time.time() + 5.0
        while not self.reducer.collectives_done() and time.time() < deadline:
            time.sleep(0.005)
        time.sleep(0.12)


CLIP_NORM = 0.1  # --clip_gradient (main.py / engine.py:105-106)


def make_optimizer(flat, reduced_after_pack):
    """AdamW(lr 7e-4, weight decay 0.1) behind clip_grad_norm_(0.1) on the flat buffer: vdetr_amd.optim.ClipAdamW (norm out of the pack
    launch, clip + update in one launch); VDETR_OWN_ADAMW=0: torch's fused AdamW with the clip coefficient as its grad_scale (A/B)"""
    if os.environ.get("VDETR_OWN_ADAMW", "1") != "0" and flat.data.is_cuda:
        from vdetr_amd.optim import ClipAdamW
        return ClipAdamW(flat, lr=7e-4, weight_decay=0.1, max_norm=CLIP_NORM, norm_from_pack=not reduced_after_pack)
    return torch.optim.AdamW([flat.param], lr=7e-4, weight_decay=0.1, capturable=True, fused=True)


def optimizer_step(opt, flat):
    if not hasattr(opt, "max_norm"):
        # clip_grad_norm_(params, 0.1): one norm over the flat gradient; the clip coefficient is applied inside the fused AdamW
        # launch (grad_scale = 1 / coefficient)
        opt.grad_scale = flat.clip_scale(CLIP_NORM)[0]
    opt.step()


class Trainer:
    """fwd + bwd (+ all-reduce) + clip + AdamW, eager or as captured hipGraphs."""

    def __init__(self, model, inputs, world, use_graph, overlap, fps_prefetch=True, criterion=None, targets=None,
                 defer_wg=True, fps_depth=None, force_dist=False, fps_at_layer=None):
        from vdetr_amd.dist import FlatParams, GradientReducer
        global flush_weight_grads
        from vdetr_amd.runtime import defer_weight_grads, flush_weight_grads
        self.model, self.inputs, self.world = model, inputs, world
        # the next scene's sampling launch (one CU for milliseconds) is forked in front of this decoder layer of the forward instead
        # of at the step's start (-1); chosen by measurement in choose_fps_depth, VDETR_BENCH_FPS_AT_LAYER forces it
        self.fps_at_layer = int(os.environ.get("VDETR_BENCH_FPS_AT_LAYER", "-1")) if fps_at_layer is None else int(fps_at_layer)
        # criterion=None: the synthetic scalar loss of SURVEY.md §8d (the headline metric); otherwise the device set
        # criterion (v-detr_amd/criterion.py) on `targets`, prepared once: their box counts do not depend on the model
        self.criterion = criterion
        self.targets = criterion.prepare_targets(targets) if criterion is not None else None
        # weight / bias gradients of the linear layers in shape-batched GEMMs after the backward (off its critical path)
        self.defer_wg = defer_wg
        defer_weight_grads(defer_wg)
        self.params = [p for p in model.parameters() if p.requires_grad]
        # parameters / gradients as views of two flat buffers: one AdamW launch, one norm, slice-shaped buckets
        # hooks + bucket views only pay off when they overlap communication with an EAGER backward; otherwise
        # gradients stay ordinary tensors and are packed with one multi-tensor copy before the all-reduce
        self.hooked = (world > 1 or force_dist) and overlap and not use_graph
        # (shape-grouped layout for the parked weight gradients; reverse parameter order where bucket hooks fire during an
        # eager backward with inline weight gradients, so that buckets complete while the backward is still running)
        self.flat = FlatParams(self.params, groups=model.flat_param_groups(), group_shapes=defer_wg or not self.hooked)
        grad_bytes = sum(p.numel() for p in self.params) * 4
        self.bucket_model = allreduce_bucket_model(grad_bytes, world)
        bucket_mb = float(os.environ.get("VDETR_BUCKET_MB", self.bucket_model["bucket_mb"]))
        self.reducer = GradientReducer(self.params, bucket_mb=bucket_mb, overlap=self.hooked,
                                       bucket_views=self.hooked, flat=self.flat, force=force_dist)
        # Captured step on several ranks: the gradient buckets are all-reduced INSIDE the hipGraph, each on the side stream as
        # soon as its slice of the flat buffer is packed, while the parked weight gradients of the next bucket are computed
        # (GradientReducer.reduce_phased); the optimizer step follows in the same graph: one replay per step, no eager gap
        # between backward, RCCL and AdamW.  VDETR_PHASED_REDUCE=0: the round-2 sequence (graph, eager all-reduce, graph).
        self.phased = use_graph and self.reducer.active and os.environ.get("VDETR_PHASED_REDUCE", "1") != "0"
        self.opt = make_optimizer(self.flat, reduced_after_pack=self.reducer.active)
        self.use_graph = use_graph
        self.g_main = self.g_opt = None
        self.loss = None
        # FPS of the NEXT scene runs on a side stream while this scene is in the decoder (the sampling depends on
        # voxel coordinates only).  Every step still runs one full FPS kernel; it just no longer serialises a
        # one-CU kernel in front of a 256-CU step.
        self.fps_prefetch = fps_prefetch
        # lookahead 2 (graph mode): the FPS of scene i+2 is launched OUTSIDE the captured step on one of two side streams
        # (one launch per step, each has two steps to finish), so the ~9.5 ms single-CU kernel no longer has to fit INSIDE a
        # step the way a forked branch of the hipGraph must.  What a data loader with a two-scene queue provides.
        # Measured: C4 (80k points, FPS 10.7 ms) 12.15 -> 11.14 ms per step; C2 (FPS 9.5 ms < step) 10.65 -> 11.12 ms (two
        # CUs busy, eager launches between the replays): used when the sampling does not fit inside a step.
        # Which one: main() MEASURES it (`choose_fps_depth`: when the sampling alone takes more than 0.8 of the captured
        # depth-1 step, a depth-2 trainer is captured as well and the faster one kept); VDETR_FPS_DEPTH=1 / 2 forces it.
        depth = str(fps_depth) if fps_depth is not None else os.environ.get("VDETR_FPS_DEPTH", "auto")
        self.fps_depth2 = fps_prefetch and use_graph and depth == "2"
        if fps_prefetch:
            self.side = torch.cuda.Stream()
            self.cur_inds = model.sample_indices(inputs)
        if self.fps_depth2:
            self.fps_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            self.fps_ring = [self.cur_inds.clone(), self.cur_inds.clone()]
            self.fps_events = [None, None]
            self.fps_tick = 0

    def _fwd_bwd(self):
        self.reducer.zero_grad()
        for f in self.inputs["backbone_features"]:
            f.grad = None
        if self.fps_depth2:
            self.inputs["fps_inds"] = self.cur_inds  # filled by step() from the ring before the replay
        elif self.fps_prefetch:
            main = torch.cuda.current_stream()
            box, hook = {}, None

            def launch_fps(*_):
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    box["inds"] = self.model.sample_indices(self.inputs)  # (the synthetic bench feeds the same scene again)
            at = self.fps_at_layer
            layers = getattr(getattr(self.model, "decoder", None), "layers", None)
            if at >= 0 and layers is not None and at < len(layers):
                # the sampling kernel holds one CU for ~4.7 ms: forked in front of decoder layer `at` instead of at the step's start
                hook = layers[at].register_forward_pre_hook(launch_fps)
            else:
                launch_fps()
            self.inputs["fps_inds"] = self.cur_inds
        from vdetr_amd.runtime import ts_mark  # (no-ops unless VDETR_TS_PROBE=1: tools/probes/step_timeline.py)
        ts_mark("step start")
        out = self.model(self.inputs)
        if self.fps_prefetch and not self.fps_depth2:
            if hook is not None:
                hook.remove()
            next_inds = box["inds"]
            with torch.cuda.stream(self.side):
                ts_mark("side: sampling done")
        self.loss = loss_fn(out) if self.criterion is None else self.criterion(out, self.targets)[0]
        ts_mark("forward + loss done")
        self.loss.backward()
        ts_mark("backward enqueued: main chain done")
        if self.phased:
            self.reducer.reduce_phased()  # parked weight gradients, slice packs and all-reduces, bucket by bucket
        else:
            if self.defer_wg:
                flush_weight_grads()
            ts_mark("flush done (side branch joined)")
            if not self.hooked:
                self.flat.pack_grads()  # one launch; (hooked eager mode accumulates straight into the flat buffer)
        if self.fps_prefetch and not self.fps_depth2:
            main.wait_stream(self.side)
            self.cur_inds.copy_(next_inds)
        ts_mark("gradients packed, sampling joined")

    def _update(self):
        optimizer_step(self.opt, self.flat)
        from vdetr_amd.runtime import ts_mark
        ts_mark("optimizer done")

    def reset_state(self, snap):
        """parameters and buffers back to `snap` (model_state()), AdamW moments and step count to zero — in place, the captured
        graphs keep their pointers.  The step's duration depends on the weights (the synthetic loss is unbounded below: the boxes
        grow with every step, and with them the work of the box kernels): forms of the step are compared from the SAME state."""
        with torch.no_grad():
            for t, s0 in zip(list(self.model.parameters()) + list(self.model.buffers()), snap):
                t.copy_(s0)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()

    quiesce_collectives = quiesce_collectives  # (module-level: BackboneTrainer takes it too)
    _capture_stream = None  # one per process: a second Trainer on the same model meets the first one's AccumulateGrad nodes

    def capture(self):
        if Trainer._capture_stream is None:
            Trainer._capture_stream = torch.cuda.Stream()
        s = Trainer._capture_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):  # warm allocator, lazy inits, LDS attribute grants
                self._fwd_bwd()
                if self.reducer.active and not self.phased:
                    self.reducer.reduce_all()  # real updates: the replicas must stay identical
                self._update()
        torch.cuda.current_stream().wait_stream(s)
        self.quiesce_collectives()
        # Capture on the stream the warm-up ran on: the warm-up's autograd graph (kept alive by tensors the modules hold)
        # owns one AccumulateGrad node per parameter, bound to the stream it was created on; under a different capture
        # stream the engine runs those nodes on the old stream and stitches cross-stream dependencies into the graph.
        same = os.environ.get("VDETR_CAPTURE_SAME_STREAM", "1") != "0"
        self.g_main = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local", **({"stream": s} if same else {})):
            self._fwd_bwd()
            if self.phased or not self.reducer.active:
                self._update()
        if self.reducer.active and not self.phased:
            self.reducer.reduce_all()  # the flat gradient buffer, in slices; not captured (RCCL outside the graph)
            self.quiesce_collectives()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local", **({"stream": s} if same else {})):
                self._update()

    def _fps_lookahead(self):
        """launch the sampling of scene i+2 on side stream i % 2; hand the indices of scene i (launched two steps ago on
        the same stream) to the captured step"""
        k = self.fps_tick & 1
        main = torch.cuda.current_stream()
        if self.fps_events[k] is not None:
            main.wait_event(self.fps_events[k])
            self.cur_inds.copy_(self.fps_ring[k])
        done_reading = torch.cuda.Event()
        done_reading.record(main)
        st = self.fps_streams[k]
        st.wait_event(done_reading)  # the ring slot is free again
        with torch.cuda.stream(st):
            self.fps_ring[k].copy_(self.model.sample_indices(self.inputs))  # (the synthetic bench feeds the same scene again)
            ev = torch.cuda.Event()
            ev.record(st)
        self.fps_events[k] = ev
        self.fps_tick += 1

    def step(self):
        if self.g_main is not None:
            if self.fps_depth2:
                self._fps_lookahead()
            self.g_main.replay()
            if self.g_opt is not None:
                self.reducer.reduce_all()
                self.g_opt.replay()
        else:
            self._fwd_bwd()
            if self.hooked:
                self.reducer.finish()
            elif self.reducer.active and not self.phased:
                self.reducer.reduce_all()
            self._update()


class BackboneTrainer:
    """The training step INCLUDING the sparse-convolution backbone on raw points:
    voxelise -> MinkResNet34 + FPN (eager: its tensor sizes follow the scene) -> FPS tokens -> projection + decoder + loss +
    backward of that part as ONE captured hipGraph (fixed sizes) -> backbone backward (eager) -> pack, clip, AdamW.
    What depends on the point coordinates only — voxel sites, kernel maps, compacted row lists, FPS indices — is built
    ahead of time (`geometry_ms`), as a data loader / side stream would for the next scene."""
    quiesce_collectives = quiesce_collectives


    def __init__(self, cfg_name, device, npoints=40000, seed=0, force_dist=False, scene_seed=None):
        from vdetr_amd import pointnet2_utils as PU
        from vdetr_amd import sparse_ops as S
        from vdetr_amd.dataset_config import ScannetDatasetConfig
        from vdetr_amd.dist import FlatParams, GradientReducer
        from vdetr_amd.model_vdetr import build_vdetr, default_args
        global flush_weight_grads
        from vdetr_amd.runtime import defer_weight_grads, flush_weight_grads
        _, bs, npre, nq, nl, angle_type, _ = CONFIGS[cfg_name]
        # The decoder's table gradient stays in line here: this step already keeps a loader stream (next scene's geometry, a
        # 5 ms one-CU sampling kernel) next to the eager backbone, and a further queue for the captured step's side branch lands
        # behind one of them (tools/backbone_step_probe.py: thread 22.0 -> 22.7 ms, inline 22.0 -> 26.4 with it on)
        from vdetr_amd import attention as _A
        self._async_table_mode = _A.set_async_table_grad(os.environ.get("VDETR_BENCH_BB_ASYNC_TABLE", "0"))
        torch.manual_seed(seed)
        self.model = model = build_vdetr(default_args(dec_nlayers=nl, nqueries=nq, preenc_npoints=npre, angle_type=angle_type),
                                         ScannetDatasetConfig(), "minkowski").to(device).train()
        with torch.no_grad():
            for h in model.decoder.mlp_heads:
                for k in ("center_head", "size_head"):
                    h[k].layers[-1].weight.add_(0.01 * torch.randn_like(h[k].layers[-1].weight))
        cloud = make_room_cloud(npoints, seed if scene_seed is None else scene_seed, device)  # same weights, own scene per rank
        self.inputs = {"point_clouds": [cloud], "point_cloud_dims_min": cloud.min(0)[0][None], "point_cloud_dims_max": cloud.max(0)[0][None]}
        for rep in range(2):  # the second pass is the timed one (the first also tunes the library GEMMs of the dry run)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            geo = model.prepare_geometry(self.inputs)
            xyz4 = (S.unpack_keys(geo.keys[4])[:, 1:].float() * model.voxel_size).contiguous()
            self.inds = PU.furthest_point_sample_varlen([xyz4], npre)
            torch.cuda.synchronize()
            self.geometry_ms = (time.perf_counter() - t0) * 1e3
        self.inputs["geometry"] = geo
        self.voxels = {ts: int(k.shape[0]) for ts, k in sorted(geo.keys.items())}
        self.static_xyz = torch.zeros((1, npre, 3), device=device)
        self.static_feat = torch.zeros((npre, 1, 256), device=device, requires_grad=True)
        self.dec_inputs = dict(self.inputs, enc_xyz=self.static_xyz, enc_features=self.static_feat, enc_inds=self.inds)
        defer_weight_grads(True)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.dec_params = [p for n, p in model.named_parameters() if n.startswith(("decoder.", "encoder_to_decoder"))]
        self.bb_params = [p for n, p in model.named_parameters() if not n.startswith(("decoder.", "encoder_to_decoder"))]
        # (the decoder's part shape by shape for its batched weight gradients, the backbone's in REVERSE parameter order: its
        # gradient buckets then complete one after the other while its eager backward pass is still running)
        self.flat = FlatParams(self.params, groups=model.flat_param_groups(), first=self.dec_params, group_shapes="first")
        # data parallel: the decoder's gradients (47 MB) are final after its captured backward and are all-reduced on the side
        # stream while the backbone's backward (the larger half of the step) runs; the backbone's 268 MB follow in 64 MB buckets
        first_bb = next(p for p in self.flat.params if any(p is q for q in self.bb_params))
        self.reducer = GradientReducer(self.params, bucket_mb=float(os.environ.get("VDETR_BUCKET_MB", "64")), overlap=False,
                                       bucket_views=False, flat=self.flat, force=force_dist, break_before=[first_bb])
        self.dec_buckets = self.reducer.buckets_of(self.dec_params)
        self.bb_buckets = [k for k in range(len(self.reducer.buckets)) if k not in self.dec_buckets]
        assert not set(self.reducer.buckets_of(self.bb_params)) & set(self.dec_buckets), "decoder / backbone gradients share a bucket"
        # every backbone bucket leaves as soon as its last gradient has arrived, under the rest of the backbone's backward
        # (DistributedDataParallel's bucket hooks in the reference, main.py:515-517); VDETR_BB_OVERLAP=0: all of them after it
        self.bb_overlap = self.reducer.active and os.environ.get("VDETR_BB_OVERLAP", "1") != "0"
        if self.bb_overlap:
            self.reducer.launch_when_complete(self.bb_buckets)
        self.opt = make_optimizer(self.flat, reduced_after_pack=self.reducer.active)
        self.graph, self.loss = None, None
        # the NEXT scene's geometry and FPS indices are built on a side stream while this scene trains (the synthetic bench
        # feeds the same cloud again, so "next" is recomputed from it every step: its cost is inside ms_per_step)
        # (VDETR_BENCH_GEOMETRY: "inline" = by the launching thread once the step is enqueued, "thread" = by a loader thread)
        # (VDETR_BENCH_SIDE_PRIORITY=-1: high-priority side streams, A/B switch.  Measured worse with the loader thread: 35.8 vs
        # 26.7 ms per step — the geometry kernels and the sampling kernel then win every dispatch against the step's kernels.)
        prio = int(os.environ.get("VDETR_BENCH_SIDE_PRIORITY", "0"))
        self.side = torch.cuda.Stream(priority=prio)
        self.sides = [torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)]  # scene i+2 is prepared while i+1's sampling may still run
        self._queue, self._tick = [], 0
        self._next, self._worker = None, None

    def _prepare_next(self, stream=None):
        from vdetr_amd import pointnet2_utils as PU
        from vdetr_amd import sparse_ops as S
        torch.cuda.set_device(self.static_xyz.device)
        with torch.cuda.stream(stream or self.side):
            geo = self.model.prepare_geometry({k: v for k, v in self.inputs.items() if k != "geometry"})
            xyz4 = (S.unpack_keys(geo.keys[4])[:, 1:].float() * self.model.voxel_size).contiguous()
            self._next = (geo, PU.furthest_point_sample_varlen([xyz4], self.inds.shape[1]))
            if stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                return self._next + (ev,)

    def _decoder_fwd_bwd(self):
        for p in self.dec_params:
            p.grad = None
        self.static_feat.grad = None
        out = self.model(self.dec_inputs)
        self.loss = loss_fn(out)
        self.loss.backward()
        flush_weight_grads()

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self._decoder_fwd_bwd()
        torch.cuda.current_stream().wait_stream(s)
        self.quiesce_collectives()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
            self._decoder_fwd_bwd()

    def step(self):
        from vdetr_amd import pointnet2_utils as PU
        if len(self._queue) >= 2:  # inline mode: the scene prepared one full step ago
            geo, inds, ev = self._queue.pop(0)
            torch.cuda.current_stream().wait_event(ev)
            self.inputs["geometry"] = geo
            self.inds.copy_(inds)
        if self._worker is not None:  # thread mode: what the loader thread prepared during the previous step
            self._worker.join()
            self._worker = None
            torch.cuda.current_stream().wait_stream(self.side)
            self.inputs["geometry"], inds = self._next
            self.inds.copy_(inds)
        import threading
        mode = os.environ.get("VDETR_BENCH_GEOMETRY", "thread")  # A/B switch: thread | inline | static (first scene's geometry reused)
        marks = getattr(self, "marks", None)  # (tools/backbone_step_probe.py --timeline: events between the phases of the step)
        def mark():
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        mark()
        for p in self.bb_params:
            p.grad = None
        xyz, feats = self.model.backbone_forward(self.inputs)[0]
        enc_rows = PU.gather_rows([feats.contiguous()], self.inds)                  # [1, m, 256], differentiable
        with torch.no_grad():
            self.static_xyz.copy_(PU.gather_rows([xyz.contiguous()], self.inds))
            self.static_feat.copy_(enc_rows.permute(1, 0, 2))
        mark()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._decoder_fwd_bwd()
        mark()
        if mode == "thread":
            # (A/B variant) started HERE, not at the top of the step: the forward pass is ~500 launches of 5-100 us, the host barely
            # keeps ahead of the device there and a second Python thread taking turns on the interpreter lock starves it (backbone
            # forward 5.8 -> 9.9 ms measured); now the device has the forward and the captured decoder step (14 ms) queued
            self._worker = threading.Thread(target=self._prepare_next, daemon=True)
            self._worker.start()
        if self.reducer.active:
            self.reducer.pack_and_launch(self.dec_buckets)  # overlaps the backbone's backward
            if self.bb_overlap:
                self.reducer.begin_watch()
        enc_rows.backward(self.static_feat.grad.permute(1, 0, 2))  # (backbone buckets are packed and sent from its hooks)
        mark()
        if self.reducer.active:
            pending = self.reducer.pending_watched() if self.bb_overlap else self.bb_buckets
            self.sent_in_backward = len(self.bb_buckets) - len(pending)  # buckets whose all-reduce started under the backward pass
            self.reducer.pack_and_launch(pending)
            self.reducer.finish()
        else:
            self.flat.pack_grads()
        optimizer_step(self.opt, self.flat)
        mark()
        if mode == "inline":
            # (A/B variant) the step is enqueued: the host builds the geometry of scene i+2 now, on a side stream (two alternate:
            # scene i+1's one-CU sampling kernel may still be running on the other); lookahead 2 = a two-scene loader queue.
            # 26.0-26.2 ms per step in a fresh process or with GPU_MAX_HW_QUEUES=16, 32.7 inside this file's process with HIP's
            # default of 4 hardware queues (its two side streams then share a queue with the captured decoder step: the 9 ms
            # sampling kernel sits in front of the step's kernels); the loader thread, on ONE side stream: 25.7-26.3.
            self._queue.append(self._prepare_next(self.sides[self._tick & 1]))
            self._tick += 1

    def calibrate_side_stream(self, candidates=5, steps=6):
        """HIP maps the streams of a process onto 4 hardware queues; a side stream that lands on the queue of the step's own
        streams puts the next scene's 9 ms sampling kernel IN FRONT of the step's kernels (25 -> 29-33 ms per step, depending on
        how many streams the process created before: measured in this file's own process).  Which queue a stream gets cannot be
        asked, but it can be measured: a few steps with each of several candidate streams, keep the fastest.  Once per process."""
        best = (None, float("inf"))
        self.calibration_ms = []
        for cand in [self.side] + [torch.cuda.Stream() for _ in range(candidates - 1)]:
            self.close()
            self.side = cand
            for _ in range(3):
                self.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            self.calibration_ms.append(round(ms, 2))
            if ms < best[1]:
                best = (cand, ms)
        self.close()
        self.side = best[0]
        return best[1]

    def close(self):
        if self._worker is not None:
            self._worker.join()
            self._worker = None
        torch.cuda.synchronize()
        if self._async_table_mode is not None:
            from vdetr_amd import attention as _A
            _A.set_async_table_grad(self._async_table_mode)
            self._async_table_mode = None


# ---------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernels, measured live with HIP events on the launch stream
# ---------------------------------------------------------------------------------------------------------------
def _pmc_traffic():
    """The newest profiles/rNN_pmc_traffic.json (tools/pmc_traffic.py writes it from a round's final PMC passes)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    with open(os.environ.get("VDETR_PMC_TRAFFIC") or cands[-1]) as fh:
        return json.load(fh)


def kernel_rooflines(cfg_name, device, reps=20):
    import ctypes
    from vdetr_amd import _lib as L
    from vdetr_amd import attention as A
    _, bs, nK, nQ, *_ = CONFIGS[cfg_name]
    B, H = bs, 4
    g = torch.Generator().manual_seed(0)
    xyz, _ = make_scene(40000, 0, device)
    from vdetr_amd.pc_util import morton_argsort
    kxyz = xyz[torch.randperm(xyz.shape[0], generator=g)[:nK].to(device)][None].repeat(B, 1, 1).contiguous()
    kxyz = torch.gather(kxyz, 1, morton_argsort(kxyz).unsqueeze(-1).expand(-1, -1, 3)).contiguous()  # as the decoder does
    center = kxyz[:, torch.randperm(nK, generator=g)[:nQ].to(device)]
    half = (0.1 + torch.rand((B, nQ, 1, 3), generator=g)).to(device)
    signs = torch.tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]],
                         dtype=torch.float32, device=device)
    cos_sin = None
    if CONFIGS[cfg_name][5] == "object_coords":  # C5: rotated boxes — the corners turn about the vertical axis and the kernels
        ang = ((torch.rand((B, nQ), generator=g) * 2 - 1) * 3.1).to(device)  # get the (cos, sin) operand: general RPE kernels
        c, sn = torch.cos(ang)[:, :, None], torch.sin(ang)[:, :, None]
        off = half * signs
        # corners = centre + R(angle)^T (+-half): what the box decode writes, so that the kernels' turn by the angle (rpe_rotate)
        # brings them back onto the axes
        off = torch.stack((off[..., 0] * c + off[..., 1] * sn, -off[..., 0] * sn + off[..., 1] * c, off[..., 2]), -1)
        verts = (center[:, :, None, :] + off).contiguous()
        cos_sin = torch.stack((c[..., 0], sn[..., 0]), -1).contiguous()
    else:
        verts = (center[:, :, None, :] + half * signs).contiguous()
    q = torch.randn((B, nQ, 256), generator=g).to(device)
    k = torch.randn((B, nK, 64), generator=g).to(device)
    v = torch.randn((B, nK, 64), generator=g).to(device)
    table = torch.randn((8, 10, 10, 10, 4), generator=g).to(device)
    rng = A.begin_step(device)
    d = A._desc(L.VDETR_ATTN_SHARED_KV, B, H, nQ, nK, 0.125, table, A.RPEConfig(), verts, kxyz, cos_sin, None, 0.1, rng, 1)
    lib = L.lib()
    out = torch.empty_like(q)
    lse = torch.empty((B, nQ, H), device=device)
    scores = torch.empty((B, nQ, H, nK), device=device)
    dout = torch.randn((B, nQ, H * 64), generator=g).to(device) * 1e-3
    delta = torch.zeros((B, nQ, H), device=device)
    aux = torch.zeros(8, dtype=torch.int32, device=device)  # norm maxima, query counters, non-box count (vdetr_hip.h)
    dtable = torch.zeros_like(table)
    dscore = torch.empty_like(scores)  # dS [B, nQ, H, nK], written by the key-side pass
    wsf = lib.vdetr_attn_fwd_workspace_bytes(ctypes.byref(d))
    wsb = lib.vdetr_attn_bwd_workspace_bytes(ctypes.byref(d))
    ws = L.workspace(max(wsf, wsb), device)
    st = L.stream_ptr()

    def fwd():
        L.check(lib.vdetr_attn_fwd_f32(ctypes.byref(d), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse),
                                       L.ptr(scores), L.ptr(ws), wsf, st), "attn_fwd")

    def bwd_prep():  # as attention.py's backward does before the score stage (not part of the timed launch)
        aux.zero_()
        d.bwd_aux = aux.data_ptr()
        L.check(lib.vdetr_attn_delta_f32(ctypes.byref(d), L.ptr(dout), L.ptr(out), L.ptr(v), L.ptr(delta), st), "attn_delta")

    # what the step runs (attention.py): the key-side pass writes dS, the table kernels read it
    dkv = torch.empty((2, B, nK, 64), device=device)
    wkv = lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(d))
    ws_kv = L.workspace(wkv, device)

    def kv():
        L.check(lib.vdetr_attn_bwd_kv_f32(ctypes.byref(d), L.ptr(q), L.ptr(v), L.ptr(dout), L.ptr(scores), L.ptr(lse), L.ptr(delta),
                                          L.ptr(dscore), L.ptr(dkv[0]), L.ptr(dkv[1]), L.ptr(ws_kv), wkv, st), "attn_bwd_kv")

    def bwd():
        L.check(lib.vdetr_attn_bwd_table_f32(ctypes.byref(d), L.ptr(dscore), L.ptr(dtable), L.ptr(ws), wsb, st), "attn_bwd_table")

    def timeit(fn, prep=None):
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
                ts.append(e0.elapsed_time(e1) * 1e-3)
        return float(np.mean(ts))

    t_fwd = timeit(fwd)
    bwd_prep()
    t_kv = timeit(kv)
    # the launch as the step issues it: on the side stream the table gradient takes ASYNC_TABLE_GRID of the 256 CUs
    side = A._async_wanted(B, nQ, nK) and 2 <= A.ASYNC_TABLE_GRID < 256
    if side:
        d.table_grid = A.side_table_grid(device)
    try:
        t_bwd = timeit(bwd, bwd_prep)
    finally:
        d.table_grid = 0
    # (VDETR_ROOFLINE_STEP_GRID_ONLY=1: only the launch the step issues — so that a rocprofv3 run of tools/kernel_bench.py averages
    # ONE configuration of the kernel, the one `launch_us` reports)
    t_bwd_full = timeit(bwd, bwd_prep) if side and os.environ.get("VDETR_ROOFLINE_STEP_GRID_ONLY") != "1" else t_bwd
    pairs = B * nQ * nK
    flops = 4.0 * H * pairs * 64                       # QK^T + PV (MFMA-eligible), SURVEY.md §8d
    bytes_bwd = 4.0 * H * pairs                        # dS read once (fp32); the kernel is VALU-bound, see valu_issue
    bytes_kv = 2.0 * 4.0 * H * pairs                   # S read, dS written (fp32); operands and dK / dV are ~1 % of that
    fwd_obj = {"kernel": "attn_fwd_kernel<shared_kv,rpe> (3DV-RPE cross-attention forward)", "bound": "mfma",  # (name completed below)
               "achieved": flops / t_fwd / 1e12, "peak": 157.3, "unit": "TFLOP/s",
               "frac": flops / t_fwd / 1e12 / 157.3, "traffic": None, "launch_us": t_fwd * 1e6,
               "rpe_lookups_per_s": 8.0 * pairs / t_fwd,
               "note": "algorithmic QK^T + PV flops over the launch time (incl. the K/V operand pack and the key-split combine launches) against "
                       "the dense f32 matrix peak: the contract is f32 arithmetic.  The default kernel forms the products on the bf16 matrix "
                       "unit from split f32 operands (18 instructions per 16-key tile instead of 32 f32 ones); most of the launch is the "
                       "RPE look-up (VALU + LDS), see DESIGN.md 4.1"}
    kv_obj = {"kernel": "attn_bwd_kv_kernel (dO V^T, softmax backward, dV, dK in one pass over the scores; + its operand-packing launch)",
              "bound": "hbm", "achieved": bytes_kv / t_kv / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": bytes_kv / t_kv / 1e9 / 8000.0,
              "traffic": None, "launch_us": t_kv * 1e6}
    # the name of the kernel that did the work comes from the library and the launch's own gate word, not from a string here
    kbox, kgen = ctypes.c_char_p(), ctypes.c_char_p()
    L.check(lib.vdetr_attn_bwd_table_kernel_names(ctypes.byref(d), ctypes.byref(kbox), ctypes.byref(kgen)), "kernel names")
    gate = aux.cpu()
    ran_box = kbox.value is not None and int(gate[4]) == 0 and int(gate[5]) != 0
    bwd_kernel = (kbox.value if ran_box else kgen.value).decode()
    fwd_name = "attn_fwd_rpe_pipe_kernel, persistent workgroups" if A.FWD_KERNEL == 0 else "attn_fwd_rpe_auto_kernel"
    fwd_kernel = (f"{fwd_name} (box body" + (", one rotation per pair)" if cos_sin is not None else ")")) if ran_box else f"{fwd_name} (general body" + (", rotated)" if cos_sin is not None else ")")
    bwd_obj = {"kernel": f"{bwd_kernel} (RPE table gradient from dS; " + (("rotated boxes: axis-aligned in the turned frame)" if cos_sin is not None else "axis-aligned boxes)") if ran_box else "general vertices" + (" + rotation)" if cos_sin is not None else ")")), "bound": "hbm",
               "achieved": bytes_bwd / t_bwd / 1e9, "peak": 8000.0, "unit": "GB/s",
               "frac": bytes_bwd / t_bwd / 1e9 / 8000.0, "traffic": None, "launch_us": t_bwd * 1e6,
               "rpe_scatter_per_s": 8.0 * pairs / t_bwd,
               "placement": (f"side stream of the captured step, {A.ASYNC_TABLE_GRID} persistent workgroups = CUs, concurrent with the "
                             f"backward chain on the other {256 - A.ASYNC_TABLE_GRID} (launch alone on all 256 CUs: {t_bwd_full * 1e6:.1f} us)")
                            if side else "in line on the main stream, 256 workgroups"}
    # HBM traffic per launch measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (profiles/README.md)
    try:
        tr = _pmc_traffic()
        if cfg_name == "c2":
            fwd_obj["traffic"] = tr["attn_fwd_kernel<false,true,true>"]["bytes"]
            bkey = next((k for k in tr if bwd_kernel.split("<")[0] in k), None)
            bwd_obj["traffic"] = tr[bkey]["bytes"] if bkey else None
            if "attn_bwd_kv_kernel" in tr:
                kv_obj["traffic"] = tr["attn_bwd_kv_kernel"]["bytes"]
            vi = tr[bkey].get("valu_wave_insts") if bkey else None
            if vi:  # what actually bounds the kernel: VALU issue (1024 SIMDs x 2.4 GHz / 4 cycles per wave instruction)
                bwd_obj["valu_issue"] = {"wave_insts": vi, "limit_us": vi / 614.4e9 * 1e6, "frac": vi / 614.4e9 / t_bwd,
                                         "note": "SQ_INSTS_VALU per launch (offline PMC pass) / chip issue rate / launch time: "
                                                 "the kernel is VALU-bound, the HBM fraction above is low by construction"}
    except (OSError, KeyError, IndexError, ValueError):
        pass
    fwd_obj["kernel"] = f"{fwd_kernel}: 3DV-RPE cross-attention forward"
    bwd_obj["key_side_pass"] = kv_obj
    return fwd_obj, bwd_obj


def fps_roofline(cfg_name, device, reps=5):
    """The sampling launch alone (HIP events on the launching stream) at the configuration's cloud size.  Its §8(d) figure is
    the STREAMING-EQUIVALENT traffic of the reference kernel, 20 B x n x (m-1) (12 B xyz + 4 B read + 4 B write of the running
    distance per point per round, sampling_gpu.cu:98-112); this kernel moves far less (it skips whole buckets), so the
    fraction says how far one CU's latency chain is from an HBM stream, not how busy HBM is."""
    from vdetr_amd import pointnet2_utils as PU
    npts, _, npre, *_ = CONFIGS[cfg_name]
    xyz, _ = make_scene(npts, 0, device)
    x = xyz[None].contiguous()
    n, m = int(x.shape[1]), min(npre, int(x.shape[1]))
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        PU.furthest_point_sample(x, m)
        e1.record()
        e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.mean(ts))
    eq_bytes = 20.0 * n * (m - 1)
    traffic = None
    try:
        if cfg_name == "c2":
            traffic = _pmc_traffic()["fps_rows_kernel"]["bytes"]
    except (OSError, KeyError, IndexError, ValueError):
        pass
    return {"kernel": "fps_rows_kernel (furthest point sampling, one workgroup per scene)", "bound": "hbm",
            "achieved": eq_bytes / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": eq_bytes / t / 1e9 / 8000.0, "traffic": traffic,
            "launch_us": t * 1e6, "us_per_round": t * 1e6 / max(m - 1, 1), "points": n, "samples": m,
            "note": "achieved = streaming-equivalent bytes of the reference kernel (20 B x n x (m-1)) / launch time; the kernel is a "
                    "serial latency chain on ONE CU (m-1 dependent rounds), hidden behind the decoder on a side stream"}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a port: this repo's host modules with the attention / pointnet2 entry points routed to
# the CPU restatement), on a bounded sample of the same workload
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg_name, quick=False):
    import vdetr_amd.attention as A
    from oracle import pointnet2_oracle as O
    from functools import partial
    from oracle.attention_oracle import fused_attention_reference as _ref
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.model_vdetr import build_decoder, default_args
    # the bias through eight F.grid_sample passes, i.e. the ops the reference itself runs on CPU
    fused_attention_reference = partial(_ref, rpe_impl="grid_sample")
    npts, bs, npre, nq, nl, angle_type, _ = CONFIGS[cfg_name]
    # torch's intra-op pool stops scaling on these small tensors well before a 256-thread host is full (measured on
    # the GPU box's 2 x EPYC 9575F: 1 RPE layer fwd+bwd 4.2 s @8, 4.0 s @16/32, 4.5 s @64, 100 s @256 threads)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    xyz, _ = make_scene(npts, 0, "cpu")
    t0 = time.perf_counter()
    idx = O.furthest_point_sampling(xyz[None].numpy(), npre)
    t_fps = time.perf_counter() - t0
    kxyz = xyz[torch.from_numpy(idx[0]).long()][None]
    import vdetr_amd.box_decode as BD
    from oracle.box_oracle import decode_boxes_reference
    saved = (A.fused_attention, A.begin_step, A.current_rng)
    saved_bd = BD.decode_boxes
    A.fused_attention, A.begin_step, A.current_rng = fused_attention_reference, (lambda dev: None), (lambda dev: None)
    BD.decode_boxes = decode_boxes_reference
    import vdetr_amd.add_ln as ALN
    from oracle import add_ln_oracle
    saved_ln = (ALN.layer_norm, ALN.add_dropout_layer_norm)
    ALN.layer_norm, ALN.add_dropout_layer_norm = add_ln_oracle.layer_norm, add_ln_oracle.add_dropout_layer_norm
    try:
        times = {}
        # the whole decoder (FFN stage + all RPE layers) once; --quick-cpu-baseline: 1 and 2 RPE layers, scaled linearly
        for layers in ((2, 3) if quick else (nl,)):
            torch.manual_seed(0)
            dec = build_decoder(default_args(dec_nlayers=layers, nqueries=nq), ScannetDatasetConfig()).train()
            feats = torch.randn((kxyz.shape[1], 1, 256), requires_grad=True)
            dims = [kxyz.min(1)[0], kxyz.max(1)[0]]
            scene = dims[1] - dims[0]
            enc = {"center_normalized": (kxyz - dims[0][:, None]) / scene[:, None],
                   "size_normalized": torch.ones_like(kxyz) / scene[:, None]}
            t0 = time.perf_counter()
            out, _ = dec(None, feats, kxyz, kxyz, dims, query_pos=kxyz, enc_box_predictions=enc, enc_box_features=feats)
            loss_fn(out).backward()
            times[layers] = time.perf_counter() - t0
            del dec, out, feats
    finally:
        A.fused_attention, A.begin_step, A.current_rng = saved
        BD.decode_boxes = saved_bd
        ALN.layer_norm, ALN.add_dropout_layer_norm = saved_ln
    what = (f"FPS {npts}->{npre} pts with the C oracle on 1 thread ({t_fps:.2f} s) + decoder fwd+bwd at full nQ={nq} / nK={npre} through "
            f"the torch CPU oracle (RPE via F.grid_sample, as the reference) on {cores} threads")
    if quick:
        per_layer = max(times[3] - times[2], 1e-9)
        full = t_fps + times[2] + (nl - 2) * per_layer
        return {"value": 1.0 / full, "unit": "scenes/s", "cores": cores, "kind": "port", "extrapolated": True,
                "kind_note": "port, EXTRAPOLATED: timed on a bounded sample (below) and scaled linearly to the full layer count",
                "sample": f"EXTRAPOLATED from a bounded sample: {what}, with 1 and 2 of {nl - 1} RPE layers ({times[2]:.1f} s, {times[3]:.1f} s), "
                          f"extrapolated linearly to {nl - 1} layers = {full:.1f} s/scene"}
    full = t_fps + times[nl]
    return {"value": 1.0 / full, "unit": "scenes/s", "cores": cores, "kind": "port", "extrapolated": False,
            "kind_note": "port: the CPU restatement of the same step, one whole scene timed (no extrapolation)",
            "sample": f"one scene of the workload, one step: {what}, all {nl - 1} RPE layers + the FFN stage and the {nl} head stages "
                      f"({times[nl]:.1f} s) = {full:.1f} s/scene"}


def self_launch(ngpus):
    """Run this script as `ngpus` ranks of ONE node (a child `python -m torch.distributed.run`, rendezvous on 127.0.0.1 at a
    free port), pass rank 0's JSON line through and return the children's exit code.  The parent never initialises the GPU
    and never replaces itself (no exec): it waits for the child and leaves with its code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // ngpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] starting {ngpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        proc.wait()
        return 130


def launch_check(a):
    """What the ranks of `--launch-check` run: the process group of `--backend` over the launcher's environment, one
    all-reduce, rank 0's JSON line.  No GPU work (gloo) — it exists so that the start-up path of `--gpus N` has a CPU test."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.distributed.init_process_group(backend=a.backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if a.backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        t = t.cuda()
    torch.distributed.all_reduce(t)
    from vdetr_amd.dist import avg_reduce_supported
    avg = avg_reduce_supported()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": a.backend, "sum_of_ranks_plus_1": float(t.item()),
                          "reduce_op_avg": bool(avg)}), flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    return 0


def main():
    # a run that is still going after VDETR_BENCH_WATCHDOG seconds (default 900) prints every thread's Python stack to stderr: where a
    # hung run hangs (round 6 saw two `--config c5` runs out of ~25 sit forever without a line of output)
    import faulthandler
    wd = float(os.environ.get("VDETR_BENCH_WATCHDOG", "900"))
    if wd > 0:
        faulthandler.dump_traceback_later(wd, repeat=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="auto", choices=["auto", "f32", "bf16"], help="arithmetic of the cross attention's QK^T / PV "
                    "(auto: bf16 for c4, as BASELINE names it, f32 otherwise); softmax, RPE table and accumulators are f32 either way")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of captured hipGraphs")
    ap.add_argument("--sync-bn", action="store_true", help="batch statistics over all ranks, as the reference's SyncBatchNorm conversion (main.py:512-514)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="time 1 and 2 RPE layers on the CPU and scale to the full count, instead of the whole decoder once (~30 s)")
    ap.add_argument("--loss", default="synthetic", choices=["synthetic", "criterion"],
                    help="synthetic: scalar loss of SURVEY 8d (headline); criterion: the device set criterion on synthetic boxes")
    ap.add_argument("--no-defer-wg", action="store_true", help="weight gradients inside the backward, one GEMM per layer")
    ap.add_argument("--no-criterion-leg", action="store_true", help="skip the extra N=1 measurement with the set criterion")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the extra N=1 measurement with exact f32 products in the attention (arith.exact_f32)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-backbone-leg", action="store_true", help="skip the extra N=1 measurement with the sparse-conv backbone")
    ap.add_argument("--no-fps-prefetch", action="store_true", help="run FPS serially in front of the decoder")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="library heuristics instead of per-shape tuned GEMM solutions")
    ap.add_argument("--with-backbone-dist", action="store_true", help="N>1: also time the step with the sparse-conv backbone on "
                    "every rank (decoder gradients all-reduced under the backbone's backward)")
    ap.add_argument("--force-dist", action="store_true", help="N=1 only: a 1-rank RCCL communicator, so that the multi-GPU step "
                    "(captured all-reduces on the side stream) runs on a single GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test "
                    "the N>1 code path with several ranks on one GPU)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, form the process group, all-reduce one "
                    "number, print a JSON line and leave: the launcher path without a GPU (tests/test_dist_gloo.py)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` with nobody else having started the ranks: start them here, one process per GPU, as the
        # reference's main.py:588-593 spawns its workers itself.  Nothing in this parent has touched the GPU.
        sys.exit(self_launch(a.gpus))
    if a.launch_check:
        sys.exit(launch_check(a))

    from vdetr_amd.dist import broadcast_parameters, init_distributed
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if a.backend != "nccl":
        os.environ["LOCAL_RANK"] = str(local)
    rank, local, world = init_distributed(a.backend)
    if a.force_dist and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group(backend=a.backend, rank=0, world_size=1,
                                             **({"device_id": torch.device("cuda", local)} if a.backend == "nccl" else {}))
    assert world == a.gpus or (world == 1 and a.gpus == 1), f"--gpus {a.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local)
    if not a.no_gemm_tuning:
        try:
            from vdetr_amd.runtime import enable_gemm_tuning
            enable_gemm_tuning(rank)
        except Exception as exc:  # tuning is an optimisation: the library heuristics still work
            print(f"[bench] GEMM tuning unavailable ({exc}); continuing with library defaults", file=sys.stderr)

    model = build_model(a.config, device)
    dtype = a.dtype if a.dtype != "auto" else ("bf16" if a.config == "c4" else "f32")
    if dtype == "bf16":  # BASELINE config 4: q / k / v of the 3DV-RPE cross attention stored as bf16, QK^T / PV on the bf16 matrix cores
        from vdetr_amd.vdetr_transformer import set_attention_dtype
        set_attention_dtype(model, torch.bfloat16)
    # the fallback chain of the captured step (see relaunch() below): level 0 = one graph with the gradient all-reduce inside,
    # 1 = the all-reduce outside the graph (two graphs per step), 2 = no graph.  A level above 0 is always a child process.
    fallback = int(os.environ.get("VDETR_BENCH_FALLBACK", "0"))
    if fallback >= 1:
        os.environ["VDETR_PHASED_REDUCE"] = "0"
    use_graph = not a.no_graph and fallback < 2
    if world > 1:
        broadcast_parameters(model)
    if a.sync_bn:  # batch statistics over all ranks inside the fused BatchNorm launches (bn_act.set_sync), graph-capturable
        from vdetr_amd import bn_act as _bna
        _bna.set_sync(True, force=a.force_dist)
    inputs = make_inputs(a.config, device, rank)
    def make_trainer(with_criterion, fps_at_layer=None):
        crit = targets = None
        if with_criterion:
            from vdetr_amd.criterion import build_criterion, default_criterion_args
            crit = build_criterion(default_criterion_args(), model.dataset_config)
            targets = make_targets(a.config, device, rank)
            crit_holder[0], crit_holder[1] = crit, targets
        return Trainer(model, inputs, world, use_graph, overlap=True, fps_prefetch=not a.no_fps_prefetch, criterion=crit,
                       targets=targets, defer_wg=not a.no_defer_wg, force_dist=a.force_dist, fps_at_layer=fps_at_layer)

    def replay_ms(tr, reps=6, settle=3):
        for _ in range(settle):
            tr.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            tr.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def model_state():
        with torch.no_grad():
            return [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())]

    def choose_fps_depth(tr):
        """depth 1 (the next scene's sampling as a forked branch of the captured step) or depth 2 (two scenes' samplings in
        flight on side streams outside the graph)?  Measured, not guessed: the one-CU sampling kernel takes 2.1 ms (4k points)
        to 10.7 ms (80k), the rest of the step 3 to 11 ms.  Every rank decides for itself (the graphs are rank-local)."""
        if os.environ.get("VDETR_FPS_DEPTH", "auto") != "auto" or a.no_fps_prefetch or tr.fps_depth2:
            return tr
        ncap = [0]

        def captured(t):
            """t.capture(), then the ranks agree THROUGH THE STORE whether it worked everywhere, before any of them goes on into a
            device collective (over_ranks): a rank whose capture failed can issue no device work and would wait alone at the
            outer agreement while the healthy ranks hang in the all-reduce until the communicator's timeout."""
            err = None
            try:
                t.capture()
            except Exception as e:
                err = e
            ncap[0] += 1
            if any_rank_failed(err is not None, f"depth_capture{ncap[0]}"):
                raise err if err is not None else RuntimeError("a capture of the sampling look-ahead choice failed on another rank")
            return t
        snap = model_state()
        t1 = replay_ms(tr)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        model.sample_indices(inputs)
        ev0.record()
        model.sample_indices(inputs)
        ev1.record()
        ev1.synchronize()
        t_fps = ev0.elapsed_time(ev1)

        def over_ranks(x):  # the ranks must take the same branches (the steps below contain collectives): max over the ranks
            if world == 1:
                return x
            t = torch.tensor([x], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        t1, t_fps = over_ranks(t1), over_ranks(t_fps)
        if t_fps <= 0.8 * t1:
            # One scene's sampling in flight is enough.  WHERE it is forked matters: the kernel holds a CU for t_fps, and every launch
            # of the chain whose workgroups are one-per-CU (and most library GEMM grids at this size) pays a second round next to it
            # — 0.55 ms of the C2 step (tools/probes/step_bounds.py nofps).  Forked in front of decoder layer k of the forward
            # instead of at the step's start it overlaps less of the forward, as long as it still ends before the step does:
            # largest k with (start of layer k) + 1.12 t_fps (its in-step duration) <= t1 - 1.2 ms, the forward taking ~0.37 t1.  That step is
            # captured and timed as well; kept only if measured faster.
            nl = len(getattr(getattr(model, "decoder", None), "layers", ()))
            if os.environ.get("VDETR_BENCH_FPS_AT_LAYER") is not None or nl == 0 or tr.fps_at_layer >= 0:
                return tr
            tr.reset_state(snap)
            t1 = over_ranks(replay_ms(tr, reps=12, settle=25))  # (the first replays after a capture run ~1.5 % slow)
            tr.reset_state(snap)
            k = fps_fork_layer(t1, t_fps, nl)
            if k < 1:  # (no room behind a later fork / the one CU is held for a small part of the step anyway)
                return tr

            crit = tr.criterion and crit_holder[0]

            def build(at):
                # (every Trainer re-lays the parameters into flat buffers of its own: only the last one built has a live graph)
                t = Trainer(model, inputs, world, True, overlap=True, fps_prefetch=True, criterion=crit,
                            targets=crit_holder[1], defer_wg=not a.no_defer_wg, fps_depth=1, force_dist=a.force_dist, fps_at_layer=at)
                captured(t)
                t.reset_state(snap)
                return t
            # the latest layer the bound allows and the two in front of it: captured and timed, the fastest measured form kept
            timed = []
            for at in [x for x in (k, k - 1, k - 2) if x >= 1]:
                trk = build(at)
                timed.append((over_ranks(replay_ms(trk, reps=12, settle=25)), at))
            tk, best = min(timed)
            if rank == 0:
                print(f"[bench] sampling {t_fps:.2f} ms alone; captured step {t1:.2f} ms with it forked at the start, "
                      + ", ".join(f"{t:.2f} ms in front of decoder layer {at}" for t, at in timed)
                      + f": {'layer ' + str(best) if tk < 0.998 * t1 else 'start'}", file=sys.stderr)
            if tk < 0.998 * t1:
                if best != timed[-1][1]:
                    trk = build(best)  # (only the last Trainer built has a live graph)
                trk.reset_state(snap)
                return trk
            return build(-1)
        crit = tr.criterion and crit_holder[0]
        tr2 = Trainer(model, inputs, world, True, overlap=True, fps_prefetch=True, criterion=crit,
                      targets=crit_holder[1], defer_wg=not a.no_defer_wg, fps_depth=2, force_dist=a.force_dist)
        captured(tr2)
        t2 = over_ranks(replay_ms(tr2))
        tr2.reset_state(snap)
        if rank == 0:
            print(f"[bench] sampling {t_fps:.2f} ms vs captured step {t1:.2f} ms (one scene's sampling in flight) / {t2:.2f} ms (two): "
                  f"fps_lookahead {2 if t2 < 0.97 * t1 else 1}", file=sys.stderr)
        if t2 < 0.97 * t1:  # (a tie goes to the simpler form: one side stream, no eager launches between the replays)
            return tr2
        # (tr2 has re-laid the parameters into flat buffers of its own: the first trainer's graph points at the old ones)
        tr1 = Trainer(model, inputs, world, True, overlap=True, fps_prefetch=True, criterion=crit,
                      targets=crit_holder[1], defer_wg=not a.no_defer_wg, fps_depth=1, force_dist=a.force_dist)
        captured(tr1)
        tr1.reset_state(snap)
        return tr1

    crit_holder = [None, None]
    trainer = make_trainer(a.loss == "criterion")
    graph_ok = False
    def all_ranks(ok):
        """every rank takes the same branch of the fallback chain: a rank that replays a graph with collectives inside while
        another one issues them eagerly would wait for each other forever"""
        if world > 1:
            t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
            return bool(t.item())
        return ok

    def any_rank_failed(failed, tag):
        if world == 1:
            return failed
        return agree_any_failed(torch.distributed.distributed_c10d._get_default_store(), world, failed, tag)

    def child_port(parent_port, level):
        """rendezvous port of the child processes: rank 0 asks the OS for a free one and publishes it in the parent's store (the
        ranks of one node share 127.0.0.1); `parent_port + 1` may belong to somebody else"""
        if world == 1:
            return free_port(parent_port + 1)
        store = torch.distributed.distributed_c10d._get_default_store()
        key = f"bench_child_port_{level}"
        if rank == 0:
            store.set(key, str(free_port(parent_port + 1)))
        store.wait([key])
        return int(store.get(key))

    def relaunch(level):
        """The next form of the step — level 1: collectives outside the graph, level 2: no graph — in a CHILD process per rank
        (same RANK / WORLD_SIZE, rendezvous one port further; the parent only waits and leaves with the child's exit code: it
        never replaces itself).  The child writes the JSON line."""
        import subprocess
        env = dict(os.environ)
        env["VDETR_BENCH_FALLBACK"] = str(level)
        if "MASTER_PORT" in env:
            env["MASTER_PORT"] = str(child_port(int(env["MASTER_PORT"]), level))
        env.pop("TORCHELASTIC_RUN_ID", None)
        for k in ("TORCHELASTIC_USE_AGENT_STORE",):  # the child forms its own store on the new port
            env[k] = "False"
        if rank == 0:
            print(f"[bench] capture failed on at least one rank: every rank restarts the step in a child process at fallback level {level} "
                  f"({'collectives outside the graph' if level == 1 else 'no graph'})", file=sys.stderr, flush=True)
        sys.stdout.flush()
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env)
        code = proc.wait()
        os._exit(code)  # (no interpreter shutdown: the failed capture's graph object cannot be destroyed)

    if use_graph:
        err = None
        try:
            trainer.capture()
        except Exception as e:  # capture is an optimisation: report and fall back
            err = e
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {str(e).splitlines()[0]})", file=sys.stderr, flush=True)
        if any_rank_failed(err is not None, "capture"):
            relaunch(fallback + 1)
        graph_ok = True
        try:
            trainer = choose_fps_depth(trainer)
        except Exception as e:  # (a capture inside the choice failed: the same dead end, the child keeps depth 1)
            print(f"[bench] rank {rank}: sampling look-ahead choice failed ({type(e).__name__}: {str(e).splitlines()[0]})", file=sys.stderr, flush=True)
            os.environ["VDETR_FPS_DEPTH"] = "1"
            err = e
        if any_rank_failed(err is not None, "depth"):
            os.environ["VDETR_FPS_DEPTH"] = "1"
            relaunch(fallback)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # Part of the set-up, like the eager steps in front of the capture: the first replays of a freshly captured step run ~1.5 %
    # slower than the steady state (20 timed steps after 0 / 3 / 5 / 30 warm-up steps: 8.19 / 8.15 / 8.10 / 8.06 ms), whatever W
    # the caller asks for.  Real training steps on every rank; recorded in config.settle_steps.
    settle = int(os.environ.get("VDETR_BENCH_SETTLE", "25")) if graph_ok else 0
    for _ in range(settle):
        trainer.step()
    for _ in range(a.warmup):
        trainer.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(trainer.loss.item())
    assert np.isfinite(loss), "non-finite loss"

    npts, bs, npre, nq, nl, _, desc = CONFIGS[a.config]
    result = {
        "metric": "scenes/sec (train fwd+bwd) 40k-pt ScanNet", "value": world * bs * a.steps / dt, "unit": "scenes/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": desc, "global_batch": world * bs, "voxels_per_scene": int(inputs["backbone_xyz"][0].shape[0]),
                   "keys": npre, "queries": nq, "rpe_layers": nl - 1, "parallelism": f"dp{world}",
                   "step": "FPS+gather, projection, decoder fwd, loss, backward, grad all-reduce (N>1), clip, AdamW",
                   "hip_graph": graph_ok, "fallback_level": fallback, "settle_steps": settle, "sync_bn": bool(a.sync_bn), "fps_prefetch": not a.no_fps_prefetch,
                   "fps_lookahead": 2 if getattr(trainer, "fps_depth2", False) and graph_ok else (1 if not a.no_fps_prefetch else 0),
                   "fps_fork_layer": getattr(trainer, "fps_at_layer", -1),
                   "grad_allreduce_bytes": trainer.reducer.grad_bytes(), "grad_allreduce_buckets": len(trainer.reducer.buckets),
                   "grad_allreduce_model": trainer.bucket_model,
                   "grad_allreduce": ("none (1 rank)" if not trainer.reducer.active else
                                      "inside the hipGraph, bucket by bucket on a side stream while the next bucket's weight gradients are computed"
                                      if trainer.phased and graph_ok else
                                      "bucket hooks on a side stream during backward" if trainer.hooked else "after the replayed backward")},
        "loss": loss,
        "arith": {"activations": "f32" if dtype == "f32" else "f32 tensors everywhere; q / k / v of the cross attention ROUNDED to bf16 (nearest even) on the way into QK^T / PV (vdetr_attn_desc.fwd_kernel 3)",
                  "qk_pv": ("3DV-RPE cross attention forward: f32 operands as bf16 parts on the bf16 matrix unit, f32 accumulate — QK^T three parts, "
                            "six terms (2^-24 per product: f32 accuracy), PV two parts, three terms (2^-16 per product, output 8e-6 relative); "
                            "query self-attention and VDETR_FWD_KERNEL=2: v_mfma_f32_16x16x4_f32 (exact f32)") if dtype == "f32" else "v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16 on one bf16 part per operand, f32 accumulate; backward as in the f32 configuration (self-attention: f32)", "softmax_log2_table_lookup": "f32",
                  "dtable_products": "exact f32 outer products on v_mfma_f32_16x16x4_f32 (attn_bwd_box4_kernel: axis-aligned and rotated boxes; "
                                     "arbitrary vertices take the general kernel: split-bf16 2^-15)",
                  "dtable_accum": "int32 fixed point in LDS",
                  "backward_contractions": "dO V^T, dV = P^T dO, dK = dS^T q: f32 operands as hi + lo bf16 on v_mfma_f32_32x32x16_bf16, "
                                           "three cross terms (2^-16 per product), f32 accumulate; dQ = dS K: library f32 GEMM",
                  "note": "dtype f32 is the arithmetic of every tensor the model sees; the RPE-table gradient's group sums are rounded "
                          "into an int32 fixed-point histogram whose scale comes from a worst-case bound (DESIGN.md 4.4): relative L2 against the "
                          "fp64 oracle at this layer size in tests/test_gpu_attention.py::test_full_size_forward_backward_vs_oracle"},
    }
    if a.config == "c2":  # decoder fwd+bwd = 216 GFLOP per scene at the full configuration
        result["end_to_end"] = {"gflop_per_scene": 216.0, "achieved_tflops": 216.0e-3 * result["value"],
                                "note": "algorithmic decoder flops (SURVEY 8d) x scenes/s, all GPUs"}
    def leg(name, fn):
        """The headline line must be printed whatever happens in a secondary measurement."""
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            result[name] = {"error": f"{type(exc).__name__}: {exc}"}
            print(f"[bench] {name} leg failed: {type(exc).__name__}: {exc}", file=sys.stderr)

    def roofline_leg():
        fwd_obj, bwd_obj = kernel_rooflines(a.config, device)
        fps_obj = fps_roofline(a.config, device)
        layers = nl - 1
        fwd_obj["step_us"], bwd_obj["step_us"] = layers * fwd_obj["launch_us"], layers * bwd_obj["launch_us"]
        # one workgroup per scene: the launch time does not grow with the batch
        fps_obj["step_us"] = fps_obj["launch_us"]
        # `roofline` = the kernel with the largest share of the step's CRITICAL PATH: the main stream's cross-attention kernels.
        # The sampling of the next scene runs on a side stream under them (one CU): reported as `side_stream`, with the share of
        # the step its launch would take if it were not hidden.
        for o in (fps_obj, bwd_obj, fwd_obj):
            o["share_of_step"] = o["step_us"] * 1e-3 / result["ms_per_step"]
        fps_obj["on_critical_path"] = bool(a.no_fps_prefetch)
        # (round 4: the table gradient runs on a side branch of the captured step over 192 of the 256 CUs, NEXT TO the backward
        # chain — not hidden like the one-CU sampling: it takes three quarters of the chip from the chain while it runs, and the
        # two are about balanced, so it stays the kernel this line's `roofline` describes)
        bwd_obj["on_critical_path"] = "shares the chip with the backward chain" if "side stream" in bwd_obj.get("placement", "") else True
        objs = sorted((bwd_obj, fwd_obj) + ((fps_obj,) if a.no_fps_prefetch else ()), key=lambda o: -o["step_us"])
        result["roofline"] = objs[0]
        result["roofline_secondary"] = objs[1]
        if len(objs) > 2:
            result["roofline_tertiary"] = objs[2]
        else:
            result["side_stream"] = fps_obj

    def cpu_leg():
        result["cpu_baseline"] = cpu_baseline(a.config, quick=a.quick_cpu_baseline)

    def backbone_leg():
        # the step with the sparse-convolution backbone in front, on a synthetic 40k-point room scan
        from vdetr_amd.runtime import defer_weight_grads
        bt = BackboneTrainer(a.config, device, force_dist=a.force_dist, scene_seed=rank)
        if world > 1:
            from vdetr_amd.dist import broadcast_parameters as _bcast
            _bcast(bt.model)
        try:
            if use_graph:
                bt.capture()
            if os.environ.get("VDETR_BENCH_GEOMETRY", "thread") == "thread":
                bt.calibrate_side_stream()
            for _ in range(max(a.warmup, 8)):  # (the two-scene queue and the allocator pools of its two streams take a few steps to settle)
                bt.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                bt.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            if world > 1:  # the slowest rank's time, as for the headline
                tt = torch.tensor([ms], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                ms = float(tt.item())
            if os.environ.get("VDETR_BENCH_TIMELINE"):  # device time of the step's phases (events on the main stream)
                bt.marks = []
                for _ in range(10):
                    bt.step()
                torch.cuda.synchronize()
                m, bt.marks = bt.marks, None
                names = ["backbone fwd", "decoder graph", "backbone bwd", "pack+clip+adamw", "gap to next step"]
                acc = [0.0] * 5
                for i in range(0, len(m) - 5, 5):
                    for j in range(5):
                        acc[j] += m[i + j].elapsed_time(m[i + j + 1])
                print("[bench] with_backbone timeline (ms):", {k: round(v / ((len(m) - 5) // 5), 2) for k, v in zip(names, acc)}, file=sys.stderr)
            bloss = float(bt.loss.item())
            assert np.isfinite(bloss), "non-finite loss with the backbone"
            result["with_backbone"] = {
                "ms_per_step": ms, "scenes_per_s": world * 1e3 / ms, "n_gpus": world, "geometry_ms": bt.geometry_ms, "input_points": 40000,
                "loader_stream_calibration_ms": getattr(bt, "calibration_ms", None),
                "grad_allreduce_bytes": bt.reducer.grad_bytes() if bt.reducer.active else 0,
                "grad_allreduce": ("decoder bucket(s) on the side stream under the backbone's backward; backbone buckets of 64 MB sent from "
                                   "post-accumulate hooks while that backward is still running" if getattr(bt, "bb_overlap", False) else
                                   "after the backbone's backward" if bt.reducer.active else "none (1 rank)"),
                "grad_allreduce_buckets": len(bt.reducer.buckets) if bt.reducer.active else 0,
                "backbone_buckets_sent_during_backward": getattr(bt, "sent_in_backward", 0),
                "voxels_per_stride": bt.voxels, "backbone_parameters": sum(p.numel() for p in bt.bb_params), "loss": bloss,
                "note": "raw points -> voxels -> MinkResNet34 + FPN (HIP kernel maps, fused pair-list convolutions — wide layers as split-bf16 products (2^-16 each) on the bf16 matrix unit, f32 accumulate —, fused BatchNorm; eager) -> FPS tokens -> "
                        "decoder step (captured hipGraph) -> backbone backward -> clip + AdamW over all 79 M parameters; "
                        "the geometry (voxel sites, kernel maps, pair lists) and FPS indices of the scene after next are built from its "
                        "coordinates on a side stream inside every timed step (a two-scene loader queue); geometry_ms = that work alone"}
        finally:
            bt.close()
            defer_weight_grads(not a.no_defer_wg)

    def criterion_leg():
        # the same step with the reference's real loss: matcher + Hungarian + losses on the device
        t2 = make_trainer(True, fps_at_layer=getattr(trainer, "fps_at_layer", None))  # (forked where the headline step forks it)
        if use_graph:
            t2.capture()
        for _ in range(3):
            t2.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            t2.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        closs = float(t2.loss.item())
        assert np.isfinite(closs), "non-finite criterion loss"
        # the same criterion through the oracle (torch CPU + scipy: what the reference runs, minus its host syncs)
        from oracle import criterion_oracle as CO
        with torch.no_grad():
            out = model(inputs)
        cpu = lambda o: {k: v.detach().cpu().requires_grad_(v.is_floating_point()) for k, v in o.items()  # noqa: E731
                         if torch.is_tensor(v) and not k.startswith("_")}
        oc = {"outputs": cpu(out["outputs"]), "aux_outputs": [cpu(o) for o in out["aux_outputs"]],
              "seed_xyz": out["seed_xyz"].cpu(), "enc_outputs": cpu(out["enc_outputs"])}
        tc = {k: v.cpu() for k, v in make_targets(a.config, device, rank).items()}
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        t0 = time.perf_counter()
        ref_loss = CO.set_criterion(oc, tc)[0]
        ref_loss.backward()
        cpu_ms = (time.perf_counter() - t0) * 1e3
        result["criterion"] = {"ms_per_step": ms, "cpu_oracle_ms": cpu_ms, "scenes_per_s": bs / ms * 1e3, "added_ms": ms - result["ms_per_step"],
                               "loss": closs, "note": "same step with the set criterion (focal + L1 + GIoU on Hungarian "
                               "matches, 9 stages, 24 boxes/scene x repeat 5) instead of the synthetic scalar loss; "
                               "cpu_oracle_ms = the criterion alone (fwd+bwd) through oracle/criterion_oracle.py"}

    def exact_leg():
        # The label `dtype: f32`, made auditable: the same step with EXACT f32 products in the attention — the
        # forward on v_mfma_f32_16x16x4_f32 (vdetr_attn_desc.fwd_kernel 2) and the shared-K/V backward as library f32 GEMMs around
        # the element-wise kernel (no split-bf16 operand anywhere) — captured and timed like the headline; and what the split forms
        # differ by from it, measured on one cross-attention layer of this configuration's size (no dropout: same masks either way).
        from vdetr_amd import attention as _A
        keep = (_A.FWD_KERNEL, _A.FUSED_KV_BWD)
        try:
            B_, nQ_, nK_ = 1, nq, npre
            g = torch.Generator().manual_seed(11)
            kxyz = (1 + torch.rand((B_, nK_, 3), generator=g) * torch.tensor([8.0, 6.0, 3.0])).to(device)
            center = kxyz[:, torch.randperm(nK_, generator=g)[:nQ_].to(device)]
            half = (0.1 + torch.rand((B_, nQ_, 1, 3), generator=g)).to(device)
            signs = torch.tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]],
                                 dtype=torch.float32, device=device)
            verts = (center[:, :, None, :] + half * signs).contiguous()
            base = [torch.randn(sh, generator=g).to(device) for sh in ((B_, nQ_, 256), (B_, nK_, 64), (B_, nK_, 64), (8, 10, 10, 10, 4))]
            wout = torch.randn((B_, nQ_, 256), generator=g).to(device)
            got = {}
            for name, (fk, fused) in {"split": (0, True), "exact": (2, False)}.items():
                _A.FWD_KERNEL, _A.FUSED_KV_BWD = fk, fused
                ts = [t.clone().requires_grad_(True) for t in base]
                o = _A.fused_attention(ts[0], ts[1], ts[2], num_heads=4, scale=0.125, shared_kv=True, table=ts[3], rpe=_A.RPEConfig(),
                                       vertices=verts, xyz=kxyz)
                (o * wout).sum().backward()
                got[name] = [o.detach()] + [t.grad for t in ts]
            rel = lambda x, y: float((x - y).abs().max() / y.abs().max())  # noqa: E731
            diffs = dict(zip(("out", "dq", "dk", "dv", "dtable"), (rel(x, y) for x, y in zip(got["split"], got["exact"]))))
            _A.FWD_KERNEL, _A.FUSED_KV_BWD = 2, False
            t3 = make_trainer(False, fps_at_layer=getattr(trainer, "fps_at_layer", None))
            if use_graph:
                t3.capture()
            for _ in range(3):
                t3.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                t3.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            assert np.isfinite(float(t3.loss.item())), "non-finite loss in the exact-f32 step"
            result["arith"]["exact_f32"] = {"ms_per_step": ms, "value": bs * world / ms * 1e3,
                                            "what": "attention forward on v_mfma_f32_16x16x4_f32 (fwd_kernel 2), shared-K/V backward as library "
                                                    "f32 GEMMs (no split-bf16 products); everything else as in the headline step"}
            result["arith"]["max_rel_vs_exact"] = dict(diffs, note=f"one cross-attention layer ({nQ_} queries x {nK_} keys, boxes), split forms "
                                                       "vs exact f32 products, max |diff| / max |exact| per tensor")
        finally:
            _A.FWD_KERNEL, _A.FUSED_KV_BWD = keep

    # The secondary measurements must not cost the headline: if they are still running after VDETR_BENCH_LEG_TIMEOUT seconds (a hung
    # GPU — round 6 met a library GEMM that never returns, DESIGN.md 4.7 — or a CPU oracle on a loaded host), a watchdog thread prints the
    # line as it stands and leaves.  One rank only: with several, the legs below are the roofline micro-benchmark alone.
    import threading
    emitted, emit_lock = [False], threading.Lock()

    def emit():
        with emit_lock:
            if not emitted[0]:
                emitted[0] = True
                print(json.dumps(result), flush=True)

    legs_done = threading.Event()
    leg_timeout = float(os.environ.get("VDETR_BENCH_LEG_TIMEOUT", "600"))

    def legs_watchdog():
        if not legs_done.wait(timeout=leg_timeout):
            result["secondary_measurements"] = {"error": f"still running after {leg_timeout:.0f} s: the headline measurement is complete, the "
                                                         "line is printed without what is missing and the process leaves"}
            print(f"[bench] secondary measurements still running after {leg_timeout:.0f} s: printing the line and leaving", file=sys.stderr)
            emit()
            sys.stdout.flush()
            os._exit(0)
    if rank == 0 and world == 1 and leg_timeout > 0:
        threading.Thread(target=legs_watchdog, daemon=True).start()

    if world > 1 and a.with_backbone_dist and a.config == "c2":
        leg("with_backbone", backbone_leg)  # collective: every rank runs it (own scene each); opt-in, it follows the headline
    if rank == 0:
        if not a.no_roofline:
            leg("roofline", roofline_leg)
        # the GPU legs first: the CPU baseline leaves 32 OpenMP workers behind that take turns with the launch threads
        # (with_backbone measured 44.7 ms per step after it, 30.3 before)
        if world == 1 and a.config == "c2" and not a.no_backbone_leg:
            leg("with_backbone", backbone_leg)
        if world == 1 and dtype == "f32" and a.loss == "synthetic" and not a.no_exact_leg:
            leg("exact_f32_error", exact_leg)  # (writes result["arith"]["exact_f32"]; this key only appears with an error)
        if world == 1 and a.loss == "synthetic" and not a.no_criterion_leg:
            leg("criterion", criterion_leg)  # (times the criterion's CPU oracle as well: after the backbone leg for the same reason)
        if not a.no_cpu_baseline and world == 1:
            leg("cpu_baseline", cpu_leg)
    legs_done.set()
    if rank == 0:
        emit()
    # Leave in order: everybody done, the group destroyed.  Only a process whose captured graphs hold RCCL nodes then ends
    # without the interpreter's teardown (their destructors were seen to crash at exit, after the line above); it runs what
    # atexit would have run first (the GEMM-tuning results), and a shutdown that failed is reported through the exit code.
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rc = 0
        try:
            torch.cuda.synchronize()
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] process-group shutdown: {type(exc).__name__}: {exc}", file=sys.stderr)
            rc = 3
        rccl_in_graph = bool(graph_ok and getattr(trainer, "phased", False) and trainer.reducer.active and a.backend == "nccl")
        if (rccl_in_graph and os.environ.get("VDETR_BENCH_NORMAL_EXIT") != "1") or rc:  # (NORMAL_EXIT: under a profiler that writes its trace at exit)
            try:
                from vdetr_amd.runtime import finish_gemm_tuning
                finish_gemm_tuning()
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] GEMM-tuning results not published: {type(exc).__name__}: {exc}", file=sys.stderr)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(rc)


if __name__ == "__main__":
    main()
