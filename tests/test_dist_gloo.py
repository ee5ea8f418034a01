"""N>1 path on CPU: two gloo ranks, scene-sharded gradients averaged by GradientReducer (the RCCL path on GPUs)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, q):
    overlap = mode == "hooks"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vdetr_amd.dist import GradientReducer, broadcast_parameters, init_distributed
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    for p in model[3].parameters():
        p.requires_grad_(True)
    broadcast_parameters(model)
    red = GradientReducer(model.parameters(), bucket_mb=0.0003, overlap=overlap,  # several tiny buckets
                          bucket_views=mode != "pack")
    assert len(red.buckets) > 1
    torch.manual_seed(rank)        # each rank sees its own "scene"
    x = torch.randn(5, 8)
    for step in range(2):
        red.zero_grad()
        out = model[2](model[1](model[0](x)))  # model[3] is unused: its gradient stays zero, bucket flushed by finish()
        out.square().sum().backward()
        if mode == "hooks":
            red.finish()
        elif mode == "after":
            red.reduce_all()
        else:
            red.pack_and_reduce()
    # numpy (pickled by value): torch tensors would travel as shared-memory handles that die with the worker
    q.put((rank, [(torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy() for p in model.parameters()],
           [p.detach().numpy().copy() for p in model.parameters()], x.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conv = lambda r: (r[0], [torch.from_numpy(a) for a in r[1]], [torch.from_numpy(a) for a in r[2]], torch.from_numpy(r[3]))
    (_, g0, w0, x0), (_, g1, w1, x1) = conv(res[0]), conv(res[1])
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)            # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, atol=1e-7)   # both ranks hold the same averaged gradient
    # and it IS the average of the two local gradients
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    with torch.no_grad():
        for p, w in zip(model.parameters(), w0):
            p.copy_(w)
    exp = None
    for x in (x0, x1):
        model.zero_grad()
        model[2](model[1](model[0](x))).square().sum().backward()
        gs = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in model.parameters()]
        exp = gs if exp is None else [a + b for a, b in zip(exp, gs)]
    for a, e in zip(g0, exp):
        assert torch.allclose(a, e / 2, atol=1e-6)


def test_gradient_reducer_overlap_hooks():
    _run("hooks")


def test_gradient_reducer_after_backward():
    _run("after")


def test_gradient_reducer_pack_after_backward():
    """gradients stay ordinary tensors and are packed into the buckets with one multi-tensor copy"""
    _run("pack")


def test_flat_params_update_matches_per_parameter_update():
    """FlatParams + one-tensor fused AdamW with the clip coefficient as grad_scale == clip_grad_norm_ + AdamW over the
    individual parameters (engine.py:105-107)."""
    import copy
    from vdetr_amd.dist import FlatParams
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.LayerNorm(3))
    m1.add_module("unused", torch.nn.Linear(2, 2))  # parameters without a gradient
    m2 = copy.deepcopy(m1)
    x = torch.randn(11, 7)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2, weight_decay=0.1)
    flat = FlatParams(m2.parameters())
    o2 = torch.optim.AdamW([flat.param], lr=1e-2, weight_decay=0.1, fused=True)
    for step in range(3):
        for m in (m1, m2):
            for p in m.parameters():
                p.grad = None
            m[:4](x * (step + 1)).square().sum().backward()
        # parameters that never receive a gradient: AdamW skips them, the flat update sees a zero gradient -> the
        # decoupled weight decay still applies there; compare the used ones and check the unused ones only decay
        torch.nn.utils.clip_grad_norm_([p for p in m1.parameters() if p.grad is not None], 0.1)
        o1.step()
        flat.pack_grads()
        o2.grad_scale = flat.clip_scale(0.1)[0]
        o2.step()
        for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
            if not n.startswith("unused"):
                torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-7, msg=f"step {step} {n}")
    assert all(p.data_ptr() >= flat.data.data_ptr() for p in m2.parameters())  # still views of the flat buffer


def test_flat_params_filter_biases_wd_and_state_interop():
    """optimizer.py:12-21 (--filter_biases_wd: no decay for biases / 1-D parameters) through the one-tensor optimizer, and
    the flat AdamW state <-> per-parameter state mapping a reference optimizer checkpoint needs."""
    import copy
    from vdetr_amd.dist import FlatParams
    torch.manual_seed(1)
    m1 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    m2 = copy.deepcopy(m1)
    x = torch.randn(9, 6)
    nodecay = [p for n, p in m1.named_parameters() if p.ndim == 1 or n.endswith("bias")]
    decay = [p for n, p in m1.named_parameters() if not (p.ndim == 1 or n.endswith("bias"))]
    lr, wd = 1e-2, 0.1
    o1 = torch.optim.AdamW([{"params": nodecay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}], lr=lr)
    flat = FlatParams(m2.parameters())
    o2 = torch.optim.AdamW([flat.param], lr=lr, weight_decay=0.0, fused=True)
    vec = flat.decay_vector(list(m2.named_parameters()), lr, wd)
    for step in range(3):
        for m in (m1, m2):
            for p in m.parameters():
                p.grad = None
            m(x * (step + 1)).square().sum().backward()
        o1.step()
        flat.pack_grads()
        with torch.no_grad():
            flat.data.mul_(vec)
        o2.step()
        for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-7, msg=f"step {step} {n}")
    # flat state -> per-parameter state, in model.parameters() order
    states = flat.per_param_optimizer_state(o2, list(m2.parameters()))
    for p1, d in zip(m1.parameters(), states):
        torch.testing.assert_close(d["exp_avg"], o1.state[p1]["exp_avg"], rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(d["exp_avg_sq"], o1.state[p1]["exp_avg_sq"], rtol=1e-5, atol=1e-10)
        assert float(d["step"]) == float(o1.state[p1]["step"]) == 3.0
    # ... and back: a fresh one-tensor optimizer resumed from the per-parameter states continues identically
    m3 = copy.deepcopy(m2)
    flat3 = FlatParams(m3.parameters())
    o3 = torch.optim.AdamW([flat3.param], lr=lr, weight_decay=0.0, fused=True)
    flat3.load_per_param_optimizer_state(o3, [{k: o1.state[p][k] for k in ("step", "exp_avg", "exp_avg_sq")} for p in m1.parameters()],
                                         list(m3.parameters()))
    vec3 = flat3.decay_vector(list(m3.named_parameters()), lr, wd)
    for m in (m1, m3):
        for p in m.parameters():
            p.grad = None
        m(x * 0.5).square().sum().backward()
    o1.step()
    flat3.pack_grads()
    with torch.no_grad():
        flat3.data.mul_(vec3)
    o3.step()
    for (n, a), (_, b) in zip(m1.named_parameters(), m3.named_parameters()):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-7, msg=f"resumed {n}")


def _multi_fire_worker(rank, world, port, q):
    """hooks mode with parameters that receive gradient SEVERAL times per step (a shared weight + late accumulation after
    loss.backward(), as the deferred weight-gradient flushes do): compared with pack_and_reduce."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vdetr_amd import runtime
    from vdetr_amd.dist import GradientReducer, broadcast_parameters, init_distributed
    init_distributed("gloo")
    res = {}
    for mode in ("hooks", "pack"):
        torch.manual_seed(7)
        lin_a, lin_b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
        model = torch.nn.ModuleList([lin_a, lin_b])
        broadcast_parameters(model)
        red = GradientReducer(model.parameters(), bucket_mb=0.00005, overlap=mode == "hooks", bucket_views=mode == "hooks")
        runtime.defer_weight_grads(True)   # hooks must not launch buckets before the late gradients arrive
        try:
            torch.manual_seed(rank)
            x = torch.randn(3, 4)
            for step in range(2):
                red.zero_grad()
                h = lin_a(lin_a(x))            # lin_a used twice -> its hook fires once, after both uses accumulated
                y = lin_b(h.detach())
                h.square().sum().backward()
                # "flush": a second autograd pass that adds to lin_b AND again to lin_a
                (y.square().sum() + lin_a(x).sum()).backward()
                if mode == "hooks":
                    red.finish()
                else:
                    red.pack_and_reduce()
            res[mode] = [p.grad.numpy().copy() for p in model.parameters()]
        finally:
            runtime.defer_weight_grads(False)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_hooks_with_deferred_multi_fire_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_multi_fire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for rank, r in res:
        for a, b in zip(r["hooks"], r["pack"]):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(res[0][1]["hooks"], res[1][1]["hooks"]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)   # both ranks hold the same averaged gradient


def _loss_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vdetr_amd.dist import all_reduce_average, init_distributed, reduce_dict
    init_distributed("gloo")
    loss = torch.tensor(float(rank + 1))
    avg = all_reduce_average(loss.clone())
    separate = {"b": torch.tensor(2.0 * (rank + 1)), "a": torch.tensor(10.0 * (rank + 1))}
    table = torch.arange(12, dtype=torch.float32).reshape(3, 4) * (rank + 1)     # the criterion's loss table layout
    views = {"loss_x": table[0, 1], "loss_y_0": table[2, 3], "loss_z": table[1, 0]}
    q.put((rank, float(avg), {k: float(v) for k, v in reduce_dict(separate).items()},
           {k: float(v) for k, v in reduce_dict(views).items()}, {k: float(v) for k, v in reduce_dict(views, average=False).items()}))
    dist.destroy_process_group()


def test_loss_reductions_match_reference_semantics():
    """all_reduce_average / reduce_dict (engine.py:97-98) on two gloo ranks, incl. the one-collective path for a dictionary
    whose values are views of one loss table."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, avg, sep, views, sums in res:
        assert avg == 1.5
        assert sep == {"a": 15.0, "b": 3.0}
        assert views == {"loss_x": 1.5, "loss_y_0": 16.5, "loss_z": 6.0}
        assert sums == {"loss_x": 3.0, "loss_y_0": 33.0, "loss_z": 12.0}


class _ParkedMLP(torch.nn.Module):
    """linear layers through helpers._Linear (weight gradients parked while runtime.defer_weight_grads is on), an
    in_proj-style weight used as three views, and a LayerNorm-free tail with an ordinary autograd gradient"""

    def __init__(self):
        super().__init__()
        self.fcs = torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(6)])
        self.in_proj = torch.nn.Parameter(torch.randn(48, 16) * 0.1)
        self.scale = torch.nn.Parameter(torch.ones(16))

    def forward(self, x):
        from vdetr_amd.helpers import _Linear
        for fc in self.fcs:
            x = torch.relu(_Linear.apply(x, fc.weight, fc.bias))
        q, k, v = self.in_proj.view(3, 16, 16).unbind(0)
        x = _Linear.apply(x, q, None) + _Linear.apply(x, k, None) * 0.5 + _Linear.apply(x, v, None) * 0.25
        return x * self.scale


def _phased_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vdetr_amd import runtime
    from vdetr_amd.dist import FlatParams, GradientReducer, broadcast_parameters, init_distributed
    init_distributed("gloo")
    torch.manual_seed(7)
    model = _ParkedMLP()
    broadcast_parameters(model)
    params = list(model.parameters())
    flat = FlatParams(params)
    red = GradientReducer(params, bucket_mb=0.002, overlap=False, bucket_views=False, flat=flat)  # ~500 floats per bucket
    assert len(red.buckets) >= 3
    torch.manual_seed(rank)
    x = torch.randn(9, 16)
    runtime.defer_weight_grads(True)
    res = {}
    for mode in ("one_shot", "phased"):
        red.zero_grad()
        model(x).square().sum().backward()
        if mode == "one_shot":       # all parked gradients, one pack, then every bucket
            runtime.flush_weight_grads()
            red.pack_and_reduce()
        else:                        # bucket by bucket: parked gradients of bucket k, pack its slice, its all-reduce
            red.reduce_phased()
        res[mode] = [p.grad.detach().numpy().copy() for p in params]
    runtime.defer_weight_grads(False)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_phased_flush_pack_reduce_equals_one_shot_bit_for_bit():
    """reduce_phased (what the captured multi-GPU step runs: per bucket parked weight gradients -> slice pack -> all-reduce
    overlapping the next bucket) delivers exactly the gradients of flush + pack + reduce_all, on both ranks."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_phased_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r in res:
        for a, b in zip(r["one_shot"], r["phased"]):
            assert np.array_equal(a, b), f"rank {rank}"
        assert all(np.abs(a).sum() > 0 for a in r["phased"])
    for a, b in zip(res[0][1]["phased"], res[1][1]["phased"]):
        assert np.array_equal(a, b)   # both ranks hold the same averaged gradient


def test_flat_layout_parts_and_forced_bucket_breaks():
    """FlatParams(first=...) lays one part of the model out in front of the other and GradientReducer(break_before=...) ends
    a bucket there, so that the part whose gradients are final first owns whole buckets; pack_and_launch over all buckets
    fills the flat gradient buffer exactly as pack_grads does (one rank: no collective)."""
    from vdetr_amd.dist import FlatParams, GradientReducer
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 8))       # "backbone"
    b = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 3))       # "decoder": final first
    params = list(a.parameters()) + list(b.parameters())
    flat = FlatParams(params, first=list(b.parameters()))
    nb = sum(p.numel() for p in b.parameters())
    offs_b = sorted(flat.offsets[id(p)] for p in b.parameters())
    offs_a = sorted(flat.offsets[id(p)] for p in a.parameters())
    assert offs_b[-1] < offs_a[0] and offs_a[0] >= nb
    first_a = next(p for p in flat.params if any(p is q for q in a.parameters()))
    red = GradientReducer(params, bucket_mb=1.0, overlap=False, bucket_views=False, flat=flat, break_before=[first_a])
    assert len(red.buckets) == 2
    assert red.buckets_of(list(b.parameters())) == [0] and red.buckets_of(list(a.parameters())) == [1]
    x = torch.randn(4, 8)
    b(a(x)).square().sum().backward()
    flat.pack_grads()
    ref = flat.grad.clone()
    flat.grad.fill_(7.0)
    red.pack_and_launch([0])
    red.pack_and_launch([1])
    red.finish()
    for p in params:  # (the alignment padding between parameters is not part of any span: compare the views)
        o = flat.offsets[id(p)]
        assert torch.equal(flat.grad[o:o + p.numel()], ref[o:o + p.numel()])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE / RANK in the environment starts two ranks itself (as the reference's
    main.py:588-593 spawns its workers), passes rank 0's JSON line through and leaves with the children's exit code.  Run here
    through the launcher's CPU path (--launch-check: process group + one all-reduce over gloo, no GPU work)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["launch_check"] and d["n_gpus"] == 2 and d["sum_of_ranks_plus_1"] == 3.0 and d["reduce_op_avg"] is False
    # a rank that fails makes the launcher fail
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "no-such-backend", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def _agree_worker(rank, world, port, fails, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = dist.distributed_c10d._get_default_store()
    out = []
    for tag, failed_on in fails:   # one agreement per tag, as bench.py makes them ("capture", then "depth")
        if rank == 1:
            import time
            time.sleep(0.2)        # the ranks do not arrive together
        out.append(bench.agree_any_failed(store, world, rank in failed_on, tag, timeout_s=60))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_ranks_agree_on_a_failed_capture():
    """bench.agree_any_failed: after a hipGraph capture, the ranks agree through the process group's store (no device work: a rank
    whose capture failed cannot issue any) whether ANY of them failed — all then take the same branch of the fallback chain
    (restart at the next level / go on).  Two ranks over gloo: nobody failed, rank 1 only, rank 0 only, both; the answers are the
    same on both ranks and tags do not leak into each other."""
    cases = [("capture", ()), ("depth", (1,)), ("third", (0,)), ("fourth", (0, 1)), ("fifth", ())]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, cases, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == [False, True, True, True, False]


def test_bench_child_rendezvous_port_is_a_free_one():
    """bench.free_port: the children of the fallback chain rendezvous on `parent port + 1` only if nothing listens there"""
    import socket
    import bench
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as held:
        held.bind(("127.0.0.1", 0))
        held.listen(1)
        busy = held.getsockname()[1]
        got = bench.free_port(busy)
        assert got != busy and got > 0
    free = _free_port()
    assert bench.free_port(free) == free


def test_bench_sampling_fork_layer_rule():
    """bench.fps_fork_layer: the LATEST layer in front of which the next scene's sampling branch may be forked, from the two measured
    times (bench captures and times that layer and the two in front of it).  Round 6's C2 step (6.4-6.55 ms, sampling 4.22 ms) ->
    layer 3, i.e. layers 3 / 2 / 1 are tried (measured best: layer 1); C4 7.41 / 5.9 -> no room, the start; C5 22.0 / 3.8 -> the held
    CU is a small part of the step, the start."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert bench.fps_fork_layer(6.55, 4.22, 9) == 3 and bench.fps_fork_layer(6.4, 4.22, 9) == 3
    assert bench.fps_fork_layer(6.0, 4.22, 9) == 2           # (a faster step still has its candidates)
    assert bench.fps_fork_layer(7.41, 5.9, 9) == -1
    assert bench.fps_fork_layer(22.0, 3.8, 9) == -1
    assert bench.fps_fork_layer(30.0, 10.0, 3) == 2          # never beyond the last layer
    assert bench.fps_fork_layer(6.9, 4.2, 0) == -1 and bench.fps_fork_layer(0.0, 1.0, 9) == -1
    for t1 in (3.0, 6.9, 12.0, 25.0):                        # the branch always ends before the step does
        for frac in (0.3, 0.5, 0.7, 0.8):
            k = bench.fps_fork_layer(t1, frac * t1, 9)
            assert k == -1 or (0.03 + 0.044 * k) * t1 + 1.12 * frac * t1 <= t1 - 0.5 + 1e-9


def test_avg_reduce_verdict_is_collective_and_cached():
    """dist.avg_reduce_supported: gloo has no ReduceOp.AVG -> False on every rank without a collective; GradientReducer then
    divides and sums (the path every CPU test of this file takes)."""
    from vdetr_amd import dist as D
    assert D.avg_reduce_supported() is False  # no process group in this process


def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vdetr_amd import bn_act as BNA
    from vdetr_amd import sparse_ops as S
    from vdetr_amd.dist import init_distributed
    from vdetr_amd.helpers import GenericMLP
    init_distributed("gloo")
    BNA.set_sync(True)
    g = torch.Generator().manual_seed(5)
    full = torch.randn((7, 6), generator=g, dtype=torch.float64) * 3 + 1     # rows of ALL ranks: rank 0 holds 7, rank 1 holds none
    wout = torch.randn((7, 6), generator=g, dtype=torch.float64)
    mine = slice(0, 7) if rank == 0 else slice(7, 7)
    res = {}
    # (1) the sparse backbone's BatchNorm on a path the fused launch does not take (CPU rows; 6 channels: not a multiple of 4),
    #     one rank with an EMPTY tensor: it must still join the collectives
    bn = torch.nn.BatchNorm1d(6).double().train()
    x = full[mine].clone().requires_grad_(True)
    y = S.bn_act(x, bn, "relu")
    (y * wout[mine]).sum().backward()
    res["sp"] = (y.detach().numpy(), x.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy(), bn.running_mean.numpy().copy(),
                 bn.running_var.numpy().copy(), int(bn.num_batches_tracked))
    # (2) a GenericMLP reached through its nn.Sequential (CPU: no fused launch): [B, C, N] with different N per rank
    torch.manual_seed(11)
    mlp = GenericMLP(input_dim=4, hidden_dims=[5], output_dim=3, norm_fn_name="bn1d", activation="relu", use_conv=True).double().train()
    xin = torch.randn((1, 4, 9), generator=g, dtype=torch.float64)
    cols = slice(0, 6) if rank == 0 else slice(6, 9)
    xm = xin[:, :, cols].clone().requires_grad_(True)
    ym = mlp(xm)
    ym.square().sum().backward()
    bnm = [m for m in mlp.layers if isinstance(m, torch.nn.BatchNorm1d)][0]
    res["mlp"] = (ym.detach().numpy(), xm.grad.numpy(), bnm.weight.grad.numpy(), bnm.running_mean.numpy().copy(), bnm.running_var.numpy().copy())
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batch_norm_fallback_paths_match_one_big_batch():
    """with bn_act.set_sync every BatchNorm call must use cross-replica statistics, not only the fused launches.  Two
    gloo ranks (one holding an EMPTY tensor in the sparse case, 6 + 3 columns in the dense one) reproduce the BatchNorm of the
    concatenated batch: outputs, input gradients, summed parameter gradients, running statistics."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process statement of the same
    g = torch.Generator().manual_seed(5)
    full = (torch.randn((7, 6), generator=g, dtype=torch.float64) * 3 + 1).requires_grad_(True)
    wout = torch.randn((7, 6), generator=g, dtype=torch.float64)
    bn = torch.nn.BatchNorm1d(6).double().train()
    y = torch.relu(bn(full))
    (y * wout).sum().backward()
    sp0, sp1 = res[0]["sp"], res[1]["sp"]
    np.testing.assert_allclose(sp0[0], y.detach().numpy(), rtol=1e-10, atol=1e-12)
    assert sp1[0].shape == (0, 6)
    np.testing.assert_allclose(sp0[1], full.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sp0[2] + sp1[2], bn.weight.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sp0[3] + sp1[3], bn.bias.grad.numpy(), rtol=1e-9, atol=1e-12)
    for r in (sp0, sp1):  # both ranks hold the SAME running statistics: those of the whole batch
        np.testing.assert_allclose(r[4], bn.running_mean.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r[5], bn.running_var.numpy(), rtol=1e-10, atol=1e-12)
        assert r[6] == 1
    from vdetr_amd.helpers import GenericMLP
    torch.manual_seed(11)
    mlp = GenericMLP(input_dim=4, hidden_dims=[5], output_dim=3, norm_fn_name="bn1d", activation="relu", use_conv=True).double().train()
    xin = torch.randn((1, 4, 9), generator=g, dtype=torch.float64).requires_grad_(True)
    ym = mlp(xin)
    ym.square().sum().backward()
    bnm = [m for m in mlp.layers if isinstance(m, torch.nn.BatchNorm1d)][0]
    m0, m1 = res[0]["mlp"], res[1]["mlp"]
    np.testing.assert_allclose(np.concatenate((m0[0], m1[0]), 2), ym.detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.concatenate((m0[1], m1[1]), 2), xin.grad.numpy(), rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(m0[2] + m1[2], bnm.weight.grad.numpy(), rtol=1e-8, atol=1e-11)
    for r in (m0, m1):
        np.testing.assert_allclose(r[3], bnm.running_mean.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r[4], bnm.running_var.numpy(), rtol=1e-10, atol=1e-12)


def _watch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vdetr_amd.dist import FlatParams, GradientReducer, broadcast_parameters, init_distributed
    init_distributed("gloo")
    torch.manual_seed(3)
    bb = torch.nn.Sequential(*[torch.nn.Linear(8, 8) for _ in range(6)])      # "backbone": eager backward, watched buckets
    dec = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 2))   # "decoder": gradients final first
    unused = torch.nn.Linear(8, 8)                                            # never receives a gradient: its bucket is flushed after
    params = list(bb.parameters()) + list(unused.parameters()) + list(dec.parameters())
    broadcast_parameters(torch.nn.ModuleList([bb, dec, unused]))
    flat = FlatParams(params, first=list(dec.parameters()), group_shapes="first")
    # the backbone part lies in REVERSE parameter order behind the decoder part
    offs = [flat.offsets[id(p)] for p in bb.parameters()]
    assert offs == sorted(offs, reverse=True)
    first_bb = next(p for p in flat.params if not any(p is d for d in dec.parameters()))
    red = GradientReducer(params, bucket_mb=0.0004, overlap=False, bucket_views=False, flat=flat, break_before=[first_bb])
    dec_b = red.buckets_of(list(dec.parameters()))
    bb_b = [k for k in range(len(red.buckets)) if k not in dec_b]
    assert len(bb_b) >= 3
    launched_in_backward = []
    orig = red._launch
    red._launch = lambda k: (launched_in_backward.append(k), orig(k))[1]
    red.launch_when_complete(bb_b)
    torch.manual_seed(10 + rank)
    x = torch.randn(5, 8)
    out = []
    for step in range(2):
        for p in params:
            p.grad = None
        launched_in_backward.clear()
        red.begin_watch()
        dec(bb(x)).square().sum().backward()
        during = list(launched_in_backward)
        red.pack_and_launch(dec_b)
        pend = red.pending_watched()
        red.pack_and_launch(pend)
        red.finish()
        out.append((during, pend, flat.grad.numpy().copy()))
    # reference: local gradients, to be averaged by the parent
    for p in params:
        p.grad = None
    dec(bb(x)).square().sum().backward()
    q.put((rank, out, [(flat.offsets[id(p)], (torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy()) for p in params], bb_b))
    dist.barrier()
    dist.destroy_process_group()


def test_watched_buckets_leave_during_the_backward_pass():
    """GradientReducer.launch_when_complete (the backbone's gradient buckets in BackboneTrainer): with the backbone part of the
    flat buffer in reverse parameter order, every watched bucket whose parameters all receive a gradient is packed and its
    all-reduce started from a post-accumulate hook INSIDE backward(), in the order the backward pass completes them; a bucket
    with an unused parameter is left for pending_watched(); the flat gradient equals the average of the ranks' gradients."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_watch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, _, bb_b in res:
        for step, (during, pend, _) in enumerate(out):
            assert set(during) <= set(bb_b), (during, bb_b)
            assert during == sorted(during)                                         # back to front = ascending bucket index
            assert len(pend) >= 1 and not set(pend) & set(during)                   # the bucket of the unused parameter (+ held back)
        # step 0: the bucket of the unused parameter holds back what lies behind it in the fixed (ascending) issue order — every
        # rank issues the same sequence whatever order its autograd ran in; the ranks then agree that it never completes, and
        # from step 1 on the others leave while backward() is still running
        during1, pend1, _ = out[1]
        assert len(during1) >= 2, (during1, bb_b)
        assert len(pend1) <= len(out[0][1])
    for (off, g0), (_, g1) in zip(res[0][2], res[1][2]):
        avg = ((g0 + g1) / 2).reshape(-1)
        for rank, out, _, _ in res:
            for _, _, flatgrad in out:
                np.testing.assert_allclose(flatgrad[off:off + avg.size], avg, rtol=1e-6, atol=1e-7)
