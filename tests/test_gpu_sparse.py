"""Sparse-convolution backbone on the GPU (HIP index kernels through the C-ABI + library GEMMs) vs the CPU oracle."""
import copy

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _oracle_gather_sum(d, inv, offset_major=False, flat=False):
    if flat:  # rows of a flat [P, C] source: present it as [P, K, C] with the row repeated along K
        return O.gather_sum(d[:, None, :].expand(-1, inv.shape[0], -1), inv)
    return O.gather_sum(d.permute(1, 0, 2) if offset_major else d, inv)


def _cloud(n, seed, batch=2, extent=40, ts=1):
    rng = np.random.default_rng(seed)
    c = np.concatenate((rng.integers(0, batch, (n, 1)), rng.integers(-extent, extent, (n, 3)) * ts), 1)
    return O.unpack_keys_np(np.unique(O.pack_keys_np(c)))


@pytest.mark.parametrize("n,ks,stride,ts", [(500, 3, 1, 1), (3000, 3, 2, 1), (2000, 3, 2, 4), (700, 1, 2, 2), (1, 3, 1, 1)])
def test_kernel_and_inverse_maps_bit_exact(n, ks, stride, ts):
    from vdetr_amd import sparse_ops as S
    coords = _cloud(n, n, ts=ts)
    if n > 1:
        coords[0, 1:] = [-32768 + 0, 0, 32767 - (32767 % ts)]  # sites at the edge of the key range: neighbours fall outside
        coords = O.unpack_keys_np(np.unique(O.pack_keys_np(coords)))
    out_coords = O.strided_coords(coords, ts * stride) if stride > 1 else coords
    ik, ok = torch.from_numpy(O.pack_keys_np(coords)), torch.from_numpy(O.pack_keys_np(out_coords))
    off = torch.from_numpy(O.region_offsets(ks) * ts)
    ref = O.kernel_map(ik, ok, off)
    got = S.kernel_map(ik.to(DEV), ok.to(DEV), off.to(DEV))
    assert torch.equal(got.cpu(), ref)
    assert torch.equal(S.inverse_map(got, ik.shape[0]).cpu(), O.inverse_map(ref, ik.shape[0]))
    assert torch.equal(S.unpack_keys(ik.to(DEV)).cpu(), torch.from_numpy(coords).int())
    assert torch.equal(S.pack_keys(torch.from_numpy(coords).to(DEV)).cpu(), ik)


def test_transposed_map_bit_exact():
    from vdetr_amd import sparse_ops as S
    fine = _cloud(4000, 7, ts=2)
    coarse = O.strided_coords(fine, 4)
    ik, ok = torch.from_numpy(O.pack_keys_np(coarse)), torch.from_numpy(O.pack_keys_np(fine))
    off = torch.from_numpy(-O.region_offsets(2) * 2)
    got = S.kernel_map(ik.to(DEV), ok.to(DEV), off.to(DEV))
    assert torch.equal(got.cpu(), O.kernel_map(ik, ok, off))
    assert bool(((got >= 0).sum(0) == 1).all())


@pytest.mark.parametrize("n,cin,cout,ks", [(800, 3, 16, 3), (2500, 64, 64, 3), (600, 32, 48, 1), (900, 8, 4, 2), (1500, 128, 192, 3)])
def test_sparse_conv_forward_backward(n, cin, cout, ks, monkeypatch):
    from vdetr_amd import sparse_ops as S
    torch.manual_seed(n)
    coords = _cloud(n, n + 1, extent=14)
    keys = torch.from_numpy(O.pack_keys_np(coords))
    off = torch.from_numpy(O.region_offsets(ks))
    nbr = O.kernel_map(keys, keys, off)
    inv = O.inverse_map(nbr, keys.shape[0])
    f = torch.randn(keys.shape[0], cin)
    w = torch.randn(ks ** 3, cin, cout) / np.sqrt(cin * ks ** 3)
    g = torch.randn(keys.shape[0], cout)

    fr, wr = f.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = O.sparse_conv(fr, wr, nbr)
    (ref * g.double()).sum().backward()
    for path in ("pairs", "plan", "sorted", "im2col"):  # fused pair-list kernels (default), batched library GEMMs over
        fd, wd = f.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)  # compacted row lists (two groupings), im2col
        if path == "pairs":
            pp = S.PairPlan(nbr.to(DEV), keys.shape[0])
            assert pp.P == int((nbr >= 0).sum()) and pp.seg[-1] == pp.P
            got = S.sparse_conv(fd, wd, nbr.to(DEV), inv.to(DEV), pp)
        elif path != "im2col":
            plan = S.ConvPlan(nbr.to(DEV), keys.shape[0])
            assert plan.pairs == int((nbr >= 0).sum()) and plan.padded_pairs >= plan.pairs
            if path == "sorted":  # force the count-sorted grouping (normally chosen for wide layers only)
                monkeypatch.setattr(S.ConvPlan, "SORTED_MIN_CHANNELS", 0)
                assert sum(g["n"] * g["M"] for g in plan.grouping("sorted")) <= plan.padded_pairs or ks < 3
            got = S.sparse_conv(fd, wd, nbr.to(DEV), inv.to(DEV), plan)
            monkeypatch.undo()
        else:
            pad = (-cin) % 4
            got = S._SparseConvFn.apply(torch.nn.functional.pad(fd, (0, pad)).contiguous(), torch.nn.functional.pad(wd, (0, 0, 0, pad)),
                                        nbr.to(DEV), inv.to(DEV))
        (got * g.to(DEV)).sum().backward()
        for name, a, b in (("out", got, ref), ("dfeats", fd.grad, fr.grad), ("dweight", wd.grad, wr.grad)):
            scale = float(b.abs().max())
            assert float((a.detach().cpu().double() - b.detach()).abs().max()) <= 1e-4 * scale + 1e-6, (path, name)
    # gather kernels alone: exact copies / fixed-order sums
    col = S.gather_cols(torch.nn.functional.pad(f, (0, (-cin) % 4)).to(DEV).contiguous(), nbr.to(DEV))
    assert torch.equal(col.cpu(), O.gather_cols(torch.nn.functional.pad(f, (0, (-cin) % 4)), nbr))
    dc = torch.randn(col.shape)
    torch.testing.assert_close(S.gather_sum(dc.to(DEV), inv.to(DEV)).cpu(), O.gather_sum(dc, inv), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,c,act,res,training", [(5000, 64, "relu", True, True), (777, 256, "elu", False, True),
                                                  (130, 16, None, True, True), (3000, 512, "relu", False, False), (1, 8, "relu", False, True)])
def test_fused_batchnorm_activation(n, c, act, res, training):
    """csrc/sp_bn.hip against nn.BatchNorm1d + residual + activation in float64: outputs, all gradients, running statistics"""
    from vdetr_amd import sparse_ops as S
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 3 + torch.linspace(-50, 50, c)  # large per-channel means: E[x^2] - mean^2 would cancel
    r = torch.randn(n, c) if res else None
    g = torch.randn(n, c)
    bn_ref = torch.nn.BatchNorm1d(c).double()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-1, 1)
        bn_ref.running_mean.uniform_(-1, 1); bn_ref.running_var.uniform_(0.5, 2)
    bn_gpu = torch.nn.BatchNorm1d(c).to(DEV)
    bn_gpu.load_state_dict({k: v.float() for k, v in bn_ref.state_dict().items()})
    bn_ref.train(training); bn_gpu.train(training)
    fn = {"relu": torch.relu, "elu": torch.nn.functional.elu, None: lambda t: t}[act]

    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    yr = fn(bn_ref(xr) + (rr if res else 0)) if n > 1 or not training else None
    if yr is None:
        return  # nn.BatchNorm1d refuses a single row in training mode; the kernel handles it (variance 0)
    (yr * g.double()).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    rg = r.to(DEV).requires_grad_(True) if res else None
    yg = S.bn_act(xg, bn_gpu, act, rg)
    (yg * g.to(DEV)).sum().backward()
    checks = [("y", yg, yr), ("dx", xg.grad, xr.grad), ("dgamma", bn_gpu.weight.grad, bn_ref.weight.grad),
              ("dbeta", bn_gpu.bias.grad, bn_ref.bias.grad), ("running_mean", bn_gpu.running_mean, bn_ref.running_mean),
              ("running_var", bn_gpu.running_var, bn_ref.running_var)]
    if res:
        checks.append(("dres", rg.grad, rr.grad))
    for name, a, b in checks:
        scale = float(b.detach().abs().max())
        assert float((a.detach().cpu().double() - b.detach()).abs().max()) <= 2e-4 * scale + 1e-5, name
    assert int(bn_gpu.num_batches_tracked) == int(bn_ref.num_batches_tracked)


def test_sparse_ops_refuse_cpu_tensors():
    from vdetr_amd import sparse_ops as S
    with pytest.raises(RuntimeError, match="CPU not supported"):
        S.gather_cols(torch.zeros(4, 4), torch.zeros((1, 4), dtype=torch.int32))


def test_backbone_gpu_equals_cpu_oracle(monkeypatch):
    """MinkResNet18 (narrow) + FPN neck pieces: the GPU run against the same modules with the native entry points routed to
    the oracle on CPU — outputs of every stage, loss gradient of the stem kernel."""
    from vdetr_amd import minkowski as ME
    from vdetr_amd import sparse_ops as S
    from vdetr_amd.mink_resnet import MinkResNet
    torch.manual_seed(0)
    pts = torch.rand(3000, 3) * torch.tensor([2.0, 1.5, 0.8])
    data = [(pts / 0.01, pts), (pts[:1200] / 0.01 + 5, pts[:1200] * 0.5)]
    net = MinkResNet(18, 3, inplanes=16, num_stages=4, stem_bn=True).train()
    up = ME.MinkowskiConvolutionTranspose(128, 64, kernel_size=2, stride=2, dimension=3)
    net_g, up_g = copy.deepcopy(net).to(DEV), copy.deepcopy(up).to(DEV)

    def run(net, up, dev):
        coords, feats = ME.batch_sparse_collate([(c.to(dev), f.to(dev)) for c, f in data])
        outs = net(ME.SparseTensor(feats, coordinates=coords))
        y = outs[2] + up(outs[3])
        loss = sum(o.F.square().mean() for o in outs) + y.F.square().mean()
        loss.backward()
        return outs, y, loss

    outs_g, y_g, loss_g = run(net_g, up_g, DEV)
    with monkeypatch.context() as mp:
        mp.setattr(S, "kernel_map", lambda ik, ok, off: O.kernel_map(ik, ok, off))
        mp.setattr(S, "inverse_map", lambda nbr, nin: O.inverse_map(nbr, nin))
        mp.setattr(S, "gather_cols", lambda f, nbr: O.gather_cols(f, nbr).contiguous())
        mp.setattr(S, "gather_sum", _oracle_gather_sum)
        mp.setattr(S, "pairs_gemm", lambda x, arow, w, plan, tr: O.pairs_gemm(x, arow, w, plan.seg, tr))
        mp.setattr(S, "pairs_wgrad", lambda x, dy, plan, cin, cout: O.pairs_wgrad(x, dy, plan.pin, plan.pout, plan.seg, plan.K))
        outs_c, y_c, loss_c = run(net, up, "cpu")
    for og, oc in zip(outs_g + [y_g], outs_c + [y_c]):
        assert torch.equal(og.keys.cpu(), oc.keys)
        scale = float(oc.F.abs().max())
        assert float((og.F.detach().cpu() - oc.F.detach()).abs().max()) <= 1e-3 * scale
    assert abs(float(loss_g) - float(loss_c)) <= 1e-4 * abs(float(loss_c))
    gg, gc = net_g.conv1.kernel.grad.cpu(), net.conv1.kernel.grad
    assert float((gg - gc).abs().max()) <= 2e-3 * float(gc.abs().max())


def test_full_model_with_sparse_backbone_runs():
    """ModelVDETR.forward on raw point clouds through the sparse ResNet34 + FPN backbone, FPS and the decoder; gradients
    reach the stem kernel."""
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.model_vdetr import build_vdetr, default_args
    torch.manual_seed(0)
    model = build_vdetr(default_args(nqueries=32, dec_nlayers=2, preenc_npoints=256), ScannetDatasetConfig(), "minkowski").to(DEV).train()
    g = torch.Generator().manual_seed(1)
    clouds = []
    for b in range(2):  # points on two planes: a floor and a wall
        n = 6000 - 1500 * b
        u = torch.rand((n, 2), generator=g)
        floor = torch.stack((u[:, 0] * 4, u[:, 1] * 3, torch.zeros(n)), 1)
        wall = torch.stack((u[:, 0] * 4, torch.zeros(n), u[:, 1] * 2.5), 1)
        clouds.append(torch.cat((floor[: n // 2], wall[n // 2:])).to(DEV) + 1.0)
    inputs = {"point_clouds": clouds, "point_cloud_dims_min": torch.stack([c.min(0)[0] for c in clouds]),
              "point_cloud_dims_max": torch.stack([c.max(0)[0] for c in clouds])}
    out = model(inputs)
    assert out["outputs"]["sem_cls_logits"].shape[:2] == (2, 32) and out["seed_xyz"].shape == (2, 256, 3)
    # seed coordinates are sites of the 4 cm lattice inside the scene
    s = out["seed_xyz"] / 0.04
    assert float((s - s.round()).abs().max()) < 1e-3
    loss = out["outputs"]["sem_cls_logits"].sum() + out["outputs"]["center_normalized"].sum()
    loss.backward()
    assert float(model.pre_encoder.conv1.kernel.grad.abs().sum()) > 0
    assert float(model.out_block_0[0].kernel.grad.abs().sum()) > 0
    # geometry built ahead of time (coordinates only) gives the same backbone output and leaves the BN statistics alone
    rm = model.pre_encoder.norm1.bn.running_mean.clone()
    geo = model.prepare_geometry(inputs)
    assert torch.equal(rm, model.pre_encoder.norm1.bn.running_mean) and model.training
    model.eval()
    with torch.no_grad():
        a = model.backbone_forward(inputs)
        b = model.backbone_forward(dict(inputs, geometry=geo))
    for (xa, fa), (xb, fb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(fa, fb)
    # ... and the same gradients: its weight-gradient chunk tables were requested by the geometry-only pass and built by
    # finalize(), the plain pass builds them inside the backward
    model.train()
    grads = []
    for extra in ({}, {"geometry": model.prepare_geometry(inputs)}):
        model.zero_grad(set_to_none=True)
        feats = model.backbone_forward(dict(inputs, **extra))
        sum(f.square().sum() for _, f in feats).backward()
        grads.append([model.pre_encoder.conv1.kernel.grad.clone(), model.pre_encoder.layer3[0].conv1.kernel.grad.clone(),
                      model.out_block_0[0].kernel.grad.clone()])
    for ga, gb in zip(*grads):
        assert torch.equal(ga, gb)
    # a manager built on a loader stream and dropped right after the step: backbone_forward ties its memory to the
    # consuming stream (CoordinateManager.use_on), so the loader's next scene cannot be handed its blocks early
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        geo2 = model.prepare_geometry(inputs)
    torch.cuda.current_stream().wait_stream(side)
    main = torch.cuda.current_stream().cuda_stream
    model.zero_grad(set_to_none=True)
    feats = model.backbone_forward(dict(inputs, geometry=geo2))
    held = geo2.device_tensors()
    assert len(held) > 20 and all((t.untyped_storage().data_ptr(), main) in geo2._recorded for t in held)
    del geo2
    with torch.cuda.stream(side):  # the loader's next scene allocates on its own stream while the step is still queued
        geo3 = model.prepare_geometry(inputs)
    sum(f.square().sum() for _, f in feats).backward()
    torch.cuda.synchronize()
    for ga, gb in zip(grads[0], [model.pre_encoder.conv1.kernel.grad, model.pre_encoder.layer3[0].conv1.kernel.grad,
                                 model.out_block_0[0].kernel.grad]):
        assert torch.equal(ga, gb)
    del geo3


@pytest.mark.parametrize("n,ks,seed", [(3000, 3, 0), (700, 2, 1), (5, 3, 2), (20000, 3, 3)])
def test_pair_plan_device_builder_equals_host_statement(n, ks, seed):
    """vdetr_sp_pair_plan_i32 (three launches, no host round trip) against PairPlan._build_host on the CPU copy of the same map:
    identical pair lists, slots and tables (integer work: bit-exact)."""
    from vdetr_amd import sparse_ops as S
    rng = np.random.default_rng(seed)
    coords = np.unique(rng.integers(0, 24, size=(n, 3)), axis=0).astype(np.int32)
    coords = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords], 1)
    keys = torch.sort(S.pack_keys(torch.from_numpy(coords)))[0].cuda()
    offs = torch.tensor(O.region_offsets(ks), dtype=torch.int32).cuda()
    sub = keys[::2].contiguous()  # different input / output site sets
    for ink, outk in ((keys, keys), (keys, sub), (sub, keys)):
        nbr = S.kernel_map(ink, outk, offs)
        dev = S.PairPlan(nbr, ink.shape[0])
        assert dev._pending is not None            # nothing has waited for the device yet
        host = S.PairPlan(nbr.cpu(), ink.shape[0])
        assert dev.P == host.P and dev.counts == host.counts and dev.seg == host.seg and dev.ntiles == host.ntiles
        for name in ("pin", "pout", "slot", "islot", "tiles"):
            assert torch.equal(getattr(dev, name).cpu(), getattr(host, name)), name
        c_dev, c_host = dev.wgrad_chunks(64, 64), host.wgrad_chunks(64, 64)
        assert torch.equal(c_dev[0].cpu(), c_host[0]) and torch.equal(c_dev[1].cpu(), c_host[1]) and c_dev[2] == c_host[2]


def test_persistent_conv_kernel_work_counters_survive_reuse():
    """The persistent forward / input-gradient kernel draws its work items from a (work, exit) counter pair out of a ring of 4096
    that the launch's last workgroup resets.  More launches than the ring has slots, on two streams whose launches overlap:
    every result must equal the first one bit for bit (a counter left non-zero would silently drop work items)."""
    from vdetr_amd import sparse_ops as S
    rng = np.random.default_rng(0)
    coords = np.unique(rng.integers(0, 14, size=(1500, 3)), axis=0).astype(np.int32)
    coords = np.concatenate([np.zeros((coords.shape[0], 1), np.int32), coords], 1)
    keys = torch.sort(S.pack_keys(torch.from_numpy(coords)))[0].cuda()
    offs = torch.tensor(O.region_offsets(3), dtype=torch.int32).cuda()
    nbr = S.kernel_map(keys, keys, offs)
    plan = S.PairPlan(nbr, keys.shape[0]).finalize()
    assert plan.ntiles > 27
    g = torch.Generator().manual_seed(1)
    x = torch.randn((keys.shape[0], 128), generator=g).cuda()
    w = (torch.randn((27, 128, 128), generator=g) / 30).cuda()
    ref_f = S.pairs_gemm(x, plan.pin, w, plan, False)
    ref_d = S.pairs_gemm(x, plan.pout, w, plan, True)
    want = S.gather_sum(ref_f, plan.slot, flat=True)
    oracle = O.sparse_conv(x.cpu(), w.cpu(), nbr.cpu())
    assert float((want.cpu() - oracle).abs().max()) <= 2e-5 * float(oracle.abs().max())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    bad = 0
    for i in range(2300):  # 2 x 2300 launches > 4096 ring slots
        with torch.cuda.stream(streams[i & 1]):
            yf = S.pairs_gemm(x, plan.pin, w, plan, False)
            yd = S.pairs_gemm(x, plan.pout, w, plan, True)
            if i % 97 == 0 or i > 2290:
                bad += int(not torch.equal(yf, ref_f)) + int(not torch.equal(yd, ref_d))
    torch.cuda.synchronize()
    assert bad == 0


def test_sparse_conv_full_size_linearity_and_adjoints():
    """BASELINE-size geometry (40k-point room scan, stride-4 sites, 3x3x3 kernel, 64 -> 128 channels), where the CPU oracle
    would take minutes: size-independent properties instead.  Linearity of the forward map; the input gradient is its adjoint
    (<conv(x), y> = <x, dgrad(y)>); the weight gradient is the adjoint in W (<conv_W(x), y> = <W, dW>)."""
    import bench
    from vdetr_amd import minkowski as ME
    from vdetr_amd import sparse_ops as S
    dev = torch.device("cuda")
    cloud = bench.make_room_cloud(40000, 0, dev)
    coords, _ = ME.batch_sparse_collate([(cloud / 0.01, cloud)])
    cm = ME.CoordinateManager(dev)
    cm.insert_points(coords)
    k2 = cm.strided(cm.keys[1], 1, 2)
    k4 = cm.strided(k2, 2, 4)
    nbr, inv, plan = cm.kernel_map(k4, k4, 4, 4, 3, False)
    n = k4.shape[0]
    assert n > 30000 and plan.P > 3 * n
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, 64), generator=g).to(dev).requires_grad_(True)
    z = torch.randn((n, 64), generator=g).to(dev)
    w = (torch.randn((27, 64, 128), generator=g) / 20).to(dev).requires_grad_(True)
    y = torch.randn((n, 128), generator=g).to(dev)
    out = S.sparse_conv(x, w, nbr, inv, plan)
    lin = S.sparse_conv((2.0 * x + 3.0 * z).detach(), w.detach(), nbr, inv, plan)
    ref = 2.0 * out.detach() + 3.0 * S.sparse_conv(z, w.detach(), nbr, inv, plan)
    assert float((lin - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    inner = (out.detach().double() * y.double()).sum()
    out.backward(y)
    via_x = (x.detach().double() * x.grad.double()).sum()
    via_w = (w.detach().double() * w.grad.double()).sum()
    assert abs(float(via_x - inner)) <= 1e-5 * abs(float(inner)) + 1e-3
    assert abs(float(via_w - inner)) <= 1e-5 * abs(float(inner)) + 1e-3
    # every output row that has no neighbour at all is exactly zero, and the centre offset alone reproduces x W[13]
    centre_only = S.sparse_conv(x.detach(), w.detach() * (torch.arange(27, device=dev) == 13).float()[:, None, None], nbr, inv, plan)
    assert float((centre_only - x.detach() @ w.detach()[13]).abs().max()) <= 1e-5 * float(centre_only.abs().max())


def test_fused_batchnorm_full_size_properties():
    """36k rows x 128 channels (a stride-4 feature table of a 40k-point scene): per-channel mean / variance of the normalised
    output, running statistics, and the two orthogonality relations of BatchNorm's input gradient (sum_r dx = 0,
    sum_r dx * xhat = 0 per channel), with a badly centred input (mean 50 x std: E[x^2] - mean^2 would cancel)."""
    from vdetr_amd import sparse_ops as S
    g = torch.Generator().manual_seed(2)
    N, C = 36363, 128
    x = (torch.randn((N, C), generator=g) * 0.7 + 35.0).cuda().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
    y = S.bn_act(x, bn)
    yd = y.detach().double()
    assert float((yd.mean(0) - bn.bias.detach().double()).abs().max()) < 2e-5
    assert float((yd.var(0, unbiased=False).sqrt() - bn.weight.detach().double().abs()).abs().max()) < 2e-4
    xd = x.detach().double()
    assert float((bn.running_mean.double() - 0.1 * xd.mean(0)).abs().max()) < 1e-4
    assert float((bn.running_var.double() - (0.9 + 0.1 * xd.var(0, unbiased=True))).abs().max()) < 1e-4
    dy = torch.randn((N, C), generator=g).cuda()
    y.backward(dy)
    dx = x.grad.double()
    xhat = (xd - xd.mean(0)) / xd.var(0, unbiased=False).add(bn.eps).sqrt()
    scale = float(dx.abs().sum(0).max())
    assert float(dx.sum(0).abs().max()) < 1e-5 * scale
    assert float((dx * xhat).sum(0).abs().max()) < 1e-5 * scale
    assert float((bn.bias.grad.double() - dy.double().sum(0)).abs().max()) < 1e-4 * float(dy.double().sum(0).abs().max() + 1)


def test_kernel_map_full_size_symmetry():
    """same-stride 3x3x3 map of a 40k-point scene's stride-4 sites: offset k and its opposite 26 - k are inverse relations
    (u reads i through k  <=>  i reads u through 26 - k), the centre offset is the identity, and the device-built pair lists
    agree with the map entry by entry (integer work: exact)."""
    import bench
    from vdetr_amd import minkowski as ME
    dev = torch.device("cuda")
    cloud = bench.make_room_cloud(40000, 1, dev)
    coords, _ = ME.batch_sparse_collate([(cloud / 0.01, cloud)])
    cm = ME.CoordinateManager(dev)
    cm.insert_points(coords)
    k4 = cm.strided(cm.strided(cm.keys[1], 1, 2), 2, 4)
    nbr, _, plan = cm.kernel_map(k4, k4, 4, 4, 3, False)
    n = k4.shape[0]
    ar = torch.arange(n, device=dev, dtype=torch.int32)
    assert torch.equal(nbr[13], ar)
    for k in (0, 5, 12):
        u = torch.nonzero(nbr[k] >= 0)[:, 0]
        i = nbr[k][u].long()
        assert torch.equal(nbr[26 - k][i].long(), u)
        assert int((nbr[k] >= 0).sum()) == int((nbr[26 - k] >= 0).sum())
    plan.finalize()
    assert plan.P == int((nbr >= 0).sum()) and plan.counts[13] == n
    p = torch.arange(plan.P, device=dev)
    kidx = torch.repeat_interleave(torch.arange(27, device=dev), torch.tensor(plan.counts, device=dev))
    assert torch.equal(nbr[kidx, plan.pout.long()], plan.pin)                     # pair p is (pin, pout) of offset k(p)
    assert torch.equal(plan.slot[kidx, plan.pout.long()].long(), p)              # slot finds the pair back from its output row
    assert torch.equal(plan.islot[kidx, plan.pin.long()].long(), p)              # islot from its input row
    assert int((plan.slot >= 0).sum()) == plan.P and int((plan.islot >= 0).sum()) == plan.P
