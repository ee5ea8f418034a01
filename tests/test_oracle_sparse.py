"""Sparse-convolution oracle (oracle/sparse_oracle.py) pinned against torch's DENSE convolutions, and the host modules of
v-detr_amd/minkowski.py / mink_resnet.py on CPU with the four native entry points routed to the oracle (test fixture only)."""
import numpy as np
import pytest
import torch

from oracle import sparse_oracle as O


def _oracle_gather_sum(d, inv, offset_major=False, flat=False):
    if flat:  # rows of a flat [P, C] source: present it as [P, K, C] with the row repeated along K
        return O.gather_sum(d[:, None, :].expand(-1, inv.shape[0], -1), inv)
    return O.gather_sum(d.permute(1, 0, 2) if offset_major else d, inv)


def _cloud(n, seed, batch=2, extent=12, ts=1):
    rng = np.random.default_rng(seed)
    c = np.concatenate((rng.integers(0, batch, (n, 1)), rng.integers(-extent, extent, (n, 3)) * ts), 1)
    return O.unpack_keys_np(np.unique(O.pack_keys_np(c)))


@pytest.mark.parametrize("stride", [1, 2])
def test_sparse_conv_equals_dense_conv3d(stride):
    torch.manual_seed(0)
    coords = _cloud(300, 1)
    feats = torch.randn(coords.shape[0], 5, dtype=torch.float64)
    w = torch.randn(27, 5, 7, dtype=torch.float64)
    out_coords, ref = O.dense_conv_reference(coords, feats, w, 3, stride)
    nbr = O.kernel_map(O.pack_keys_np(coords), O.pack_keys_np(out_coords), O.region_offsets(3))
    got = O.sparse_conv(feats, w, nbr)
    assert got.shape == ref.shape and (nbr >= 0).sum() > 0
    torch.testing.assert_close(got, ref, rtol=1e-10, atol=1e-10)


def test_strided_conv_on_coarse_lattice_equals_dense():
    """tensor stride 2 input: offsets are +-2, outputs on multiples of 4"""
    torch.manual_seed(1)
    coords = _cloud(200, 2, ts=2)
    feats = torch.randn(coords.shape[0], 4, dtype=torch.float64)
    w = torch.randn(27, 4, 6, dtype=torch.float64)
    out_coords, ref = O.dense_conv_reference(coords, feats, w, 3, 2, in_ts=2)
    nbr = O.kernel_map(O.pack_keys_np(coords), O.pack_keys_np(out_coords), O.region_offsets(3) * 2)
    torch.testing.assert_close(O.sparse_conv(feats, w, nbr), ref, rtol=1e-10, atol=1e-10)
    assert (out_coords[:, 1:] % 4 == 0).all()


def test_transposed_conv_equals_dense_conv_transpose3d():
    torch.manual_seed(2)
    fine = _cloud(250, 3, ts=2)
    coarse = O.strided_coords(fine, 4)
    feats = torch.randn(coarse.shape[0], 6, dtype=torch.float64)
    w = torch.randn(8, 6, 3, dtype=torch.float64)
    ref = O.dense_transpose_reference(coarse, feats, w, fine, 2)
    nbr = O.kernel_map(O.pack_keys_np(coarse), O.pack_keys_np(fine), -O.region_offsets(2) * 2)
    assert ((nbr >= 0).sum(0) == 1).all()  # every fine site has exactly one parent
    torch.testing.assert_close(O.sparse_conv(feats, w, nbr), ref, rtol=1e-10, atol=1e-10)


def test_inverse_map_and_adjoint():
    torch.manual_seed(3)
    coords = _cloud(150, 4)
    keys = O.pack_keys_np(coords)
    nbr = O.kernel_map(keys, keys, O.region_offsets(3))
    inv = O.inverse_map(nbr, len(keys))
    # stride-1 maps are symmetric: u reads i through k  <=>  i reads u through 26 - k
    assert torch.equal(inv, nbr.flip(0))
    feats = torch.randn(len(keys), 8, dtype=torch.float64, requires_grad=True)
    col = O.gather_cols(feats, nbr)
    g = torch.randn_like(col)
    (col * g).sum().backward()
    torch.testing.assert_close(O.gather_sum(g, inv), feats.grad)


@pytest.fixture
def cpu_sparse_ops(monkeypatch):
    from vdetr_amd import sparse_ops as S
    monkeypatch.setattr(S, "kernel_map", lambda ik, ok, off: O.kernel_map(ik, ok, off))
    monkeypatch.setattr(S, "inverse_map", lambda nbr, nin: O.inverse_map(nbr, nin))
    monkeypatch.setattr(S, "gather_cols", lambda f, nbr: O.gather_cols(f, nbr).contiguous())
    monkeypatch.setattr(S, "gather_sum", _oracle_gather_sum)
    monkeypatch.setattr(S, "pairs_gemm", lambda x, arow, w, plan, tr: O.pairs_gemm(x, arow, w, plan.seg, tr))
    monkeypatch.setattr(S, "pairs_wgrad", lambda x, dy, plan, cin, cout: O.pairs_wgrad(x, dy, plan.pin, plan.pout, plan.seg, plan.K))
    return S


def test_minkowski_modules_on_cpu(cpu_sparse_ops):
    """module wiring: strides, coordinate maps, skip connections, state-dict layout of the reference's backbone"""
    from vdetr_amd import minkowski as ME
    from vdetr_amd.mink_resnet import MinkResNet
    torch.manual_seed(0)
    pts = torch.rand(400, 3) * torch.tensor([1.2, 0.9, 0.5])
    coords, feats = ME.batch_sparse_collate([(pts / 0.01, pts), (pts[:150] / 0.01 + 3, pts[:150])])
    x = ME.SparseTensor(feats, coordinates=coords)
    assert x.F.shape[0] == x.keys.shape[0] <= 550 and torch.equal(x.keys, torch.sort(x.keys)[0])
    net = MinkResNet(18, 3, inplanes=8, num_stages=4, stem_bn=True)
    outs = net(x)
    assert [o.tensor_stride for o in outs] == [4, 8, 16, 32]
    assert [o.F.shape[1] for o in outs] == [8, 16, 32, 64]
    for o in outs:
        assert (o.C[:, 1:] % o.tensor_stride == 0).all() and o.F.shape[0] == o.keys.shape[0]
    up = ME.MinkowskiConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
    y = outs[2] + up(outs[3])              # lands on the encoder's sites of stride 16
    assert y.tensor_stride == 16 and y.F.shape == outs[2].F.shape
    gen = ME.MinkowskiGenerativeConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
    z = gen(outs[3])
    assert z.F.shape[0] == 8 * outs[3].F.shape[0] and z.tensor_stride == 16
    y.F.sum().backward()
    assert net.conv1.kernel.grad is not None and float(net.conv1.kernel.grad.abs().sum()) > 0
    keys = set(net.state_dict())
    assert {"conv1.kernel", "norm1.bn.running_mean", "layer1.0.downsample.0.kernel", "layer1.0.downsample.1.bn.weight",
            "layer4.1.conv2.kernel"} <= keys
    assert net.layer1[0].downsample[0].kernel.shape == (1, 8, 8) and net.conv1.kernel.shape == (27, 3, 8)


def test_model_with_sparse_backbone_state_dict_layout():
    """the reference's checkpoint keys for the backbone + neck (pre_encoder.*, up_block_{1,2,3}.*, out_block_0.*)"""
    from vdetr_amd.dataset_config import ScannetDatasetConfig
    from vdetr_amd.model_vdetr import build_vdetr, default_args
    m = build_vdetr(default_args(nqueries=16, dec_nlayers=2), ScannetDatasetConfig(), "minkowski")
    sd = m.state_dict()
    for k, shape in {"pre_encoder.conv1.kernel": (27, 3, 64), "pre_encoder.layer4.2.conv2.kernel": (27, 512, 512),
                     "pre_encoder.layer2.0.downsample.0.kernel": (1, 64, 128), "up_block_3.0.kernel": (8, 512, 256),
                     "up_block_1.3.kernel": (27, 64, 64), "out_block_0.0.kernel": (27, 64, 256),
                     "out_block_0.1.bn.running_var": (256,)}.items():
        assert tuple(sd[k].shape) == shape, k
    n = sum(p.numel() for name, p in m.named_parameters() if not name.startswith(("decoder", "encoder_to")))
    assert 67_000_000 < n < 68_500_000  # ResNet34 + FPN: the ~67 M parameters of SURVEY.md §2d


def test_pair_plan_tables_cover_every_pair_once():
    """host statement of the pair lists (PairPlan on CPU tensors): tiles of <= 128 pairs and weight-gradient chunks of equal
    length partition every offset's segment exactly; the chunk partials of an offset are a contiguous range"""
    import numpy as np
    from vdetr_amd import sparse_ops as S
    rng = np.random.default_rng(3)
    K, nout, nin = 27, 700, 650
    nbr = torch.from_numpy(np.where(rng.random((K, nout)) < 0.3, rng.integers(0, nin, (K, nout)), -1).astype(np.int32))
    nbr[5] = -1                      # an offset without pairs
    nbr[13] = torch.arange(nout, dtype=torch.int32) % nin  # a full one (the centre offset of a same-stride map)
    plan = S.PairPlan(nbr, nin)
    counts = (nbr >= 0).sum(1).tolist()
    assert plan.counts == counts and plan.P == sum(counts) and plan.seg[-1] == plan.P
    tiles = plan.tiles.numpy()
    assert plan.ntiles == sum(-(-c // 128) for c in counts)
    covered = np.zeros(plan.P, np.int32)
    for k, start, n in tiles[:plan.ntiles]:
        assert 0 < n <= 128 and plan.seg[k] <= start and start + n <= plan.seg[k + 1]
        covered[start:start + n] += 1
    assert (covered == 1).all()
    for cin, cout in ((64, 64), (256, 256), (128, 512)):
        chunks, cseg, n = plan.wgrad_chunks(cin, cout)
        chunks, cseg = chunks.numpy(), cseg.numpy()
        assert cseg[0] == 0 and cseg[-1] == n and (np.diff(cseg) >= 0).all() and cseg[6] - cseg[5] == 0
        covered[:] = 0
        L = chunks[:n, 2].max()
        for k, start, ln, slot in chunks[:n]:
            assert cseg[k] <= slot < cseg[k + 1] and 0 < ln <= L
            covered[start:start + ln] += 1
        assert (covered == 1).all()
        # equal length: only the last chunk of an offset may be shorter
        for k in range(K):
            lens = chunks[cseg[k]:cseg[k + 1], 2]
            assert (lens[:-1] == L).all() if len(lens) > 1 else True


def test_morton_codes_order_points_along_the_curve():
    """pc_util.morton_codes (the tensor-expression statement of csrc/morton.hip): 30-bit codes, monotone along each axis alone,
    and the stable argsort keeps equal codes in index order"""
    from vdetr_amd import pc_util
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(2, 500, 3, generator=g) * torch.tensor([8.0, 6.0, 3.0])
    codes = pc_util.morton_codes(xyz)
    assert codes.dtype == torch.int64 and int(codes.min()) >= 0 and int(codes.max()) < (1 << 30)
    line = torch.zeros(1, 64, 3)
    line[0, :, 1] = torch.linspace(0, 1, 64)
    c = pc_util.morton_codes(line)[0]
    assert (c[1:] >= c[:-1]).all()
    dup = xyz[:1].clone()
    dup[0, 100] = dup[0, 7]
    order = pc_util.morton_argsort(dup)[0].tolist()
    assert order.index(7) < order.index(100)
