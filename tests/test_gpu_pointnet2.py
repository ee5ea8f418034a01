"""HIP pointnet2 ops vs the CPU oracle, through the Python API that mirrors pointnet2_utils.py (which calls the
C-ABI).  Indices: bit-exact.  Floats: bit-exact too where the summation order is fixed (gather, group,
three_interpolate, three_nn distances); atomically accumulated gradients within 1e-5 relative.
"""
import numpy as np
import pytest
import torch

from oracle import pointnet2_oracle as O
from test_oracle_pointnet2 import grid_cloud

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("n,m,seed", [(3000, 128, 0), (6000, 1024, 1), (700, 700, 2), (65, 40, 3), (512, 100, 4),
                                      (1, 1, 5), (63, 63, 6), (20000, 2048, 7)])
def test_fps_grid_clouds_bit_exact(n, m, seed):
    from vdetr_amd import pointnet2_utils as PU
    x = grid_cloud(n, seed)
    ref = O.furthest_point_sampling(x[None], m)
    got = PU.furthest_point_sample(cu(x[None]), m).cpu().numpy()
    assert got.dtype == np.int32
    bad = np.nonzero(ref != got)[1]
    assert bad.size == 0, f"first mismatch at sample {bad[:1]} of {m}: ref {ref[0, bad[:3]]} got {got[0, bad[:3]]}"


def test_fps_batch_random_and_skip_rule():
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(0)
    x = rng.normal(size=(3, 5000, 3)).astype(np.float32) * 2
    x[1, :40] *= 0.004            # a cluster inside the origin-skip ball
    x[2, 0] = 0                   # index 0 itself is skipped but still emitted first
    ref = O.furthest_point_sampling(x, 600)
    got = PU.furthest_point_sample(cu(x), 600).cpu().numpy()
    assert np.array_equal(ref, got)
    z = torch.zeros((1, 100, 3), device=DEV)
    assert PU.furthest_point_sample(z, 5).cpu().tolist() == [[0, 0, 0, 0, 0]]


def test_fps_full_size_scene():
    """BASELINE config 2 size: 40k-point voxelised scene -> 4096 samples, against the oracle."""
    from vdetr_amd import pointnet2_utils as PU
    x = grid_cloud(40000, 11)
    ref = O.furthest_point_sampling(x[None], 4096)
    got = PU.furthest_point_sample(cu(x[None]), 4096).cpu().numpy()
    assert np.array_equal(ref, got)
    assert len(set(got[0].tolist())) == 4096  # size-independent property: samples are distinct


def test_fps_c4_size_scene():
    """BASELINE config 4 size: 80k-point voxelised scene -> 4096 samples, against the oracle (two buckets per owner
    lane in fps_rows.hip)."""
    from vdetr_amd import pointnet2_utils as PU
    x = grid_cloud(80000, 12)
    ref = O.furthest_point_sampling(x[None], 4096)
    got = PU.furthest_point_sample(cu(x[None]), 4096).cpu().numpy()
    bad = np.nonzero(ref != got)[1]
    assert bad.size == 0, f"first mismatch at sample {bad[:1]}: ref {ref[0, bad[:3]]} got {got[0, bad[:3]]}"


@pytest.mark.parametrize("kind", ["line", "plane", "duplicates", "two_clusters", "huge_coords"])
def test_fps_degenerate_clouds_bit_exact(kind):
    """Clouds that defeat the cell grid (everything in a few cells, empty axes, ties everywhere): the buckets are
    still runs of 64 sorted points with exact boxes, so the result must not change."""
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(5)
    n, m = 5000, 300
    if kind == "line":
        x = np.zeros((n, 3), np.float32); x[:, 0] = rng.permutation(n) * 0.01 + 1.0
    elif kind == "plane":
        x = np.stack([rng.integers(0, 60, n) * 0.05 + 1, rng.integers(0, 60, n) * 0.05 + 1, np.full(n, 2.0)], 1).astype(np.float32)
    elif kind == "duplicates":
        x = np.repeat(rng.uniform(1, 3, (50, 3)).astype(np.float32), 100, 0)
        rng.shuffle(x)
    elif kind == "two_clusters":
        x = np.concatenate([rng.normal(1, 1e-3, (n // 2, 3)), rng.normal(500, 1e-3, (n - n // 2, 3))]).astype(np.float32)
    else:  # squared distances beyond the reference's 1e10 initial value
        x = (rng.uniform(-1, 1, (n, 3)) * 3e5).astype(np.float32)
    ref = O.furthest_point_sampling(x[None], m)
    got = PU.furthest_point_sample(cu(x[None]), m).cpu().numpy()
    assert np.array_equal(ref, got), np.nonzero(ref != got)[1][:3]


@pytest.mark.parametrize("n,kind", [(65536, "uniform"), (50000, "uniform"), (60000, "clustered"), (126000, "uniform"), (4097, "uniform")])
def test_fps_bucket_layouts_bit_exact(n, kind):
    """The bucket layouts of fps_rows.hip's prologue (round 3): tree leaves of <= 64 points when they fit the owner lanes'
    slots (4097, 40 k), nodes of <= 128 / 256 points chopped into runs when only those fit (50 k, 65,536 = exactly one full
    slot of runs, a clustered cloud whose leaves are small), plain runs (126 k).  600 samples each against the C oracle."""
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(n)
    if kind == "uniform":
        x = rng.uniform(1, 9, size=(1, n, 3)).astype(np.float32)
    else:  # 200 tight clusters + 10 % background: many nearly-empty leaves next to crowded cells
        c = rng.uniform(1, 9, size=(200, 3))
        x = np.concatenate([c[rng.integers(0, 200, n - n // 10)] + rng.normal(0, 0.02, (n - n // 10, 3)),
                            rng.uniform(1, 9, size=(n // 10, 3))]).astype(np.float32)[None]
    ref = O.furthest_point_sampling(x, 600)
    got = PU.furthest_point_sample(cu(x), 600).cpu().numpy()
    bad = np.nonzero(ref != got)[1]
    assert bad.size == 0, f"first mismatch at sample {bad[:1]}: ref {ref[0, bad[:3]]} got {got[0, bad[:3]]}"


def test_fps_200k_property():
    """Four buckets per owner lane (the largest geometry of fps_rows.hip): the defining FPS property + the oracle on a prefix."""
    from vdetr_amd import pointnet2_utils as PU
    x = np.random.default_rng(4).uniform(1, 9, size=(1, 200000, 3)).astype(np.float32)
    got = PU.furthest_point_sample(cu(x), 256).cpu().numpy()[0]
    assert np.array_equal(got, O.furthest_point_sampling(x, 256)[0])


def test_fps_large_property():
    """n beyond one bucket per lane-slot (bucket size > 64): distinct samples, first index 0, and the min pairwise
    distance of the sample set is non-increasing in sampling order (the defining FPS property)."""
    from vdetr_amd import pointnet2_utils as PU
    x = np.random.default_rng(3).uniform(1, 9, size=(1, 300000, 3)).astype(np.float32)
    got = PU.furthest_point_sample(cu(x), 512).cpu().numpy()[0]
    assert got[0] == 0 and len(set(got.tolist())) == 512
    p = x[0, got].astype(np.float64)
    d = ((p[:, None] - p[None]) ** 2).sum(-1)
    radii = [d[j, :j].min() for j in range(1, 512)]
    assert all(radii[i] >= radii[i + 1] - 1e-9 for i in range(len(radii) - 1))


def test_gather_group_interpolate_and_grads():
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(1)
    b, c, n, m = 2, 37, 1000, 300
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    f = cu(pts).requires_grad_(True)
    out = PU.gather_operation(f, cu(idx))
    assert np.array_equal(out.detach().cpu().numpy(), O.gather_points(pts, idx))
    g = rng.normal(size=(b, c, m)).astype(np.float32)
    out.backward(cu(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.gather_points_grad(g, idx, n), rtol=1e-5, atol=1e-5)

    gidx = rng.integers(0, n, size=(b, 50, 7)).astype(np.int32)
    f = cu(pts).requires_grad_(True)
    out = PU.grouping_operation(f, cu(gidx))
    assert np.array_equal(out.detach().cpu().numpy(), O.group_points(pts, gidx))
    g = rng.normal(size=(b, c, 50, 7)).astype(np.float32)
    out.backward(cu(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.group_points_grad(g, gidx, n), rtol=1e-5, atol=1e-5)

    iidx = rng.integers(0, n, size=(b, 400, 3)).astype(np.int32)
    w = rng.random(size=(b, 400, 3)).astype(np.float32)
    f = cu(pts).requires_grad_(True)
    out = PU.three_interpolate(f, cu(iidx), cu(w))
    assert np.array_equal(out.detach().cpu().numpy(), O.three_interpolate(pts, iidx, w))
    g = rng.normal(size=(b, c, 400)).astype(np.float32)
    out.backward(cu(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), O.three_interpolate_grad(g, iidx, w, n), rtol=1e-5, atol=1e-5)


def test_group_points_row_in_lds():
    """group_points with the channel row staged in LDS (n <= 40,000, b*c >= 32): 16-byte and scalar index walks, one to eight
    slices of the index tensor per row, rows at every offset inside a 16-byte line (odd n), and the strided form beyond."""
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(6)
    for b, c, n, npts, ns in [(1, 128, 39642, 256, 64), (2, 16, 4001, 100, 16), (1, 33, 999, 41, 7), (1, 300, 2048, 64, 8),
                              (1, 40, 45000, 64, 16), (1, 5, 3000, 30, 4)]:
        pts = rng.normal(size=(b, c, n)).astype(np.float32)
        gidx = rng.integers(0, n, size=(b, npts, ns)).astype(np.int32)
        gidx[:, :, ns // 2:] = gidx[:, :, :1]  # padded rows
        got = PU._ext.group_points(cu(pts), cu(gidx)).cpu().numpy()
        assert np.array_equal(got, O.group_points(pts, gidx)), (b, c, n, npts, ns)


def test_scatter_gradients_through_lds():
    """gather_points_grad / group_points_grad as the op the reference's binding is (the result written whole: the output buffer
    starts as NaN here): rows accumulated in LDS (b*c >= 64), one and two ranges per row (n below / above 40,000), pair counts that
    are no multiple of a wave, ball-query rows padded with their first hit, and the memset + atomics pair below 64 rows."""
    from vdetr_amd import _lib as L
    rng = np.random.default_rng(5)
    for b, c, n, m, npts, ns in [(1, 64, 4000, 333, 70, 9), (2, 40, 45001, 1000, 130, 16), (1, 7, 3000, 257, 33, 5),
                                 (1, 128, 39642, 4096, 256, 64)]:
        idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
        idx[:, : m // 4] = idx[:, m // 4: 2 * (m // 4)]  # repeated samples
        g = rng.normal(size=(b, c, m)).astype(np.float32)
        out = torch.full((b, c, n), float("nan"), device=DEV)
        dg, didx = cu(g), cu(idx)
        L.check(L.lib().vdetr_gather_points_grad_set_f32(L.ptr(dg), L.ptr(didx), L.ptr(out), b, c, n, m, L.stream_ptr()), "t")
        np.testing.assert_allclose(out.cpu().numpy(), O.gather_points_grad(g, idx, n), rtol=1e-5, atol=1e-5)
        acc = torch.ones((b, c, n), device=DEV)  # the accumulating entry point keeps its meaning
        L.check(L.lib().vdetr_gather_points_grad_f32(L.ptr(dg), L.ptr(didx), L.ptr(acc), b, c, n, m, L.stream_ptr()), "t")
        np.testing.assert_allclose(acc.cpu().numpy(), 1.0 + O.gather_points_grad(g, idx, n), rtol=1e-5, atol=1e-5)

        gidx = rng.integers(0, n, size=(b, npts, ns)).astype(np.int32)
        hits = rng.integers(1, ns + 1, size=(b, npts))
        for bi in range(b):
            for j in range(npts):
                gidx[bi, j, hits[bi, j]:] = gidx[bi, j, 0]  # what ball_query leaves behind a row's last hit
        gg = rng.normal(size=(b, c, npts, ns)).astype(np.float32)
        out = torch.full((b, c, n), float("nan"), device=DEV)
        dgg, dgidx = cu(gg), cu(gidx)
        L.check(L.lib().vdetr_group_points_grad_set_f32(L.ptr(dgg), L.ptr(dgidx), L.ptr(out), b, c, n, npts, ns,
                                                        L.stream_ptr()), "t")
        np.testing.assert_allclose(out.cpu().numpy(), O.group_points_grad(gg, gidx, n), rtol=1e-5, atol=2e-5)
    # nothing to scatter: zeros, not whatever the buffer held
    out = torch.full((1, 64, 100), float("nan"), device=DEV)
    L.check(L.lib().vdetr_gather_points_grad_set_f32(None, None, L.ptr(out), 1, 64, 100, 0, L.stream_ptr()), "t")
    assert not out.any()


def test_reference_interpolation_test_vector():
    """pointnet2_test.py:15-27 (gradcheck of three_interpolate, atol=rtol=1e-1) on the HIP kernels"""
    from torch.autograd import gradcheck
    from vdetr_amd import pointnet2_utils as PU
    feats = torch.randn(1, 2, 4, device=DEV, requires_grad=True)

    def interpolate_func(inputs):
        idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32, device=DEV)
        weight = torch.tensor([[[1, 1, 1], [2, 2, 2]]], dtype=torch.float32, device=DEV)
        return PU.three_interpolate(inputs, idx, weight)

    assert gradcheck(interpolate_func, feats, atol=1e-1, rtol=1e-1, eps=1e-2)


def test_three_nn_and_ball_query():
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(2)
    known = grid_cloud(1500, 3)[None][:, :1024]
    unk = grid_cloud(3000, 4)[None][:, :2048]
    d2, idx = O.three_nn(unk, known)
    dist, gi = PU.three_nn(cu(unk), cu(known))
    assert np.array_equal(gi.cpu().numpy(), idx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    # m < 3
    dist, gi = PU.three_nn(cu(unk[:, :10]), cu(known[:, :2]))
    assert gi[0, :, 2].cpu().tolist() == [0] * 10 and torch.isinf(dist[0, :, 2]).all()

    xyz = grid_cloud(20000, 5)[None]
    new_xyz = xyz[:, :700].copy()
    new_xyz[0, 5] = [100, 100, 100]  # no neighbour
    for radius, ns in [(0.2, 64), (0.05, 16), (1.0, 3)]:
        ref = O.ball_query(new_xyz, xyz, radius, ns)
        got = PU.ball_query(radius, ns, cu(xyz), cu(new_xyz)).cpu().numpy()
        assert np.array_equal(ref, got), (radius, ns)
    assert got[0, 5].tolist() == [0, 0, 0]


def test_ball_query_tile_edges():
    """The cloud goes through LDS in tiles of 2048 points shared by the eight queries of a workgroup: clouds shorter than a wave,
    one point past a tile, two scenes per launch, a query count that leaves waves of the last workgroup without a query, rows that
    fill up inside the first tile (the workgroup stops early) next to rows that never do."""
    from vdetr_amd import pointnet2_utils as PU
    for n, m, b, radius, ns in [(37, 5, 1, 0.3, 4), (2048, 8, 1, 0.2, 16), (2049, 9, 2, 0.2, 64), (6000, 13, 2, 0.08, 32),
                                (4097, 1, 3, 2.0, 128), (8192, 64, 1, 0.04, 8)]:
        xyz = np.stack([grid_cloud(3 * n, 11 + s)[:n] for s in range(b)])
        assert xyz.shape == (b, n, 3)
        new_xyz = xyz[:, -m:].copy()  # the queries' own points sit at the END of the cloud
        new_xyz[:, 0] += 50.0         # one query per scene with no neighbour at all
        ref = O.ball_query(new_xyz, xyz, radius, ns)
        got = PU.ball_query(radius, ns, cu(xyz), cu(new_xyz)).cpu().numpy()
        assert np.array_equal(ref, got), (n, m, b, radius, ns)
        assert not got[:, 0].any()


def test_query_and_group_module():
    from vdetr_amd import pointnet2_utils as PU
    xyz = cu(grid_cloud(4000, 6)[None])
    new_xyz = xyz[:, :128].contiguous()
    feats = torch.randn(1, 5, xyz.shape[1], device=DEV)
    out = PU.QueryAndGroup(0.3, 8, use_xyz=True)(xyz, new_xyz, feats)
    assert out.shape == (1, 8, 128, 8)
    idx = PU.ball_query(0.3, 8, xyz, new_xyz).long()
    exp = torch.gather(feats[0], 1, idx[0].reshape(1, -1).expand(5, -1)).view(5, 128, 8)
    assert torch.equal(out[0, 3:], exp)
    assert PU.GroupAll()(xyz, None, feats).shape == (1, 8, 1, xyz.shape[1])


def test_fps_varlen_batch_equals_per_scene_sampling():
    """One launch over scenes of different sizes (tie-heavy voxel grids, a 40k scene, a tiny one) == the fixed-size entry
    point on each scene alone == the oracle, bit for bit."""
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(3)
    sizes = [2978, 40000, 513, 64, 7001]
    clouds = [grid_cloud(n, 10 + i) if i != 2 else (rng.normal(size=(n, 3)) * 2).astype(np.float32) for i, n in enumerate(sizes)]
    m = 64
    got = PU.furthest_point_sample_varlen([cu(c) for c in clouds], m).cpu().numpy()
    assert got.shape == (len(clouds), m) and got.dtype == np.int32
    for i, c in enumerate(clouds):
        alone = PU.furthest_point_sample(cu(c[None]), m).cpu().numpy()[0]
        assert np.array_equal(got[i], alone), f"scene {i} (n={c.shape[0]})"
        if c.shape[0] <= 8000:
            assert np.array_equal(got[i], O.furthest_point_sampling(c[None], m)[0]), f"scene {i} vs oracle"
    # the full-size sampling in a mixed batch
    big = PU.furthest_point_sample_varlen([cu(clouds[1]), cu(clouds[4])], 4096).cpu().numpy()
    assert np.array_equal(big[0], PU.furthest_point_sample(cu(clouds[1][None]), 4096).cpu().numpy()[0])
    assert np.array_equal(big[1], PU.furthest_point_sample(cu(clouds[4][None]), 4096).cpu().numpy()[0])


@pytest.mark.parametrize("c", [3, 256, 10])
def test_gather_rows_and_grad(c):
    from vdetr_amd import pointnet2_utils as PU
    g = torch.Generator().manual_seed(c)
    sizes, m = [500, 1300, 64], 200
    rows = [torch.randn((n, c), generator=g).to(DEV).requires_grad_(True) for n in sizes]
    idx = torch.stack([torch.randint(0, n, (m,), generator=g) for n in sizes]).to(torch.int32)
    idx[0, :10] = 7                      # repeated index: the gradient accumulates
    idx = idx.to(DEV)
    out = PU.gather_rows(rows, idx)
    want = torch.stack([r[idx[i].long()] for i, r in enumerate(rows)])
    assert torch.equal(out, want)
    w = torch.randn(out.shape, generator=g).to(DEV)
    (out * w).sum().backward()
    for i, r in enumerate(rows):
        ref = torch.zeros_like(r)
        ref.index_add_(0, idx[i].long(), w[i])
        torch.testing.assert_close(r.grad, ref, rtol=1e-5, atol=1e-6)
    ref_np = O.gather_rows([r.detach().cpu().numpy() for r in rows], idx.cpu().numpy())
    assert np.array_equal(out.detach().cpu().numpy(), ref_np)


@pytest.mark.parametrize("b,c,m,n", [(1, 256, 1024, 2048), (2, 37, 700, 1500), (1, 16, 1024, 1024)])
def test_three_interpolate_through_lds(b, c, m, n):
    """known-point sets whose channel rows fit in LDS take the staged kernel (pointnet2.hip: three_interpolate_lds_kernel): bit-equal to
    the oracle (same contraction order), full and partial channel strips, n not a multiple of the workgroup"""
    from oracle import pointnet2_oracle as O
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(b * 100 + c)
    pts = rng.normal(size=(b, c, m)).astype(np.float32)
    iidx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random(size=(b, n, 3)).astype(np.float32)
    out = PU.three_interpolate(torch.from_numpy(pts).cuda(), torch.from_numpy(iidx).cuda(), torch.from_numpy(w).cuda())
    assert np.array_equal(out.cpu().numpy(), O.three_interpolate(pts, iidx, w))


@pytest.mark.parametrize("b,c,n,m", [(1, 256, 39642, 4096), (2, 70, 2503, 300), (1, 64, 20481, 1024), (3, 33, 1000, 1000)])
def test_gather_points_through_lds(b, c, n, m):
    """rows streamed through LDS in 80 KB chunks (pointnet2.hip: gather_points_lds_kernel): bit-equal to the oracle, rows that are
    not 16-byte aligned (odd n), one / two / three chunks, duplicates in idx"""
    from vdetr_amd import pointnet2_utils as PU
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    idx[:, :3] = np.array([0, n - 1, n - 1])
    out = PU._ext.gather_points(torch.from_numpy(pts).cuda(), torch.from_numpy(idx).cuda())
    assert np.array_equal(out.cpu().numpy(), O.gather_points(pts, idx))
