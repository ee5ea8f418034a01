"""PositionEmbeddingCoordsSine on the device (csrc/pos_embed.hip through the module) against the vectors the
reference's own module produced (tests/golden/pos_embed.npz, oracle/make_golden.py) and against the module's host
restatement on larger clouds.  Tolerance 1e-4 (sin/cos of arguments up to 2 pi * |gauss_B| * sqrt(3))."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_close, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _modules():
    from oracle.param_fill import fill_module
    from vdetr_amd.position_embedding import PositionEmbeddingCoordsSine
    four = fill_module(PositionEmbeddingCoordsSine(d_pos=256, pos_type="fourier", normalize=True))
    sine = PositionEmbeddingCoordsSine(pos_type="sine", normalize=True)
    return four, sine


def test_pos_embed_vs_reference_vectors():
    g = load_golden("pos_embed")
    four, sine = _modules()
    four = four.to(DEV)
    xyz, rng = t(g["xyz"]).to(DEV), [t(g["rmin"]).to(DEV), t(g["rmax"]).to(DEV)]
    keep = xyz.clone()
    for name, out in (("fourier", four(xyz, input_range=rng)), ("fourier_64", four(xyz, num_channels=64, input_range=rng)),
                      ("sine_256", sine(xyz, num_channels=256, input_range=rng)), ("sine_100", sine(xyz, num_channels=100, input_range=rng))):
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == g[name].shape
        assert_close(out.cpu(), g[name], 1e-4, 1e-4, name)
    assert torch.equal(xyz, keep)  # the input is not modified


@pytest.mark.parametrize("normalize", [True, False])
def test_pos_embed_device_equals_host_restatement(normalize):
    from vdetr_amd.position_embedding import PositionEmbeddingCoordsSine
    gen = torch.Generator().manual_seed(3)
    B, N = 3, 4096
    xyz = torch.rand((B, N, 3), generator=gen) * torch.tensor([8.0, 6.0, 3.0]) + 1.0
    rng = [xyz.amin(1), xyz.amax(1)]
    four = PositionEmbeddingCoordsSine(d_pos=256, pos_type="fourier", normalize=normalize)
    sine = PositionEmbeddingCoordsSine(pos_type="sine", normalize=normalize, scale=(2 * np.pi if normalize else None))
    kw = dict(input_range=rng) if normalize else {}
    dkw = dict(input_range=[r.to(DEV) for r in rng]) if normalize else {}
    if not normalize:
        xyz = xyz * 0.1  # raw metres times 2 pi times the projection: keep the arguments where fp32 sin is comparable
    assert_close(four.to(DEV)(xyz.to(DEV), **dkw).cpu(), four.cpu()(xyz, **kw).numpy(), 2e-4, 2e-4, "fourier")
    assert_close(sine(xyz.to(DEV), num_channels=288, **dkw).cpu(), sine(xyz, num_channels=288, **kw).numpy(), 2e-4, 2e-4, "sine")
