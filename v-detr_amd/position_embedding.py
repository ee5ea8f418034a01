"""Fixed (non-learned) coordinate embeddings — same constructor, buffer name (``gauss_B``) and forward signature as
the reference's models/position_embedding.py:21-148.  Dead code in the default config (querypos_mlp=True), kept for
the querypos_mlp=False path of ModelVDETR (model_vdetr.py:121-132).

Device tensors go through one HIP launch per call (csrc/pos_embed.hip, ``vdetr_pos_embed_{fourier,sine}_f32``) and
come back as the reference's layout, a contiguous (B, d_pos, N).  Host tensors take the op-by-op restatement below,
which is what the CPU tests pin against the reference's golden vectors."""
import math

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .pc_util import shift_scale_points


def _range_ptrs(xyz, normalize, input_range):
    if not normalize:
        return None, None
    lo, hi = (r.to(device=xyz.device, dtype=torch.float32).contiguous() for r in input_range)
    assert lo.shape == (xyz.shape[0], 3) and hi.shape == (xyz.shape[0], 3)
    return lo, hi


class PositionEmbeddingCoordsSine(nn.Module):
    def __init__(self, temperature=10000, normalize=False, scale=None, pos_type="fourier", d_pos=None, d_in=3,
                 gauss_scale=1.0):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        assert pos_type in ("sine", "fourier")
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.pos_type = pos_type
        if pos_type == "fourier":
            assert d_pos is not None and d_pos % 2 == 0
            # random projection matrix: a checkpointed BUFFER (position_embedding.py:45-48)
            self.register_buffer("gauss_B", torch.empty((d_in, d_pos // 2)).normal_() * gauss_scale)
            self.d_pos = d_pos

    @torch.no_grad()
    def get_sine_embeddings(self, xyz, num_channels, input_range):
        """position_embedding.py:51-96: per-axis sin/cos with geometric frequencies, remainder channels go to the
        first axes in steps of two."""
        if xyz.is_cuda:
            assert xyz.shape[2] == 3 and num_channels % 2 == 0
            L.require_float(xyz, "xyz")
            x = xyz.contiguous()
            lo, hi = _range_ptrs(x, self.normalize, input_range)
            out = x.new_empty((x.shape[0], num_channels, x.shape[1]))
            L.check(L.lib().vdetr_pos_embed_sine_f32(L.ptr(x), x.shape[0], x.shape[1], L.ptr(lo), L.ptr(hi), num_channels,
                                                     float(self.temperature), float(self.scale or 0.0), L.ptr(out),
                                                     L.stream_ptr()), "pos_embed_sine")
            return out
        xyz = xyz.clone()
        if self.normalize:
            xyz = shift_scale_points(xyz, src_range=input_range)
        naxis = xyz.shape[2]
        ndim = num_channels // naxis
        ndim -= ndim % 2
        rems = num_channels - ndim * naxis
        embeds = []
        for d in range(naxis):
            cdim = ndim
            if rems > 0:
                cdim += 2
                rems -= 2
            dim_t = torch.arange(cdim, dtype=torch.float32, device=xyz.device)
            dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / cdim)
            raw = xyz[:, :, d]
            if self.scale:
                raw = raw * self.scale
            pos = raw[:, :, None] / dim_t
            embeds.append(torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2))
        return torch.cat(embeds, dim=2).permute(0, 2, 1)

    @torch.no_grad()
    def get_fourier_embeddings(self, xyz, num_channels=None, input_range=None):
        """position_embedding.py:98-127: sin/cos of 2*pi*xyz_norm @ gauss_B -> (B, d_pos, N)."""
        if num_channels is None:
            num_channels = self.gauss_B.shape[1] * 2
        bsize, npoints = xyz.shape[0], xyz.shape[1]
        assert num_channels > 0 and num_channels % 2 == 0
        d_in, d_out = self.gauss_B.shape[0], num_channels // 2
        assert d_out <= self.gauss_B.shape[1] and d_in == xyz.shape[-1]
        if xyz.is_cuda:
            assert d_in == 3
            L.require_float(xyz, "xyz")
            x, gb = xyz.contiguous(), self.gauss_B.contiguous()
            lo, hi = _range_ptrs(x, self.normalize, input_range)
            out = x.new_empty((bsize, num_channels, npoints))
            L.check(L.lib().vdetr_pos_embed_fourier_f32(L.ptr(x), bsize, npoints, L.ptr(lo), L.ptr(hi), L.ptr(gb), gb.shape[1],
                                                        d_out, L.ptr(out), L.stream_ptr()), "pos_embed_fourier")
            return out
        xyz = xyz.clone()
        if self.normalize:
            xyz = shift_scale_points(xyz, src_range=input_range)
        xyz = xyz * (2 * np.pi)
        proj = torch.mm(xyz.view(-1, d_in), self.gauss_B[:, :d_out]).view(bsize, npoints, d_out)
        return torch.cat((proj.sin(), proj.cos()), dim=2).permute(0, 2, 1)

    def forward(self, xyz, num_channels=None, input_range=None):
        assert isinstance(xyz, torch.Tensor) and xyz.ndim == 3
        if self.pos_type == "sine":
            return self.get_sine_embeddings(xyz, num_channels, input_range)
        return self.get_fourier_embeddings(xyz, num_channels, input_range)

    def extra_repr(self):
        st = f"type={self.pos_type}, scale={self.scale}, normalize={self.normalize}"
        if hasattr(self, "gauss_B"):
            st += f", gaussB={self.gauss_B.shape}, gaussBsum={self.gauss_B.sum().item()}"
        return st
