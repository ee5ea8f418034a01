"""Sparse-convolution primitives of the backbone on top of lib/libvdetr_hip.so.

A sparse tensor is a sorted int64 KEY vector [N] (``pack_keys``: batch, x, y, z as biased 16-bit fields, so ascending key
order is lexicographic coordinate order) plus a point-major feature table [N, C] — the layout the hot path's FPS and row
gathers consume.  The four native entry points (kernel map, inverse map, column gather, adjoint gather: csrc/sparse_conv.hip)
are geometry / HBM-stream kernels; the contraction itself is ONE library GEMM per layer over the gathered columns
(``col [N, K*Cin] @ W [K*Cin, Cout]``), its two gradients two more.  No CPU path: every entry point raises on CPU tensors.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib as L

KEY_BIAS = 32768


def pack_keys(coords, check=True):
    """[N,4] integer (batch, x, y, z) -> int64 keys (include/vdetr_hip.h: vdetr_sp_kernel_map_i32).  ``check`` validates the
    16-bit field range (a device->host sync: used on the raw input coordinates only)."""
    c = coords.to(torch.int64)
    if check and c.numel():
        lo, hi = int(c[:, 1:].min()), int(c[:, 1:].max())
        if lo < -KEY_BIAS or hi >= KEY_BIAS or int(c[:, 0].min()) < 0 or int(c[:, 0].max()) >= 32768:
            raise ValueError(f"voxel coordinates outside the 16-bit key range: [{lo}, {hi}]")
    return (c[:, 0] << 48) | ((c[:, 1] + KEY_BIAS) << 32) | ((c[:, 2] + KEY_BIAS) << 16) | (c[:, 3] + KEY_BIAS)


def unpack_keys(keys):
    """int64 keys -> [N,4] int32 (batch, x, y, z)."""
    b = keys >> 48
    x = ((keys >> 32) & 0xFFFF) - KEY_BIAS
    y = ((keys >> 16) & 0xFFFF) - KEY_BIAS
    z = (keys & 0xFFFF) - KEY_BIAS
    return torch.stack((b, x, y, z), dim=1).to(torch.int32)


def kernel_map(in_keys, out_keys, offsets):
    """nbr [K, Nout] int32: row of the (sorted) input site at out_keys[u] + offsets[k], -1 where unoccupied."""
    L.require_gpu(in_keys, "in_keys")
    assert in_keys.dtype == torch.int64 and out_keys.dtype == torch.int64 and offsets.dtype == torch.int32
    K, nout = offsets.shape[0], out_keys.shape[0]
    nbr = torch.empty((K, nout), dtype=torch.int32, device=in_keys.device)
    L.check(L.lib().vdetr_sp_kernel_map_i32(L.ptr(in_keys.contiguous()), in_keys.shape[0], L.ptr(out_keys.contiguous()), nout,
                                            L.ptr(offsets.contiguous()), K, L.ptr(nbr), L.stream_ptr()), "sp_kernel_map")
    return nbr


def inverse_map(nbr, nin):
    """inv [K, Nin] int32: the output row that reads input row i through offset k (unique on a lattice), -1 if none."""
    L.require_gpu(nbr, "nbr")
    K, nout = nbr.shape
    inv = torch.full((K, nin), -1, dtype=torch.int32, device=nbr.device)
    L.check(L.lib().vdetr_sp_inverse_map_i32(L.ptr(nbr), K, nout, nin, L.ptr(inv), L.stream_ptr()), "sp_inverse_map")
    return inv


def gather_cols(feats, nbr):
    """col [Nout, K, C] = feats[nbr[k, u]] (zeros where -1); C % 4 == 0."""
    L.require_gpu(feats, "feats")
    L.require_float(feats, "feats")
    L.require_contiguous(feats, "feats")
    K, nout = nbr.shape
    C = feats.shape[1]
    col = torch.empty((nout, K, C), dtype=torch.float32, device=feats.device)
    L.check(L.lib().vdetr_sp_gather_cols_f32(L.ptr(feats), L.ptr(nbr), K, nout, C, L.ptr(col), L.stream_ptr()), "sp_gather_cols")
    return col


def gather_sum(dcol, inv):
    """din [Nin, C] = sum_k dcol[inv[k, i], k] (the adjoint of ``gather_cols``, as a gather)."""
    L.require_gpu(dcol, "dcol")
    L.require_float(dcol, "dcol")
    L.require_contiguous(dcol, "dcol")
    K, nin = inv.shape
    C = dcol.shape[2]
    din = torch.empty((nin, C), dtype=torch.float32, device=dcol.device)
    L.check(L.lib().vdetr_sp_gather_sum_f32(L.ptr(dcol), L.ptr(inv), K, nin, C, L.ptr(din), L.stream_ptr()), "sp_gather_sum")
    return din


class _SparseConvFn(Function):
    """out [Nout, Cout] = sum_k feats[nbr[k]] @ W[k]   (W [K, Cin, Cout]); gradients w.r.t. feats and W."""

    @staticmethod
    def forward(ctx, feats, weight, nbr, inv):
        K, cin, cout = weight.shape
        col = gather_cols(feats, nbr)                       # [Nout, K, Cin]
        out = col.view(col.shape[0], K * cin) @ weight.reshape(K * cin, cout)
        ctx.save_for_backward(col, weight, inv)
        return out

    @staticmethod
    def backward(ctx, dout):
        col, weight, inv = ctx.saved_tensors
        K, cin, cout = weight.shape
        dout = dout.contiguous()
        dw = dfeats = None
        if ctx.needs_input_grad[1]:
            dw = (col.view(col.shape[0], K * cin).t() @ dout).view(K, cin, cout)
        if ctx.needs_input_grad[0]:
            dcol = (dout @ weight.reshape(K * cin, cout).t()).view(-1, K, cin)
            dfeats = gather_sum(dcol, inv)
        return dfeats, dw, None, None


def sparse_conv(feats, weight, nbr, inv):
    """feats [Nin, Cin] (any Cin: padded to a multiple of 4 here), weight [K, Cin, Cout], nbr [K, Nout], inv [K, Nin]."""
    cin = feats.shape[1]
    pad = (-cin) % 4
    if pad:
        feats = torch.nn.functional.pad(feats, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, pad))
    return _SparseConvFn.apply(feats.contiguous(), weight, nbr, inv)
