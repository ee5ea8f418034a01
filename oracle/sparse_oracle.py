"""TEST INFRASTRUCTURE — CPU restatement of the sparse-convolution backbone primitives (v-detr_amd/csrc/sparse_conv.hip,
v-detr_amd/sparse_ops.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference: the backbone is MinkowskiEngine's generalized sparse convolution (models/mink_resnet.py:38-84,
models/model_vdetr.py:141-176,248-280).  MinkowskiEngine is an un-vendored dependency (README.md:47-53: `git clone` of
NVIDIA/MinkowskiEngine master, no pinned commit) and cannot be built here (CUDA only): PARITY UNPINNED against its binary.
Restated: its published operator (Choy et al., CVPR 2019, eq. 3)  out[u] = sum_{i in N(u) ∩ occupied} W_i in[u + i],
with the conventions listed in v-detr_amd/minkowski.py.  What IS pinned (tests/test_oracle_sparse.py): the arithmetic
against torch's dense conv3d / conv_transpose3d on the densified grid — a sparse convolution restricted to its output
sites equals the dense one with zeros at the unoccupied input sites.
"""
import numpy as np
import torch

KEY_BIAS = 32768


def pack_keys_np(coords):
    c = np.asarray(coords, dtype=np.int64)
    return (c[:, 0] << 48) | ((c[:, 1] + KEY_BIAS) << 32) | ((c[:, 2] + KEY_BIAS) << 16) | (c[:, 3] + KEY_BIAS)


def unpack_keys_np(keys):
    k = np.asarray(keys, dtype=np.int64)
    return np.stack((k >> 48, ((k >> 32) & 0xFFFF) - KEY_BIAS, ((k >> 16) & 0xFFFF) - KEY_BIAS, (k & 0xFFFF) - KEY_BIAS), 1)


def kernel_map(in_keys, out_keys, offsets):
    """nbr [K, Nout] int32 through a dictionary (no ordering assumption on in_keys)."""
    ik, ok, off = (np.asarray(t.cpu() if torch.is_tensor(t) else t) for t in (in_keys, out_keys, offsets))
    index = {int(k): i for i, k in enumerate(ik)}
    oc = unpack_keys_np(ok)
    nbr = np.full((off.shape[0], ok.shape[0]), -1, dtype=np.int32)
    for k in range(off.shape[0]):
        q = oc.copy()
        q[:, 1:] += off[k][None].astype(np.int64)
        inside = ((q[:, 1:] >= -KEY_BIAS) & (q[:, 1:] < KEY_BIAS)).all(1)
        qk = pack_keys_np(q)
        nbr[k] = [index.get(int(v), -1) if ins else -1 for v, ins in zip(qk, inside)]
    return torch.from_numpy(nbr)


def inverse_map(nbr, nin):
    nbr = nbr.cpu().numpy()
    inv = np.full((nbr.shape[0], nin), -1, dtype=np.int32)
    for k in range(nbr.shape[0]):
        u = np.nonzero(nbr[k] >= 0)[0]
        assert len(np.unique(nbr[k][u])) == len(u), "a site is read twice through one offset: not a lattice map"
        inv[k, nbr[k][u]] = u
    return torch.from_numpy(inv)


def gather_cols(feats, nbr):
    """col [Nout, K, C]; differentiable w.r.t. feats (torch index ops)"""
    pad = torch.cat((feats, feats.new_zeros((1, feats.shape[1]))))          # row -1 -> zeros
    idx = torch.where(nbr >= 0, nbr.long(), torch.full_like(nbr, feats.shape[0], dtype=torch.int64))
    return pad[idx].permute(1, 0, 2)


def gather_sum(dcol, inv):
    K, nin = inv.shape
    out = dcol.new_zeros((nin, dcol.shape[2]))
    for k in range(K):
        i = torch.nonzero(inv[k] >= 0)[:, 0]
        out[i] += dcol[inv[k][i].long(), k]
    return out


def sparse_conv(feats, weight, nbr):
    """out [Nout, Cout] = sum_k feats[nbr[k]] @ W[k]  (autograd through torch)."""
    col = gather_cols(feats, nbr)
    return torch.einsum("nkc,kcd->nd", col, weight)


def pairs_gemm(x, arow, weight, seg, transposed):
    """restatement of vdetr_sp_pairs_gemm_f32: y[p] = x[arow[p]] @ W[k(p)] (or its transpose), k(p) from the segment table"""
    K = weight.shape[0]
    kidx = torch.repeat_interleave(torch.arange(K), torch.tensor([seg[k + 1] - seg[k] for k in range(K)]))
    w = weight.transpose(1, 2) if transposed else weight
    return torch.einsum("pc,pcd->pd", x[arow.long()], w[kidx])


def pairs_wgrad(x, dy, pin, pout, seg, K):
    out = []
    for k in range(K):
        a, b = seg[k], seg[k + 1]
        out.append(x[pin[a:b].long()].t() @ dy[pout[a:b].long()])
    return torch.stack(out)


def region_offsets(kernel_size):
    r = range(-(kernel_size // 2), kernel_size // 2 + 1) if kernel_size % 2 else range(kernel_size)
    return np.array([(x, y, z) for z in r for y in r for x in r], dtype=np.int32)


def strided_coords(coords, new_ts):
    c = np.asarray(coords, dtype=np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], new_ts) * new_ts
    return unpack_keys_np(np.unique(pack_keys_np(c)))


def dense_conv_reference(coords, feats, weight, kernel_size, stride, in_ts=1):
    """The same convolution through torch's DENSE conv3d: coords [N,4] (sorted by key), feats [N,Cin], weight [K,Cin,Cout]
    -> (out_coords [M,4], out [M,Cout]) on the sites a strided sparse convolution writes."""
    c = np.asarray(coords, dtype=np.int64)
    assert (c[:, 1:] % in_ts == 0).all()
    B = int(c[:, 0].max()) + 1
    g = c[:, 1:] // in_ts
    lo = g.min(0) - kernel_size
    g0 = g - lo
    shape = g0.max(0) + 1 + 2 * kernel_size
    cin, cout = weight.shape[1], weight.shape[2]
    dense = torch.zeros((B, cin, shape[0], shape[1], shape[2]), dtype=feats.dtype)
    dense[c[:, 0], :, g0[:, 0], g0[:, 1], g0[:, 2]] = feats
    # weight index k = ix + ks*(iy + ks*iz) -> conv3d weight [Cout, Cin, kx, ky, kz] over dims (x, y, z)
    w = weight.reshape(kernel_size, kernel_size, kernel_size, cin, cout).permute(4, 3, 2, 1, 0)  # [out, in, ix, iy, iz]
    full = torch.nn.functional.conv3d(dense, w.contiguous(), padding=kernel_size // 2 if kernel_size % 2 else 0)
    out_coords = strided_coords(c, in_ts * stride) if stride > 1 else c
    og = out_coords[:, 1:] // in_ts - lo
    if kernel_size % 2 == 0:
        raise NotImplementedError
    out = full[out_coords[:, 0], :, og[:, 0], og[:, 1], og[:, 2]]
    return out_coords, out


def dense_transpose_reference(coarse_coords, feats, weight, fine_coords, out_ts):
    """kernel_size 2, stride 2 transposed convolution onto the existing fine sites, through conv_transpose3d."""
    c = np.asarray(coarse_coords, dtype=np.int64)
    f = np.asarray(fine_coords, dtype=np.int64)
    in_ts = out_ts * 2
    B = int(max(c[:, 0].max(), f[:, 0].max())) + 1
    lo = np.minimum(c[:, 1:].min(0), np.floor_divide(f[:, 1:].min(0), in_ts) * in_ts)
    cg = (c[:, 1:] - lo) // in_ts
    shape = np.maximum(cg.max(0), (f[:, 1:] - lo).max(0) // in_ts) + 1
    cin, cout = weight.shape[1], weight.shape[2]
    dense = torch.zeros((B, cin, shape[0], shape[1], shape[2]), dtype=feats.dtype)
    dense[c[:, 0], :, cg[:, 0], cg[:, 1], cg[:, 2]] = feats
    w = weight.reshape(2, 2, 2, cin, cout).permute(3, 4, 2, 1, 0)  # conv_transpose3d weight [in, out, ix, iy, iz]
    full = torch.nn.functional.conv_transpose3d(dense, w.contiguous(), stride=2)
    fg = (f[:, 1:] - lo) // out_ts
    return full[f[:, 0], :, fg[:, 0], fg[:, 1], fg[:, 2]]
