"""Point / box coordinate helpers the decoder needs (subset of the reference's utils/pc_util.py:38-73 and
utils/box_util.py:294-358, without their import-time plyfile/trimesh dependencies)."""
import torch


def shift_scale_points(pred_xyz, src_range, dst_range=None):
    """Affine map of (B,N,3) points from src_range=[min (B,3), max (B,3)] to dst_range (default [0,1]^3)
    (pc_util.py:38-66; same operation order: ((x - smin) * ddiff) / sdiff + dmin)."""
    if dst_range is None:
        dst_range = [torch.zeros_like(src_range[0]), torch.ones_like(src_range[0])]
    if pred_xyz.ndim == 4:
        src_range = [x[:, None] for x in src_range]
        dst_range = [x[:, None] for x in dst_range]
    assert src_range[0].shape[0] == pred_xyz.shape[0] and dst_range[0].shape[0] == pred_xyz.shape[0]
    assert src_range[0].shape[-1] == pred_xyz.shape[-1]
    src_diff = src_range[1][:, None, :] - src_range[0][:, None, :]
    dst_diff = dst_range[1][:, None, :] - dst_range[0][:, None, :]
    return ((pred_xyz - src_range[0][:, None, :]) * dst_diff) / src_diff + dst_range[0][:, None, :]


def scale_points(pred_xyz, mult_factor):
    """pc_util.py:69-73"""
    if pred_xyz.ndim == 4:
        mult_factor = mult_factor[:, None]
    return pred_xyz * mult_factor[:, None, :]


def flip_axis_to_camera_tensor(pc):
    """depth (x right, y forward, z up) -> camera (x, -z, y)   (box_util.py:294-301)"""
    return torch.stack((pc[..., 0], -pc[..., 2], pc[..., 1]), dim=-1)


def roty_batch_tensor(t):
    """Rotation about the camera y axis, [..., 3, 3] (box_util.py:304-316)."""
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack((torch.stack((c, zero, s), -1), torch.stack((zero, one, zero), -1),
                        torch.stack((-s, zero, c), -1)), -2)


# corner sign pattern of get_3d_box_batch_tensor (box_util.py:338-346): x=+-l/2, y=+-h/2, z=+-w/2 (camera frame)
_SX = (1, 1, -1, -1, 1, 1, -1, -1)
_SY = (1, 1, 1, 1, -1, -1, -1, -1)
_SZ = (1, -1, -1, 1, 1, -1, -1, 1)


_sign_cache = {}


def _corner_signs(ref):
    """(3, 8) half-extent sign table on ref's device/dtype, cached: no host->device copy per call (and none
    inside a captured hipGraph)."""
    key = (ref.device, ref.dtype)
    if key not in _sign_cache:
        _sign_cache[key] = torch.tensor([_SX, _SY, _SZ], dtype=ref.dtype, device=ref.device) * 0.5
    return _sign_cache[key]


def get_3d_box_batch_tensor(box_size, angle, center):
    """(.., 3) size (l,w,h), (..) yaw, (.., 3) camera-frame centre -> (.., 8, 3) corners (box_util.py:319-352).
    ``angle=None`` = yaw 0 everywhere (the encoder proposals, model_vdetr.py:360-362 pass a zero tensor): the rotation is the
    identity and the corners are centre + half-extent * sign in ONE launch behind the (l, h, w) gather — the same values as
    the general form gives for a zero tensor (products with 1 and 0 and sums with 0 are exact), 4 launches instead of 18."""
    if angle is None:
        key = (box_size.device, box_size.dtype, "cam")
        if key not in _sign_cache:
            _sign_cache[key] = (torch.tensor([_SX, _SY, _SZ], dtype=box_size.dtype, device=box_size.device) * 0.5).t().contiguous()
        lhw = torch.stack((box_size[..., 0], box_size[..., 2], box_size[..., 1]), dim=-1)
        return torch.addcmul(center.unsqueeze(-2), lhw.unsqueeze(-2), _sign_cache[key])  # [.., 8, 3]
    sx, sy, sz = _corner_signs(box_size)
    l, w, h = box_size[..., 0:1], box_size[..., 1:2], box_size[..., 2:3]
    lx, ly, lz = l * sx, h * sy, w * sz                     # (.., 8) each
    # local @ roty(angle)^T written out (R = [[c, 0, s], [0, 1, 0], [-s, 0, c]]): as a batched 8 x 3 x 3 matmul this was a
    # 48 us library launch per call for 0.3 MFLOP, plus the 8 launches that assemble R
    c, s = torch.cos(angle).unsqueeze(-1), torch.sin(angle).unsqueeze(-1)
    corners = torch.stack((lx * c + lz * s, ly, lz * c - lx * s), dim=-1)
    return corners + center.unsqueeze(-2)


def morton_argsort(xyz):
    """[B,N,3] -> [B,N] permutation that orders the points along a 30-bit Morton (Z-order) curve of their bounding
    box.  Attention is invariant to the order of its keys; the HIP kernels are not indifferent to it: neighbouring
    lanes = neighbouring keys = the same RPE table cell (LDS broadcast in the forward, wave-level aggregation of
    the table gradient in the backward)."""
    if xyz.is_cuda:  # one launch: bounding box, codes and (up to 8192 points) the sort, csrc/morton.hip
        from . import _lib as L
        L.require_float(xyz, "xyz")
        x = xyz.contiguous()
        B, N = x.shape[0], x.shape[1]
        lib = L.lib()
        if N <= lib.vdetr_morton_sort_max():
            order = torch.empty((B, N), dtype=torch.int64, device=x.device)
            L.check(lib.vdetr_morton_order_f32(L.ptr(x), B, N, None, L.ptr(order), L.stream_ptr()), "morton_order")
            return order
        codes = torch.empty((B, N), dtype=torch.int32, device=x.device)
        L.check(lib.vdetr_morton_order_f32(L.ptr(x), B, N, L.ptr(codes), None, L.stream_ptr()), "morton_order")
        return torch.argsort(codes, dim=1, stable=True)
    return torch.argsort(morton_codes(xyz), dim=1, stable=True)


def morton_codes(xyz):
    """[B,N,3] -> [B,N] int64 30-bit Morton codes as tensor expressions (host-side statement of csrc/morton.hip)"""
    lo = xyz.min(dim=1, keepdim=True)[0]
    ext = (xyz.max(dim=1, keepdim=True)[0] - lo).clamp(min=1e-6)
    q = ((xyz - lo) / ext * 1023.0).long().clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    return spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
