"""Point-cloud sampling / grouping ops — same Python API as the reference's
``third_party/pointnet2/pointnet2_utils.py`` (names, argument order, autograd behaviour), running on the HIP
kernels of ``lib/libvdetr_hip.so`` instead of the ``pointnet2._ext`` CUDA extension.

API map (reference line → here):
  furthest_point_sample  :77   gather_operation :114   three_nn :146 (returns sqrt distances, :139)
  three_interpolate      :203  grouping_operation :254 ball_query :288 (radius, nsample, xyz, new_xyz)
  QueryAndGroup :291     GroupAll :376
``_ext`` below mirrors the nine raw extension functions of ``_ext_src/src/bindings.cpp:9-22``.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L


class _Ext:
    """The nine functions the reference binds in bindings.cpp:9-22, same names and argument order."""

    @staticmethod
    def _chk_f(t, name):
        L.require_gpu(t, name)
        L.require_contiguous(t, name)
        L.require_float(t, name)

    @staticmethod
    def _chk_i(t, name):
        L.require_gpu(t, name)
        L.require_contiguous(t, name)
        L.require_int(t, name)

    # at::Tensor furthest_point_sampling(at::Tensor points, const int nsamples)   sampling.cpp:67
    def furthest_point_sampling(self, points, nsamples):
        self._chk_f(points, "points")
        b, n, _ = points.shape
        out = torch.zeros((b, nsamples), dtype=torch.int32, device=points.device)
        lib = L.lib()
        nbytes = lib.vdetr_fps_workspace_bytes(b, n)
        ws = L.workspace(nbytes, points.device)
        L.check(lib.vdetr_furthest_point_sampling_f32(L.ptr(points), b, n, nsamples, L.ptr(out), L.ptr(ws), nbytes,
                                                      L.stream_ptr()), "furthest_point_sampling")
        return out

    # at::Tensor gather_points(at::Tensor points, at::Tensor idx)   sampling.cpp:17
    def gather_points(self, points, idx):
        self._chk_f(points, "points")
        self._chk_i(idx, "idx")
        b, c, n = points.shape
        m = idx.shape[1]
        out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
        L.check(L.lib().vdetr_gather_points_f32(L.ptr(points), L.ptr(idx), L.ptr(out), b, c, n, m, L.stream_ptr()),
                "gather_points")
        return out

    # at::Tensor gather_points_grad(at::Tensor grad_out, at::Tensor idx, const int n)   sampling.cpp:44
    def gather_points_grad(self, grad_out, idx, n):
        self._chk_f(grad_out, "grad_out")
        self._chk_i(idx, "idx")
        b, c, m = grad_out.shape
        out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)  # written whole by the op
        L.check(L.lib().vdetr_gather_points_grad_set_f32(L.ptr(grad_out), L.ptr(idx), L.ptr(out), b, c, n, m,
                                                         L.stream_ptr()), "gather_points_grad")
        return out

    # std::vector<at::Tensor> three_nn(at::Tensor unknowns, at::Tensor knows)   interpolate.cpp:17
    def three_nn(self, unknowns, knows):
        self._chk_f(unknowns, "unknowns")
        self._chk_f(knows, "knows")
        b, n, _ = unknowns.shape
        m = knows.shape[1]
        idx = torch.zeros((b, n, 3), dtype=torch.int32, device=unknowns.device)
        dist2 = torch.zeros((b, n, 3), dtype=torch.float32, device=unknowns.device)
        L.check(L.lib().vdetr_three_nn_f32(L.ptr(unknowns), L.ptr(knows), L.ptr(dist2), L.ptr(idx), b, n, m,
                                           L.stream_ptr()), "three_nn")
        return dist2, idx

    # at::Tensor three_interpolate(at::Tensor points, at::Tensor idx, at::Tensor weight)   interpolate.cpp:46
    def three_interpolate(self, points, idx, weight):
        self._chk_f(points, "points")
        self._chk_i(idx, "idx")
        self._chk_f(weight, "weight")
        b, c, m = points.shape
        n = idx.shape[1]
        out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
        L.check(L.lib().vdetr_three_interpolate_f32(L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(out), b, c, m, n,
                                                    L.stream_ptr()), "three_interpolate")
        return out

    # at::Tensor three_interpolate_grad(grad_out, idx, weight, const int m)   interpolate.cpp:75
    def three_interpolate_grad(self, grad_out, idx, weight, m):
        self._chk_f(grad_out, "grad_out")
        self._chk_i(idx, "idx")
        self._chk_f(weight, "weight")
        b, c, n = grad_out.shape
        out = torch.zeros((b, c, m), dtype=torch.float32, device=grad_out.device)
        L.check(L.lib().vdetr_three_interpolate_grad_f32(L.ptr(grad_out), L.ptr(idx), L.ptr(weight), L.ptr(out), b, c,
                                                         n, m, L.stream_ptr()), "three_interpolate_grad")
        return out

    # at::Tensor ball_query(at::Tensor new_xyz, at::Tensor xyz, const float radius, const int nsample)  ball_query.cpp:11
    def ball_query(self, new_xyz, xyz, radius, nsample):
        self._chk_f(new_xyz, "new_xyz")
        self._chk_f(xyz, "xyz")
        b, m, _ = new_xyz.shape
        n = xyz.shape[1]
        idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
        L.check(L.lib().vdetr_ball_query_f32(L.ptr(new_xyz), L.ptr(xyz), L.ptr(idx), b, n, m, float(radius),
                                             int(nsample), L.stream_ptr()), "ball_query")
        return idx

    # at::Tensor group_points(at::Tensor points, at::Tensor idx)   group_points.cpp:14
    def group_points(self, points, idx):
        self._chk_f(points, "points")
        self._chk_i(idx, "idx")
        b, c, n = points.shape
        _, npoints, nsample = idx.shape
        out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
        L.check(L.lib().vdetr_group_points_f32(L.ptr(points), L.ptr(idx), L.ptr(out), b, c, n, npoints, nsample,
                                               L.stream_ptr()), "group_points")
        return out

    # at::Tensor group_points_grad(at::Tensor grad_out, at::Tensor idx, const int n)   group_points.cpp:40
    def group_points_grad(self, grad_out, idx, n):
        self._chk_f(grad_out, "grad_out")
        self._chk_i(idx, "idx")
        b, c, npoints, nsample = grad_out.shape
        out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)  # written whole by the op
        L.check(L.lib().vdetr_group_points_grad_set_f32(L.ptr(grad_out), L.ptr(idx), L.ptr(out), b, c, n, npoints, nsample,
                                                        L.stream_ptr()), "group_points_grad")
        return out


    # ---- variable-length batches of point-major tables (see furthest_point_sample_varlen / gather_rows below) ----
    @staticmethod
    def _ptr_array(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def furthest_point_sampling_varlen(self, xyz_list, npoint):
        for i, x in enumerate(xyz_list):
            self._chk_f(x, f"xyz[{i}]")
            if x.dim() != 2 or x.shape[1] != 3:
                raise RuntimeError(f"xyz[{i}] must be [n,3]")
        B, dev = len(xyz_list), xyz_list[0].device
        counts = (ctypes.c_int32 * B)(*[int(x.shape[0]) for x in xyz_list])
        idx = torch.empty((B, npoint), dtype=torch.int32, device=dev)
        lib = L.lib()
        nbytes = lib.vdetr_fps_varlen_workspace_bytes(counts, B)
        ws = L.workspace(nbytes, dev)
        L.check(lib.vdetr_furthest_point_sampling_varlen_f32(self._ptr_array(xyz_list), counts, B, npoint, L.ptr(idx),
                                                             L.ptr(ws), nbytes, L.stream_ptr()), "furthest_point_sampling_varlen")
        return idx

    def gather_rows(self, rows, idx):
        self._chk_i(idx, "idx")
        B, m = idx.shape
        if len(rows) != B:
            raise RuntimeError(f"gather_rows: {len(rows)} tables for a batch of {B}")
        C = rows[0].shape[1]
        for i, r in enumerate(rows):
            self._chk_f(r, f"rows[{i}]")
            if r.dim() != 2 or r.shape[1] != C:
                raise RuntimeError(f"rows[{i}] must be [n,{C}]")
        out = torch.empty((B, m, C), dtype=torch.float32, device=idx.device)
        L.check(L.lib().vdetr_gather_rows_f32(self._ptr_array(rows), L.ptr(idx), L.ptr(out), B, C, m, L.stream_ptr()),
                "gather_rows")
        return out

    def gather_rows_grad(self, grad_out, idx, counts):
        self._chk_f(grad_out, "grad_out")
        self._chk_i(idx, "idx")
        B, m, C = grad_out.shape
        grads = [torch.zeros((int(n), C), dtype=torch.float32, device=grad_out.device) for n in counts]
        L.check(L.lib().vdetr_gather_rows_grad_f32(L.ptr(grad_out), L.ptr(idx), self._ptr_array(grads), B, C, m,
                                                   L.stream_ptr()), "gather_rows_grad")
        return grads

_ext = _Ext()


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices; non-differentiable (pointnet2_utils.py:48-71)."""
        fps_inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(fps_inds)
        return fps_inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (pointnet2_utils.py:80-111)."""
        _, C, N = features.size()
        ctx.for_backwards = (idx, C, N)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        return _ext.gather_points_grad(grad_out.contiguous(), idx, N), None


gather_operation = GatherOperation.apply


# ------------------------------------------------------------------------------------------------------------------
# Variable-length batches of point-major tables (not in the reference's extension): what ModelVDETR's encoder uses.
# The backbone hands over per scene xyz [n_i,3] and features [n_i,C]; the reference loops over the scenes in Python,
# transposes each feature table to (1,C,n_i) and launches FPS + two gathers per scene (model_vdetr.py:285-316).
# ------------------------------------------------------------------------------------------------------------------
def furthest_point_sample_varlen(xyz_list, npoint):
    """xyz_list: B tensors [n_i,3] (float32, GPU, contiguous) -> int32 [B,npoint], indices relative to each scene.
    One launch for the whole batch; per scene the result is furthest_point_sample(xyz_i[None], npoint)."""
    return _ext.furthest_point_sampling_varlen(list(xyz_list), int(npoint))


class GatherRows(Function):
    """out[i,j,:] = rows_i[idx[i,j], :] for B point-major tables rows_i [n_i,C]; differentiable w.r.t. the tables."""

    @staticmethod
    def forward(ctx, idx, *rows):
        ctx.idx, ctx.shapes = idx, [tuple(r.shape) for r in rows]
        return _ext.gather_rows(list(rows), idx)

    @staticmethod
    def backward(ctx, grad_out):
        need = ctx.needs_input_grad[1:]
        if not any(need):
            return (None,) * (1 + len(ctx.shapes))
        grads = _ext.gather_rows_grad(grad_out.contiguous(), ctx.idx, [sh[0] for sh in ctx.shapes])
        return (None, *[g if n else None for g, n in zip(grads, need)])


def gather_rows(rows, idx):
    """rows: B tensors [n_i,C]; idx int32 [B,m] -> [B,m,C]"""
    return GatherRows.apply(idx, *rows)


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """-> (dist (B,n,3) L2 distances, idx (B,n,3)); sqrt taken here as in pointnet2_utils.py:137-139."""
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n) (pointnet2_utils.py:149-200)."""
        B, c, m = features.size()
        ctx.three_interpolate_for_backward = (idx, weight, m)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample) (pointnet2_utils.py:206-251)."""
        _, C, N = features.size()
        ctx.for_backwards = (idx, N)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        return _ext.group_points_grad(grad_out.contiguous(), idx, N), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> (B,npoint,nsample) int32 (pointnet2_utils.py:257-285)."""
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping (pointnet2_utils.py:291-373), same constructor flags and return convention."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if self.sample_uniformly:
            # host-side resampling of the padded slots, as in the reference (:335-344)
            unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
            for i_batch in range(idx.shape[0]):
                for i_region in range(idx.shape[1]):
                    unique_ind = torch.unique(idx[i_batch, i_region, :])
                    num_unique = unique_ind.shape[0]
                    unique_cnt[i_batch, i_region] = num_unique
                    sample_ind = torch.randint(0, num_unique, (self.nsample - num_unique,), dtype=torch.long,
                                               device=idx.device)
                    idx[i_batch, i_region, :] = torch.cat((unique_ind, unique_ind[sample_ind]))
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B,3,npoint,nsample)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Groups all features (pointnet2_utils.py:376-422)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
