"""numpy front-end of ``pointnet2_oracle.c`` (CPU restatement of third_party/pointnet2/_ext_src/src/*.cu).

TEST INFRASTRUCTURE ONLY — see the header of the C file.  PARITY UNPINNED against the CUDA binary (it cannot be
built or run here); pinned by hand-computed known-answer tests in tests/test_oracle_pointnet2.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_pointnet2.so")
_lib = None


def build(force=False):
    """(Re)build the shared object with gcc when missing, stale, or built for another CPU's flags."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "liboracle_pointnet2.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_fps_bucketed.restype = ctypes.c_long
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def furthest_point_sampling(xyz, m, variant="literal"):
    """xyz (b,n,3) f32 -> (b,m) i32.  variant: 'literal' (thread/tree simulation) or 'keyed' (closed form)."""
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    idx = np.zeros((b, max(m, 0)), np.int32)
    fn = {"literal": lib().oracle_fps, "keyed": lib().oracle_fps_keyed}[variant]
    fn(_p(xyz), b, n, m, _p(idx))
    return idx


def furthest_point_sampling_bucketed(xyz, m, order, bucket=64):
    """Single cloud (n,3); returns (idx (m,), number of distance evaluations)."""
    xyz = _f(xyz)
    order = _i(order)
    idx = np.zeros((max(m, 0),), np.int32)
    ev = lib().oracle_fps_bucketed(_p(xyz), xyz.shape[0], m, _p(order), bucket, _p(idx))
    return idx, int(ev)


def gather_points(points, idx):
    points, idx = _f(points), _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_gather_points(_p(points), _p(idx), _p(out), b, c, n, m)
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f(grad_out), _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_gather_points_grad(_p(grad_out), _p(idx), _p(out), b, c, n, m)
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, xyz = _f(new_xyz), _f(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().oracle_ball_query(_p(new_xyz), _p(xyz), _p(idx), b, n, m, ctypes.c_float(radius), nsample)
    return idx


def group_points(points, idx):
    points, idx = _f(points), _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().oracle_group_points(_p(points), _p(idx), _p(out), b, c, n, npoints, nsample)
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f(grad_out), _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_group_points_grad(_p(grad_out), _p(idx), _p(out), b, c, n, npoints, nsample)
    return out


def three_nn(unknown, known):
    """Returns (dist2, idx): SQUARED distances, as the extension does (the sqrt is in pointnet2_utils.py:139)."""
    unknown, known = _f(unknown), _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().oracle_three_nn(_p(unknown), _p(known), _p(dist2), _p(idx), b, n, m)
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_three_interpolate(_p(points), _p(idx), _p(weight), _p(out), b, c, m, n)
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f(grad_out), _i(idx), _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), _p(out), b, c, n, m)
    return out


# ---- variable-length batches of point-major tables (the product's encoder path; no counterpart in the reference's
# ---- extension: per scene these are the reference ops above on that scene alone, model_vdetr.py:285-316) -------------
def furthest_point_sampling_varlen(xyz_list, m):
    return np.concatenate([furthest_point_sampling(np.ascontiguousarray(x)[None], m) for x in xyz_list], 0)


def gather_rows(rows, idx):
    return np.stack([np.asarray(r)[np.asarray(idx)[i]] for i, r in enumerate(rows)], 0).astype(np.float32)


def gather_rows_grad(grad_out, idx, counts):
    out = []
    for i, n in enumerate(counts):
        g = np.zeros((int(n), grad_out.shape[2]), dtype=np.float32)
        np.add.at(g, np.asarray(idx)[i], grad_out[i])
        out.append(g)
    return out
