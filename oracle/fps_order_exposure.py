"""How many furthest-point-sampling picks depend on the floating-point contraction order of the squared distance?

TEST INFRASTRUCTURE (oracle/).  The reference computes ``(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)`` in CUDA
(third_party/pointnet2/_ext_src/src/sampling_gpu.cu:103-107) and nvcc contracts it (-fmad=true) in an order that cannot be
observed in this image (no nvcc, no CUDA device).  ``pointnet2_oracle.c`` and the HIP kernels pin
    order 0:  t = dy*dy;  t = fma(dx,dx,t);  t = fma(dz,dz,t)      (what LLVM's DAG combiner emits for the expression)
The other candidates:
    order 1:  t = dx*dx;  t = fma(dy,dy,t);  t = fma(dz,dz,t)      (the first product rounded)
    order 2:  (dx*dx + dy*dy) + dz*dz                              (no contraction, nvcc -fmad=false)
On voxel-grid clouds the winner of a round is often decided between points whose distances differ in the last bit, so
the orders may pick different points; one different pick changes every later round.  This script builds the oracle in
the three orders (into oracle/_variants/, git-ignored) and reports, per scene of BASELINE.json's configurations, the
number of sampled indices that differ from order 0 and the first round at which they part.

    python oracle/fps_order_exposure.py > profiles/r06_fps_order_exposure.txt
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
VAR = os.path.join(HERE, "_variants")


def build(order):
    os.makedirs(VAR, exist_ok=True)
    so = os.path.join(VAR, f"liboracle_pointnet2_order{order}.so")
    src = os.path.join(HERE, "pointnet2_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", *fma,
                               f"-DORACLE_SQ3_ORDER={order}", "-o", so, src, "-lm"])
    return ctypes.CDLL(so)


def fps(lib, xyz, m):
    xyz = np.ascontiguousarray(xyz, np.float32)
    idx = np.zeros((1, m), np.int32)
    lib.oracle_fps_keyed(xyz.ctypes.data_as(ctypes.c_void_p), 1, xyz.shape[0], m, idx.ctypes.data_as(ctypes.c_void_p))
    return idx[0]


def grid_scene(npoints, seed, voxel=0.04):
    """bench.make_scene's cloud: uniform in an 8 x 6 x 3 m room + 1 m, on the 4 cm grid, shuffled"""
    rng = np.random.default_rng(seed)
    pts = rng.uniform([0, 0, 0], [8, 6, 3], (npoints, 3)) + 1.0
    vox = np.unique(np.round(pts / voxel).astype(np.int64), axis=0)
    rng.shuffle(vox)
    return (vox * voxel).astype(np.float32)


def lattice(nx, ny, nz, step=0.04, seed=0):
    """a FULL lattice: every distance is shared by many points (the tie-heaviest cloud there is)"""
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
    np.random.default_rng(seed).shuffle(g)
    return ((g + 25) * step).astype(np.float32)


def scenes():
    yield "C1 (4k points, m = min(4096, n))", grid_scene(4000, 0), None
    yield "C2 (40k points -> 4096)", grid_scene(40000, 0), 4096
    for r in (1, 2, 3):
        yield f"C2, scene seed {r} (rank {r} of C3)", grid_scene(40000, r), 4096
    yield "C4 (80k points -> 4096)", grid_scene(80000, 0), 4096
    for i in range(4):
        yield f"C5 scene {i} (20k points -> 4096)", grid_scene(20000, i), 4096
    yield "full 32 x 32 x 16 lattice (16384 points -> 4096)", lattice(32, 32, 16), 4096
    yield "raw uniform cloud, not on a grid (40k -> 4096)", (np.random.default_rng(5).uniform([0, 0, 0], [8, 6, 3], (40000, 3)) + 1).astype(np.float32), 4096


def main():
    libs = {o: build(o) for o in (0, 1, 2)}
    print("# furthest-point sampling: sampled indices that differ from contraction order 0 (the pinned one), out of m;")
    print("# `first` = the first round whose pick differs (every later round then works on a different set)")
    print(f"{'scene':58s} {'n':>6s} {'m':>5s} | {'order 1: differ':>15s} {'first':>6s} {'same set':>9s} | {'order 2: differ':>15s} {'first':>6s} {'same set':>9s}")
    worst = 0
    for name, xyz, m in scenes():
        n = xyz.shape[0]
        m = min(4096, n) if m is None else m
        base = fps(libs[0], xyz, m)
        cols = []
        for o in (1, 2):
            got = fps(libs[o], xyz, m)
            diff = np.nonzero(got != base)[0]
            same_set = len(np.intersect1d(got, base))
            cols.append((len(diff), int(diff[0]) if len(diff) else -1, same_set))
            worst = max(worst, len(diff))
        print(f"{name:58s} {n:6d} {m:5d} | {cols[0][0]:15d} {cols[0][1]:6d} {cols[0][2]:9d} | {cols[1][0]:15d} {cols[1][1]:6d} {cols[1][2]:9d}")
    print(f"# largest number of differing picks over the scenes: {worst}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
