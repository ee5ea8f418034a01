#!/usr/bin/env python3
"""Host model of fps_rows.hip's rounds (numpy, CPU only): which buckets survive the box test in every round, and how
they spread over the waves, for alternative bucket layouts and bucket -> wave maps.  Used to decide what to build
(DESIGN.md 4.1); the sample ORDER is the oracle's, the statistics are what the kernel would see.

    python tools/fps_model.py [npoints] [nsamples]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def split_sequence(ext, bits=15):
    cell = ext.astype(np.float64).copy()
    seq = []
    for _ in range(bits):
        a = int(np.argmax(cell >= cell.max()))  # first longest
        seq.append(a)
        cell[a] *= 0.5
    return seq


def codes_of(x, bits=15):
    lo, hi = x.min(0), x.max(0)
    ext = np.maximum(hi - lo, 1e-30)
    seq = split_sequence(ext, bits)
    nb = [seq.count(a) for a in range(3)]
    c = [np.minimum(((x[:, a] - lo[a]) * ((1 << nb[a]) / ext[a])).astype(np.int64), (1 << nb[a]) - 1) for a in range(3)]
    rem = list(nb)
    code = np.zeros(x.shape[0], np.int64)
    for a in seq:
        rem[a] -= 1
        code = (code << 1) | ((c[a] >> rem[a]) & 1)
    return code


def fixed_runs(order, bp=64):
    """the kernel's layout: consecutive runs of bp points of the Z-ordered cloud"""
    n = order.shape[0]
    return [order[i:i + bp] for i in range(0, n, bp)]


def tree_leaves(code_sorted, order, bits=15, cap=64):
    """buckets = maximal nodes of the binary tree over the code with <= cap points (a full cell with more is chopped)"""
    out = []

    def rec(lo_i, hi_i, depth):
        cnt = hi_i - lo_i
        if cnt == 0:
            return
        if cnt <= cap or depth == bits:
            for s in range(lo_i, hi_i, cap):
                out.append(order[s:min(s + cap, hi_i)])
            return
        shift = bits - 1 - depth
        prefix = code_sorted[lo_i] >> (shift + 1)
        mid_code = ((prefix << 1) | 1) << shift
        mid = lo_i + int(np.searchsorted(code_sorted[lo_i:hi_i], mid_code, side="left"))
        rec(lo_i, mid, depth + 1)
        rec(mid, hi_i, depth + 1)

    rec(0, code_sorted.shape[0], 0)
    return out


def simulate(x, samples, buckets, wave_of, waves=16, label=""):
    nb = len(buckets)
    lo = np.stack([x[b].min(0) for b in buckets])
    hi = np.stack([x[b].max(0) for b in buckets])
    t = np.full(x.shape[0], 1e10, np.float32)
    bmax = np.full(nb, 1e10, np.float32)
    pad = max(len(b) for b in buckets)
    idx = np.full((nb, pad), -1, np.int64)
    for i, b in enumerate(buckets):
        idx[i, :len(b)] = b
    surv_total, max_per_wave, pts = [], [], []
    hist = np.zeros(12, np.int64)
    for j in range(1, samples.shape[0]):
        c = x[samples[j - 1]]
        d = np.maximum(np.maximum(lo - c, c - hi), 0.0).astype(np.float32)
        bd = d[:, 1] * d[:, 1] + d[:, 0] * d[:, 0] + d[:, 2] * d[:, 2]
        s = np.nonzero(bd < bmax)[0]
        if s.size:
            ii = idx[s]
            valid = ii >= 0
            p = x[np.where(valid, ii, 0)]
            dd = ((p - c) ** 2).sum(-1).astype(np.float32)
            tt = np.where(valid, np.minimum(t[np.where(valid, ii, 0)], dd), -np.inf)
            t[ii[valid]] = tt[valid]
            bmax[s] = tt.max(1)
        if j > 64:  # the first rounds touch everything: not what the steady state looks like
            surv_total.append(s.size)
            pts.append(int((idx[s] >= 0).sum()) if s.size else 0)
            load = np.bincount(wave_of[s], minlength=waves)
            max_per_wave.append(load.max())
            hist[min(load.max(), 11)] += 1
    print(f"{label:44s} buckets {nb:5d}  survivors/round {np.mean(surv_total):6.2f}  points/round {np.mean(pts):7.1f}  "
          f"E[max per wave] {np.mean(max_per_wave):5.2f}  P(max>=3) {np.mean(np.array(max_per_wave) >= 3):.2f}  "
          f"hist(max)= {hist[:7].tolist()}")


def lattice_colour(x, buckets, waves=16):
    """bucket centre on a lattice of the mean bucket pitch, colour = (i + 3 j + 9 k) mod waves"""
    cen = np.stack([0.5 * (x[b].min(0) + x[b].max(0)) for b in buckets])
    lo, hi = x.min(0), x.max(0)
    vol = np.prod(np.maximum(hi - lo, 1e-9))
    h = (vol / len(buckets)) ** (1.0 / 3.0)
    ijk = np.floor((cen - lo) / h).astype(np.int64)
    return (ijk[:, 0] + 3 * ijk[:, 1] + 9 * ijk[:, 2]) % waves


def main():
    import bench
    from oracle import pointnet2_oracle as O
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    xyz, _ = bench.make_scene(npts, 0, "cpu")
    x = xyz.numpy().astype(np.float32)
    t0 = time.time()
    samples = O.furthest_point_sampling(x[None], m)[0]
    code = codes_of(x)
    order = np.argsort(code, kind="stable")
    cs = code[order]
    print(f"n = {x.shape[0]}, m = {m}, oracle {time.time() - t0:.1f} s")
    runs = fixed_runs(order)
    g = np.arange(len(runs))
    simulate(x, samples, runs, g % 16, label="64-point runs, wave = g % 16 (the kernel)")
    simulate(x, samples, runs, lattice_colour(x, runs), label="64-point runs, lattice colouring")
    rng = np.random.default_rng(0)
    simulate(x, samples, runs, rng.integers(0, 16, len(runs)), label="64-point runs, random wave")
    leaves = tree_leaves(cs, order)
    gl = np.arange(len(leaves))
    simulate(x, samples, leaves, gl % 16, label="tree leaves <= 64, wave = g % 16")
    simulate(x, samples, leaves, lattice_colour(x, leaves), label="tree leaves <= 64, lattice colouring")
    leaves32 = tree_leaves(cs, order, cap=32)
    simulate(x, samples, leaves32, np.arange(len(leaves32)) % 16, label="tree leaves <= 32, wave = g % 16")


if __name__ == "__main__":
    main()
