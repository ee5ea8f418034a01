#!/usr/bin/env python3
"""A/B of the furthest-point-sampling kernels on the GPU box: for every workgroup size (4, 8, 16 waves) of
fps_rows.hip and for fps.hip's kernel, time the C2 (40k -> 4096) and C4 (80k -> 4096) scenes, compare the indices
with the C oracle bit for bit, and (with --debug) print the in-kernel phase counters.

    python tools/fps_variants.py [--debug] [--quick]

The geometry is read from the environment once per process (VDETR_FPS_IMPL / _WAVES), so every variant runs
in a child process; this parent never touches the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(sizes, m, debug):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bench
    from oracle import pointnet2_oracle as O
    from vdetr_amd import pointnet2_utils as PU
    out = []
    for npts in sizes:
        xyz, _ = bench.make_scene(npts, 0, "cuda")
        x = xyz[None].contiguous()
        n = x.shape[1]
        mm = min(m, n)
        got = PU.furthest_point_sample(x, mm)  # warm-up (+ the debug print, if enabled)
        torch.cuda.synchronize()
        rec = {"n": n, "m": mm}
        if not debug:
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                got = PU.furthest_point_sample(x, mm)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            rec["ms"] = min(ts)
            rec["us_per_round"] = min(ts) * 1e3 / max(mm - 1, 1)
            ref = O.furthest_point_sampling(x.cpu().numpy(), mm)
            bad = np.nonzero(ref != got.cpu().numpy())[1]
            rec["exact"] = bool(bad.size == 0)
            if bad.size:
                rec["first_bad"] = int(bad[0])
        out.append(rec)
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if "--child" in sys.argv:
        i = sys.argv.index("--child")
        sizes = [int(v) for v in sys.argv[i + 1].split(",")]
        return child(sizes, int(sys.argv[i + 2]), "--debug" in sys.argv)
    debug = "--debug" in sys.argv
    sizes = "40000,80000" if "--quick" not in sys.argv else "40000"
    if "--sizes" in sys.argv:
        sizes = sys.argv[sys.argv.index("--sizes") + 1]
    variants = [("v2", {"VDETR_FPS_IMPL": "2"})]
    for wv in (16, 8, 4):
        variants.append((f"rows W={wv}", {"VDETR_FPS_WAVES": str(wv)}))
    for tr in (0, 2):
        variants.append((f"rows tree={tr}", {"VDETR_FPS_TREE": str(tr)}))
    variants.append(("default", {}))
    if "--only" in sys.argv:
        keep = sys.argv[sys.argv.index("--only") + 1].split(";")
        variants = [v for v in variants if v[0] in keep]
    for name, env in variants:
        e = dict(os.environ)
        for k in ("VDETR_FPS_IMPL", "VDETR_FPS_WAVES", "VDETR_FPS_DEBUG", "VDETR_FPS_TREE"):
            e.pop(k, None)
        e.update(env)
        if debug:
            e["VDETR_FPS_DEBUG"] = "1"
        cmd = [sys.executable, os.path.abspath(__file__), "--child", sizes, "4096"] + (["--debug"] if debug else [])
        try:
            p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            print(f"{name}: TIMEOUT", flush=True)
            continue
        res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        print(f"== {name}: rc={p.returncode} " + (res[0][7:] if res else p.stdout[-400:] + p.stderr[-1500:]), flush=True)
        if debug:
            print("\n".join(ln for ln in p.stderr.splitlines() if "fps" in ln), flush=True)


if __name__ == "__main__":
    main()
