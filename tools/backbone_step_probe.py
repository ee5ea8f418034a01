#!/usr/bin/env python3
"""Where does the step with the sparse backbone spend its wall time?  Host enqueue time vs device time, with the next scene's
geometry built by the loader thread or reused (VDETR_BENCH_GEOMETRY=static).   python tools/backbone_step_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def run(tr, steps):
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        tr.step()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    try:
        from vdetr_amd.runtime import enable_gemm_tuning
        enable_gemm_tuning(0)
    except Exception:
        pass
    tr = bench.BackboneTrainer("c2", dev)
    tr.capture()
    for mode in os.environ.get("PROBE_MODES", "static thread inline static inline").split():
        if ":" in mode:
            sys.setswitchinterval(float(mode.split(":")[1]))
            mode = "thread"
        os.environ["VDETR_BENCH_GEOMETRY"] = mode
        tr.close()
        ms, host = run(tr, 10)
        print(f"geometry={mode:7s}  {ms:7.2f} ms/step   host enqueue {host:7.2f} ms/step")
    if "--syncdebug" in sys.argv:  # which operators of a steady-state step synchronise the host with the device?
        os.environ["VDETR_BENCH_GEOMETRY"] = "static"
        tr.close()
        for _ in range(3):
            tr.step()
        torch.cuda.synchronize()
        import warnings
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        tr.step()
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    if "--timeline" in sys.argv:
        for mode in ("static", "inline", "thread"):
            os.environ["VDETR_BENCH_GEOMETRY"] = mode
            tr.close()
            for _ in range(4):
                tr.step()
            tr.marks = []
            for _ in range(10):
                tr.step()
            torch.cuda.synchronize()
            m, tr.marks = tr.marks, None
            names = ["backbone fwd", "decoder graph", "backbone bwd", "pack+clip+adamw", "gap to next step"]
            acc = [0.0] * 5
            for i in range(0, len(m) - 5, 5):
                for j in range(5):
                    acc[j] += m[i + j].elapsed_time(m[i + j + 1])
            n = (len(m) - 5) // 5
            print(mode, {k: round(v / n, 2) for k, v in zip(names, acc)})
    if "--host" in sys.argv:  # pure host cost of enqueuing one step: the device queue is empty when the step starts
        os.environ["VDETR_BENCH_GEOMETRY"] = "static"
        tr.close()
        for _ in range(3):
            tr.step()
        hs = []
        for _ in range(8):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            tr.step()
            hs.append((time.perf_counter() - h0) * 1e3)
        torch.cuda.synchronize()
        print("host enqueue of one step on an idle device (ms):", [round(h, 2) for h in hs])
        import cProfile
        import pstats
        pr = cProfile.Profile()
        for _ in range(5):
            torch.cuda.synchronize()
            pr.enable()
            tr.step()
            pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    if "--phases" in sys.argv:  # host wall time of the pieces of the inline geometry build while the device is busy
        from vdetr_amd import minkowski as ME
        acc = {}

        def timed(name, fn):
            def wrap(*a, **k):
                t0 = time.perf_counter()
                r = fn(*a, **k)
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
                return r
            return wrap
        ME.CoordinateManager.insert_points = timed("insert_points", ME.CoordinateManager.insert_points)
        ME.CoordinateManager.strided = timed("strided", ME.CoordinateManager.strided)
        ME.CoordinateManager.kernel_map = timed("kernel_map", ME.CoordinateManager.kernel_map)
        ME.CoordinateManager.finalize = timed("finalize", ME.CoordinateManager.finalize)
        ME.CoordinateManager.scene_counts = timed("scene_counts", ME.CoordinateManager.scene_counts)
        ME.batch_sparse_collate = timed("collate", ME.batch_sparse_collate)
        tr._prepare_next = timed("prepare_next_total", tr._prepare_next)
        os.environ["VDETR_BENCH_GEOMETRY"] = "inline"
        tr.close()
        for _ in range(3):
            tr.step()
        acc.clear()
        for _ in range(10):
            tr.step()
        torch.cuda.synchronize()
        print({k: round(v / 10 * 1e3, 2) for k, v in acc.items()})
    if "--cprofile" in sys.argv:
        import cProfile
        import pstats
        os.environ["VDETR_BENCH_GEOMETRY"] = "static"
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            tr.step()
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(45)
    tr.close()


if __name__ == "__main__":
    main()
