#!/usr/bin/env python3
"""bench.py's timed region with the driver's step counts (--steps 20 --warmup 5) next to the default (200 / 20):
per-frame device intervals from the timing ring and host submission stamps, to see where a short run loses the overlap
of two frames in flight (clock ramp after the scene set-up, host gaps, second-slot set-up).

usage: python tools/exp_bench_ramp.py [--out gpurun_out/bench_ramp.json] [--math strict]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from bhr_amd import workloads


def region(r, wl, warm, steps):
    for _ in range(warm):
        r.render_async(wl["cam_pos"], wl["fov"])
    r.timing_reset()
    r.sync()
    host = []
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_async(wl["cam_pos"], wl["fov"])
        host.append((time.perf_counter() - t0) * 1e3)
    r.sync()
    wall = (time.perf_counter() - t0) * 1e3
    ft = r.frame_times(min(steps, 500))
    return {"warmup": warm, "steps": steps, "wall_ms": wall, "ms_per_step": wall / steps, "fps": steps / wall * 1e3,
            "host_submit_ms": [round(h, 4) for h in host[:40]], "host_submit_last_ms": host[-1],
            "frames": [[round(float(x), 4) for x in row] for row in ft[:40]],
            "march_ms_first10": [round(float(b - a), 4) for a, b, _ in ft[:10]],
            "march_ms_last10": [round(float(b - a), 4) for a, b, _ in ft[-10:]],
            "device_span_ms": float(ft[-1, 2])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_ramp.json"))
    ap.add_argument("--math", default=None)
    ap.add_argument("--spin-up-ms", type=float, default=0.0, help="un-timed frames for this long before the first region (bench.py's spin-up)")
    a = ap.parse_args()
    wl = bench.WORKLOADS["fhd"]
    res = {}
    for slots in (2, 1):
        t0 = time.perf_counter()
        r, _, _, _ = workloads.make_scene(wl, math=a.math, frame_slots=slots)
        setup = time.perf_counter() - t0
        runs = []
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < a.spin_up_ms:
            for _ in range(8):
                r.render_async(wl["cam_pos"], wl["fov"])
            r.sync()
        # the driver's counts FIRST, straight after the scene set-up (as in a fresh bench.py process), then again, then long
        for warm, steps in ((5, 20), (5, 20), (20, 200), (5, 20)):
            runs.append(region(r, wl, warm, steps))
            print(f"slots {slots} warm {warm} steps {steps}: {runs[-1]['fps']:.0f} fps, march first10 {runs[-1]['march_ms_first10']}", flush=True)
        res[f"slots{slots}"] = {"setup_s": setup, "runs": runs}
        r.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
