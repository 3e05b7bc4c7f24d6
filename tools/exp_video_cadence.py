"""Frame cadence inside the video driver's loop from the library's own event ring (no profiler): start-to-start interval,
march bracket and frame bracket of the last frames, beside the host's loop time.  Usage: python tools/exp_video_cadence.py [n]"""
import os, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import drivers
from bhr_amd.output import FrameSink, DEVICE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tmp = tempfile.mkdtemp(prefix="bhr_video_")
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math="hybrid")
factories = drivers.init_lifecycle_system(r, r.dtex_h, r.dtex_w, seed=42)
for mode in ("full", "no_sink", "no_texture"):
    sink = FrameSink(r, slots=0, workers=4, level=DEVICE) if mode != "no_sink" else None
    for phase in (0, 1):
        if phase == 1:
            r.sync(); r.timing_reset(); t0 = time.perf_counter()
        for f in range(300 if phase == 0 else n):
            if mode != "no_texture":
                drivers.advance_lifecycle_frame(r, factories, f * 0.1, 0.1, recompute_stats=(f % 60 == 0), compose=True)
            r.render_async(drivers.orbit_position([6, 0, 0.5], f, 3600, 360.0), 90, frame=0)
            if sink is not None:
                sink.submit(os.path.join(tmp, f"frame_{f:04d}.png"))
                if (f + 1) % 50 == 0:
                    sink.drain()
    if sink is not None:
        sink.drain()
    r.sync()
    dt = (time.perf_counter() - t0) / n * 1e3
    ft = r.frame_times(min(n, 400))
    starts = ft[:, 0]
    iv = np.diff(starts)
    print(f"{mode}: host loop {dt:.3f} ms/frame; device start-to-start median {np.median(iv):.3f} (p10 {np.quantile(iv, 0.1):.3f}, p90 {np.quantile(iv, 0.9):.3f}); "
          f"march bracket median {np.median(ft[:, 1] - ft[:, 0]):.3f}; frame bracket median {np.median(ft[:, 2] - ft[:, 0]):.3f}", flush=True)
    if sink is not None:
        sink.close()
r.close(); shutil.rmtree(tmp)
