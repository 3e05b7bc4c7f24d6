"""Timing of the device lens flare at the 4k config (HIP events around the two launches)."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bhr_amd import HipRenderer, scenes, _lib
for (w, h) in ((1920, 1080), (3840, 2160)):
    r = HipRenderer(w, h, scenes.analytic_skybox(), scenes.noisy_disk(), disk_tilt=25.0, anti_alias="lod_radius")
    cam = [6, 0, 0.5]
    for flare in (False, True):
        for _ in range(3):
            r.render_async(cam, 90, lens_flare=flare)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            r.render_async(cam, 90, lens_flare=flare)
        r.sync()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        c = r.counters()
        print(f"{w}x{h} flare={flare}: {dt:.3f} ms/frame wall, march {c['march_ms']:.3f} post {c['bloom_ms']:.3f}", flush=True)
    r.close()
