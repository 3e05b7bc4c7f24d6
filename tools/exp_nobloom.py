"""Upper bound of any bloom optimisation on the headline: fhd frames per second with two frames in flight, with and
without the bloom passes (skip_bloom still runs the combine in the V kernel).  Usage: python tools/exp_nobloom.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhr_amd import workloads
wl = dict(width=1920, height=1080, cam_pos=[6, 0, 0.5], fov=90, step_size=0.1, disk_tilt=0.0, anti_alias="disabled")
r, _, _, _ = workloads.make_scene(wl)
for rep in range(2):
    for skip in (False, True):
        for _ in range(30):
            r.render_async(wl["cam_pos"], wl["fov"], skip_bloom=skip)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(300):
            r.render_async(wl["cam_pos"], wl["fov"], skip_bloom=skip)
        r.sync()
        print(f"skip_bloom={skip}: {300 / (time.perf_counter() - t0):.1f} fps", flush=True)
r.close()
