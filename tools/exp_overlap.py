"""How much throughput is left in the tails?  One context rendering N frames back to back against two contexts
(two streams) rendering N/2 frames each concurrently, same scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bhr_amd.workloads import make_scene
wl = bench.WORKLOADS["fhd"]
for math in ("strict", "fast"):
    a, sky, tex, _ = make_scene(wl, n_stars=2000, math=math)
    b, _, _, _ = make_scene(wl, n_stars=2000, math=math)
    cam, fov = wl["cam_pos"], wl["fov"]
    for r in (a, b):
        for _ in range(10):
            r.render_async(cam, fov)
        r.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        a.render_async(cam, fov)
    a.sync()
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n // 2):
        a.render_async(cam, fov)
        b.render_async(cam, fov)
    a.sync(); b.sync()
    t2 = time.perf_counter() - t0
    print(f"{math}: one stream {n / t1:.0f} fps ({t1 / n * 1e3:.3f} ms/frame), two streams {n / t2:.0f} fps ({t2 / n * 1e3:.3f} ms/frame): "
          f"{(t1 / t2 - 1) * 100:+.1f} %", flush=True)
    a.close(); b.close()
