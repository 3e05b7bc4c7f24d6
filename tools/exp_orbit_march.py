import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from bhr_amd import drivers
r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math="hybrid")
n_r, n_phi = r.dtex_h, r.dtex_w
factories = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
drivers.advance_lifecycle_frame(r, factories, 0.0, 0.1, recompute_stats=True, compose=True)
def run(tag, cams, math=None):
    for c in cams[:20]: r.render_async(c, 90, math=math)
    r.sync(); r.timing_reset()
    t0 = time.perf_counter()
    for c in cams: r.render_async(c, 90, math=math)
    r.sync()
    dt = (time.perf_counter() - t0) / len(cams) * 1e3
    c = r.counters()
    print(tag, "ms/frame %.3f" % dt, "march_ms %.3f" % (c["march_ms_sum"] / max(c["frames_timed"], 1)), "steps/frame %.4g" % (c["ray_steps_sum"] / max(c["frames_timed"], 1)), r.hybrid_info() if (math or r.math) == "hybrid" else "", flush=True)
N = 400
static = [[6, 0, 0.5]] * N
orbit = [drivers.orbit_position([6, 0, 0.5], f, N, 360.0) for f in range(N)]
orbit_small = [drivers.orbit_position([6, 0, 0.5], f, 3600, 360.0) for f in range(N)]
for m in ("hybrid", "fast", "strict"):
    run("static " + m, static, m)
    run("orbit  " + m, orbit, m)
    run("orbit/3600 " + m, orbit_small, m)
r.close()
