"""cProfile of the per-frame lifecycle host work at fhd (configs[4])."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhr_amd import drivers
r, _, n_r, n_phi = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=100, tex_w=256, tex_h=128)
f = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
for k in range(5):
    drivers.advance_lifecycle_frame(r, f, k * 0.1, 0.1)
r.sync()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for k in range(5, 105):
    drivers.advance_lifecycle_frame(r, f, k * 0.1, 0.1, recompute_stats=False)
r.sync()
pr.disable()
print(f"{(time.perf_counter() - t0) * 10:.2f} ms/frame (profiled)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
