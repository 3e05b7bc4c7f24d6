"""Texture pipeline kernels at the fhd texture size (416 x 2912): background / entity / compose time per call (HIP-side
wall time of 200 back-to-back calls).  Usage: python tools/exp_bg.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bhr_amd import drivers
r, _, n_r, n_phi = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=100, tex_w=256, tex_h=128)
fac = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
drivers.advance_lifecycle_frame(r, fac, 0.0, 0.0, recompute_stats=True)
for name, fn in (("generate_background", lambda k: r.generate_background(0.1 * k)),
                 ("compose_interactive_texture", lambda k: r.compose_interactive_texture()),
                 ("accumulate_entity_layer", lambda k: r.accumulate_entity_layer(fac, 0.0))):
    for k in range(10):
        fn(k)
    r.sync()
    t0 = time.perf_counter()
    for k in range(200):
        fn(k)
    r.sync()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call ({n_r} x {n_phi})", flush=True)
r.close()
