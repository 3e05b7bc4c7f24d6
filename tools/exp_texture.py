import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd
from bhr_amd import HipRenderer, scenes
from bhr_amd.drivers import init_lifecycle_system, advance_lifecycle_frame
n_r, n_phi = 416, 2912
r = HipRenderer(1920, 1080, scenes.analytic_skybox(64, 128), np.zeros((n_r, n_phi, 4), dtype=np.float32))
fac = init_lifecycle_system(r, n_r, n_phi, seed=42)
bg, cm = [], []
for k in range(10):
    r.generate_background(0.1 * k); r.compose_interactive_texture(); c = r.counters(); bg.append(c["background_ms"]); cm.append(c["compose_ms"])
print("background_ms", np.median(bg), "compose+mips_ms", np.median(cm))
