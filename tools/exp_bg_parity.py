#!/usr/bin/env python3
"""Background planes against the reference-statement fixture (max / median / q99 per plane) and the kernel's time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import HipRenderer
texg = np.load(os.path.join(ROOT, "tests", "golden", "texture_ref.npz"))
for t in (0.0, 5.0, 36.5):
    ref = texg[f"f32_bg_t{t:g}"]
    n_r, n_phi = ref.shape[1:]
    hip = HipRenderer(32, 18, np.zeros((8, 16, 3), np.float32), np.zeros((n_r, n_phi, 4), np.float32), r_disk_inner=2.0, r_disk_outer=15.0)
    hip.init_background_layer(n_r, n_phi, seed=42)
    hip.generate_background(t)
    got = hip.read_comp()
    hip.close()
    for idx in (0, 3, 4, 11, 12):
        d = np.abs(got[idx] - ref[idx])
        print(f"t={t:g} plane {idx}: max {d.max():.3g} median {np.median(d):.3g} q99 {np.quantile(d, 0.99):.3g} identical {float((d == 0).mean()):.4f}")
hip = HipRenderer(32, 18, np.zeros((8, 16, 3), np.float32), np.zeros((416, 2912, 4), np.float32), r_disk_inner=2.0, r_disk_outer=15.0)
hip.init_background_layer(416, 2912, seed=42)
for _ in range(5):
    hip.generate_background(1.0)
hip.sync()
t0 = time.perf_counter()
for k in range(50):
    hip.generate_background(1.0 + 0.1 * k)
hip.sync()
print("background kernel at 416x2912:", (time.perf_counter() - t0) / 50 * 1e3, "ms;", hip.counters()["background_ms"])
hip.close()
