import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd
from bhr_amd import HipRenderer, scenes
from bhr_amd.drivers import init_lifecycle_system, advance_lifecycle_frame
from bhr_amd.textures import compute_disk_texture_resolution
for name, (W, H) in (("fhd", (1920, 1080)), ("4k", (3840, 2160))):
    n_phi, n_r = compute_disk_texture_resolution(W, H, [6, 0, 0.5], 90, 2.0, 15.0)
    for dev in (True, False):
        r = HipRenderer(W, H, scenes.analytic_skybox(64, 128), np.zeros((n_r, n_phi, 4), np.float32))
        r.device_lifecycle = dev
        t0 = time.perf_counter(); fac = init_lifecycle_system(r, n_r, n_phi); t_init = time.perf_counter() - t0
        ts = []
        for k in range(1, 9):
            t0 = time.perf_counter(); advance_lifecycle_frame(r, fac, 0.1 * k, 0.1); r.sync(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); r.recompute_interactive_stats(); t_stats = time.perf_counter() - t0
        print(f"{name} device_lifecycle={dev}: init {t_init*1e3:.0f} ms, advance {np.median(ts)*1e3:.2f} ms/frame, stats {t_stats*1e3:.2f} ms")
        r.close()
