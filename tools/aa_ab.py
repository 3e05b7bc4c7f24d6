import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, bhr_amd
from bhr_amd import HipRenderer, scenes
from bhr_amd.textures import compute_disk_texture_resolution
res = {}
for name, (W, H, tilt) in {"fhd_aa": (1920, 1080, 0.0), "4k_aa": (3840, 2160, 25.0)}.items():
    n_phi, n_r = compute_disk_texture_resolution(W, H, [6, 0, 0.5], 90, 2.0, 15.0)
    r = HipRenderer(W, H, scenes.analytic_skybox(1024, 2048), scenes.noisy_disk(n_r, n_phi), disk_tilt=tilt, anti_alias="lod_radius", frame_slots=1)
    for _ in range(3): r.render_async([6, 0, 0.5], 90)
    r.timing_reset()
    for _ in range(20): r.render_async([6, 0, 0.5], 90)
    c = r.counters(); res[name] = (round(c["march_ms_sum"] / c["frames_timed"], 3), c["march_vgprs"], c["ray_steps"])
    img = r.render([6, 0, 0.5], 90); res[name + "_sum"] = float(img.astype(np.float64).sum())
    r.close()
print(os.environ.get("BHR_HIP_LIBRARY", "default")[-14:], res)
