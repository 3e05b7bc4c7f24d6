"""Numbers quoted in DESIGN.md: march/bloom per config and math mode, readback, lifecycle host time."""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bhr_amd
from bhr_amd import HipRenderer, scenes, _lib
from bhr_amd.drivers import init_lifecycle_system, advance_lifecycle_frame
from bhr_amd.textures import compute_disk_texture_resolution

def timed(r, cam, fov, n=20, **kw):
    for _ in range(3): r.render_async(cam, fov, **kw)
    r.timing_reset()
    for _ in range(n): r.render_async(cam, fov, **kw)
    c = r.counters()
    return c["march_ms_sum"] / c["frames_timed"], c["bloom_ms_sum"] / c["frames_timed"], c["ray_steps"], c["march_vgprs"]

res = {}
cfgs = {"fhd": (1920, 1080, 0.1, 0.0, "disabled"), "fhd_aa": (1920, 1080, 0.1, 0.0, "lod_radius"),
        "4k_tilt25_aa": (3840, 2160, 0.1, 25.0, "lod_radius"), "8k_step005": (7680, 4320, 0.05, 0.0, "disabled")}
for name, (W, H, step, tilt, aa) in cfgs.items():
    n_phi, n_r = compute_disk_texture_resolution(W, H, [6, 0, 0.5], 90, 2.0, 15.0)
    sky, tex = scenes.analytic_skybox(1024, 2048), scenes.noisy_disk(n_r, n_phi)
    for math in ("strict", "fast"):
        r = HipRenderer(W, H, sky, tex, step_size=step, disk_tilt=tilt, anti_alias=aa, math=math, frame_slots=1)   # isolated launches
        m, b, steps, vg = timed(r, [6, 0, 0.5], 90, n=10 if W > 4000 else 20)
        res[f"{name}/{math}"] = dict(march_ms=round(m, 3), bloom_ms=round(b, 3), gsteps_per_s=round(steps / m / 1e6, 1), steps_per_ray=round(steps / W / H, 2), vgprs=vg)
        if name == "fhd" and math == "strict":
            t0 = time.perf_counter(); 
            for _ in range(10): r.render([6, 0, 0.5], 90)
            res["fhd_render_with_f32_readback_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
            t0 = time.perf_counter()
            for _ in range(10): r.render_async([6, 0, 0.5], 90); r.read_final_u8()
            res["fhd_render_with_u8_readback_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        r.close()
# lifecycle host cost at fhd
n_phi, n_r = compute_disk_texture_resolution(1920, 1080, [6, 0, 0.5], 90, 2.0, 15.0)
r = HipRenderer(1920, 1080, scenes.analytic_skybox(64, 128), np.zeros((n_r, n_phi, 4), np.float32))
fac = init_lifecycle_system(r, n_r, n_phi)
t0 = time.perf_counter()
for k in range(5): advance_lifecycle_frame(r, fac, 0.1 * k, 0.1)
r.sync(); res["lifecycle_advance_ms_fhd"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
t0 = time.perf_counter(); r.recompute_interactive_stats(); res["stats_ms_fhd"] = round((time.perf_counter() - t0) * 1e3, 2)
print(json.dumps(res, indent=1))
