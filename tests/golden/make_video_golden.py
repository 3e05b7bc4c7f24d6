#!/usr/bin/env python3
"""Generates tests/golden/video_ref.npz: frames of the reference's OWN video loop (BASELINE.json configs[4] at the
e2e size) -- `_init_lifecycle_system`, then per frame `_advance_lifecycle_frame(renderer, factories, t, dt,
recompute_stats=(frame % 60 == 0))` and `renderer.render(orbit camera, fov, frame=0)` exactly as render_video does
(render.py:4419-4453) -- with its Taichi kernels running as plain Python on tests/golden/ti_shim.py in binary32 mode
(see make_kernel_golden.py).  What it pins beyond e2e_ref.npz: the texture at t > 0 (background kernel with rotation,
Keplerian roll of the entities, births and deaths of the populations), the statistics cadence, the orbit camera.

Frames that are not stored only tick the populations (the reference's kernels are pure functions of the populations,
t and the statistics, which are recomputed at frame 0 only within these frames).  ~20 minutes per stored frame.

Run in the build container only:  python tests/golden/make_video_golden.py [--out DIR] [--frames 0,2,7]
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import ti_shim  # noqa: E402

ti_shim.install()
ti_shim.set_default_fp("f32")
sys.path.insert(0, REF)
import render as ref  # noqa: E402

# the e2e scene (tests/e2e_render.py:27-43) as a 24-frame full orbit
W, H, POV, FOV = 320, 180, [6.0, 0.0, 0.5], 60.0
SCENE = dict(step_size=0.1, r_max=10, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15)
N_FRAMES, ORBIT_DEGREES, SPEED, N_STARS = 24, 360.0, 0.1, 100


def main():
    out_dir = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else HERE
    stored = [int(x) for x in sys.argv[sys.argv.index("--frames") + 1].split(",")] if "--frames" in sys.argv else [0, 2, 7]
    # render.py:4626-4644 (_make_renderer_with_placeholder)
    skybox, _, _ = ref.load_or_generate_skybox(None, 2048, 1024, N_STARS)
    n_phi, n_r = ref.compute_disk_texture_resolution(W, H, POV, FOV, SCENE["r_disk_inner"], SCENE["r_disk_outer"])
    renderer = ref.TaichiRenderer(W, H, skybox, np.zeros((n_r, n_phi, 4), dtype=np.float32), step_size=SCENE["step_size"],
                                  r_max=SCENE["r_max"], device="cpu", r_disk_inner=SCENE["r_disk_inner"],
                                  r_disk_outer=SCENE["r_disk_outer"], disk_tilt=SCENE["disk_tilt"], lens_flare=False,
                                  anti_alias="disabled", aa_strength=1.0, disk_rotation_speed=SPEED, ignore_taichi_cache=True)
    # render.py:4419-4422, 4436-4453
    factories = ref._init_lifecycle_system(renderer, renderer.dtex_h, renderer.dtex_w, seed=42)
    dt = SPEED
    orbit_radius = float(np.linalg.norm(POV))
    angle_step = ORBIT_DEGREES / N_FRAMES
    out = dict(frames=np.array(stored), n_frames=N_FRAMES, orbit_degrees=ORBIT_DEGREES, speed=SPEED, n_stars=N_STARS,
               sky_sha256=hashlib.sha256(np.ascontiguousarray(skybox).tobytes()).hexdigest(), tex_shape=np.array([n_r, n_phi]))
    for frame in range(max(stored) + 1):
        t = frame * dt
        if frame not in stored:
            assert frame % 60 != 0
            for f in factories.values():
                f.tick(now=t, dt=dt)
            continue
        t0 = time.time()
        angle_rad = np.radians(frame * angle_step)
        cam_pos = [orbit_radius * np.cos(angle_rad), orbit_radius * np.sin(angle_rad), POV[2]]
        ref._advance_lifecycle_frame(renderer, factories, t, dt, recompute_stats=(frame % 60 == 0))
        img = np.ascontiguousarray(renderer.render(cam_pos, FOV, frame=0), dtype=np.float32)
        out[f"final_{frame}"] = img
        out[f"cam_{frame}"] = np.array(cam_pos, dtype=np.float64)
        out[f"disk_tex_{frame}"] = renderer.disk_texture_field.to_numpy()
        out[f"alive_{frame}"] = np.array([len(factories[k].alive_entities) for k in ("filament", "hotspot", "rt_spike")])
        print(f"frame {frame}: {time.time() - t0:.0f} s, md5 {hashlib.md5(img.tobytes()).hexdigest()}", flush=True)
    out["stats"] = renderer._param_stats_field.to_numpy()
    out["row_stats"] = renderer._param_row_stats_field.to_numpy()
    np.savez_compressed(os.path.join(out_dir, "video_ref.npz"), **out)


if __name__ == "__main__":
    main()
