"""Scene set-up for bench.py / profiling: the BASELINE.json workloads with all inputs resident in HBM."""
from __future__ import annotations

import numpy as np

from .renderer import HipRenderer
from .textures import compute_disk_texture_resolution


def make_scene(wl: dict, device_index: int = 0, n_stars: int = 6000, math=None, frame_slots=None):
    """Renderer for workload ``wl`` (see bench.WORKLOADS) with the reference's default scene:
    procedural skybox (generate_skybox(2048, 1024, seed 42)) and the lifecycle disk texture at
    t = 0 (render_image, render.py:4044-4069).  Returns (renderer, skybox, disk_tex, note)."""
    from .drivers import init_lifecycle_system, advance_lifecycle_frame

    W, H = wl["width"], wl["height"]
    r_in, r_out = 2.0, 15.0
    n_phi, n_r = compute_disk_texture_resolution(W, H, wl["cam_pos"], wl["fov"], r_in, r_out)
    placeholder = np.zeros((n_r, n_phi, 4), dtype=np.float32)
    r = HipRenderer(W, H, np.zeros((1024, 2048, 3), dtype=np.float32), placeholder, step_size=wl["step_size"], r_max=10.0,
                    r_disk_inner=r_in, r_disk_outer=r_out, disk_tilt=wl["disk_tilt"], anti_alias=wl["anti_alias"],
                    device_index=device_index, frame_slots=frame_slots, **({} if math is None else {"math": math}))
    r.build_procedural_skybox(seed=42, n_stars=n_stars)              # random tables on the host, texels on the device
    sky = r.read_skybox()
    factories = init_lifecycle_system(r, n_r, n_phi, seed=42)
    advance_lifecycle_frame(r, factories, t=0.0, dt=0.0, recompute_stats=True)
    tex = r.disk_texture_field.to_numpy()
    note = (f"procedural skybox 1024x2048 seed 42 ({n_stars} stars) + lifecycle disk texture "
            f"{n_r}x{n_phi} seed 42 at t=0, generated on the device")
    return r, sky, tex, note


def plan_blocks(wl: dict, n: int, device_index: int = 0, balance: bool = True, math=None):
    """The n row blocks of workload ``wl``: cut by the cost profile of a probe frame rendered on ``device_index``
    (multigpu.balanced_row_blocks) -- deterministic, so every rank of a one-process-per-tile run computes the same cut."""
    from . import multigpu
    W, H = wl["width"], wl["height"]
    kw = dict(step_size=wl["step_size"], r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=wl["disk_tilt"])
    blocks = multigpu.aligned_row_blocks(H, n)
    if balance and n > 1 and H >= 64 * n:
        per_row, band_rows = multigpu.probe_row_costs(W, H, wl["cam_pos"], wl["fov"], device_index=device_index, math=math, **kw)
        blocks = multigpu.balanced_row_blocks(H, n, per_row, band_rows, fixed_cost_per_row=0.1 * float(per_row.mean()))
    return blocks


def make_tile(wl: dict, rows, device_index: int, n_stars: int = 6000, math=None):
    """One row-block renderer of workload ``wl`` with its own copy of the deterministic scene."""
    from .drivers import init_lifecycle_system, advance_lifecycle_frame
    W, H = wl["width"], wl["height"]
    r_in, r_out = 2.0, 15.0
    n_phi, n_r = compute_disk_texture_resolution(W, H, wl["cam_pos"], wl["fov"], r_in, r_out)
    kw = dict(step_size=wl["step_size"], r_max=10.0, r_disk_inner=r_in, r_disk_outer=r_out, disk_tilt=wl["disk_tilt"])
    r = HipRenderer(W, H, np.zeros((1024, 2048, 3), dtype=np.float32), np.zeros((n_r, n_phi, 4), dtype=np.float32),
                    anti_alias=wl["anti_alias"], device_index=device_index, rows=rows, frame_slots=1, **kw,
                    **({} if math is None else {"math": math}))
    r.build_procedural_skybox(seed=42, n_stars=n_stars)
    factories = init_lifecycle_system(r, n_r, n_phi, seed=42)
    advance_lifecycle_frame(r, factories, t=0.0, dt=0.0, recompute_stats=True)
    return r, (n_r, n_phi)


def make_tiles(wl: dict, devices, n_stars: int = 6000, math=None, balance: bool = True):
    """Row-block renderers for ONE frame of workload ``wl`` (BASELINE.json configs[3]): block k on HIP device
    ``devices[k]``, every device with its own copy of the deterministic scene, rows cut by the cost profile of a
    probe frame (multigpu.balanced_row_blocks).  Returns (tiles, blocks, note); render with multigpu.group_render."""
    n = len(devices)
    blocks = plan_blocks(wl, n, devices[0], balance, math=math)
    tiles, dims = [], None
    for dev, rows in zip(devices, blocks):
        r, dims = make_tile(wl, rows, dev, n_stars=n_stars, math=math)
        tiles.append(r)
    note = (f"{n} row blocks {blocks} cut by cost, one context per device {list(devices)}, scene generated on every device "
            f"(skybox 1024x2048 seed 42, lifecycle disk texture {dims[0]}x{dims[1]} seed 42 at t=0)")
    return tiles, blocks, note
