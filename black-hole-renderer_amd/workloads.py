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


def plan_blocks(wl: dict, n: int, device_index: int = 0, balance: bool = True, math=None, refine: int = 1, return_costs: bool = False):
    """The n row blocks of workload ``wl``: cut by the cost profile of a probe frame rendered on ``device_index``
    (multigpu.balanced_row_blocks), then refined by ``refine`` rounds of timing every block's march on that device with
    placeholder textures (make_tiles adds a round on the real scene).  The timings make the cut run-dependent: ranks of a
    one-process-per-tile run take rank 0's (bench.tile_leg_per_rank)."""
    from . import multigpu
    W, H = wl["width"], wl["height"]
    kw = dict(step_size=wl["step_size"], r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=wl["disk_tilt"])
    blocks = multigpu.aligned_row_blocks(H, n)
    per_row = None
    if balance and n > 1 and H >= 64 * n:
        per_row, band_rows = multigpu.probe_row_costs(W, H, wl["cam_pos"], wl["fov"], device_index=device_index, math=math, **kw)
        per_row = per_row + 0.1 * float(per_row.mean())              # per-row constant: set-up and shading work that does not scale with the steps
        blocks = multigpu.balanced_row_blocks(H, n, per_row, band_rows)
        # The probe counts ray-steps; what has to be equal is TIME -- a block's ragged end, the strict share of a hybrid frame
        # and the waves' lifetimes do not scale with its steps (measured on the 8k frame in 8 blocks: blocks of equal cost took
        # 1.14 ... 1.28 ms of march, 12 % apart).  Every block's march is timed alone on this device, the cost density of its
        # rows scaled by measured / predicted, and the frame cut again.
        for _ in range(refine):
            ms = measure_block_marches(wl, blocks, device_index, math)
            blocks, per_row = rebalance(H, n, blocks, per_row, ms)
    return (blocks, per_row) if return_costs else blocks


def rebalance(H: int, n: int, blocks, per_row, ms):
    """Blocks cut again after their marches took ``ms``: the cost density of every block's rows is scaled by its share of the
    time over its share of the predicted cost (clipped: one round never moves a cut by more than a quarter)."""
    from . import multigpu
    ms = np.asarray(ms, dtype=np.float64)
    if ms.min() < 0.15:                                           # small frames: a launch's fixed cost, not the rays, is what the timer sees
        return blocks, per_row
    per_row = per_row.copy()
    cost = np.array([per_row[a:b].sum() for a, b in blocks])
    scale = np.clip((ms / ms.sum()) / (cost / cost.sum()), 0.8, 1.25)
    for (a, b), f in zip(blocks, scale):
        per_row[a:b] *= f
    return multigpu.balanced_row_blocks(H, n, per_row, 1), per_row


def measure_block_marches(wl: dict, blocks, device_index: int = 0, math=None, reps: int = 7) -> np.ndarray:
    """March time (ms) of every row block of ``wl`` rendered alone on ``device_index`` with placeholder textures."""
    kw = dict(step_size=wl["step_size"], r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=wl["disk_tilt"])
    out = []
    for rows in blocks:
        r = HipRenderer(wl["width"], wl["height"], np.zeros((8, 16, 3), np.float32), np.zeros((32, 64, 4), np.float32),
                        anti_alias=wl["anti_alias"], device_index=device_index, rows=rows, frame_slots=1, **kw,
                        **({} if math is None else {"math": math}))
        try:
            import time
            t0 = time.perf_counter()                              # the context's set-up idled the chip: ~30 ms of load bring the clocks back
            while time.perf_counter() - t0 < 0.03:
                for _ in range(4):
                    r.render_async(wl["cam_pos"], wl["fov"], skip_bloom=True)
                r.sync()
            ms = []
            for _ in range(reps):
                r.render_async(wl["cam_pos"], wl["fov"], skip_bloom=True)
                ms.append(r.counters()["march_ms"])
            out.append(float(np.median(ms)))
        finally:
            r.close()
    return np.array(out)


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


def make_tiles(wl: dict, devices, n_stars: int = 6000, math=None, balance: bool = True, refine_on_scene: bool = True):
    """Row-block renderers for ONE frame of workload ``wl`` (BASELINE.json configs[3]): block k on HIP device
    ``devices[k]``, every device with its own copy of the deterministic scene, rows cut by the cost profile of a
    probe frame (multigpu.balanced_row_blocks).  Returns (tiles, blocks, note); render with multigpu.group_render."""
    n = len(devices)
    blocks, per_row = plan_blocks(wl, n, devices[0], balance, math=math, return_costs=True)

    def build(blocks):
        tiles, dims = [], None
        for dev, rows in zip(devices, blocks):
            r, dims = make_tile(wl, rows, dev, n_stars=n_stars, math=math)
            tiles.append(r)
        return tiles, dims

    tiles, dims = build(blocks)
    if per_row is not None and refine_on_scene:
        # one more round on the REAL scene, timing the block's whole frame -- march AND post-pass: the march of the rows that see
        # the disk's near side magnified gathers from a far larger part of the 300 MB texture than a placeholder has (the centre
        # block of the 8k frame: 8 % slower than the placeholder timing said), and the post-pass costs per ROW, whatever the
        # rays do (the sky-only bottom block of the 8k frame, cut to equal march time, had 608 rows against 520 and was the
        # slowest tile by 6 %).  Worth rebuilding the tiles for when a block is more than 3 % off the mean.
        import time
        ms = []
        for t in tiles:
            t.set_outputs("u8")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.03:
                for _ in range(4):
                    t.render_async(wl["cam_pos"], wl["fov"])
                t.sync()
            one = []
            for _ in range(7):
                t.render_async(wl["cam_pos"], wl["fov"])
                one.append(t.counters()["frame_ms"])
            ms.append(float(np.median(one)))
            t.set_outputs("f32")                                    # the library's default
        ms = np.array(ms)
        if ms.min() >= 0.15 and ms.max() > 1.03 * ms.mean():
            blocks2, _ = rebalance(wl["height"], n, blocks, per_row, ms)
            if blocks2 != blocks:
                for t in tiles:
                    t.close()
                blocks = blocks2
                tiles, dims = build(blocks)
    note = (f"{n} row blocks {blocks} cut by cost, one context per device {list(devices)}, scene generated on every device "
            f"(skybox 1024x2048 seed 42, lifecycle disk texture {dims[0]}x{dims[1]} seed 42 at t=0)")
    return tiles, blocks, note
