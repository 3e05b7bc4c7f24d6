"""Multi-GPU partitioning of the path.

* still image : contiguous row blocks, one context per device, driven from ONE process through
  bhr_group_render (halo exchange + gather with hipMemcpyPeerAsync over xGMI, no collective);
* video       : independent frames, frame f on rank f % world (one process per GPU, drivers.render_video).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib


def row_blocks(height: int, n: int) -> List[Tuple[int, int]]:
    """n contiguous row blocks covering [0, height); sizes differ by at most one row, larger first."""
    if n < 1 or n > height:
        raise ValueError(f"cannot cut {height} rows into {n} blocks")
    base, extra = divmod(height, n)
    cuts, r = [], 0
    for k in range(n):
        size = base + (1 if k < extra else 0)
        cuts.append((r, r + size))
        r += size
    return cuts


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    """Round-robin frame shard of BASELINE.json configs[4]."""
    return range(rank, n_frames, world)


def group_render(tiles: Sequence, cam_pos, fov: float, frame: int = 0, skip_differentials=False,
                 skip_bloom=False, lens_flare=False) -> np.ndarray:
    """Render one frame with the row-block renderers ``tiles`` (HipRenderer objects whose rows tile
    the image in order).  Returns the gathered (H, W, 3) float32 frame."""
    first = tiles[0]
    lib = _lib.load()
    arr = (C.c_void_p * len(tiles))(*[t._ctx for t in tiles])
    out = np.empty((first.height, first.width, 3), dtype=np.float32)
    cam = first.camera_uniforms(cam_pos, fov, frame)
    flags = first._flags(skip_differentials, skip_bloom)
    if lens_flare:
        flags |= _lib.LENS_FLARE
    _lib.check(lib.bhr_group_render(arr, len(tiles), C.byref(cam), flags, _lib.fptr(out)))
    return out


def render_image_tiled(width, height, cam_pos, fov, gpus, lens_flare=False, devices=None, **kw) -> np.ndarray:
    """render_image over ``gpus`` row blocks.  ``devices`` maps block k to a HIP device ordinal
    (default k); every device builds the same deterministic scene."""
    from .drivers import make_renderer, init_lifecycle_system, advance_lifecycle_frame, use_analytic_disk
    disk_model = kw.pop("disk_model", "texture")
    if devices is None and os.environ.get("BHR_TILE_DEVICES"):     # e.g. "0,0": rehearse two tiles on one card
        devices = [int(d) for d in os.environ["BHR_TILE_DEVICES"].split(",")]
    devices = list(range(gpus)) if devices is None else list(devices)
    if len(devices) != gpus:
        raise ValueError(f"{gpus} row blocks need {gpus} device ordinals, got {devices}")
    tiles = []
    for k, rows in enumerate(row_blocks(height, gpus)):
        r, use_lifecycle, n_r, n_phi = make_renderer(width, height, cam_pos, fov, device_index=devices[k],
                                                     rows=rows, lens_flare=False, **kw)
        if use_analytic_disk(r, disk_model):
            pass
        elif use_lifecycle:
            factories = init_lifecycle_system(r, n_r, n_phi, seed=42)
            advance_lifecycle_frame(r, factories, t=0.0, dt=0.0, recompute_stats=True)
        tiles.append(r)
    img = group_render(tiles, cam_pos, fov, lens_flare=lens_flare)
    for t in tiles:
        t.close()
    return img
