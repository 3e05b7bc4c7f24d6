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


def aligned_row_blocks(height: int, n: int, quantum: int = 8) -> List[Tuple[int, int]]:
    """The even cut with its cuts on multiples of ``quantum`` rows (the march's 8x8 tiles: a block whose first row is not a
    multiple of 8 tiles its rows differently from the whole frame, and the hybrid arithmetic chooses strict / fast per
    tile); plain row_blocks where the frame is too small for that."""
    if n * quantum > height:
        return row_blocks(height, n)
    return balanced_row_blocks(height, n, np.ones((height + quantum - 1) // quantum), quantum, quantum)


def balanced_row_blocks(height: int, n: int, band_costs: Sequence[float], band_rows: int, quantum: int = 8,
                        fixed_cost_per_row: float = 0.0) -> List[Tuple[int, int]]:
    """n contiguous row blocks of about equal COST.  ``band_costs[k]`` is the cost (ray-steps) of rows
    [k * band_rows, (k + 1) * band_rows) of the image; cuts fall on multiples of ``quantum`` rows (the march works
    in 8x8 tiles) and every block gets at least one quantum.  ``fixed_cost_per_row`` adds a per-row constant
    (set-up and shading work that does not scale with the step count)."""
    if n < 1 or n * quantum > height:
        raise ValueError(f"cannot cut {height} rows into {n} blocks of at least {quantum} rows")
    costs = np.asarray(band_costs, dtype=np.float64)
    if costs.ndim != 1 or len(costs) * band_rows < height or (costs < 0).any():
        raise ValueError("band_costs must be non-negative and cover the image")
    per_row = np.repeat(costs / band_rows, band_rows)[:height] + fixed_cost_per_row
    if per_row.sum() <= 0:
        return row_blocks(height, n)
    cum = np.concatenate([[0.0], np.cumsum(per_row)])
    cuts = [0]
    for k in range(1, n):
        target = cum[-1] * k / n
        r = int(np.searchsorted(cum, target))
        r = int(round(r / quantum)) * quantum
        lo = cuts[-1] + quantum                                   # at least one quantum per block ...
        hi = height - (n - k) * quantum                           # ... including the blocks still to come
        cuts.append(min(max(r, lo), hi))
    cuts.append(height)
    return [(cuts[k], cuts[k + 1]) for k in range(n)]


STRICT_STEP_COST = 2.2      # cost of a ray-step of the strict kernel in units of the fast kernel's (207 / 98 VALU per wave-step)


def probe_row_costs(width: int, height: int, cam_pos, fov: float, scale: int = None, device_index: int = 0, math=None, **kw):
    """(band_costs, band_rows) of a width x height frame from a frame ``scale`` times smaller: the number of
    steps a ray takes depends on the camera, the step size and the escape radius, not on the textures, so the
    probe runs with placeholder textures (a fraction of a millisecond).  math="hybrid": the probe marches hybrid and
    the steps its strict tiles took (the rows through the photon ring) count STRICT_STEP_COST times.  scale: 8, or 2 for
    hybrid -- the strict set is chosen per 8x8 tile with a margin of one tile, so a coarse probe would see a thicker ring
    than the full frame has (a 4k probe of an 8k frame takes 1.5 ms)."""
    from .renderer import HipRenderer
    if scale is None:
        scale = 2 if math == "hybrid" else 8
    keep = {k: kw[k] for k in ("step_size", "r_max", "r_disk_inner", "r_disk_outer", "disk_tilt") if k in kw}
    pw, ph = max(8, width // scale), max(8, height // scale)
    probe = HipRenderer(pw, ph, np.zeros((8, 16, 3), np.float32), np.zeros((32, 64, 4), np.float32),
                        device_index=device_index, math="hybrid" if math == "hybrid" else "fast", **keep)
    try:
        if math == "hybrid":
            fast, strict = probe.row_costs(cam_pos, fov, split=True)
            costs = fast.astype(np.float64) + STRICT_STEP_COST * strict.astype(np.float64)
        else:
            costs = probe.row_costs(cam_pos, fov).astype(np.float64)
    finally:
        probe.close()
    band_rows = 8 * height / ph                                   # probe band of 8 rows -> rows of the full image
    # resample onto whole rows of the full image
    per_row = np.interp((np.arange(height) + 0.5) / band_rows, np.arange(len(costs)) + 0.5, costs / band_rows)
    return per_row, 1


def frames_of_rank(n_frames: int, rank: int, world: int) -> range:
    """Round-robin frame shard of BASELINE.json configs[4]."""
    return range(rank, n_frames, world)


def group_render(tiles: Sequence, cam_pos, fov: float, frame: int = 0, skip_differentials=False,
                 skip_bloom=False, lens_flare=False, gather: str = "host", schedule: str = "auto", live=None,
                 time_march: bool = False, wait: bool = True):
    """Render one frame with the row-block renderers ``tiles`` (HipRenderer objects whose rows tile
    the image in order).  gather="host": returns the (H, W, 3) float32 frame, assembled from per-device pinned
    buffers.  gather="peer": every tile's V pass stores its f32 rows straight into the frame buffer on tiles[0]'s device
    (peer memory over xGMI) -- returns None, read_gathered(tiles) fetches the frame.  gather="peer_u8": the same with the
    quantised rows (a quarter of the bytes; read_gathered_u8).  gather="none": the rows stay in their tiles.
    schedule: "pipelined" (the rows that need no halo go through the V pass first, under the neighbours' H passes,
    csrc/group.hip), "serial" (one V launch behind the wait; same bytes) or "auto" (default: pipelined where the
    tiles sit on distinct devices, serial where they share one).  live: per-tile 0/1 -- only those tiles render, the
    others keep the buffers of the last call in which they did (bench.py times one tile of eight that way).
    time_march: also record every tile's march-end event (counters()["march_ms"]; a ~5 us bubble in the tile's stream).
    wait=False (gather "peer" / "peer_u8" of frames whose rows the kernels store themselves): return once the frame is
    submitted -- the next one may follow at once, the tiles order themselves on the device (BHR_GROUP_ASYNC);
    group_sync(tiles), read_gathered*(tiles) or any waiting call ends the frames in flight."""
    if gather not in ("host", "peer", "peer_u8", "none"):
        raise ValueError(f"gather must be 'host', 'peer', 'peer_u8' or 'none', got {gather!r}")
    if schedule not in ("auto", "pipelined", "serial"):
        raise ValueError(f"schedule must be 'auto', 'pipelined' or 'serial', got {schedule!r}")
    first = tiles[0]
    lib = _lib.load()
    arr = (C.c_void_p * len(tiles))(*[t._ctx for t in tiles])
    cam = first.camera_uniforms(cam_pos, fov, frame)
    flags = first._flags(skip_differentials, skip_bloom)
    if lens_flare:
        flags |= _lib.LENS_FLARE
    flags |= {"auto": 0, "serial": _lib.GROUP_SERIAL, "pipelined": _lib.GROUP_PIPELINED}[schedule]
    flags |= {"host": 0, "none": 0, "peer": _lib.GATHER_PEER, "peer_u8": _lib.GATHER_U8}[gather]
    if time_march:
        flags |= _lib.GROUP_TIME_MARCH
    if not wait:
        flags |= _lib.GROUP_ASYNC
    live_arr = None
    if live is not None:
        if len(live) != len(tiles):
            raise ValueError(f"live must have one entry per tile ({len(tiles)}), got {len(live)}")
        live_arr = (C.c_int32 * len(tiles))(*[1 if v else 0 for v in live])
    if gather != "host":
        _lib.check(lib.bhr_group_render_subset(arr, len(tiles), C.byref(cam), flags, None, live_arr))
        return None
    out = np.empty((first.height, first.width, 3), dtype=np.float32)
    _lib.check(lib.bhr_group_render_subset(arr, len(tiles), C.byref(cam), flags, _lib.fptr(out), live_arr))
    return out


def group_sync(tiles: Sequence) -> None:
    """Wait for every group frame the tiles have in flight (group_render(..., wait=False))."""
    arr = (C.c_void_p * len(tiles))(*[t._ctx for t in tiles])
    _lib.check(_lib.load().bhr_group_sync(arr, len(tiles)))


def read_gathered(tiles: Sequence) -> np.ndarray:
    """The (H, W, 3) frame the last group_render(..., gather="peer") left on tiles[0]'s device."""
    first = tiles[0]
    out = np.empty((first.height, first.width, 3), dtype=np.float32)
    _lib.check(_lib.load().bhr_read_gathered(first._ctx, _lib.fptr(out)))
    return out


def read_gathered_u8(tiles: Sequence) -> np.ndarray:
    """The quantised (H, W, 3) uint8 frame the last group_render(..., gather="peer_u8") left on tiles[0]'s device."""
    first = tiles[0]
    out = np.empty((first.height, first.width, 3), dtype=np.uint8)
    _lib.check(_lib.load().bhr_read_gathered_u8(first._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


class TileLink:
    """One rank's end of the one-process-per-tile row-block path (include/bhr.h: bhr_tile_export / _connect / _render):
    every rank owns one HipRenderer whose rows are its block of the frame; the neighbours' halo rows and rank 0's frame
    buffers are reached through HIP IPC memory handles, the ranks pace each other through counters in host shared memory.

    ``exchange(payload: bytes) -> list[bytes]`` is any all-gather over the ranks (torch.distributed.all_gather_object,
    files ...); ``shm_name`` names the shared-memory block (rank 0 creates it)."""

    def __init__(self, renderer, rank: int, world: int, exchange, shm_name: str, gather: str = "peer_u8"):
        from multiprocessing import shared_memory
        if gather not in ("peer", "peer_u8", "none"):
            raise ValueError(f"gather must be 'peer', 'peer_u8' or 'none', got {gather!r}")
        self.renderer, self.rank, self.world, self.gather = renderer, rank, world, gather
        self._lib = _lib.load()
        self._gflag = {"peer": _lib.GATHER_PEER, "peer_u8": _lib.GATHER_U8, "none": 0}[gather]
        nbytes = world * _lib.TILE_SHM_WORDS * 8
        self._shm = None
        if rank == 0:
            try:
                shared_memory.SharedMemory(name=shm_name).unlink()          # a crashed earlier run
            except FileNotFoundError:
                pass
            self._shm = shared_memory.SharedMemory(name=shm_name, create=True, size=nbytes)
            self._shm.buf[:nbytes] = bytes(nbytes)
        mine = _lib.TileHandles()
        _lib.check(self._lib.bhr_tile_export(renderer._ctx, self._gflag, C.byref(mine)))
        everyone = exchange(bytes(mine))                                     # also the barrier behind rank 0's create
        if len(everyone) != world:
            raise ValueError(f"exchange returned {len(everyone)} records for {world} ranks")
        if self._shm is None:
            self._shm = shared_memory.SharedMemory(name=shm_name)
            # Python < 3.13 registers ATTACHED segments with this process's resource tracker too, which then tries to unlink
            # rank 0's segment at exit and warns when it is already gone: the segment is rank 0's to remove
            try:
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        arr = (_lib.TileHandles * world)(*[_lib.TileHandles.from_buffer_copy(b) for b in everyone])
        self._words = (C.c_uint64 * (world * _lib.TILE_SHM_WORDS)).from_buffer(self._shm.buf)
        _lib.check(self._lib.bhr_tile_connect(renderer._ctx, rank, world, arr, self._words))

    def render(self, cam_pos, fov: float, frame: int = 0, skip_bloom: bool = False, time_march: bool = False) -> None:
        """One frame: this rank's tile, pipelined; returns when every rank's rows have landed on rank 0's device."""
        r = self.renderer
        cam = r.camera_uniforms(cam_pos, fov, frame)
        _lib.check(self._lib.bhr_tile_render(r._ctx, C.byref(cam), r._flags(False, skip_bloom) | self._gflag
                                             | (_lib.GROUP_TIME_MARCH if time_march else 0)))

    def read_gathered(self) -> np.ndarray:
        return (read_gathered if self.gather == "peer" else read_gathered_u8)([self.renderer])

    def close(self) -> None:
        if self._shm is not None:
            del self._words                                                  # releases the exported buffer view
            self._shm.close()
            if self.rank == 0:
                try:
                    self._shm.unlink()
                except FileNotFoundError:
                    pass
            self._shm = None


def file_exchange(directory: str, rank: int, world: int, tag: str = "h", timeout: float = 60.0):
    """An all-gather through files in ``directory`` (tests, launchers without torch.distributed)."""
    import time

    def exchange(payload: bytes):
        tmp = os.path.join(directory, f".{tag}{rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, os.path.join(directory, f"{tag}{rank}.bin"))
        out, t0 = [], time.time()
        for k in range(world):
            path = os.path.join(directory, f"{tag}{k}.bin")
            while not os.path.isfile(path):
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"rank {rank}: no record from rank {k} after {timeout:.0f} s")
                time.sleep(0.005)
            with open(path, "rb") as f:
                out.append(f.read())
        return out
    return exchange


def render_image_tiled(width, height, cam_pos, fov, gpus, lens_flare=False, devices=None, balance=True, **kw) -> np.ndarray:
    """render_image over ``gpus`` row blocks.  ``devices`` maps block k to a HIP device ordinal
    (default k); every device builds the same deterministic scene.  ``balance``: cut the rows by the cost
    profile of a probe frame (rows through the shadow and the photon ring take more steps) instead of evenly."""
    from .drivers import make_renderer, init_lifecycle_system, advance_lifecycle_frame, use_analytic_disk
    disk_model = kw.pop("disk_model", "texture")
    math = kw.pop("math", None)
    if devices is None and os.environ.get("BHR_TILE_DEVICES"):     # e.g. "0,0": rehearse two tiles on one card
        devices = [int(d) for d in os.environ["BHR_TILE_DEVICES"].split(",")]
    devices = list(range(gpus)) if devices is None else list(devices)
    if len(devices) != gpus:
        raise ValueError(f"{gpus} row blocks need {gpus} device ordinals, got {devices}")
    blocks = aligned_row_blocks(height, gpus)
    if balance and gpus > 1 and height >= 64 * gpus:
        per_row, band_rows = probe_row_costs(width, height, cam_pos, fov, device_index=devices[0], math=math, **kw)
        blocks = balanced_row_blocks(height, gpus, per_row, band_rows, fixed_cost_per_row=0.1 * float(per_row.mean()))
    tiles = []
    for k, rows in enumerate(blocks):
        r, use_lifecycle, n_r, n_phi = make_renderer(width, height, cam_pos, fov, device_index=devices[k],
                                                     rows=rows, lens_flare=False, math=math, **kw)
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
