"""HipRenderer: the reference's ``TaichiRenderer`` surface (render.py:2189-4028) on libbhr_hip.so.

Same constructor arguments, methods, attribute names and error behaviour, so the drivers and
tests that talk to ``TaichiRenderer`` read the same against this class.  All device work goes
through the C ABI in include/bhr.h; nothing here computes pixels on the host except the
lens flare, which the reference also does in NumPy (render.py:3925-4028).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .camera import build_camera
from .textures import compute_edge_alpha, keplerian_omega_rows

R_DISK_INNER_DEFAULT = 2.0   # render.py:433
R_DISK_OUTER_DEFAULT = 15.0  # render.py:434
DISK_COLOR_TEMPERATURE = 6000  # render.py:52


class _FieldView:
    """Stand-in for a Taichi field that tests read with ``.to_numpy()``."""

    def __init__(self, reader, shape):
        self._reader = reader
        self.shape = tuple(shape)

    def to_numpy(self) -> np.ndarray:
        return self._reader()


class HipRenderer:
    """Renders Schwarzschild black-hole frames on one MI355X (or one row block of a frame).

    Usage (render.py:2193-2196)::

        renderer = HipRenderer(width, height, skybox, disk_tex, ...)
        img1 = renderer.render(cam_pos=[6, 0, 0.5], fov=90)
    """

    def __init__(self, width, height, skybox, disk_tex,
                 step_size=0.1, r_max=10.0, device="hip",
                 r_disk_inner=R_DISK_INNER_DEFAULT, r_disk_outer=R_DISK_OUTER_DEFAULT,
                 disk_tilt=0.0, lens_flare=False, anti_alias="disabled", aa_strength=1.0,
                 disk_rotation_speed=0.1, ignore_taichi_cache=False,
                 device_index: int = 0, rows: Optional[Sequence[int]] = None, math: str = "strict",
                 frame_slots: Optional[int] = None, outputs: Optional[str] = None, options: Optional[dict] = None):
        if device not in ("hip", "gpu"):
            raise ValueError(f"HipRenderer runs on the GPU only (device={device!r}); there is no CPU path")
        # math="strict" (default): the RK4 loop in the reference's operation order with IEEE sqrt and
        # divide -- ray paths bit-identical to a strict f32 evaluation of render.py:2854-3006.
        # math="fast": 2-D orbital-plane state, v_rsq/v_rcp, fast-math (Taichi's fast_math=True
        # analogue); ~3x faster, deviates from strict by f32 rounding noise only.
        # math="hybrid": the strict kernel on the 8x8 tiles whose rays pass near the photon sphere (impact parameter
        # within a band around 3 sqrt(3)/2 r_s -- the only rays that amplify rounding), the fast kernel on all others.
        if math not in ("fast", "strict", "hybrid"):
            raise ValueError(f"math must be 'fast', 'strict' or 'hybrid', got {math!r}")
        self.math = math
        if anti_alias not in ("disabled", "lod_radius"):
            raise ValueError(f"anti_alias must be 'disabled' or 'lod_radius', got {anti_alias!r}")
        self.width, self.height = int(width), int(height)
        self.step_size, self.r_max = step_size, r_max
        self.r_disk_inner, self.r_disk_outer = r_disk_inner, r_disk_outer
        self.disk_tilt = disk_tilt
        self.lens_flare = lens_flare
        self.anti_alias, self.aa_strength = anti_alias, aa_strength
        self.disk_rotation_speed = disk_rotation_speed
        self.device_index = int(device_index)
        self.row0, self.row1 = (0, self.height) if rows is None else (int(rows[0]), int(rows[1]))

        self._lib = _lib.load()
        cfg = _lib.Config(self.width, self.height, self.row0, self.row1, float(step_size), float(r_max),
                          float(r_disk_inner), float(r_disk_outer), float(disk_tilt),
                          0 if anti_alias == "disabled" else 1, float(aa_strength), float(disk_rotation_speed),
                          self.device_index, {"strict": _lib.MATH_STRICT, "fast": _lib.MATH_FAST, "hybrid": _lib.MATH_HYBRID}[math])
        handle = C.c_void_p()
        # frame_slots: 2 (library default) = successive render_async calls alternate between two frame slots /
        # streams and overlap; 1 = one frame at a time on the context's stream (isolated kernel timing).  The
        # library reads BHR_FRAME_SLOTS when the context is created.
        if frame_slots not in (None, 1, 2):
            raise ValueError(f"frame_slots must be 1 or 2, got {frame_slots!r}")
        saved = os.environ.get("BHR_FRAME_SLOTS")
        if frame_slots is not None:
            os.environ["BHR_FRAME_SLOTS"] = str(frame_slots)
        try:
            _lib.check(self._lib.bhr_create(C.byref(cfg), C.byref(handle)))
        finally:
            if frame_slots is not None:
                if saved is None:
                    os.environ.pop("BHR_FRAME_SLOTS", None)
                else:
                    os.environ["BHR_FRAME_SLOTS"] = saved
        self._ctx = handle
        self.frame_slots = frame_slots if frame_slots is not None else (1 if saved == "1" else 2)
        # outputs: what a frame keeps in memory besides the bg / disk layers -- "f32" (default: what render() returns),
        # "u8" (the quantised rows only: video loop, PNG sink, u8 gather), or several joined by "+" ("f32+blur+u8");
        # anything not kept is produced on demand by the call that reads it.  options: {name: value} for bhr_set_option.
        if outputs is not None:
            self.set_outputs(outputs)
        for name, value in (options or {}).items():
            self.set_option(name, value)

        skybox = np.ascontiguousarray(skybox, dtype=np.float32)
        disk_tex = np.ascontiguousarray(disk_tex, dtype=np.float32)
        assert skybox.ndim == 3 and skybox.shape[2] == 3, "skybox must be (tex_h, tex_w, 3)"
        assert disk_tex.ndim == 3 and disk_tex.shape[2] == 4, "disk_tex must be (n_r, n_phi, 4)"
        self.tex_h, self.tex_w = skybox.shape[:2]
        self.dtex_h, self.dtex_w = disk_tex.shape[:2]
        _lib.check(self._lib.bhr_set_skybox(self._ctx, _lib.fptr(skybox), self.tex_h, self.tex_w))
        _lib.check(self._lib.bhr_set_disk_texture(self._ctx, _lib.fptr(disk_tex), self.dtex_h, self.dtex_w))
        self.num_mip_levels = int(self._lib.bhr_num_mip_levels(self._ctx))

        self.disk_texture_field = _FieldView(self._read_disk_texture, (self.dtex_h, self.dtex_w))
        self.disk_mips_field = _FieldView(self._read_mips_padded, (self.num_mip_levels, self.dtex_h, self.dtex_w))
        self._bg_ready = False
        self._parametric_gpu_ready = False

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        for ref in getattr(self, "_sinks", []):      # frame sinks hold this context: drain and free them first
            sink = ref()
            if sink is not None:
                sink.close()
        self._sinks = []
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx:
            self._lib.bhr_destroy(ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self) -> int:
        return self.row1 - self.row0

    # ------------------------------------------------------------------ textures
    def add_skybox_glow(self) -> None:
        """Milky-Way glow + clip of generate_skybox (render.py:296-341) on the device, in place; the sky given to the
        constructor must be generate_skybox(..., glow=False)."""
        _lib.check(self._lib.bhr_skybox_add_glow(self._ctx))

    def build_procedural_skybox(self, seed: int = 42, n_stars: int = 6000) -> None:
        """generate_skybox(tex_w, tex_h, seed, n_stars) (render.py:153-341) into this renderer's skybox, on the
        device: the host draws the random tables (skybox.sky_tables), bhr_skybox_build rasterises them bit-identically
        to NumPy / Pillow, bhr_skybox_add_glow adds the Milky-Way glow and clips.  The skybox given to the constructor
        only fixes the size."""
        from .skybox import STAR_PATCH_R, pillow_bilinear_coeffs, sky_tables
        t = sky_tables(self.tex_w, self.tex_h, seed, n_stars)
        ch, cw = t["coarse_u8"].shape[:2]
        kh, bh = pillow_bilinear_coeffs(cw, self.tex_w)
        kv, bv = pillow_bilinear_coeffs(ch, self.tex_h)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))   # noqa: E731
        coarse = np.ascontiguousarray(t["coarse_u8"])
        _lib.check(self._lib.bhr_skybox_build(
            self._ctx, self.tex_h, self.tex_w, coarse.ctypes.data_as(C.POINTER(C.c_uint8)), ch, cw, i32(kh), i32(bh), kh.shape[1],
            i32(kv), i32(bv), kv.shape[1], len(t["cx"]), _lib.fptr(t["cx"]), _lib.fptr(t["cy"]), _lib.fptr(t["colors"]),
            _lib.fptr(t["vals"]), STAR_PATCH_R))
        self.add_skybox_glow()

    def read_skybox(self) -> np.ndarray:
        out = np.empty((self.tex_h, self.tex_w, 3), dtype=np.float32)
        _lib.check(self._lib.bhr_get_skybox(self._ctx, _lib.fptr(out)))
        return out

    def update_disk_texture(self, new_disk_tex: np.ndarray) -> None:
        """Replace the disk texture and rebuild its mip chain (render.py:2292-2312)."""
        dtex_h, dtex_w = new_disk_tex.shape[:2]
        assert dtex_h == self.dtex_h and dtex_w == self.dtex_w, \
            f"Texture size mismatch: expected {self.dtex_h}x{self.dtex_w}, got {dtex_h}x{dtex_w}"
        tex = np.ascontiguousarray(new_disk_tex, dtype=np.float32)
        _lib.check(self._lib.bhr_set_disk_texture(self._ctx, _lib.fptr(tex), dtex_h, dtex_w))

    def _read_disk_texture(self) -> np.ndarray:
        out = np.empty((self.dtex_h, self.dtex_w, 4), dtype=np.float32)
        _lib.check(self._lib.bhr_get_disk_texture(self._ctx, _lib.fptr(out)))
        return out

    def read_mip_level(self, level: int) -> np.ndarray:
        h, w = self.dtex_h, self.dtex_w
        for _ in range(level):
            h, w = h // 2, w // 2
        out = np.empty((h, w, 4), dtype=np.float32)
        _lib.check(self._lib.bhr_get_disk_mip(self._ctx, level, _lib.fptr(out)))
        return out

    def _read_mips_padded(self) -> np.ndarray:
        """Same padded (levels, n_r, n_phi, 4) array disk_mips_field.to_numpy() yields (render.py:2244-2251)."""
        out = np.zeros((self.num_mip_levels, self.dtex_h, self.dtex_w, 4), dtype=np.float32)
        for lev in range(self.num_mip_levels):
            m = self.read_mip_level(lev)
            out[lev, :m.shape[0], :m.shape[1]] = m
        return out

    # ------------------------------------------------------------------ parametric texture state
    def upload_parametric_state(self, state) -> None:
        """Upload a 13-component rotating-texture state and its normalisation statistics
        (render.py:2314-2387).  ``state`` is any object with the DiskTextureRotatingState fields."""
        n_r, n_phi = state.n_r, state.n_phi
        packed = np.stack([
            state.temp_base, state.spiral, state.spiral_temp, state.turbulence, state.turb_temp,
            state.arcs, state.arcs_temp, state.rt_spikes, state.rt_temp, state.hotspot,
            state.hotspot_temp, state.az_hotspot, state.disturb_mod], axis=0).astype(np.float32)
        edge = np.ascontiguousarray(state.edge, dtype=np.float32)
        omega = np.ascontiguousarray(state.omega_rows, dtype=np.float32)
        _lib.check(self._lib.bhr_bg_init(self._ctx, n_r, n_phi, 0, 0.0, _lib.fptr(edge), _lib.fptr(omega)))
        _lib.check(self._lib.bhr_set_comp(self._ctx, _lib.fptr(np.ascontiguousarray(packed))))
        self._bg_n_r, self._bg_n_phi = n_r, n_phi
        self._install_field_views(n_r, n_phi, edge, omega)

        rt_weight = 0.20 if state.enable_rt else 0.0
        density = (0.15 + 0.10 * state.spiral + 0.30 * state.turbulence + 0.20 * state.hotspot
                   + 0.30 * state.arcs + rt_weight * state.rt_spikes) * state.disturb_mod
        density *= state.edge[:, None]
        density_p98 = float(np.percentile(density, 98))
        temp_struct = (state.spiral_temp + state.turb_temp + state.arcs_temp + state.rt_temp
                       + state.hotspot_temp) * state.disturb_mod
        pos = temp_struct > 0
        struct_scale = float(np.percentile(temp_struct[pos], 95)) if np.any(pos) else 1.0
        scaled = np.clip(temp_struct / (struct_scale + 1e-6) * 0.8, 0, 1.2)
        row_stats = np.stack([np.max(scaled, axis=1).astype(np.float32),
                              np.quantile(scaled, 0.7, axis=1).astype(np.float32)], axis=1).astype(np.float32)
        self._set_stats(density_p98, struct_scale, row_stats)
        self._param_enable_rt = 1 if state.enable_rt else 0
        self._param_color_temp = float(state.color_temp)
        self._parametric_gpu_ready = True

    def update_disk_texture_gpu(self, t_offset: float) -> None:
        """Compose the rotated texture and its mips on the device (render.py:3792-3817)."""
        assert self._parametric_gpu_ready, \
            "Must call upload_parametric_state() before update_disk_texture_gpu()"
        _lib.check(self._lib.bhr_compose_texture(self._ctx, float(t_offset), self._param_enable_rt,
                                                 self._param_color_temp))

    # ------------------------------------------------------------------ lifecycle texture pipeline
    def init_background_layer(self, n_r: int, n_phi: int, seed: int = 42) -> None:
        """render.py:3491-3547: draws az_freq / az_shear (in that order), uploads edge, omega rows and
        the permissive initial statistics."""
        rng = np.random.default_rng(seed)
        self._bg_az_freq = int(rng.integers(2, 5))
        self._bg_az_shear = float(rng.uniform(2.0, 4.0))
        edge = compute_edge_alpha(n_r).astype(np.float32)
        omega_rows = keplerian_omega_rows(n_r, self.r_disk_inner, self.r_disk_outer)
        _lib.check(self._lib.bhr_bg_init(self._ctx, n_r, n_phi, self._bg_az_freq, self._bg_az_shear,
                                         _lib.fptr(edge), _lib.fptr(omega_rows)))
        self._bg_omega_all_np = omega_rows
        self._bg_r_norm_all = np.linspace(0, 1, n_r)
        self._profile_pool = None
        self._bg_n_r, self._bg_n_phi = n_r, n_phi
        self._install_field_views(n_r, n_phi, edge, omega_rows)
        tb_init = np.clip(1.0 - np.linspace(0, 1, n_r), 0, 1) ** 1.3 * 0.25
        self._stats_np = np.array([0.5, 0.5], dtype=np.float32)
        self._row_stats_np = np.column_stack([np.maximum(tb_init, 0.25).astype(np.float32),
                                              np.maximum(tb_init * 0.8, 0.10).astype(np.float32)])
        self._param_enable_rt = 1
        self._param_color_temp = float(DISK_COLOR_TEMPERATURE)
        self._bg_ready = True

    def _install_field_views(self, n_r, n_phi, edge, omega):
        self._edge_np, self._omega_np = edge, omega
        self._comp_field = _FieldView(self.read_comp, (13, n_r, n_phi))
        self._edge_field = _FieldView(lambda: self._edge_np.copy(), (n_r,))
        self._omega_rows_field = _FieldView(lambda: self._omega_np.copy(), (n_r,))
        self._param_stats_field = _FieldView(lambda: self._stats_np.copy(), (2,))
        self._param_row_stats_field = _FieldView(lambda: self._row_stats_np.copy(), (n_r,))

    def _set_stats(self, density_p98: float, struct_scale: float, row_stats: np.ndarray) -> None:
        self._stats_np = np.array([density_p98, struct_scale], dtype=np.float32)
        self._row_stats_np = np.ascontiguousarray(row_stats, dtype=np.float32)
        _lib.check(self._lib.bhr_set_compose_stats(self._ctx, float(self._stats_np[0]), float(self._stats_np[1]),
                                                   _lib.fptr(self._row_stats_np)))

    def generate_background(self, t: float) -> None:
        """Background components comp[0,1,2,3,4,11,12] at time t on the device (render.py:3549-3562)."""
        assert self._bg_ready, "Must call init_background_layer() first"
        _lib.check(self._lib.bhr_generate_background(self._ctx, float(t)))

    def read_comp(self) -> np.ndarray:
        out = np.empty((13, self._bg_n_r, self._bg_n_phi), dtype=np.float32)
        _lib.check(self._lib.bhr_read_comp(self._ctx, _lib.fptr(out)))
        return out

    def accumulate_entity_layer(self, factories: dict, now: float, pairs_on_host: bool = False) -> None:
        """Rasterise the alive entities into comp[5:11] on the device (render.py:3564-3653).  The host hands over
        entity records (rebuilt only when a population changes); the library evaluates the fades and the per-(entity,
        row) scalars, csrc/lifecycle.hip does the per-texel work, nothing waits for the stream.
        ``pairs_on_host=True`` builds the (entity, row) pair tables in NumPy instead (round-1 path, same result)."""
        from . import lifecycle_device as ld
        if getattr(self, "_profile_pool", None) is None:
            self._profile_pool = ld.ProfilePool(self._lib, self._ctx, self._bg_n_phi)
            self._population_tables = ld.PopulationTables()
        if not pairs_on_host and ld.accumulate_population(self._lib, self._ctx, self._profile_pool, self._population_tables,
                                                          factories, now, self._bg_n_r, self._bg_n_phi,
                                                          self._bg_omega_all_np, self._bg_r_norm_all):
            return
        ld.accumulate_on_device(self._lib, self._ctx, self._profile_pool, factories, now, self._bg_n_r,
                                self._bg_n_phi, self._bg_omega_all_np, self._bg_r_norm_all)

    def recompute_interactive_stats(self) -> None:
        """Normalisation statistics from the current components, selected on the device (render.py:3655-3712)."""
        if self._bg_n_phi > 32768:
            raise ValueError(f"device statistics support textures up to 32768 columns, got {self._bg_n_phi}")
        from . import lifecycle_device as ld
        p98, scale, row_stats = ld.stats_on_device(self._lib, self._ctx, self._bg_n_r, self._bg_n_phi,
                                                   self._param_enable_rt)
        self._set_stats(p98, scale, row_stats)

    def compose_interactive_texture(self, solo_idx: int = -1) -> None:
        """Compose comp -> RGBA texture + mips with t_offset = 0 (render.py:3714-3767)."""
        if solo_idx >= 0:
            pairs = {0: [], 1: [2], 2: [1], 3: [4], 4: [3], 5: [6], 6: [5], 7: [8], 8: [7], 9: [10], 10: [9],
                     11: [], 12: []}
            keep = {solo_idx} | set(pairs.get(solo_idx, []))
            for i in range(13):
                if i not in keep:
                    _lib.check(self._lib.bhr_fill_comp_slice(self._ctx, i, 1.0 if i == 12 else 0.0))
            self.recompute_interactive_stats()
        _lib.check(self._lib.bhr_compose_texture(self._ctx, 0.0, self._param_enable_rt, self._param_color_temp))

    def eval_noise(self, coords: np.ndarray, mode: str = "simplex", octaves: int = 4,
                   persistence: float = 0.5, lacunarity: float = 2.0) -> np.ndarray:
        """Simplex / FBM values at (N, 3) coordinates (render.py:3769-3790)."""
        c = np.ascontiguousarray(coords, dtype=np.float32)
        out = np.empty(c.shape[0], dtype=np.float32)
        _lib.check(self._lib.bhr_eval_noise(self._ctx, _lib.fptr(c), c.shape[0], 0 if mode == "simplex" else 1,
                                            int(octaves), float(persistence), float(lacunarity), _lib.fptr(out)))
        return out

    # ------------------------------------------------------------------ analytic disk source
    def use_disk_v2(self, params=None, structure_params=None, seed: int = 42, volume: bool = False,
                    absorption: float = 4.0, grazing_gain: float = 1.0, substeps: int = 2) -> None:
        """Shade the disk from the Disk V2 model inside the march kernel instead of the texture
        (include/bhr_disk_v2.h: bhr_set_disk_source).  ``volume=False``: the model's mid-plane fields at
        every plane crossing; ``volume=True``: the finite-thickness emission-absorption integral of
        docs/design_ad_v2.md 4.3 (absorption coefficient per unit density, grazing-angle gain, pieces per
        RK4 step).  Pass ``params=None`` to go back to the texture."""
        from . import disk_v2 as dv
        lib = self._lib
        lib.bhr_set_disk_source.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.bhr_set_disk_source.restype = C.c_int32
        lib.bhr_set_disk_volume_options.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int32]
        lib.bhr_set_disk_volume_options.restype = C.c_int32
        if params is None:
            _lib.check(lib.bhr_set_disk_source(self._ctx, 0, None, 0.0, 0.0, 0.0))
            self._dv2 = None
            return
        cp, m_sh, m_hs, t_peak = dv.reference_norms(params, structure_params, seed, ctx=self._ctx)
        if volume:
            _lib.check(lib.bhr_set_disk_volume_options(self._ctx, absorption, grazing_gain, substeps))
        _lib.check(lib.bhr_set_disk_source(self._ctx, 2 if volume else 1, C.byref(cp), m_sh, m_hs, t_peak))
        self._dv2 = (cp, m_sh, m_hs, t_peak)

    # ------------------------------------------------------------------ rendering
    def camera_uniforms(self, cam_pos, fov: float, frame: int = 0) -> _lib.Camera:
        """f64 camera -> the f32 uniforms of render.py:3880-3897."""
        eye, right, up, fwd, pw, ph = build_camera(np.array(cam_pos, dtype=np.float64), fov, self.width,
                                                   self.height)
        cam = _lib.Camera()
        cam.pos[:] = list(eye.astype(np.float32))
        cam.right[:] = list(right.astype(np.float32))
        cam.up[:] = list(up.astype(np.float32))
        cam.forward[:] = list(fwd.astype(np.float32))
        cam.pixel_width, cam.pixel_height = float(pw), float(ph)
        cam.r_escape = float(max(self.r_max, float(np.linalg.norm(eye)) * 2))
        cam.t_offset = float(frame) * self.disk_rotation_speed
        return cam

    @staticmethod
    def _flags(skip_differentials: bool, skip_bloom: bool, compaction: bool = False, math=None) -> int:
        return ((_lib.SKIP_DIFFERENTIALS if skip_differentials else 0) | (_lib.SKIP_BLOOM if skip_bloom else 0)
                | (_lib.PERSISTENT if compaction else 0)
                | {None: 0, "fast": _lib.FORCE_FAST, "strict": _lib.FORCE_STRICT, "hybrid": _lib.FORCE_HYBRID}[math])

    def render_async(self, cam_pos, fov: float, frame: int = 0, skip_differentials: bool = False,
                     skip_bloom: bool = False, compaction: bool = False, math=None, lens_flare=None) -> None:
        """Launch march + bloom + combine; the frame stays in HBM (counterpart of render_to_field,
        render.py:3819-3863, without the GUI flip)."""
        cam = self.camera_uniforms(cam_pos, fov, frame)
        flags = self._flags(skip_differentials, skip_bloom, compaction, math)
        if self.lens_flare if lens_flare is None else lens_flare:
            flags |= _lib.LENS_FLARE          # device twin of _apply_lens_flare (render.py:3920-4028)
        _lib.check(self._lib.bhr_render(self._ctx, C.byref(cam), flags))

    def sync(self) -> None:
        _lib.check(self._lib.bhr_sync(self._ctx))

    def set_outputs(self, outputs: str) -> None:
        """Layers the V pass of every later frame stores: "f32", "blur", "u8" joined by "+" (bhr_set_outputs)."""
        bits = {"f32": _lib.OUTPUT_F32, "blur": _lib.OUTPUT_BLUR, "u8": _lib.OUTPUT_U8}
        mask = 0
        for part in outputs.split("+"):
            if part not in bits:
                raise ValueError(f"outputs: 'f32', 'blur', 'u8' joined by '+', got {outputs!r}")
            mask |= bits[part]
        _lib.check(self._lib.bhr_set_outputs(self._ctx, mask))

    def set_option(self, name: str, value) -> None:
        """One of the library's switches for this context (bhr_set_option; include/bhr.h lists them)."""
        _lib.check(self._lib.bhr_set_option(self._ctx, name.encode(), float(value)))

    def read_layer(self, layer: int) -> np.ndarray:
        out = np.empty((self.rows, self.width, 3), dtype=np.float32)
        _lib.check(self._lib.bhr_read_layer(self._ctx, layer, _lib.fptr(out)))
        return out

    def write_layer(self, layer: int, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, dtype=np.float32)
        if data.shape != (self.rows, self.width, 3):
            raise ValueError(f"layer must be {(self.rows, self.width, 3)}, got {data.shape}")
        _lib.check(self._lib.bhr_write_layer(self._ctx, layer, _lib.fptr(data)))

    def bloom_only(self) -> None:
        """BLUR <- bloom(DISK), FINAL <- clip(BG + DISK + BLUR, 0, 1) on the layers in the context
        (_bloom_kernel + combine, render.py:3914-3918)."""
        _lib.check(self._lib.bhr_bloom(self._ctx))

    def apply_lens_flare(self) -> None:
        """FINAL <- clip(FINAL + flare(DISK), 0, 1) on the device (_apply_lens_flare, render.py:3925-4028)."""
        _lib.check(self._lib.bhr_lens_flare(self._ctx))

    def lens_flare_sums(self) -> np.ndarray:
        """(sum glow, sum x glow, sum y glow) of the disk layer, in NumPy's summation order."""
        out = (C.c_double * 3)()
        _lib.check(self._lib.bhr_lens_flare_sums(self._ctx, out))
        return np.array(out[:], dtype=np.float64)

    def mip_lds_level(self) -> int:
        """First mip level the last anti-aliased fast march staged in LDS (option "mip_lds" / environment BHR_MIP_LDS=1), -1 if none."""
        return int(self._lib.bhr_mip_lds_level(self._ctx))

    def row_costs(self, cam_pos, fov: float, split: bool = False):
        """Ray-steps per band of 8 rows for this view (one march with BHR_ROW_COSTS, no bloom).  split=True: the pair
        (steps taken by the fast arithmetic, steps taken by the strict arithmetic) -- a hybrid frame has both."""
        cam = self.camera_uniforms(cam_pos, fov, 0)
        flags = self._flags(True, True) | _lib.ROW_COSTS
        _lib.check(self._lib.bhr_render(self._ctx, C.byref(cam), flags))
        n = (self.rows + 7) // 8
        u64p = C.POINTER(C.c_uint64)
        if split:
            fast, strict = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
            _lib.check(self._lib.bhr_get_row_costs_split(self._ctx, fast.ctypes.data_as(u64p), strict.ctypes.data_as(u64p), n))
            return fast, strict
        out = np.zeros(n, dtype=np.uint64)
        _lib.check(self._lib.bhr_get_row_costs(self._ctx, out.ctypes.data_as(u64p), n))
        return out

    def read_final_u8(self) -> np.ndarray:
        out = np.empty((self.rows, self.width, 3), dtype=np.uint8)
        _lib.check(self._lib.bhr_read_final_u8(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def render(self, cam_pos: List[float], fov: float, frame: int = 0,
               skip_differentials: bool = False, skip_bloom: bool = False) -> np.ndarray:
        """One frame -> (rows, width, 3) float32 in [0, 1]  (render.py:3865-3923)."""
        if self.lens_flare and self.rows != self.height:
            raise ValueError("lens flare needs whole-frame sums; render row blocks through "
                             "bhr_amd.multigpu.group_render(..., lens_flare=True)")
        self.render_async(cam_pos, fov, frame, skip_differentials, skip_bloom)
        return self.read_layer(_lib.LAYER_FINAL)

    def selftest(self) -> dict:
        """Device check of the strict march's exact sqrt / divide sequences (bhr_selftest)."""
        out = (C.c_uint64 * 4)()
        _lib.check(self._lib.bhr_selftest(self._ctx, out))
        return {"bad_sqrt": out[0], "bad_div": out[1], "bad_div6": out[2], "checked": out[3]}

    def stream_map(self) -> dict:
        """Which of the context's streams share a hardware queue (bhr_debug_read, which = 3: a probe of two one-lane kernels
        per pair).  Keys "a+b" over scene / slot0 / slot1 / aux0 / aux1; True = one queue.  Diagnostics."""
        geom = (C.c_int32 * 10)()
        _lib.check(self._lib.bhr_debug_read(self._ctx, 3, None, 0, geom))
        names = ["scene", "slot0", "slot1", "aux0", "aux1"]
        out, q = {}, 0
        for i in range(5):
            for j in range(i + 1, 5):
                if geom[q] >= 0:
                    out[f"{names[i]}+{names[j]}"] = bool(geom[q])
                q += 1
        return out

    def stream_calibration(self) -> dict:
        """What the context's choice of slot 1's stream was made from (bhr_debug_read, which = 4): the candidates' frame
        rates and the one kept; "done" False until the ninth two-slot frame."""
        geom = (C.c_int32 * 10)()
        _lib.check(self._lib.bhr_debug_read(self._ctx, 4, None, 0, geom))
        return {"done": bool(geom[0]), "kept": int(geom[1]), "candidates_fps": [int(geom[2 + c]) for c in range(6)]}

    def hybrid_launch_order(self) -> np.ndarray:
        """The partitioned launch order of the last math="hybrid" march: tile indices, the strict tiles first
        (bhr_debug_read, which = 2).  Diagnostics."""
        geom = (C.c_int32 * 10)()
        _lib.check(self._lib.bhr_debug_read(self._ctx, 2, None, 0, geom))
        out = np.empty(int(geom[0]), dtype=np.int32)
        _lib.check(self._lib.bhr_debug_read(self._ctx, 2, out.ctypes.data, out.nbytes, None))
        return out

    def hybrid_info(self) -> dict:
        """Tile split and band of the last math="hybrid" march (bhr_hybrid_info)."""
        t, b = (C.c_int32 * 2)(), (C.c_double * 2)()
        _lib.check(self._lib.bhr_hybrid_info(self._ctx, t, b))
        rep = (C.c_int32 * 2)()
        _lib.check(self._lib.bhr_hybrid_repairs(self._ctx, rep))
        return {"strict_tiles": int(t[0]), "tiles": int(t[1]), "band_below": float(b[0]), "band_above": float(b[1]),
                "repaired_pixels": int(rep[0]), "repair_capacity": int(rep[1])}

    def timing_reset(self) -> None:
        _lib.check(self._lib.bhr_timing_reset(self._ctx))

    def frame_times(self, n: int) -> np.ndarray:
        """(n, 3) march start / march end / frame end of the last n timed frames, ms after the oldest one's start."""
        out = np.empty((n, 3), dtype=np.float32)
        _lib.check(self._lib.bhr_timing_dump(self._ctx, _lib.fptr(out), n))
        return out

    def counters(self) -> dict:
        c = _lib.Counters()
        _lib.check(self._lib.bhr_get_counters(self._ctx, C.byref(c)))
        return {name: getattr(c, name) for name, _ in c._fields_}
