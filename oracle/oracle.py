"""ctypes front-end of the CPU oracle (oracle/bhr_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of bhr_oracle.c.  Imported by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by
the product package.

`OracleRenderer` mirrors the host-side call sequence of the reference's
``TaichiRenderer`` (render.py:2189-4028): same constructor arguments, same
``render()`` composition (render.py:3865-3923) including its quirk that the
disk layer is read back *before* the bloom kernel's in-place update, so the
CLI image is ``clip(bg + disk + blur)`` with the un-scaled blur.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "build")

NUM_MIP_EXTRA_LEVELS = 4  # generate_disk_mipmaps(levels=4) -> 5 stored levels (render.py:2239)


class _Camera(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 3), ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("cam_forward", C.c_float * 3), ("pixel_width", C.c_float), ("pixel_height", C.c_float),
                ("r_escape", C.c_float)]


class _MarchParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("h_base", C.c_float), ("r_inner", C.c_float),
                ("r_outer", C.c_float), ("t_offset", C.c_float), ("disk_tilt", C.c_float),
                ("skip_diff", C.c_int32), ("anti_alias_mode", C.c_int32), ("aa_strength", C.c_float)]


def build(force: bool = False) -> None:
    """Compile both oracle variants with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_libs: dict = {}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def load(fast=False) -> C.CDLL:
    """fast: False -> strict f32 checker; True -> -ffast-math/OpenMP build; "f64" -> binary64 build."""
    name = {False: "liboracle.so", True: "liboracle_fast.so", "f64": "liboracle_f64.so"}[fast]
    if name in _libs:
        return _libs[name]
    path = os.path.join(_BUILD, name)
    src = os.path.join(_HERE, "bhr_oracle.c")
    if not os.path.isfile(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    lib = C.CDLL(path)
    F, I32, I64 = C.POINTER(C.c_float), C.c_int32, C.c_int64
    lib.oracle_ray_march.restype = I64
    lib.oracle_ray_march.argtypes = [C.POINTER(_Camera), C.POINTER(_MarchParams), F, I32, I32, F, I32, I32,
                                     F, I32, F, F, C.POINTER(C.c_int32), I32, I32]
    lib.oracle_bloom.restype = None
    lib.oracle_bloom.argtypes = [F, F, F, I32, I32, C.c_float, C.c_float, I32, C.c_float]
    lib.oracle_compose_disk_texture.restype = None
    lib.oracle_compose_disk_texture.argtypes = [F, F, F, F, F, F, I32, I32, C.c_float, I32, C.c_float]
    lib.oracle_build_mips.restype = None
    lib.oracle_build_mips.argtypes = [F, F, I32, I32, I32]
    lib.oracle_generate_background.restype = None
    lib.oracle_generate_background.argtypes = [F, I32, I32, I32, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.oracle_eval_noise.restype = None
    lib.oracle_eval_noise.argtypes = [F, I64, I32, I32, C.c_float, C.c_float, F]
    D = C.POINTER(C.c_double)
    lib.oracle_dv2_eval.restype = None
    lib.oracle_dv2_eval.argtypes = [C.c_void_p, I32, D, D, D, I64, C.c_double, C.c_double, D]
    lib.oracle_set_volume.restype = None
    lib.oracle_set_volume.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I32]
    lib.oracle_set_escape_out.restype = None
    lib.oracle_set_escape_out.argtypes = [C.c_void_p]
    for fn, at in (("oracle_probe_g_factor", [D, I64, D]), ("oracle_probe_tint", [D, I64, D]),
                   ("oracle_probe_disk_mip", [F, I32, I32, I32, D, I64, D]),
                   ("oracle_probe_skybox", [F, I32, I32, D, I64, D])):
        getattr(lib, fn).restype = None
        getattr(lib, fn).argtypes = at
    lib.oracle_num_threads.restype = I32
    lib.oracle_set_num_threads.argtypes = [I32]
    _libs[name] = lib
    return lib


# --------------------------------------------------------------------------- host helpers
def build_camera(cam_pos, fov_deg: float, width: int, height: int):
    """f64 pin-hole camera looking at the origin (render.py:93-127)."""
    p = np.array(cam_pos, dtype=np.float64)
    fwd = -p / np.linalg.norm(p)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    rn = np.linalg.norm(right)
    right = np.array([1.0, 0.0, 0.0]) if rn < 1e-6 else right / rn
    up = np.cross(right, fwd)
    up /= np.linalg.norm(up)
    plane_h = 2.0 * np.tan(np.radians(fov_deg) / 2)
    plane_w = plane_h * (width / height)
    return p, right, up, fwd, plane_w / width, plane_h / height


def build_mips_padded(disk_tex: np.ndarray, levels: int = NUM_MIP_EXTRA_LEVELS + 1, fast: bool = False) -> np.ndarray:
    """(levels, n_r, n_phi, 4) padded mip stack as held by disk_mips_field (render.py:2239-2251)."""
    lib = load(fast)
    base = np.ascontiguousarray(disk_tex, dtype=np.float32)
    n_r, n_phi = base.shape[:2]
    mips = np.zeros((levels, n_r, n_phi, 4), dtype=np.float32)
    lib.oracle_build_mips(_fp(mips), _fp(base), n_r, n_phi, levels)
    return mips


def eval_noise(coords: np.ndarray, mode: str = "simplex", octaves: int = 4, persistence: float = 0.5,
               lacunarity: float = 2.0) -> np.ndarray:
    """render.py:3769-3790."""
    lib = load()
    c = np.ascontiguousarray(coords, dtype=np.float32)
    out = np.empty(c.shape[0], dtype=np.float32)
    lib.oracle_eval_noise(_fp(c), c.shape[0], 0 if mode == "simplex" else 1, int(octaves),
                          float(persistence), float(lacunarity), _fp(out))
    return out


def generate_background(n_r: int, n_phi: int, az_freq: int, az_shear: float, r_inner: float, r_outer: float,
                        t: float, comp: np.ndarray | None = None, fast: bool = False) -> np.ndarray:
    """Fills comp[0,1,2,3,4,11,12] (render.py:3332-3451); other planes untouched."""
    lib = load(fast)
    if comp is None:
        comp = np.zeros((13, n_r, n_phi), dtype=np.float32)
    assert comp.dtype == np.float32 and comp.flags.c_contiguous and comp.shape == (13, n_r, n_phi)
    lib.oracle_generate_background(_fp(comp), n_r, n_phi, int(az_freq), float(az_shear), float(r_inner),
                                   float(r_outer), float(t))
    return comp


def compose_disk_texture(comp, omega_rows, edge, stats, row_stats, t_offset: float, enable_rt: int = 1,
                         color_temp: float = 6000.0, fast: bool = False) -> np.ndarray:
    """render.py:3169-3257; returns (n_r, n_phi, 4) f32."""
    lib = load(fast)
    comp = np.ascontiguousarray(comp, dtype=np.float32)
    _, n_r, n_phi = comp.shape
    omega_rows = np.ascontiguousarray(omega_rows, dtype=np.float32)
    edge = np.ascontiguousarray(edge, dtype=np.float32)
    stats = np.ascontiguousarray(stats, dtype=np.float32)
    row_stats = np.ascontiguousarray(row_stats, dtype=np.float32)
    assert omega_rows.shape == (n_r,) and edge.shape == (n_r,) and stats.shape == (2,) and row_stats.shape == (n_r, 2)
    tex = np.zeros((n_r, n_phi, 4), dtype=np.float32)
    lib.oracle_compose_disk_texture(_fp(tex), _fp(comp), _fp(omega_rows), _fp(edge), _fp(stats), _fp(row_stats),
                                    n_r, n_phi, float(t_offset), int(enable_rt), float(color_temp))
    return tex


def dv2_eval(cparams, field: int, r, z=None, phi=None, norm_shear: float = 0.0, norm_hotspot: float = 0.0, fast=False):
    """Disk V2 field ``field`` (ids of include/bhr_disk_v2.h) at broadcast points; 11 / 12 return the raw
    shear / hotspot sums.  ``cparams``: a ctypes struct with the layout of bhr_disk_v2_params."""
    arrs = np.broadcast_arrays(*[np.asarray(a, dtype=np.float64) for a in (r, z, phi) if a is not None])
    shape = arrs[0].shape
    it = iter(arrs)
    get = lambda a: np.ascontiguousarray(next(it)).ravel() if a is not None else None
    rr, zz, pp = get(r), get(z), get(phi)
    out = np.empty(rr.size, dtype=np.float64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    load(fast).oracle_dv2_eval(C.byref(cparams), field, dp(rr), dp(zz), dp(pp), rr.size, norm_shear, norm_hotspot, dp(out))
    return out.reshape(shape)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def probe_g_factor(rows16, fast=False):
    """_apply_g_factor (render.py:2439-2516) on rows of base_color[3], hit_pos[3], hit_r, ray_dir_to_cam[3],
    cam_pos[3], r_inner, r_outer, tilt_rad -> (n, 3)."""
    a = np.ascontiguousarray(rows16, dtype=np.float64).reshape(-1, 16)
    out = np.empty((a.shape[0], 3), dtype=np.float64)
    load(fast).oracle_probe_g_factor(_dp(a), a.shape[0], _dp(out))
    return out


def probe_tint(temps, fast=False):
    """_color_temp_to_tint (render.py:2407-2437) -> (n, 3)."""
    a = np.ascontiguousarray(temps, dtype=np.float64).ravel()
    out = np.empty((a.size, 3), dtype=np.float64)
    load(fast).oracle_probe_tint(_dp(a), a.size, _dp(out))
    return out


def probe_disk_mip(mips_padded, rows6, fast=False):
    """_sample_disk_mip (render.py:2600-2637) on rows of hit_x, hit_y, r_inner, r_outer, t_offset, lod -> (n, 4)."""
    m = np.ascontiguousarray(mips_padded, dtype=np.float32)
    a = np.ascontiguousarray(rows6, dtype=np.float64).reshape(-1, 6)
    out = np.empty((a.shape[0], 4), dtype=np.float64)
    load(fast).oracle_probe_disk_mip(_fp(m), m.shape[0], m.shape[1], m.shape[2], _dp(a), a.shape[0], _dp(out))
    return out


def probe_skybox(skybox, dirs, fast=False):
    """_sample_skybox (render.py:2541-2566) on unit directions -> (n, 3)."""
    s = np.ascontiguousarray(skybox, dtype=np.float32)
    a = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    out = np.empty((a.shape[0], 3), dtype=np.float64)
    load(fast).oracle_probe_skybox(_fp(s), s.shape[0], s.shape[1], _dp(a), a.shape[0], _dp(out))
    return out


class OracleRenderer:
    """CPU twin of TaichiRenderer's render path (render.py:2199-2266, 3865-3923)."""

    def set_volume(self, cparams=None, norm_shear=1.0, norm_hotspot=1.0, t_peak=1.0, absorption=4.0, grazing_gain=1.0,
                   substeps=2):
        """Finite-thickness Disk V2 source for the following march() calls of THIS build of the library
        (None: back to the textured thin disk).  Mirrors bhr_set_disk_source(BHR_DISK_V2_VOLUME)."""
        self.lib.oracle_set_volume(C.byref(cparams) if cparams is not None else None, norm_shear, norm_hotspot, t_peak,
                                   absorption, grazing_gain, substeps)

    def __init__(self, width, height, skybox, disk_tex, step_size=0.1, r_max=10.0, r_disk_inner=2.0,
                 r_disk_outer=15.0, disk_tilt=0.0, anti_alias="disabled", aa_strength=1.0,
                 disk_rotation_speed=0.1, fast=False):
        self.lib = load(fast)
        self.fast = fast
        self.width, self.height = int(width), int(height)
        self.step_size, self.r_max = float(step_size), float(r_max)
        self.r_disk_inner, self.r_disk_outer = float(r_disk_inner), float(r_disk_outer)
        self.disk_tilt = float(disk_tilt)
        self.anti_alias, self.aa_strength = anti_alias, float(aa_strength)
        self.disk_rotation_speed = float(disk_rotation_speed)
        self.skybox = np.ascontiguousarray(skybox, dtype=np.float32)
        self.tex_h, self.tex_w = self.skybox.shape[:2]
        self.update_disk_texture(disk_tex)
        self.last_steps = None
        self.last_total_steps = 0

    def update_disk_texture(self, disk_tex):
        self.disk_tex = np.ascontiguousarray(disk_tex, dtype=np.float32)
        self.dtex_h, self.dtex_w = self.disk_tex.shape[:2]
        self.mips = build_mips_padded(self.disk_tex, fast=self.fast)
        self.num_mip_levels = self.mips.shape[0]

    def camera_uniforms(self, cam_pos, fov):
        p, right, up, fwd, pw, ph = build_camera(np.array(cam_pos, dtype=np.float64), fov, self.width, self.height)
        cam = _Camera()
        cam.cam_pos[:] = list(p.astype(np.float32))
        cam.cam_right[:] = list(right.astype(np.float32))
        cam.cam_up[:] = list(up.astype(np.float32))
        cam.cam_forward[:] = list(fwd.astype(np.float32))
        cam.pixel_width = float(pw)
        cam.pixel_height = float(ph)
        cam.r_escape = float(max(self.r_max, float(np.linalg.norm(p)) * 2))
        return cam

    def march(self, cam_pos, fov, frame=0, skip_differentials=False, rows=None, want_steps=True):
        """Returns (image, disk_layer) in the reference's (W, H, 3) field layout."""
        W, H = self.width, self.height
        cam = self.camera_uniforms(cam_pos, fov)
        prm = _MarchParams(W, H, self.step_size, self.r_disk_inner, self.r_disk_outer,
                           float(frame) * self.disk_rotation_speed, self.disk_tilt,
                           1 if skip_differentials else 0, 0 if self.anti_alias == "disabled" else 1,
                           self.aa_strength)
        img = np.zeros((W, H, 3), dtype=np.float32)
        disk = np.zeros((W, H, 3), dtype=np.float32)
        steps = np.zeros((W, H), dtype=np.int32) if want_steps else None
        j_lo, j_hi = (0, H) if rows is None else rows
        total = self.lib.oracle_ray_march(
            C.byref(cam), C.byref(prm), _fp(self.skybox), self.tex_h, self.tex_w, _fp(self.disk_tex),
            self.dtex_h, self.dtex_w, _fp(self.mips), self.num_mip_levels, _fp(img), _fp(disk),
            steps.ctypes.data_as(C.POINTER(C.c_int32)) if want_steps else None, j_lo, j_hi)
        self.last_steps, self.last_total_steps = steps, int(total)
        return img, disk

    def escape_directions(self, cam_pos, fov):
        """(W, H, 3) float64 unit escape directions of the rays of one frame (zeros for captured rays)."""
        buf = np.zeros((self.width, self.height, 3), dtype=np.float64)
        self.lib.oracle_set_escape_out(buf.ctypes.data_as(C.c_void_p))
        try:
            self.march(cam_pos, fov, skip_differentials=True, want_steps=False)
        finally:
            self.lib.oracle_set_escape_out(None)
        return buf

    def bloom(self, disk_layer):
        """_bloom_kernel as called at render.py:3914-3917. Returns (blur, mutated_disk_layer), (W,H,3)."""
        W, H = self.width, self.height
        layer = np.ascontiguousarray(disk_layer, dtype=np.float32).copy()
        bright = np.zeros_like(layer)
        blur = np.zeros_like(layer)
        self.lib.oracle_bloom(_fp(layer), _fp(bright), _fp(blur), W, H, 0.0, 0.4, int(W * 0.02),
                              float((W / 640.0) ** 2))
        return blur, layer

    def render(self, cam_pos, fov, frame=0, skip_differentials=False, skip_bloom=False, parts=False):
        """(H, W, 3) float32 in [0,1]; with parts=True also the (W,H,3) bg / disk / blur layers."""
        img, disk = self.march(cam_pos, fov, frame, skip_differentials)
        if skip_bloom:
            blur = np.zeros_like(img)
            final = np.clip(img + disk, 0, 1)
        else:
            blur, _ = self.bloom(disk)
            final = np.clip(img + disk + blur, 0, 1)
        out = final.transpose(1, 0, 2)
        if parts:
            return out, img, disk, blur
        return out
