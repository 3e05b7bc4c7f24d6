"""Disk V2 analytic accretion-disk model, evaluated on the MI355X in binary64.

Same surface as the reference's ``disk_v2`` package (function names, argument meaning, scalar/array
return behaviour, parameter validation; `disk_v2/__init__.py`, `params.py`, `geometry.py`,
`physical_fields.py`, `structure_modulations.py`).  All field arithmetic happens in
``csrc/disk_v2.hip`` through ``bhr_disk_v2_eval``; the host only validates parameters, broadcasts the
inputs and makes the reference's random draws (``default_rng(seed)`` for the shear texture,
``default_rng(seed + 1)`` for the hotspots inside ``structure_modulation``) so that the same seeds give
the same structures.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

MAX_TERMS = 32


@dataclass(frozen=True)
class DiskV2Params:
    """Base disk: radii, thickness law, radial power laws, edge softness (params.py:12-68)."""
    r_in: float = 2.0
    r_out: float = 10.0
    h0: float = 0.05
    beta_h: float = 0.05
    rho_power: float = 1.0
    temp_scale: float = 1.0
    omega_scale: float = 1.0
    edge_softness: float = 0.1

    def __post_init__(self) -> None:
        if self.r_in <= 0.0:
            raise ValueError("r_in must be positive")
        if self.r_out <= self.r_in:
            raise ValueError("r_out must be greater than r_in")
        if self.h0 <= 0.0:
            raise ValueError("h0 must be positive")
        if self.rho_power <= 0.0:
            raise ValueError("rho_power must be positive")
        if self.temp_scale <= 0.0:
            raise ValueError("temp_scale must be positive")
        if self.omega_scale <= 0.0:
            raise ValueError("omega_scale must be positive")
        if not 0.0 <= self.edge_softness < 0.5:
            raise ValueError("edge_softness must be in [0, 0.5)")


@dataclass(frozen=True)
class DiskV2StructureParams:
    """Multiplicative surface structure: two weak modes, a sheared Fourier texture, sparse hotspots
    (params.py:70-144).  Strength limits keep every factor positive."""
    mode1_strength: float = 0.03
    mode2_strength: float = 0.05
    shear_strength: float = 0.22
    shear_components: int = 8
    hotspot_strength: float = 0.16
    hotspot_count: int = 8
    hotspot_phi_sigma: float = 0.18
    hotspot_logr_sigma: float = 0.12
    hotspot_inner_bias: float = 2.0

    def __post_init__(self) -> None:
        if self.mode1_strength < 0.0:
            raise ValueError("mode1_strength must be non-negative")
        if self.mode2_strength < 0.0:
            raise ValueError("mode2_strength must be non-negative")
        if self.mode1_strength + self.mode2_strength >= 1.0:
            raise ValueError("mode1_strength + mode2_strength must be less than 1")
        if self.shear_strength < 0.0:
            raise ValueError("shear_strength must be non-negative")
        if self.shear_strength >= 1.0:
            raise ValueError("shear_strength must be less than 1")
        if self.shear_components <= 0:
            raise ValueError("shear_components must be positive")
        if self.hotspot_strength < 0.0:
            raise ValueError("hotspot_strength must be non-negative")
        if self.hotspot_strength >= 1.0:
            raise ValueError("hotspot_strength must be less than 1")
        if self.hotspot_count <= 0:
            raise ValueError("hotspot_count must be positive")
        if self.hotspot_phi_sigma <= 0.0:
            raise ValueError("hotspot_phi_sigma must be positive")
        if self.hotspot_logr_sigma <= 0.0:
            raise ValueError("hotspot_logr_sigma must be positive")
        if self.hotspot_inner_bias <= 0.0:
            raise ValueError("hotspot_inner_bias must be positive")
        if self.shear_components > MAX_TERMS or self.hotspot_count > MAX_TERMS:
            raise ValueError(f"at most {MAX_TERMS} shear components / hotspots on the device")


class _CParams(C.Structure):     # include/bhr_disk_v2.h: bhr_disk_v2_params
    _fields_ = [(n, C.c_double) for n in ("r_in", "r_out", "h0", "beta_h", "rho_power", "temp_scale", "omega_scale",
                                          "edge_softness", "mode1_strength", "mode2_strength", "shear_strength",
                                          "hotspot_strength", "hotspot_phi_sigma", "hotspot_logr_sigma",
                                          "hotspot_inner_bias")] + [
        ("shear_components", C.c_int32), ("hotspot_count", C.c_int32),
        ("shear_phi_freq", C.c_int32 * MAX_TERMS), ("shear_logr_freq", C.c_int32 * MAX_TERMS),
        ("shear_phase", C.c_double * MAX_TERMS), ("hotspot_phase", C.c_double * MAX_TERMS),
        ("hotspot_log_r", C.c_double * MAX_TERMS), ("hotspot_weight", C.c_double * MAX_TERMS)]


(F_H, F_MASK_R, F_W_R, F_W_Z, F_MASK_VOL, F_OMEGA, F_RHO_MID, F_T_MID, F_RHO, F_T, F_MODE, F_SHEAR, F_HOTSPOT,
 F_TOTAL) = range(14)


def shear_table(structure_params: DiskV2StructureParams, seed: int):
    """The draws of shear_modulation (structure_modulations.py:170-176): per component
    integers(2, 10), integers(1, 6), uniform(0, 2 pi)."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(structure_params.shear_components):
        rows.append((int(rng.integers(2, 10)), int(rng.integers(1, 6)), float(rng.uniform(0.0, 2.0 * np.pi))))
    return rows


def hotspot_table(params: DiskV2Params, structure_params: DiskV2StructureParams, seed: int):
    """The draws of hotspot_modulation (structure_modulations.py:247-250): per spot uniform(0, 2 pi),
    uniform(0, 1) ** inner_bias * log(r_out / r_in), uniform(0.6, 1)."""
    rng = np.random.default_rng(seed)
    span = np.log(params.r_out / params.r_in)
    rows = []
    for _ in range(structure_params.hotspot_count):
        phase = float(rng.uniform(0.0, 2.0 * np.pi))
        log_r = float((rng.uniform(0.0, 1.0) ** structure_params.hotspot_inner_bias) * span)
        rows.append((phase, log_r, float(rng.uniform(0.6, 1.0))))
    return rows


def pack_params(params: DiskV2Params, structure_params: DiskV2StructureParams | None = None, shear_seed: int = 42,
                hotspot_seed: int = 42) -> _CParams:
    sp = structure_params or DiskV2StructureParams()
    c = _CParams()
    for k in ("r_in", "r_out", "h0", "beta_h", "rho_power", "temp_scale", "omega_scale", "edge_softness"):
        setattr(c, k, float(getattr(params, k)))
    for k in ("mode1_strength", "mode2_strength", "shear_strength", "hotspot_strength", "hotspot_phi_sigma",
              "hotspot_logr_sigma", "hotspot_inner_bias"):
        setattr(c, k, float(getattr(sp, k)))
    c.shear_components, c.hotspot_count = sp.shear_components, sp.hotspot_count
    for i, (pf, lf, ph) in enumerate(shear_table(sp, shear_seed)):
        c.shear_phi_freq[i], c.shear_logr_freq[i], c.shear_phase[i] = pf, lf, ph
    for i, (ph, lr, wt) in enumerate(hotspot_table(params, sp, hotspot_seed)):
        c.hotspot_phase[i], c.hotspot_log_r[i], c.hotspot_weight[i] = ph, lr, wt
    return c


_default_ctx = None


def _context():
    """A minimal device context for field evaluation (no image buffers worth mentioning)."""
    global _default_ctx
    if _default_ctx is None:
        lib = _lib.load()
        cfg = _lib.Config(8, 8, 0, 8, 0.1, 10.0, 2.0, 15.0, 0.0, 0, 1.0, 0.1, 0, 1)
        h = C.c_void_p()
        _lib.check(lib.bhr_create(C.byref(cfg), C.byref(h)))
        _default_ctx = h
    return _default_ctx


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def evaluate(field: int, cparams: _CParams, r, z=None, phi=None, norm_shear: float = 0.0, norm_hotspot: float = 0.0,
             ctx=None, return_max: bool = False):
    """Broadcast the inputs, evaluate ``field`` on the device, return an array of the broadcast shape."""
    lib = _lib.load()
    if not hasattr(lib.bhr_disk_v2_eval, "_typed"):
        lib.bhr_disk_v2_eval.argtypes = [C.c_void_p, C.POINTER(_CParams), C.c_int32, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_double,
                                         C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.bhr_disk_v2_eval.restype = C.c_int32
        lib.bhr_disk_v2_eval._typed = True
    arrs = [np.asarray(a, dtype=np.float64) for a in (r, z, phi) if a is not None]
    shape = np.broadcast_shapes(*[a.shape for a in arrs])
    flat = lambda a: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), shape)).ravel()
    rr, zz, pp = flat(r), flat(z), flat(phi)
    out = np.empty(rr.shape[0], dtype=np.float64)
    mx = (C.c_double * 2)()
    _lib.check(lib.bhr_disk_v2_eval(ctx or _context(), C.byref(cparams), int(field), _dptr(rr), _dptr(zz), _dptr(pp),
                                    rr.shape[0], float(norm_shear), float(norm_hotspot), _dptr(out), mx))
    out = out.reshape(shape)
    return (out, (mx[0], mx[1])) if return_max else out


def _restore(value: np.ndarray, *originals, boolean: bool = False):
    """Scalars in, scalar out; arrays in, array out (_array_utils.py:34-63)."""
    if boolean:
        value = value != 0.0
    if all(np.ndim(o) == 0 for o in originals):
        return bool(value) if boolean else float(value)
    return value


def _base(params: DiskV2Params) -> _CParams:
    return pack_params(params)


# ---- geometry.py ---------------------------------------------------------------------------------
def smoothstep(edge0: float, edge1: float, x):
    """Cubic Hermite step (geometry.py:15-47); trivial, kept on the host."""
    if edge1 <= edge0:
        raise ValueError("edge1 must be greater than edge0")
    xa = np.asarray(x, dtype=np.float64)
    t = np.clip((xa - edge0) / (edge1 - edge0), 0.0, 1.0)
    return _restore(t * t * (3.0 - 2.0 * t), x)


def disk_half_thickness(r, params: DiskV2Params):
    return _restore(evaluate(F_H, _base(params), r), r)


def disk_radial_mask(r, params: DiskV2Params):
    return _restore(evaluate(F_MASK_R, _base(params), r), r, boolean=True)


def disk_radial_weight(r, params: DiskV2Params):
    return _restore(evaluate(F_W_R, _base(params), r), r)


def disk_vertical_weight(r, z, params: DiskV2Params):
    return _restore(evaluate(F_W_Z, _base(params), r, z=z), r, z)


def disk_volume_mask(r, z, params: DiskV2Params):
    return _restore(evaluate(F_MASK_VOL, _base(params), r, z=z), r, z, boolean=True)


# ---- physical_fields.py --------------------------------------------------------------------------
def angular_velocity_field(r, params: DiskV2Params):
    return _restore(evaluate(F_OMEGA, _base(params), r), r)


def midplane_density_field(r, params: DiskV2Params):
    return _restore(evaluate(F_RHO_MID, _base(params), r), r)


def midplane_temperature_field(r, params: DiskV2Params):
    return _restore(evaluate(F_T_MID, _base(params), r), r)


def density_field(r, z, params: DiskV2Params):
    return _restore(evaluate(F_RHO, _base(params), r, z=z), r, z)


def temperature_field(r, z, params: DiskV2Params):
    return _restore(evaluate(F_T, _base(params), r, z=z), r, z)


# ---- structure_modulations.py --------------------------------------------------------------------
def weak_mode_modulation(r, phi, params: DiskV2Params, structure_params: DiskV2StructureParams | None = None):
    return _restore(evaluate(F_MODE, pack_params(params, structure_params), r, phi=phi), r, phi)


def shear_modulation(r, phi, params: DiskV2Params, structure_params: DiskV2StructureParams | None = None,
                     seed: int = 42):
    return _restore(evaluate(F_SHEAR, pack_params(params, structure_params, shear_seed=seed), r, phi=phi), r, phi)


def hotspot_modulation(r, phi, params: DiskV2Params, structure_params: DiskV2StructureParams | None = None,
                       seed: int = 42):
    return _restore(evaluate(F_HOTSPOT, pack_params(params, structure_params, hotspot_seed=seed), r, phi=phi), r, phi)


def structure_modulation(r, phi, params: DiskV2Params, structure_params: DiskV2StructureParams | None = None,
                         seed: int = 42):
    """F_mode * F_shear(seed) * F_hotspot(seed + 1), 1 outside the disk (structure_modulations.py:292-334)."""
    cp = pack_params(params, structure_params, shear_seed=seed, hotspot_seed=seed + 1)
    return _restore(evaluate(F_TOTAL, cp, r, phi=phi), r, phi)


# ---- analytic disk source for the renderer ----------------------------------------------------------
def reference_norms(params: DiskV2Params, structure_params: DiskV2StructureParams | None = None, seed: int = 42,
                    n_r: int = 512, n_phi: int = 2048, ctx=None):
    """(cparams, max|raw shear|, max|raw hotspot|, peak T_mid) on the reference grid
    r = linspace(r_in, r_out, n_r) x phi = linspace(0, 2 pi, n_phi, endpoint=False): the fixed constants
    that replace the reference's "maximum over the evaluated array" when single rays are shaded."""
    cp = pack_params(params, structure_params, shear_seed=seed, hotspot_seed=seed + 1)
    r = np.linspace(params.r_in, params.r_out, n_r)
    phi = np.linspace(0.0, 2.0 * np.pi, n_phi, endpoint=False)
    rg, pg = np.meshgrid(r, phi, indexing="ij")
    _, (m_sh, m_hs) = evaluate(F_TOTAL, cp, rg, phi=pg, ctx=ctx, return_max=True)
    t_peak = float(np.max(evaluate(F_T_MID, cp, r, ctx=ctx)))
    return cp, m_sh, m_hs, t_peak


def disk_rgba(r, phi, cparams: _CParams, norm_shear: float, norm_hotspot: float, t_peak: float,
              color_temp: float = 6000.0, ctx=None) -> np.ndarray:
    """Host twin of csrc/march.hip: disk_v2_rgba (fields from the device, colour mapping in NumPy):
    (..., 4) float32.  Used to bake textures and to test the in-kernel source."""
    from .skybox import blackbody_rgb
    F = evaluate(F_TOTAL, cparams, r, phi=phi, norm_shear=norm_shear, norm_hotspot=norm_hotspot, ctx=ctx)
    T = evaluate(F_T_MID, cparams, np.broadcast_to(r, F.shape), ctx=ctx)
    rho = evaluate(F_RHO_MID, cparams, np.broadcast_to(r, F.shape), ctx=ctx)
    t = np.clip(T * F / t_peak, 0.0, 1.0).astype(np.float32)
    t_factor = np.float32((color_temp - 4500.0) / (6500.0 - 2700.0))
    T_min, T_max = np.float32(2000.0) + t_factor * np.float32(1000.0), np.float32(9000.0) + t_factor * np.float32(3000.0)
    bb = blackbody_rgb((T_min + t * (T_max - T_min)).astype(np.float64))
    bb[..., 2] = np.minimum(bb[..., 2], bb[..., 0])
    lum = np.clip(np.sqrt(t), 0, 1)
    out = np.empty(F.shape + (4,), dtype=np.float32)
    out[..., :3] = np.clip(bb * lum[..., None], 0, 1)
    out[..., 3] = np.clip(rho * F, 0.0, 1.0)
    return out


__all__ = ["DiskV2Params", "DiskV2StructureParams", "disk_half_thickness", "disk_radial_mask", "disk_radial_weight",
           "disk_vertical_weight", "disk_volume_mask", "density_field", "midplane_density_field",
           "midplane_temperature_field", "angular_velocity_field", "temperature_field", "weak_mode_modulation",
           "shear_modulation", "hotspot_modulation", "structure_modulation"]
