"""Device-side entity layer and compose statistics (include/bhr_lifecycle.h).

The host keeps what is cheap and order-sensitive -- the factories' random streams, the per-entity
scalars, the per-(entity, row) scalars that depend on NumPy's promotion rules -- and hands the
per-texel work to the GPU: ~12 000 (entity, row) pairs instead of 6 x n_r x n_phi texels cross
PCIe per frame, and the percentile statistics are selected on the device without reading the 13
component planes back.  tests/lifecycle_checker.py holds the reference-identical NumPy forms these are
tested against.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from .lifecycle import FILAMENT_BIRTH_FADE_DUR, FILAMENT_DEATH_THRESHOLD, envelope, filament_strength

FIL_DTYPE = np.dtype([("center", "<f8"), ("inv_2s_phi", "<f8"), ("coef_d", "<f8"), ("coef_t", "<f8")])
ROL_DTYPE = np.dtype([("offset", "<i8"), ("shift", "<i4"), ("plane", "<i4"), ("alpha", "<f4"), ("stride", "<i4")])
# bhr_filament_entity / bhr_rolled_entity (include/bhr_lifecycle.h): what bhr_accumulate_population consumes
FIL_ENTITY = np.dtype([("birth_time", "<f8"), ("source_phi", "<f8"), ("sigma_phi0", "<f8"), ("shear_rate", "<f8"),
                       ("peak_density", "<f8"), ("peak_temp", "<f8"), ("cooling_time", "<f8"), ("row_lo", "<i4"), ("n_rows", "<i4")])
ROL_ENTITY = np.dtype([("birth_time", "<f8"), ("lifetime", "<f8"), ("ramp_in", "<f8"), ("ramp_out", "<f8"), ("offset", "<i8"),
                       ("row_lo", "<i4"), ("n_rows", "<i4"), ("plane", "<i4"), ("pad_", "<i4")])
_PLANE = {"rt_spike": 2, "hotspot": 4}
_TWO_PI = 2 * np.pi


def _bind(lib):
    if getattr(lib, "_lifecycle_typed", False):
        return
    P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
    F, D = C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.bhr_entity_profile_upload.argtypes = [P, F, F, I32, C.POINTER(I64)]
    lib.bhr_entity_profile_reset.argtypes = [P]
    lib.bhr_accumulate_entities.argtypes = [P, P, C.POINTER(I32), P, C.POINTER(I32), D]
    lib.bhr_accumulate_population.argtypes = [P, C.c_double, P, I32, D, P, I32, F]
    lib.bhr_stats_prepare.argtypes = [P, I32, C.POINTER(C.c_uint64)]
    lib.bhr_stats_select.argtypes = [P, I32, C.c_uint64, F]
    lib.bhr_stats_row_statistics.argtypes = [P, C.c_float, I32, I32, F]
    for n in ("bhr_entity_profile_upload", "bhr_entity_profile_reset", "bhr_accumulate_entities", "bhr_accumulate_population",
              "bhr_stats_prepare", "bhr_stats_select", "bhr_stats_row_statistics"):
        getattr(lib, n).restype = I32
    lib._lifecycle_typed = True


# --------------------------------------------------------------------------- pair tables
def _filament_static(e, n_r: int, omega_rows: np.ndarray, r_norm_all: np.ndarray):
    """What does not change over a filament's life: its rows, the f64 radial weights (math.exp, as the
    scalar code of the reference) and the rows' angular velocities."""
    c = getattr(e, "_pair_static", None)
    if c is None or c[0] != n_r:
        rows = e.row_indices[(e.row_indices >= 0) & (e.row_indices < n_r)].astype(np.int64)
        sigma_r = max(e.sigma_r, 1e-6)
        inv_2s_r = 0.5 / (sigma_r * sigma_r)
        r_w = np.array([math.exp(-(r_norm_all[ri] - e.base_r) ** 2 * inv_2s_r) for ri in rows], dtype=np.float64)
        c = (n_r, rows, r_w, omega_rows[rows])
        e._pair_static = c
    return c


def filament_pairs(factory, now: float, n_r: int, omega_rows: np.ndarray, r_norm_all: np.ndarray):
    """(rows, table) for the alive filaments in list order (render.py:3606-3638).  The per-row
    scalars are evaluated exactly as the reference's NumPy rasteriser does (f32 centre under NumPy's
    weak-scalar promotion, f64 radial weight through math.exp); the per-entity scalars are computed
    in a Python loop, everything per (entity, row) in one vector pass over all pairs."""
    rows_l, rw_l, om_l, counts = [], [], [], []
    src, age32, inv2s, sc_d, sc_t = [], [], [], [], []
    for e in factory.alive_entities:
        age = now - e.birth_time
        if filament_strength(e, age) < FILAMENT_DEATH_THRESHOLD:
            continue
        _, rows, r_w, om = _filament_static(e, n_r, omega_rows, r_norm_all)
        if len(rows) == 0:
            continue
        s0 = max(e.sigma_phi0, 1e-6)
        sigma_phi = s0 + e.shear_rate * age
        amp_d = e.peak_density * s0 / sigma_phi
        amp_t = e.peak_temp * s0 / sigma_phi
        born = min(age / FILAMENT_BIRTH_FADE_DUR, 1.0) if FILAMENT_BIRTH_FADE_DUR > 0 else 1.0
        cool = math.exp(-age / e.cooling_time) if e.cooling_time > 0 else 1.0
        rows_l.append(rows); rw_l.append(r_w); om_l.append(om); counts.append(len(rows))
        src.append(e.source_phi); age32.append(age)
        inv2s.append(0.5 / (sigma_phi * sigma_phi))
        sc_d.append(amp_d * born * cool)
        sc_t.append(amp_t * born * cool)
    if not counts:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=FIL_DTYPE)
    counts = np.asarray(counts)
    rows = np.concatenate(rows_l)
    r_w = np.concatenate(rw_l)
    om = np.concatenate(om_l)
    rep = lambda v, dt: np.repeat(np.asarray(v, dtype=dt), counts)
    # (f32(source_phi) - omega[row] * f32(age)) % f32(2 pi): the same f32 operations, elementwise
    center = (rep(src, np.float32) - om * rep(age32, np.float32)) % np.float32(_TWO_PI)
    rec = np.empty(len(rows), dtype=FIL_DTYPE)
    rec["center"] = center.astype(np.float64)
    rec["inv_2s_phi"] = rep(inv2s, np.float64)
    rec["coef_d"] = rep(sc_d, np.float64) * r_w
    rec["coef_t"] = rep(sc_t, np.float64) * r_w
    return rows, rec


def rolled_pairs(factories: dict, now: float, n_r: int, n_phi: int, omega_rows: np.ndarray, pool):
    """(rows, table) for RT spikes then hotspots (render.py:3640-3649)."""
    rows_l, om_l, off_l, counts = [], [], [], []
    age32, alpha32, plane, stride_l = [], [], [], []
    for key in ("rt_spike", "hotspot"):
        factory = factories.get(key)
        if factory is None:
            continue
        for e in factory.alive_entities:
            alpha = envelope(e, now)
            if alpha <= 0:
                continue
            off, stride = pool.offset_of(e)
            c = getattr(e, "_pair_static", None)
            if c is None or c[0] != (n_r, n_phi):
                k_idx = np.nonzero((e.row_indices >= 0) & (e.row_indices < n_r))[0]
                rows = e.row_indices[k_idx].astype(np.int64)
                c = ((n_r, n_phi), rows, omega_rows[rows], k_idx.astype(np.int64) * n_phi)
                e._pair_static = c
            _, rows, om, rel = c
            if len(rows) == 0:
                continue
            rows_l.append(rows); om_l.append(om); off_l.append(off + rel); counts.append(len(rows))
            age32.append(now - e.birth_time); alpha32.append(alpha); plane.append(_PLANE[key]); stride_l.append(stride)
    if not counts:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=ROL_DTYPE)
    counts = np.asarray(counts)
    rows = np.concatenate(rows_l)
    rep = lambda v, dt: np.repeat(np.asarray(v, dtype=dt), counts)
    rec = np.empty(len(rows), dtype=ROL_DTYPE)
    rec["offset"] = np.concatenate(off_l)
    # int(age * omega[row] / (2 pi) * n_phi) with the f32 arithmetic of np.float32 scalars
    rec["shift"] = (rep(age32, np.float32) * np.concatenate(om_l) / np.float32(_TWO_PI) * np.float32(n_phi)).astype(np.int64)
    rec["plane"] = rep(plane, np.int32)
    rec["alpha"] = rep(alpha32, np.float32)
    rec["stride"] = rep(stride_l, np.int32)
    return rows, rec


def to_csr(rows: np.ndarray, table: np.ndarray, n_r: int):
    """Group by texture row, keeping the visiting order inside each row (stable sort)."""
    order = np.argsort(rows, kind="stable")
    ptr = np.searchsorted(rows[order], np.arange(n_r + 1)).astype(np.int32)
    return np.ascontiguousarray(table[order]), ptr


class ProfilePool:
    """Device copies of the pre-rasterised hotspot / RT-spike rows, uploaded once per entity."""

    def __init__(self, lib, ctx, n_phi: int):
        self.lib, self.ctx, self.n_phi = lib, ctx, n_phi
        self._live = {}

    def offset_of(self, e):
        key = id(e)
        hit = self._live.get(key)
        if hit is not None and hit[2] is e:
            return hit[0], hit[1]
        dens = np.ascontiguousarray(e.phi_density, dtype=np.float32)
        temp = np.ascontiguousarray(e.phi_temp, dtype=np.float32)
        off = C.c_int64()
        _lib.check(self.lib.bhr_entity_profile_upload(self.ctx, _lib.fptr(dens), _lib.fptr(temp), dens.shape[0],
                                                      C.byref(off)))
        self._live[key] = (int(off.value), dens.shape[0] * self.n_phi, e)
        return self._live[key][0], self._live[key][1]

    def collect(self, factories: dict) -> None:
        """Forget dead entities; rebuild the pool when three quarters of it is garbage."""
        alive = {id(e) for k in ("rt_spike", "hotspot") if k in factories for e in factories[k].alive_entities}
        dead = [k for k in self._live if k not in alive]
        for k in dead:
            del self._live[k]
        self._dead_since_reset = getattr(self, "_dead_since_reset", 0) + len(dead)
        if self._dead_since_reset > 3 * max(len(alive), 1):
            _lib.check(self.lib.bhr_entity_profile_reset(self.ctx))
            self._live.clear()
            self._dead_since_reset = 0
            self.generation = getattr(self, "generation", 0) + 1     # every offset handed out so far is void


def accumulate_on_device(lib, ctx, pool: ProfilePool, factories: dict, now: float, n_r: int, n_phi: int,
                         omega_rows: np.ndarray, r_norm_all: np.ndarray) -> None:
    _bind(lib)
    pool.collect(factories)
    fil = factories.get("filament")
    f_rows, f_tab = filament_pairs(fil, now, n_r, omega_rows, r_norm_all) if fil is not None else (
        np.zeros(0, np.int64), np.zeros(0, FIL_DTYPE))
    r_rows, r_tab = rolled_pairs(factories, now, n_r, n_phi, omega_rows, pool)
    f_tab, f_ptr = to_csr(f_rows, f_tab, n_r)
    r_tab, r_ptr = to_csr(r_rows, r_tab, n_r)
    phi = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    i32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    _lib.check(lib.bhr_accumulate_entities(
        ctx, f_tab.ctypes.data_as(C.c_void_p) if len(f_tab) else None, i32p(f_ptr),
        r_tab.ctypes.data_as(C.c_void_p) if len(r_tab) else None, i32p(r_ptr),
        phi.ctypes.data_as(C.POINTER(C.c_double))))


# --------------------------------------------------------------------------- entity records (the per-frame path)
def _contiguous(rows: np.ndarray, n_r: int):
    """(row_lo, n_rows) if the rows are a run lo, lo + 1, ... inside the texture, else None."""
    if len(rows) == 0:
        return 0, 0
    lo, n = int(rows[0]), len(rows)
    if lo < 0 or lo + n > n_r or int(rows[-1]) != lo + n - 1 or (n > 2 and not np.array_equal(rows, np.arange(lo, lo + n))):
        return None
    return lo, n


class PopulationTables:
    """Entity records for bhr_accumulate_population, rebuilt only when a population's member list changes (a birth
    or a death; every fourth frame at the reference's rates).  The records hold what is static over an entity's life;
    everything that depends on `now` is evaluated by the library."""

    def __init__(self):
        self._fil_key = self._rol_key = None
        self.fil = np.zeros(0, dtype=FIL_ENTITY)
        self.rw = np.zeros(0, dtype=np.float64)
        self.rol = np.zeros(0, dtype=ROL_ENTITY)
        self.usable = True

    def filaments(self, factory, n_r, omega_rows, r_norm_all):
        ents = factory.alive_entities if factory is not None else []
        key = [id(e) for e in ents]
        if key == self._fil_key:
            return
        recs, rws = [], []
        for e in ents:
            c = getattr(e, "_pop_record", None)
            if c is None or c[0] != n_r:
                _, rows, r_w, _ = _filament_static(e, n_r, omega_rows, r_norm_all)
                run = _contiguous(rows, n_r)
                rec = None if run is None else np.array(
                    [(e.birth_time, e.source_phi, e.sigma_phi0, e.shear_rate, e.peak_density, e.peak_temp, e.cooling_time,
                      run[0], run[1])], dtype=FIL_ENTITY).tobytes()
                c = e._pop_record = (n_r, rec, np.ascontiguousarray(r_w, dtype=np.float64).tobytes())
            if c[1] is None:
                self.usable = False
                return
            recs.append(c[1])
            rws.append(c[2])
        # records are kept as bytes: joining 200 of them costs microseconds (np.concatenate of structured arrays: 0.4 ms)
        self.fil = np.frombuffer(b"".join(recs), dtype=FIL_ENTITY)
        self.rw = np.frombuffer(b"".join(rws), dtype=np.float64)
        self._fil_key = key
        self._fil_refs = list(ents)          # ids stay unique while the objects are alive

    def rolled(self, factories, n_r, n_phi, pool):
        ents = [(k, e) for k in ("rt_spike", "hotspot") if factories.get(k) is not None for e in factories[k].alive_entities]
        generation = getattr(pool, "generation", 0)
        key = [id(pool), generation] + [id(e) for _, e in ents]
        if key == self._rol_key:
            return
        recs = []
        for kind, e in ents:
            c = getattr(e, "_pop_record", None)
            if c is None or c[0] != (n_r, n_phi, id(pool), generation):
                off, _ = pool.offset_of(e)
                run = _contiguous(np.asarray(e.row_indices), n_r)
                rec = None if run is None else np.array(
                    [(e.birth_time, e.lifetime, e.ramp_in, e.ramp_out, off, run[0], run[1], _PLANE[kind], 0)],
                    dtype=ROL_ENTITY).tobytes()
                c = e._pop_record = ((n_r, n_phi, id(pool), generation), rec)
            if c[1] is None:
                self.usable = False
                return
            recs.append(c[1])
        self.rol = np.frombuffer(b"".join(recs), dtype=ROL_ENTITY)
        self._rol_key = key
        self._rol_refs = [e for _, e in ents]


def accumulate_population(lib, ctx, pool: ProfilePool, tables: PopulationTables, factories: dict, now: float, n_r: int,
                          n_phi: int, omega_rows: np.ndarray, r_norm_all: np.ndarray) -> bool:
    """accumulate_entity_layer through bhr_accumulate_population; False if a population cannot be expressed as
    entity records (rows that are not a contiguous run -- the caller falls back to accumulate_on_device)."""
    _bind(lib)
    pool.collect(factories)
    if tables.usable:
        tables.filaments(factories.get("filament"), n_r, omega_rows, r_norm_all)
    if tables.usable:
        tables.rolled(factories, n_r, n_phi, pool)
    if not tables.usable:
        return False
    om = np.ascontiguousarray(omega_rows, dtype=np.float32)
    _lib.check(lib.bhr_accumulate_population(
        ctx, float(now), tables.fil.ctypes.data_as(C.c_void_p) if len(tables.fil) else None, len(tables.fil),
        tables.rw.ctypes.data_as(C.POINTER(C.c_double)) if len(tables.rw) else None,
        tables.rol.ctypes.data_as(C.c_void_p) if len(tables.rol) else None, len(tables.rol), _lib.fptr(om)))
    return True


# --------------------------------------------------------------------------- statistics
# NumPy 2.x evaluates quantiles of f32 data in f32: q is cast to the array dtype, the virtual index
# (n - 1) q and the interpolation weight are f32, and so is the lerp (numpy/lib/_function_base_impl.py:
# quantile/percentile, _compute_virtual_index, _get_gamma, _lerp).  The device only selects order
# statistics; these helpers reproduce the index and lerp arithmetic, and
# tests/test_golden_host.py::test_numpy_lerp_replicates_percentile_and_quantile pins them to the
# installed NumPy.
def linear_rank(n: int, q32: np.float32):
    """(lo, hi, gamma) of method='linear' for n samples and an f32 quantile."""
    vi = (n - 1) * q32                                   # python int * np.float32 -> f32
    lo = int(np.floor(vi))
    hi = min(lo + 1, n - 1)
    gamma = np.float32(np.float64(vi) - lo)              # f32 - intp -> f64, cast back to the index dtype
    return lo, hi, gamma


def percentile_q(p: float) -> np.float32:
    """np.percentile(f32 array, p): q = p / f32(100)."""
    return np.true_divide(p, np.float32(100))


def numpy_lerp(a, b, t) -> np.float32:
    a, b, t = np.float32(a), np.float32(b), np.float32(t)
    diff = np.subtract(b, a)
    if t >= 0.5:
        return np.float32(b - diff * (1 - t))
    return np.float32(a + diff * t)


def stats_on_device(lib, ctx, n_r: int, n_phi: int, enable_rt: int):
    """(density_p98, struct_scale, row_stats) as compose_statistics returns them, selected on the device."""
    _bind(lib)
    n_pos = C.c_uint64()
    _lib.check(lib.bhr_stats_prepare(ctx, int(bool(enable_rt)), C.byref(n_pos)))

    def select(which, n, q32):
        lo, hi, g = linear_rank(n, q32)
        a, b = C.c_float(), C.c_float()
        _lib.check(lib.bhr_stats_select(ctx, which, lo, C.byref(a)))
        _lib.check(lib.bhr_stats_select(ctx, which, hi, C.byref(b)))
        return float(numpy_lerp(a.value, b.value, g))

    density_p98 = max(select(0, n_r * n_phi, percentile_q(98)), 0.01)
    struct_scale = select(1, int(n_pos.value), percentile_q(95)) if n_pos.value > 0 else 1.0
    struct_scale = max(struct_scale, 0.01)
    rows = np.empty((n_r, 4), dtype=np.float32)
    div = np.float32(struct_scale + 1e-6)          # temp_struct / (struct_scale + 1e-6): weak python scalar -> f32
    lo, hi, g = linear_rank(n_phi, np.float32(0.7))
    _lib.check(lib.bhr_stats_row_statistics(ctx, float(div), lo, hi, _lib.fptr(rows)))
    row_p70 = np.array([numpy_lerp(a, b, g) for a, b in zip(rows[:, 1], rows[:, 2])], dtype=np.float32)
    row_max = np.maximum(rows[:, 0], rows[:, 3])
    row_p70 = np.maximum(row_p70, rows[:, 3] * 0.8)
    return density_p98, struct_scale, np.column_stack([row_max, row_p70]).astype(np.float32)
