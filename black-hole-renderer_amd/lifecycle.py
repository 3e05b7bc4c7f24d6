"""Entity lifecycle system that feeds the disk texture (host side).

Behavioural twin of the reference's CPU producer (render.py:493-792 entities and factories,
1667-1866 spawners, 3564-3653 rasterisation, 3655-3712 statistics).  Three populations of
transient structures live on the (r_norm, phi) texture grid:

* filaments  -- analytic Gaussian blobs sheared by differential rotation; evaluated per frame;
* hotspots   -- pre-rasterised von-Mises x Gaussian patches, rolled by their row's Keplerian shift;
* RT spikes  -- pre-rasterised radial fingers near the inner edge, rolled likewise.

Every draw from the NumPy generators happens in the reference's order, so a factory seeded
like the reference's replays its population exactly; tests/golden pins that.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Tuple

import numpy as np

FILAMENT_SHEAR_ALPHA = 0.1        # render.py:493
FILAMENT_TAU_COOL = 50.0
FILAMENT_DEATH_THRESHOLD = 0.008
FILAMENT_MAX_LIFETIME = 120.0
FILAMENT_BIRTH_FADE_DUR = 5.0

_EMPTY = np.empty((0, 0), dtype=np.float32)


@dataclass
class EntityInstance:
    """One structure on the disk (render.py:500-621)."""
    row_indices: np.ndarray
    phi_density: np.ndarray
    phi_temp: np.ndarray
    omega: float
    birth_time: float
    lifetime: float
    fade_in: float
    fade_out: float
    fade_noise: np.ndarray
    entity_type: str = "generic"
    source_phi: float = 0.0
    total_extent: float = 0.0
    alpha_shear: float = 0.0
    tau_cool: float = FILAMENT_TAU_COOL
    blob_base_r: float = 0.0
    blob_sigma_r: float = 0.0
    blob_sigma_phi0: float = 0.0
    blob_peak_density: float = 0.0
    blob_peak_temp: float = 0.0

    @property
    def total_duration(self) -> float:
        return self.fade_in + self.lifetime + self.fade_out

    def density_factor(self, age: float) -> float:
        """Filament decay: shear dilution s0 / (s0 + alpha age) times cooling exp(-age / tau)."""
        s0 = max(self.blob_sigma_phi0, 1e-6)
        stretch = s0 / (s0 + self.alpha_shear * age)
        cool = math.exp(-age / self.tau_cool) if self.tau_cool > 0 else 1.0
        return stretch * cool

    def is_dead(self, now: float) -> bool:
        age = now - self.birth_time
        if self.entity_type == "filament":
            if age >= FILAMENT_MAX_LIFETIME:
                return True
            return age >= 0 and self.density_factor(age) < FILAMENT_DEATH_THRESHOLD
        return age >= self.total_duration

    def fade_factor(self, now: float) -> float:
        """Trapezoid envelope of hotspots / RT spikes: ramp in, hold, ramp out."""
        age = now - self.birth_time
        if age < 0:
            return 0.0
        if age < self.fade_in:
            return age / self.fade_in if self.fade_in > 0 else 1.0
        held = age - self.fade_in
        if held < self.lifetime:
            return 1.0
        tail = held - self.lifetime
        if tail < self.fade_out:
            return 1.0 - tail / self.fade_out if self.fade_out > 0 else 0.0
        return 0.0


# ----------------------------------------------------------------------------- spawners
def _nearest_row(r_norm_all: np.ndarray, r: float) -> int:
    return int(np.argmin(np.abs(r_norm_all - r)))


def spawn_single_filament(rng, n_r, n_phi, r_norm_all, omega_all):
    """A compact hot blob (render.py:1667-1722).  Draw order: phi, r, sigma_r, sigma_phi, peak, temp ratio."""
    source_phi = float(rng.uniform(0, 2 * np.pi))
    r_pos = float(rng.uniform(0.05, 0.95))
    base_r = 0.05 + r_pos ** 0.6 * 0.9
    sigma_r = float(rng.uniform(0.005, 0.015))
    sigma_phi0 = float(rng.uniform(0.04, 0.10))
    peak_density = float(rng.uniform(0.5, 1.0))
    peak_temp = peak_density * float(rng.uniform(0.15, 0.35))

    rows = np.where(np.abs(r_norm_all - base_r) < 4 * sigma_r)[0]
    if len(rows) == 0:
        rows = np.array([_nearest_row(r_norm_all, base_r)])
    omega = float(omega_all[_nearest_row(r_norm_all, base_r)])
    return (rows, _EMPTY.copy(), _EMPTY.copy(), omega, source_phi, 2 * np.pi,
            sigma_r, sigma_phi0, peak_density, peak_temp, base_r)


def spawn_single_hotspot(rng, n_r, n_phi, r_norm_all, omega_all):
    """Roughly circular bright patch (render.py:1725-1793)."""
    phi = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    h_phi = float(rng.uniform(0, 2 * np.pi))
    r_rand = float(rng.uniform(0, 1))
    h_r = 0.1 + r_rand ** 0.6 * 0.85
    h_phi_width = float(rng.uniform(0.08, 0.20))
    h_r_width = 0.02 + float(rng.uniform(0, 0.03))
    h_intensity = 0.3 + (1 - h_r) * 0.6 + float(rng.uniform(0, 0.1))
    _ = 0.5 + 2.5 * float(rng.power(0.4))          # h_delta_T: drawn, not used (keeps the stream aligned)

    rows = np.where((r_norm_all >= h_r - 3 * h_r_width) & (r_norm_all <= h_r + 3 * h_r_width))[0]
    if len(rows) == 0:
        rows = np.array([_nearest_row(r_norm_all, h_r)])
    r_sub = r_norm_all[rows]

    kappa = 1.5 / (h_phi_width ** 2)
    phi_prof = np.exp(kappa * (np.cos(phi - h_phi) - 1))
    dens = np.zeros((len(rows), n_phi), dtype=np.float32)
    temp = np.zeros((len(rows), n_phi), dtype=np.float32)
    for k in range(len(rows)):
        r_prof = np.exp(-0.5 * ((r_sub[k] - h_r) / (h_r_width + 1e-8)) ** 2)
        dens[k] = phi_prof * r_prof * h_intensity
        temp[k] = dens[k] * 0.12
    dens = np.clip(dens, 0, 1)
    temp = np.clip(temp, 0, 1)
    return rows, dens, temp, float(omega_all[_nearest_row(r_norm_all, h_r)])


def spawn_single_rt_spike(rng, n_r, n_phi, r_norm_all, omega_all):
    """Rayleigh-Taylor finger near the inner edge (render.py:1796-1866)."""
    phi = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    rt_phi = float(rng.uniform(0, 2 * np.pi))
    r_base = float(np.power(rng.uniform(0.01, 0.15), 1.5))
    phi_width = float(rng.uniform(0.08, 0.20))
    r_length = float(rng.uniform(0.08, 0.20))
    intensity = float(rng.uniform(0.8, 1.0))
    delta_T = float(rng.uniform(0.5, 1.2))

    rows = np.where((r_norm_all >= max(r_base - 0.02, 0.0)) & (r_norm_all <= r_base + r_length * 2.5))[0]
    if len(rows) == 0:
        rows = np.array([_nearest_row(r_norm_all, r_base)])
    r_sub = r_norm_all[rows]

    kappa = 1.5 / (phi_width ** 2)
    phi_prof = np.exp(kappa * (np.cos(phi - rt_phi) - 1))
    dens = np.zeros((len(rows), n_phi), dtype=np.float32)
    temp = np.zeros((len(rows), n_phi), dtype=np.float32)
    for k in range(len(rows)):
        dr = r_sub[k] - r_base
        fade_out = np.clip(r_length * 2 - dr, 0, 1)
        fade_in = np.clip((r_sub[k] - r_base) / (r_length * 0.3 + 1e-8), 0, 1)
        r_prof = np.exp(-0.5 * (dr / (r_length * 0.4 + 1e-8)) ** 2) * fade_out * fade_in
        dens[k] = phi_prof * r_prof * intensity
        temp[k] = dens[k] * delta_T
    dens = np.clip(dens, 0, 1)          # the temperature plane is left unclipped, as in the reference
    omega = float(omega_all[_nearest_row(r_norm_all, r_base + r_length * 0.5)])
    return rows, dens, temp, omega


# ----------------------------------------------------------------------------- factory
class EntityFactory:
    """Keeps ``target_count`` entities alive: removes the dead, spawns replacements at the
    steady-state rate (render.py:624-792)."""

    def __init__(self, spawn_fn: Callable, target_count: int, lifetime_range: Tuple[float, float],
                 fade_in: float, fade_out: float, n_r: int, n_phi: int, r_norm_all: np.ndarray,
                 omega_all: np.ndarray, seed: int = 0, entity_type: str = "generic"):
        self.spawn_fn = spawn_fn
        self.target_count = target_count
        self.lifetime_range = lifetime_range
        self.fade_in, self.fade_out = fade_in, fade_out
        self.n_r, self.n_phi = n_r, n_phi
        self.r_norm_all, self.omega_all = r_norm_all, omega_all
        self.rng = np.random.default_rng(seed)
        self.entities: List[EntityInstance] = []
        self._spawn_debt = 0.0
        self.entity_type = entity_type

    def _make_fade_noise(self) -> np.ndarray:
        """Two-sinusoid dissolve profile along phi; consumes 4 draws (render.py:720-734)."""
        phi = np.linspace(0, 2 * np.pi, self.n_phi, endpoint=False)
        f1 = int(self.rng.integers(3, 8))
        f2 = int(self.rng.integers(8, 16))
        p1 = float(self.rng.uniform(0, 2 * np.pi))
        p2 = float(self.rng.uniform(0, 2 * np.pi))
        wave = 0.6 * np.sin(phi * f1 + p1) + 0.4 * np.sin(phi * f2 + p2)
        return np.clip(wave * 0.5 + 0.5, 0, 1).astype(np.float32)

    def _spawn_one(self, now: float) -> EntityInstance:
        """spawn parameters, then lifetime, then fade noise -- in that order (render.py:676-678)."""
        made = self.spawn_fn(self.rng, self.n_r, self.n_phi, self.r_norm_all, self.omega_all)
        lifetime = float(self.rng.uniform(*self.lifetime_range))
        if self.entity_type == "filament":
            (rows, dens, temp, omega, source_phi, extent, sigma_r, sigma_phi0, peak_d, peak_t, base_r) = made
            return EntityInstance(rows, dens, temp, omega, now, lifetime, self.fade_in, self.fade_out,
                                  self._make_fade_noise(), entity_type="filament", source_phi=source_phi,
                                  total_extent=extent, alpha_shear=FILAMENT_SHEAR_ALPHA * omega,
                                  tau_cool=FILAMENT_TAU_COOL, blob_base_r=base_r, blob_sigma_r=sigma_r,
                                  blob_sigma_phi0=sigma_phi0, blob_peak_density=peak_d, blob_peak_temp=peak_t)
        rows, dens, temp, omega = made
        return EntityInstance(rows, dens, temp, omega, now, lifetime, self.fade_in, self.fade_out,
                              self._make_fade_noise(), entity_type=self.entity_type)

    @staticmethod
    def _filament_death_age(entity) -> float:
        for t in range(1, int(FILAMENT_MAX_LIFETIME) + 1):
            if entity.density_factor(float(t)) < FILAMENT_DEATH_THRESHOLD:
                return float(t)
        return FILAMENT_MAX_LIFETIME

    def seed_initial(self, now: float) -> None:
        """Start at steady state: ages staggered uniformly over each entity's life (render.py:736-757)."""
        n = max(self.target_count, 1)
        for i in range(self.target_count):
            e = self._spawn_one(now)
            if e.entity_type == "filament":
                span = max(self._filament_death_age(e) - FILAMENT_BIRTH_FADE_DUR, 1.0)
                age = FILAMENT_BIRTH_FADE_DUR + span * (i / n)
            else:
                age = (e.fade_in + e.lifetime) * (i / n)
            e.birth_time = now - age
            self.entities.append(e)

    def tick(self, now: float, dt: float) -> None:
        """One frame: drop the dead, pay off the spawn debt (render.py:767-787)."""
        self.entities = [e for e in self.entities if not e.is_dead(now)]
        deficit = self.target_count - len(self.entities)
        if deficit <= 0:
            return
        rate = self.target_count / (sum(self.lifetime_range) / 2.0)
        self._spawn_debt += rate * dt
        n_spawn = min(int(self._spawn_debt), deficit)
        self._spawn_debt -= n_spawn
        for _ in range(n_spawn):
            self.entities.append(self._spawn_one(now))

    @property
    def alive_entities(self) -> List[EntityInstance]:
        return self.entities


def make_factories(n_r: int, n_phi: int, r_inner: float, r_outer: float, seed: int = 42) -> Dict[str, EntityFactory]:
    """The three populations of _init_lifecycle_system (render.py:4098-4123), seeded and pre-aged."""
    r_norm_all = np.linspace(0, 1, n_r)
    r_vals = r_inner + (r_outer - r_inner) * r_norm_all
    omega_all = np.sqrt(0.5 / (r_vals ** 3 + 1e-6)).astype(np.float32)
    common = dict(n_r=n_r, n_phi=n_phi, r_norm_all=r_norm_all, omega_all=omega_all)
    factories = {
        "filament": EntityFactory(spawn_single_filament, target_count=200, lifetime_range=(15.0, 60.0),
                                  fade_in=0.0, fade_out=0.0, seed=seed + 100, entity_type="filament", **common),
        "hotspot": EntityFactory(spawn_single_hotspot, target_count=30, lifetime_range=(15.0, 30.0),
                                 fade_in=4.0, fade_out=4.0, seed=seed + 200, entity_type="hotspot", **common),
        "rt_spike": EntityFactory(spawn_single_rt_spike, target_count=15, lifetime_range=(15.0, 30.0),
                                  fade_in=3.0, fade_out=3.0, seed=seed + 300, entity_type="rt_spike", **common),
    }
    for f in factories.values():
        f.seed_initial(now=0.0)
    return factories


# ----------------------------------------------------------------------------- rasterisation + statistics
_STAGING_PLANES = (("filament", 0, 1), ("rt_spike", 2, 3), ("hotspot", 4, 5))


def rasterize_entities(factories: dict, now: float, n_r: int, n_phi: int, omega_rows: np.ndarray,
                       r_norm_all: np.ndarray) -> np.ndarray:
    """staging (6, n_r, n_phi) f32 = [arcs, arcs_temp, rt, rt_temp, hotspot, hotspot_temp]
    (the NumPy half of accumulate_entity_layer, render.py:3585-3649)."""
    staging = np.zeros((6, n_r, n_phi), dtype=np.float32)
    phi = np.linspace(0, 2 * np.pi, n_phi, endpoint=False)
    two_pi = 2 * np.pi
    for key, d_idx, t_idx in _STAGING_PLANES:
        factory = factories.get(key)
        if factory is None:
            continue
        for e in factory.alive_entities:
            age = now - e.birth_time
            if e.entity_type == "filament":
                if e.density_factor(age) < FILAMENT_DEATH_THRESHOLD:
                    continue
                s0 = max(e.blob_sigma_phi0, 1e-6)
                sigma_phi = s0 + e.alpha_shear * age
                amp_d = e.blob_peak_density * s0 / sigma_phi
                amp_t = e.blob_peak_temp * s0 / sigma_phi
                born = min(age / FILAMENT_BIRTH_FADE_DUR, 1.0) if FILAMENT_BIRTH_FADE_DUR > 0 else 1.0
                cool = math.exp(-age / e.tau_cool) if e.tau_cool > 0 else 1.0
                scale_d = amp_d * born * cool
                scale_t = amp_t * born * cool
                inv_2s_phi = 0.5 / (sigma_phi * sigma_phi)
                sigma_r = max(e.blob_sigma_r, 1e-6)
                inv_2s_r = 0.5 / (sigma_r * sigma_r)
                for ri in e.row_indices:
                    if 0 <= ri < n_r:
                        r_w = math.exp(-(r_norm_all[ri] - e.blob_base_r) ** 2 * inv_2s_r)
                        center = (e.source_phi - omega_rows[ri] * age) % two_pi
                        d_phi = phi - center
                        d_phi = d_phi - two_pi * np.round(d_phi / two_pi)
                        prof = np.exp(-d_phi * d_phi * inv_2s_phi)
                        staging[d_idx, ri] += prof * (scale_d * r_w)
                        staging[t_idx, ri] += prof * (scale_t * r_w)
            else:
                alpha = e.fade_factor(now)
                if alpha <= 0:
                    continue
                for k, ri in enumerate(e.row_indices):
                    if 0 <= ri < n_r:
                        shift = int(age * omega_rows[ri] / (2 * np.pi) * n_phi)
                        staging[d_idx, ri] += np.roll(e.phi_density[k], -shift) * alpha
                        staging[t_idx, ri] += np.roll(e.phi_temp[k], -shift) * alpha
    return staging


def compose_statistics(comp: np.ndarray, edge: np.ndarray, enable_rt: int = 1):
    """(density_p98, struct_scale, row_stats (n_r, 2)) from the 13 component planes
    (recompute_interactive_stats, render.py:3666-3712): 98th percentile of the edge-weighted
    density, 95th percentile of the positive structural temperature, per-row max / 70 % quantile of
    the scaled structural temperature, all floored so that temp_base survives in empty rows."""
    sp, turb, arc, rt, hs, dm = comp[1], comp[3], comp[5], comp[7], comp[9], comp[12]
    rt_w = 0.20 if enable_rt else 0.0
    density = (0.15 + 0.10 * sp + 0.30 * turb + 0.20 * hs + 0.30 * arc + rt_w * rt) * dm
    density *= edge[:, None]
    density_p98 = max(float(np.percentile(density, 98)), 0.01)

    temp_struct = (comp[2] + comp[4] + comp[6] + comp[8] + comp[10]) * dm
    positive = temp_struct > 0
    struct_scale = float(np.percentile(temp_struct[positive], 95)) if np.any(positive) else 1.0
    struct_scale = max(struct_scale, 0.01)

    scaled = np.clip(temp_struct / (struct_scale + 1e-6) * 0.8, 0, 1.2)
    row_max = np.max(scaled, axis=1).astype(np.float32)
    row_p70 = np.quantile(scaled, 0.7, axis=1).astype(np.float32)
    tb_max = np.max(comp[0], axis=1).astype(np.float32)
    row_max = np.maximum(row_max, tb_max)
    row_p70 = np.maximum(row_p70, tb_max * 0.8)
    return density_p98, struct_scale, np.column_stack([row_max, row_p70]).astype(np.float32)
