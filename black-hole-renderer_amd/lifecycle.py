"""Host producer of the disk texture's transient structures: WHO exists and with which parameters.

The reference keeps three populations on the (r_norm, phi) texture grid (render.py:493-792 bookkeeping,
1667-1866 spawners): filaments (analytic sheared Gaussians, evaluated per frame), hotspots and Rayleigh-Taylor
spikes (rows rasterised once at birth, rolled with their Keplerian shift).  Their evolution is driven by NumPy
random streams whose draw ORDER is the contract -- a population seeded like the reference's must replay it draw
for draw (tests/golden/lifecycle.npz) -- so that part stays on the host.  Everything per texel happens on the
device: `lifecycle_device` turns the populations into (entity, row) pair tables for csrc/lifecycle.hip.

Layout: entities are plain records; what an entity *does* over its life are three small functions of (record,
time) -- `filament_strength`, `expired`, `envelope`; a `Population` owns one random stream, one kind, one list.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

FILAMENT_SHEAR_ALPHA = 0.1        # render.py:493-497
FILAMENT_TAU_COOL = 50.0
FILAMENT_DEATH_THRESHOLD = 0.008
FILAMENT_MAX_LIFETIME = 120.0
FILAMENT_BIRTH_FADE_DUR = 5.0

_NO_ROWS = np.empty((0, 0), dtype=np.float32)
_TWO_PI = 2 * np.pi


@dataclass(eq=False)
class Entity:
    """One structure: where it sits, when it was born, how it fades.  Data only."""
    kind: str
    row_indices: np.ndarray            # texture rows it touches
    phi_density: np.ndarray            # (rows, n_phi) rasterised rows (hotspot / rt_spike), empty for filaments
    phi_temp: np.ndarray
    omega: float                       # Keplerian angular velocity of its anchor row
    birth_time: float
    lifetime: float                    # plateau between the ramps (hotspot / rt_spike)
    ramp_in: float
    ramp_out: float
    fade_noise: np.ndarray             # dissolve profile along phi (drawn for every entity: keeps the stream aligned)
    # analytic blob (filaments)
    source_phi: float = 0.0
    base_r: float = 0.0
    sigma_r: float = 0.0
    sigma_phi0: float = 0.0
    peak_density: float = 0.0
    peak_temp: float = 0.0
    shear_rate: float = 0.0            # d sigma_phi / d age
    cooling_time: float = FILAMENT_TAU_COOL


def filament_strength(e: Entity, age: float) -> float:
    """Remaining strength of a filament: diluted by shear, s0 / (s0 + rate age), and cooling, exp(-age / tau)."""
    s0 = max(e.sigma_phi0, 1e-6)
    dilution = s0 / (s0 + e.shear_rate * age)
    return dilution * (math.exp(-age / e.cooling_time) if e.cooling_time > 0 else 1.0)


def expired(e: Entity, now: float) -> bool:
    age = now - e.birth_time
    if e.kind != "filament":
        return age >= e.ramp_in + e.lifetime + e.ramp_out
    return age >= FILAMENT_MAX_LIFETIME or (age >= 0 and filament_strength(e, age) < FILAMENT_DEATH_THRESHOLD)


def envelope(e: Entity, now: float) -> float:
    """Trapezoid of a hotspot / spike: 0 before birth, linear ramp in, 1 on the plateau, linear ramp out, 0."""
    t = now - e.birth_time
    if t < 0:
        return 0.0
    if t < e.ramp_in:
        return t / e.ramp_in if e.ramp_in > 0 else 1.0
    t -= e.ramp_in
    if t < e.lifetime:
        return 1.0
    t -= e.lifetime
    if t < e.ramp_out:
        return 1.0 - t / e.ramp_out if e.ramp_out > 0 else 0.0
    return 0.0


# ----------------------------------------------------------------------------- birth: one draw sequence per kind
@dataclass(frozen=True)
class Grid:
    n_r: int
    n_phi: int
    r_norm: np.ndarray       # (n_r,) normalised radius of each texture row
    omega: np.ndarray        # (n_r,) f32 Keplerian angular velocity of each row

    def nearest(self, r: float) -> int:
        return int(np.argmin(np.abs(self.r_norm - r)))

    def rows_where(self, mask: np.ndarray, fallback_r: float) -> np.ndarray:
        rows = np.where(mask)[0]
        return rows if len(rows) else np.array([self.nearest(fallback_r)])


def _von_mises_rows(g: Grid, centre_phi: float, phi_width: float, radial: np.ndarray, gain: float):
    """rows x n_phi f32 patch: exp(kappa (cos(phi - centre) - 1)) x radial[k] x gain, kappa = 1.5 / width^2."""
    phi = np.linspace(0, _TWO_PI, g.n_phi, endpoint=False)
    along_phi = np.exp(1.5 / (phi_width ** 2) * (np.cos(phi - centre_phi) - 1))
    out = np.zeros((len(radial), g.n_phi), dtype=np.float32)
    for k, w in enumerate(radial):
        out[k] = along_phi * w * gain
    return out


def draw_filament(rng, g: Grid) -> dict:
    """render.py:1667-1722.  Draws: phi, r, sigma_r, sigma_phi, peak, temperature ratio."""
    source_phi = float(rng.uniform(0, _TWO_PI))
    base_r = 0.05 + float(rng.uniform(0.05, 0.95)) ** 0.6 * 0.9
    sigma_r = float(rng.uniform(0.005, 0.015))
    sigma_phi0 = float(rng.uniform(0.04, 0.10))
    peak_density = float(rng.uniform(0.5, 1.0))
    peak_temp = peak_density * float(rng.uniform(0.15, 0.35))
    omega = float(g.omega[g.nearest(base_r)])
    return dict(row_indices=g.rows_where(np.abs(g.r_norm - base_r) < 4 * sigma_r, base_r), phi_density=_NO_ROWS.copy(),
                phi_temp=_NO_ROWS.copy(), omega=omega, source_phi=source_phi, base_r=base_r, sigma_r=sigma_r,
                sigma_phi0=sigma_phi0, peak_density=peak_density, peak_temp=peak_temp,
                shear_rate=FILAMENT_SHEAR_ALPHA * omega, cooling_time=FILAMENT_TAU_COOL)


def draw_hotspot(rng, g: Grid) -> dict:
    """render.py:1725-1793.  Draws: phi, r, phi width, r width, intensity jitter, one unused power-law draw."""
    centre_phi = float(rng.uniform(0, _TWO_PI))
    centre_r = 0.1 + float(rng.uniform(0, 1)) ** 0.6 * 0.85
    phi_width = float(rng.uniform(0.08, 0.20))
    r_width = 0.02 + float(rng.uniform(0, 0.03))
    gain = 0.3 + (1 - centre_r) * 0.6 + float(rng.uniform(0, 0.1))
    rng.power(0.4)                                   # the reference draws a temperature contrast it never uses
    rows = g.rows_where((g.r_norm >= centre_r - 3 * r_width) & (g.r_norm <= centre_r + 3 * r_width), centre_r)
    radial = [np.exp(-0.5 * ((r - centre_r) / (r_width + 1e-8)) ** 2) for r in g.r_norm[rows]]
    dens = _von_mises_rows(g, centre_phi, phi_width, radial, gain)
    temp = dens * np.float32(0.12)
    return dict(row_indices=rows, phi_density=np.clip(dens, 0, 1), phi_temp=np.clip(temp, 0, 1),
                omega=float(g.omega[g.nearest(centre_r)]))


def draw_rt_spike(rng, g: Grid) -> dict:
    """render.py:1796-1866.  Draws: phi, base radius (x^1.5), phi width, length, intensity, temperature contrast."""
    centre_phi = float(rng.uniform(0, _TWO_PI))
    base_r = float(np.power(rng.uniform(0.01, 0.15), 1.5))
    phi_width = float(rng.uniform(0.08, 0.20))
    length = float(rng.uniform(0.08, 0.20))
    gain = float(rng.uniform(0.8, 1.0))
    contrast = float(rng.uniform(0.5, 1.2))
    rows = g.rows_where((g.r_norm >= max(base_r - 0.02, 0.0)) & (g.r_norm <= base_r + length * 2.5), base_r)
    radial = []
    for r in g.r_norm[rows]:
        d = r - base_r
        radial.append(np.exp(-0.5 * (d / (length * 0.4 + 1e-8)) ** 2) * np.clip(length * 2 - d, 0, 1)
                      * np.clip(d / (length * 0.3 + 1e-8), 0, 1))
    dens = _von_mises_rows(g, centre_phi, phi_width, radial, gain)
    temp = np.zeros_like(dens)
    for k in range(len(rows)):
        temp[k] = dens[k] * contrast                 # from the unclipped density; only the density is clipped
    return dict(row_indices=rows, phi_density=np.clip(dens, 0, 1), phi_temp=temp,
                omega=float(g.omega[g.nearest(base_r + length * 0.5)]))


_DRAW = {"filament": draw_filament, "hotspot": draw_hotspot, "rt_spike": draw_rt_spike}


# ----------------------------------------------------------------------------- populations
class EntityFactory:
    """A population of one kind held at ``target_count`` (the object `accumulate_entity_layer(factories, now)`
    receives; render.py:624-792): one random stream, births at the steady-state rate, deaths by `expired`."""

    def __init__(self, kind: str, target_count: int, lifetime_range: Tuple[float, float], ramp_in: float, ramp_out: float,
                 grid: Grid, seed: int):
        self.kind, self.target_count, self.lifetime_range = kind, target_count, lifetime_range
        self.ramp_in, self.ramp_out, self.grid = ramp_in, ramp_out, grid
        self.rng = np.random.default_rng(seed)
        self.entities: List[Entity] = []
        self._owed = 0.0                              # fractional births carried from frame to frame

    @property
    def alive_entities(self) -> List[Entity]:
        return self.entities

    def _dissolve_profile(self) -> np.ndarray:
        """Two sinusoids along phi, 4 draws: integer frequencies in [3, 8) and [8, 16), two phases (render.py:720-734)."""
        phi = np.linspace(0, _TWO_PI, self.grid.n_phi, endpoint=False)
        f_lo, f_hi = int(self.rng.integers(3, 8)), int(self.rng.integers(8, 16))
        p_lo, p_hi = float(self.rng.uniform(0, _TWO_PI)), float(self.rng.uniform(0, _TWO_PI))
        return np.clip((0.6 * np.sin(phi * f_lo + p_lo) + 0.4 * np.sin(phi * f_hi + p_hi)) * 0.5 + 0.5, 0, 1).astype(np.float32)

    def _birth(self, now: float) -> Entity:
        # stream order: shape parameters, plateau length, dissolve profile (render.py:676-678)
        shape = _DRAW[self.kind](self.rng, self.grid)
        plateau = float(self.rng.uniform(*self.lifetime_range))
        return Entity(kind=self.kind, birth_time=now, lifetime=plateau, ramp_in=self.ramp_in, ramp_out=self.ramp_out,
                      fade_noise=self._dissolve_profile(), **shape)

    def seed_initial(self, now: float) -> None:
        """Steady state from frame 0: the i-th of n entities is born already i/n of the way through its life
        (filaments: through the span between their birth fade and the age at which they drop under the death
        threshold, searched in whole seconds; render.py:736-757)."""
        n = max(self.target_count, 1)
        for i in range(self.target_count):
            e = self._birth(now)
            if self.kind == "filament":
                fading_age = next((float(t) for t in range(1, int(FILAMENT_MAX_LIFETIME) + 1)
                                   if filament_strength(e, float(t)) < FILAMENT_DEATH_THRESHOLD), FILAMENT_MAX_LIFETIME)
                age = FILAMENT_BIRTH_FADE_DUR + max(fading_age - FILAMENT_BIRTH_FADE_DUR, 1.0) * (i / n)
            else:
                age = (e.ramp_in + e.lifetime) * (i / n)
            e.birth_time = now - age
            self.entities.append(e)

    def tick(self, now: float, dt: float) -> None:
        """One frame (render.py:767-787): bury the expired; while under target, births accrue at
        target / mean plateau per unit time and whole ones are paid out, at most the deficit."""
        self.entities = [e for e in self.entities if not expired(e, now)]
        missing = self.target_count - len(self.entities)
        if missing <= 0:
            return
        self._owed += self.target_count / (sum(self.lifetime_range) / 2.0) * dt
        births = min(int(self._owed), missing)
        self._owed -= births
        self.entities.extend(self._birth(now) for _ in range(births))


# kind -> (target count, plateau range, ramp in, ramp out, seed offset)   (render.py:4098-4123)
POPULATIONS = {"filament": (200, (15.0, 60.0), 0.0, 0.0, 100), "hotspot": (30, (15.0, 30.0), 4.0, 4.0, 200),
               "rt_spike": (15, (15.0, 30.0), 3.0, 3.0, 300)}


def make_factories(n_r: int, n_phi: int, r_inner: float, r_outer: float, seed: int = 42) -> Dict[str, EntityFactory]:
    """The three populations of _init_lifecycle_system, seeded base + 100 / 200 / 300 and pre-aged at t = 0."""
    r_norm = np.linspace(0, 1, n_r)
    radius = r_inner + (r_outer - r_inner) * r_norm
    grid = Grid(n_r, n_phi, r_norm, np.sqrt(0.5 / (radius ** 3 + 1e-6)).astype(np.float32))
    out = {}
    for kind, (count, plateau, ramp_in, ramp_out, offset) in POPULATIONS.items():
        out[kind] = EntityFactory(kind, count, plateau, ramp_in, ramp_out, grid, seed + offset)
        out[kind].seed_initial(now=0.0)
    return out
