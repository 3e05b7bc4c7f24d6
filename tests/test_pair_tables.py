"""Pair tables handed to bhr_accumulate_entities: the vectorised builders give, bit for bit, what the
per-entity scalar expressions of the reference give (render.py:3606-3649)."""
import math

import numpy as np


class _Pool:
    def __init__(self, n_phi):
        self.n_phi, self.next, self.seen = n_phi, 0, {}

    def offset_of(self, e):
        if id(e) not in self.seen:
            self.seen[id(e)] = (self.next, len(e.row_indices) * self.n_phi)
            self.next += 2 * len(e.row_indices) * self.n_phi
        return self.seen[id(e)]


def _filament_rows_scalar(e, now, n_r, omega_rows, r_norm_all):
    from bhr_amd.lifecycle import FILAMENT_BIRTH_FADE_DUR
    age = now - e.birth_time
    s0 = max(e.sigma_phi0, 1e-6)
    sigma_phi = s0 + e.shear_rate * age
    amp_d, amp_t = e.peak_density * s0 / sigma_phi, e.peak_temp * s0 / sigma_phi
    born = min(age / FILAMENT_BIRTH_FADE_DUR, 1.0) if FILAMENT_BIRTH_FADE_DUR > 0 else 1.0
    cool = math.exp(-age / e.cooling_time) if e.cooling_time > 0 else 1.0
    sigma_r = max(e.sigma_r, 1e-6)
    inv_2s_r = 0.5 / (sigma_r * sigma_r)
    out = []
    for ri in e.row_indices:
        if 0 <= ri < n_r:
            r_w = math.exp(-(r_norm_all[ri] - e.base_r) ** 2 * inv_2s_r)
            center = (np.float32(e.source_phi) - omega_rows[ri] * np.float32(age)) % np.float32(2 * np.pi)
            out.append((int(ri), float(center), 0.5 / (sigma_phi * sigma_phi), amp_d * born * cool * r_w, amp_t * born * cool * r_w))
    return out


def test_pair_tables_match_scalar_expressions():
    from bhr_amd.lifecycle import FILAMENT_DEATH_THRESHOLD, envelope, filament_strength, make_factories
    from bhr_amd.lifecycle_device import filament_pairs, rolled_pairs
    n_r, n_phi = 96, 256
    fac = make_factories(n_r, n_phi, 2.0, 15.0, seed=42)
    r_norm_all = np.linspace(0, 1, n_r)
    r_phys = 2.0 + r_norm_all * 13.0
    omega_rows = np.sqrt(0.5 / (r_phys ** 3 + 1e-6)).astype(np.float32)
    pool = _Pool(n_phi)
    for step in range(12):
        now = 0.37 * step
        for f in fac.values():
            f.tick(now=now, dt=0.37)
        rows, tab = filament_pairs(fac["filament"], now, n_r, omega_rows, r_norm_all)     # second call reuses the caches
        want = []
        for e in fac["filament"].alive_entities:
            if filament_strength(e, now - e.birth_time) >= FILAMENT_DEATH_THRESHOLD:
                want += _filament_rows_scalar(e, now, n_r, omega_rows, r_norm_all)
        assert len(want) == len(rows) > 0
        np.testing.assert_array_equal(rows, [w[0] for w in want])
        for k, name in enumerate(("center", "inv_2s_phi", "coef_d", "coef_t")):
            np.testing.assert_array_equal(tab[name], np.array([w[k + 1] for w in want], dtype=np.float64))

        rows, tab = rolled_pairs(fac, now, n_r, n_phi, omega_rows, pool)
        k = 0
        for key, plane in (("rt_spike", 2), ("hotspot", 4)):
            for e in fac[key].alive_entities:
                alpha = envelope(e, now)
                if alpha <= 0:
                    continue
                off, stride = pool.offset_of(e)
                age = now - e.birth_time
                for j, ri in enumerate(e.row_indices):
                    if 0 <= ri < n_r:
                        shift = int(np.float32(age) * omega_rows[ri] / np.float32(2 * np.pi) * np.float32(n_phi))
                        got = tab[k]
                        assert (rows[k], got["offset"], got["shift"], got["plane"], got["stride"]) == (ri, off + j * n_phi, shift, plane, stride)
                        assert got["alpha"] == np.float32(alpha)
                        k += 1
        assert k == len(rows) > 0
