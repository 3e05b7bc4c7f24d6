"""TEST INFRASTRUCTURE -- reference-identical NumPy forms of the entity layer and the compose statistics
(render.py:3564-3653 accumulate_entity_layer, 3655-3712 recompute_interactive_stats), pinned bit for bit by
tests/golden/lifecycle.npz (generated from the reference's own functions).  The product computes both on the
device (csrc/lifecycle.hip, bhr_amd.lifecycle_device); only tests/ import this module."""
import math

import numpy as np

from bhr_amd.lifecycle import FILAMENT_BIRTH_FADE_DUR, FILAMENT_DEATH_THRESHOLD, envelope, filament_strength

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
        for e in factory.entities:
            age = now - e.birth_time
            if e.kind == "filament":
                if filament_strength(e, age) < FILAMENT_DEATH_THRESHOLD:
                    continue
                s0 = max(e.sigma_phi0, 1e-6)
                sigma_phi = s0 + e.shear_rate * age
                amp_d = e.peak_density * s0 / sigma_phi
                amp_t = e.peak_temp * s0 / sigma_phi
                born = min(age / FILAMENT_BIRTH_FADE_DUR, 1.0) if FILAMENT_BIRTH_FADE_DUR > 0 else 1.0
                cool = math.exp(-age / e.cooling_time) if e.cooling_time > 0 else 1.0
                scale_d = amp_d * born * cool
                scale_t = amp_t * born * cool
                inv_2s_phi = 0.5 / (sigma_phi * sigma_phi)
                sigma_r = max(e.sigma_r, 1e-6)
                inv_2s_r = 0.5 / (sigma_r * sigma_r)
                for ri in e.row_indices:
                    if 0 <= ri < n_r:
                        r_w = math.exp(-(r_norm_all[ri] - e.base_r) ** 2 * inv_2s_r)
                        center = (e.source_phi - omega_rows[ri] * age) % two_pi
                        d_phi = phi - center
                        d_phi = d_phi - two_pi * np.round(d_phi / two_pi)
                        prof = np.exp(-d_phi * d_phi * inv_2s_phi)
                        staging[d_idx, ri] += prof * (scale_d * r_w)
                        staging[t_idx, ri] += prof * (scale_t * r_w)
            else:
                alpha = envelope(e, now)
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
