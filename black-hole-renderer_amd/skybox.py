"""Procedural equirectangular skybox (host NumPy; one-off asset step).

Behavioural twin of the reference's generator (render.py:136-368): the same random stream
(PCG64 seeded with ``seed``, draws in the same order) and the same arithmetic, so a given
``(tex_w, tex_h, seed, n_stars)`` produces the reference's texture.  Ingredients: a faint
low-frequency nebula, ``n_stars`` Gaussian star blobs whose positions favour the galactic
plane, whose brightness follows a Salpeter IMF seen through a magnitude cut-off and whose
colour is a desaturated black body, and a Milky-Way glow with a four-arm modulation.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

# render.py:61-77
STAR_BRIGHTNESS_MIN = 0.03
STAR_BRIGHTNESS_MAX = 1.0
STAR_BRIGHTNESS_GAIN = 1.8
STAR_COLOR_SATURATION = 0.3
STAR_SIZE_MIN = 0.5
STAR_SIZE_MAX = 1.7
MILKY_WAY_GLOW = 0.10
GALACTIC_CENTER_GLOW = 0.08

_GAL_INCLINATION = np.radians(62.87)   # galactic plane vs. equator
_GAL_CENTER_RA = np.radians(266.4)
_GAL_CENTER_DEC = np.radians(-28.9)


def blackbody_rgb(T: np.ndarray) -> np.ndarray:
    """Tanner Helland colour-temperature fit, vectorised (render.py:136-150)."""
    t = T / 100.0
    red = np.where(t <= 66, 1.0, np.clip(1.292936 * np.power(np.maximum(t - 60, 1e-6), -0.1332047592), 0, 1))
    green = np.where(t <= 66,
                     np.clip(0.390082 * np.log(np.maximum(t, 1e-6)) - 0.631841, 0, 1),
                     np.clip(1.129891 * np.power(np.maximum(t - 60, 1e-6), -0.0755148492), 0, 1))
    blue = np.where(t >= 66, 1.0,
                    np.where(t <= 19, 0.0, np.clip(0.543207 * np.log(np.maximum(t - 10, 1e-6)) - 1.19625, 0, 1)))
    return np.stack([red, green, blue], axis=-1).astype(np.float32)


def _galactic_latitude_sin(dec, ra):
    return np.sin(dec) * np.cos(_GAL_INCLINATION) - np.cos(dec) * np.sin(_GAL_INCLINATION) * np.sin(ra - _GAL_CENTER_RA)


def _star_positions(rng, n_stars):
    """Rejection-sample directions: uniform on the sphere, thinned away from the galactic plane
    and boosted towards the galactic centre."""
    phis, thetas = [], []
    batch = n_stars * 3
    have = 0
    while have < n_stars:
        z = rng.uniform(-1, 1, batch)
        phi = rng.uniform(0, 2 * np.pi, batch)
        theta = np.arccos(np.clip(z, -1, 1))
        dec = np.pi / 2 - theta
        b = np.arcsin(np.clip(_galactic_latitude_sin(dec, phi), -1, 1))
        prob = 0.15 + 0.85 * np.exp(-0.5 * (b / np.radians(8)) ** 2)
        cos_sep = (np.sin(dec) * np.sin(_GAL_CENTER_DEC)
                   + np.cos(dec) * np.cos(_GAL_CENTER_DEC) * np.cos(phi - _GAL_CENTER_RA))
        sep = np.arccos(np.clip(cos_sep, -1, 1))
        prob += 0.3 * np.exp(-0.5 * (sep / np.radians(20)) ** 2)
        prob = prob / prob.max()
        keep = rng.random(batch) < prob
        room = n_stars - have
        phis.extend(phi[keep][:room])
        thetas.extend(theta[keep][:room])
        have = len(phis)
    return np.array(phis[:n_stars]), np.array(thetas[:n_stars])


def _star_masses_and_magnitudes(rng, n_stars):
    """Salpeter IMF (dN/dM ~ M^-2.35 on [0.08, 50] M_sun), main-sequence luminosities, random
    distances; keep what is brighter than apparent magnitude 8."""
    alpha, m_lo, m_hi = 2.35, 0.08, 50.0
    pool = n_stars * 30
    u = rng.random(pool)
    mass = (m_lo ** (1 - alpha) + u * (m_hi ** (1 - alpha) - m_lo ** (1 - alpha))) ** (1 / (1 - alpha))
    lum_exp = np.where(mass < 0.43, 2.3, np.where(mass < 2.0, 4.0, np.where(mass < 55.0, 3.5, 1.0)))
    luminosity = np.power(mass, lum_exp)
    abs_mag = -2.5 * np.log10(luminosity + 1e-30) + 4.83
    dist_pc = np.clip(rng.exponential(scale=200.0, size=pool), 1.0, 5000.0)
    app_mag = abs_mag + 5.0 * np.log10(dist_pc / 10.0)
    visible = np.where(app_mag <= 8.0)[0]
    if len(visible) >= n_stars:
        pick = rng.choice(visible, size=n_stars, replace=False)
    else:
        pick = np.argsort(app_mag)[:n_stars]
    return mass[pick], app_mag[pick]


STAR_PATCH_R = 4            # 9 x 9 blob


def sky_tables(tex_w: int = 2048, tex_h: int = 1024, seed: int = 42, n_stars: int = 6000) -> dict:
    """Everything of generate_skybox that comes out of the random stream (render.py:153-262), in its draw order:
    the nebula noise quantised to u8 at 1/16 resolution, and per star its texel centre, colour and the values of its
    Gaussian blob.  The per-texel work on these tables is `rasterize_sky` on the host or bhr_skybox_build on the device."""
    rng = np.random.default_rng(seed)
    coarse = rng.random((tex_h // 16, tex_w // 16, 3)).astype(np.float32) * 0.06
    coarse_u8 = (coarse * 255).astype(np.uint8)

    phi_s, theta_s = _star_positions(rng, n_stars)
    cx = (phi_s / (2 * np.pi) * tex_w).astype(np.float32)
    cy = (theta_s / np.pi * tex_h).astype(np.float32)

    mass, app_mag = _star_masses_and_magnitudes(rng, n_stars)
    mag_norm = (app_mag - app_mag.min()) / (app_mag.max() - app_mag.min() + 1e-30)
    brightness = (STAR_BRIGHTNESS_MAX - (STAR_BRIGHTNESS_MAX - STAR_BRIGHTNESS_MIN) * mag_norm).astype(np.float32)
    brightness = np.clip(brightness * STAR_BRIGHTNESS_GAIN, 0, 1)
    sigma = (STAR_SIZE_MIN + (STAR_SIZE_MAX - STAR_SIZE_MIN) * brightness).astype(np.float32)

    temp_K = np.clip(5778.0 * np.power(mass, 0.57), 2000, 50000)
    colors = blackbody_rgb(temp_K)
    colors = STAR_COLOR_SATURATION * colors + (1 - STAR_COLOR_SATURATION) * np.ones_like(colors)

    offs = np.arange(-STAR_PATCH_R, STAR_PATCH_R + 1, dtype=np.float32)
    dy_grid, dx_grid = np.meshgrid(offs, offs, indexing="ij")
    dy, dx = dy_grid.ravel(), dx_grid.ravel()
    d2 = dx[None, :] ** 2 + dy[None, :] ** 2
    vals = brightness[:, None] * np.exp(-d2 / (2 * sigma[:, None] ** 2))
    return dict(tex_w=tex_w, tex_h=tex_h, coarse_u8=coarse_u8, cx=cx, cy=cy, colors=np.ascontiguousarray(colors, dtype=np.float32),
                vals=np.ascontiguousarray(vals, dtype=np.float32), dx=dx, dy=dy)


def pillow_bilinear_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (src/libImaging/Resample.c):
    -> (k (out_size, ksize) int32 weights with 22 fractional bits, bounds (out_size, 2) int32 = first source index,
    tap count).  tests/test_golden_host.py checks the two-pass resize built from these against Image.resize."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        w = 1.0 - np.abs((np.arange(n) + xmin - center + 0.5) * ss)
        w = np.where(w > 0.0, w, 0.0)
        tot = w.sum()
        kk[xx, :n] = w / tot if tot != 0.0 else w
        bounds[xx] = (xmin, n)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << 22), 0.5 + kk * (1 << 22)).astype(np.int64)     # C (int): toward zero
    return np.ascontiguousarray(fixed.astype(np.int32)), bounds


def rasterize_sky(t: dict) -> np.ndarray:
    """Host form of the per-texel work (the reference's own NumPy / Pillow calls): nebula upsampled through 8-bit
    with Image.resize, star blobs through np.add.at.  (tex_h, tex_w, 3) float32, before the glow."""
    from PIL import Image
    tex_w, tex_h = t["tex_w"], t["tex_h"]
    sky = np.full((tex_h, tex_w, 3), 0.003, dtype=np.float32)
    sky += np.array(Image.fromarray(t["coarse_u8"]).resize((tex_w, tex_h), Image.Resampling.BILINEAR)) / 255.0 * 0.04
    cx, cy, dx, dy, vals, colors = t["cx"], t["cy"], t["dx"], t["dy"], t["vals"], t["colors"]
    n_patch = len(dy)
    px = (cx[:, None] + dx[None, :]).astype(int) % tex_w          # wrap in u
    py = (cy[:, None] + dy[None, :]).astype(int)                  # clip in v
    ok = (py >= 0) & (py < tex_h)
    contrib = np.repeat(colors, n_patch, axis=0)[ok.ravel()] * vals[ok][:, None]
    np.add.at(sky, (py[ok], px[ok]), contrib)
    return sky


def generate_skybox(tex_w: int = 2048, tex_h: int = 1024, seed: int = 42, n_stars: int = 6000, glow: bool = True) -> np.ndarray:
    """(tex_h, tex_w, 3) float32 in [0, 1], seamless in u -- entirely on the host (the reference's result, pinned by
    tests/golden/skybox.npz; the product builds the same texture on the device: HipRenderer.build_procedural_skybox).
    ``glow=False`` stops before the Milky-Way glow and the final clip."""
    sky = rasterize_sky(sky_tables(tex_w, tex_h, seed, n_stars))
    if not glow:
        return sky

    # Milky-Way glow in galactic coordinates
    v_grid = np.linspace(0, np.pi, tex_h)
    u_grid = np.linspace(0, 2 * np.pi, tex_w)
    uu, vv = np.meshgrid(u_grid, v_grid)
    dec = np.pi / 2 - vv
    b = np.arcsin(np.clip(_galactic_latitude_sin(dec, uu), -1, 1))
    sin_l_cos_b = (np.cos(dec) * np.cos(_GAL_INCLINATION) * np.sin(uu - _GAL_CENTER_RA)
                   + np.sin(dec) * np.sin(_GAL_INCLINATION))
    cos_l_cos_b = np.cos(dec) * np.cos(uu - _GAL_CENTER_RA)
    lon = np.arctan2(sin_l_cos_b, cos_l_cos_b)
    glow = MILKY_WAY_GLOW * np.exp(-0.5 * (b / np.radians(6)) ** 2)
    glow += GALACTIC_CENTER_GLOW * np.exp(-0.5 * (lon ** 2 + b ** 2) / np.radians(15) ** 2)
    arms = 0.4 + 0.6 * (0.5 + 0.5 * np.cos(4 * lon + np.radians(30)))
    near_plane = np.exp(-0.5 * (b / np.radians(8)) ** 2)
    glow *= (1.0 - near_plane) + near_plane * arms
    sky += glow[:, :, None] * np.array([1.0, 0.95, 0.85])

    return np.clip(sky, 0, 1)


def load_or_generate_skybox(skybox_path: Optional[str], tex_w: int = 2048, tex_h: int = 1024,
                            n_stars: int = 6000, glow: bool = True) -> Tuple[np.ndarray, int, int]:
    """Image file if it exists, otherwise the procedural sky (render.py:344-368).
    Returns (texture, tex_h, tex_w).  ``glow=False``: see generate_skybox; an image file is returned as it is
    and ``load_or_generate_skybox.procedural`` tells the caller which of the two it got."""
    load_or_generate_skybox.procedural = not (skybox_path and os.path.isfile(skybox_path))
    if skybox_path and os.path.isfile(skybox_path):
        from PIL import Image
        print(f"Loading skybox: {skybox_path}")
        tex = np.array(Image.open(skybox_path).convert("RGB"), dtype=np.float32) / 255.0
        tex_h, tex_w = tex.shape[:2]
        return tex, tex_h, tex_w
    if skybox_path:
        print(f"Texture not found: {skybox_path}, generating procedural skybox...")
    else:
        print("Generating procedural skybox...")
    return generate_skybox(tex_w=tex_w, tex_h=tex_h, n_stars=n_stars, glow=glow), tex_h, tex_w
