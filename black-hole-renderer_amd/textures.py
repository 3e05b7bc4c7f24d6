"""Host-side disk-texture helpers on the render path (render.py:437-459, 1113-1149)."""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import numpy as np


def compute_edge_alpha(height: int, inner_soft: float = 0.1, outer_soft: float = 0.3) -> np.ndarray:
    """Radial edge softening: cubic ramp over the inner 10 %, quadratic over the outer 30 %
    (render.py:437-445).  f32 in, f32 out."""
    v = np.linspace(0, 1, height).astype(np.float32)
    alpha = np.ones_like(v)
    lo = v < inner_soft
    hi = v > (1 - outer_soft)
    alpha[lo] = (v[lo] / inner_soft) ** 3.0
    alpha[hi] = ((1 - v[hi]) / outer_soft) ** 2
    return alpha


def load_disk_texture(path: Optional[str]) -> Optional[np.ndarray]:
    """RGB image file -> (h, w, 4) f32 with the edge alpha in channel 3 (render.py:448-459)."""
    if not (path and os.path.isfile(path)):
        return None
    from PIL import Image
    print(f"Loading disk texture: {path}")
    rgb = np.array(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    h, w = rgb.shape[:2]
    alpha = np.repeat(compute_edge_alpha(h).astype(np.float32)[:, None], w, axis=1)
    return np.concatenate([rgb, alpha[:, :, None]], axis=2)


def generate_disk_mipmaps(base_tex: np.ndarray, levels: int = 4) -> List[np.ndarray]:
    """NumPy twin of the device mip chain: 2x2 box means with floor halving (render.py:1113-1125)."""
    chain = [base_tex.copy()]
    for _ in range(levels):
        src = chain[-1]
        h, w = src.shape[:2]
        if h < 2 or w < 2:
            break
        nh, nw = h // 2, w // 2
        s = src[:nh * 2, :nw * 2]
        chain.append(((s[0::2, 0::2] + s[1::2, 0::2] + s[0::2, 1::2] + s[1::2, 1::2]) / 4.0).astype(np.float32))
    return chain


def compute_disk_texture_resolution(width: int, height: int, cam_pos, fov: float, r_inner: float,
                                    r_outer: float, rs: float = 1.0) -> Tuple[int, int]:
    """(n_phi, n_r) from the disk's angular size on screen (render.py:1128-1149):
    about one azimuthal texel per covered pixel column, half a radial texel per covered row,
    floors of 256 / 128, both rounded up to multiples of 16."""
    dist = math.sqrt(cam_pos[0] ** 2 + cam_pos[1] ** 2 + cam_pos[2] ** 2)
    ang_radius = math.atan(r_outer / dist)
    fov_rad = fov * math.pi / 180.0
    n_phi = max(256, int(width * ((2 * ang_radius) / fov_rad)))
    n_r = max(128, int(height * (ang_radius / fov_rad) * 0.5))
    n_phi += (16 - n_phi % 16) % 16
    n_r += (16 - n_r % 16) % 16
    return n_phi, n_r


def keplerian_omega_rows(n_r: int, r_inner: float, r_outer: float) -> np.ndarray:
    """omega(r) = sqrt(0.5 / (r^3 + 1e-6)) on linspace(r_inner, r_outer, n_r), f32 (render.py:3517-3519)."""
    r_norm = np.linspace(0, 1, n_r)
    r_vals = r_inner + (r_outer - r_inner) * r_norm
    return np.sqrt(0.5 / (r_vals ** 3 + 1e-6)).astype(np.float32)
