"""TEST INFRASTRUCTURE -- NumPy restatement of the reference's lens flare (render.py:3925-4028),
pinned bit for bit by tests/golden/misc.npz (generated from the reference's _apply_lens_flare).
The product path is the device kernel in csrc/flare.hip; only tests/ may import this module.


The reference works on (W, H, 3) arrays; frames here are (H, W, 3).  The effect is evaluated
on a (W, H, 3) copy so that every reduction runs over the same memory order as the reference
and the result is reproducible against its output.
"""
from __future__ import annotations

import numpy as np

_GHOST_TINT = np.array([1.0, 0.9, 0.7])
_RING_TINTS = (np.array([0.3, 0.4, 1.0]), np.array([0.5, 0.5, 0.9]), np.array([0.7, 0.5, 0.8]))
_HEX_TINT = np.array([0.6, 0.7, 1.0])
_STREAK_TINT = np.array([1.0, 0.95, 0.9])


def _flare_wh(final: np.ndarray, disk: np.ndarray) -> np.ndarray:
    """final, disk: (W, H, 3).  Returns clip(final + flare, 0, 1)."""
    w, h, _ = final.shape
    scale = min(w, h) / 360.0                      # sizes are quoted for a 360-line frame

    glow = np.max(disk, axis=2)                    # per-pixel disk brightness, (W, H)
    total = np.sum(glow)
    if total < 0.01:
        return final
    xs, ys = np.mgrid[0:w, 0:h]
    src_x = np.sum(xs * glow) / total              # brightness centroid of the disk layer
    src_y = np.sum(ys * glow) / total
    mid_x, mid_y = w / 2, h / 2
    strength = min(total / (w * h * 0.3), 1.0) * 1.5

    def along_axis(t):
        """Point at fraction t on the line source -> screen centre."""
        return src_x + (mid_x - src_x) * t, src_y + (mid_y - src_y) * t

    flare = np.zeros((w, h, 3), dtype=np.float32)

    # eight ghosts marching towards the centre, quadratic falloff
    for g in range(8):
        gx, gy = along_axis((g + 1) * 0.15)
        radius = (25 + g * 30) * scale
        dist = np.sqrt((xs - gx) ** 2 + (ys - gy) ** 2)
        inside = dist < radius
        alpha = np.zeros((w, h), dtype=np.float32)
        alpha[inside] = (1 - dist[inside] / radius) ** 2 * (1 - g * 0.08) * strength
        for c in range(3):
            flare[:, :, c] += alpha * _GHOST_TINT[c]

    # three diffraction rings with slightly different tints
    for k in range(3):
        rx, ry = along_axis(0.35 + k * 0.15)
        ring_r = (60 + k * 40) * scale
        ring_w = (6 + k * 3) * scale
        dist = np.sqrt((xs - rx) ** 2 + (ys - ry) ** 2)
        off = np.abs(dist - ring_r)
        alpha = np.clip(1 - off / ring_w, 0, 1) ** 2 * 0.5 * strength * (1 - k * 0.25)
        for c in range(3):
            flare[:, :, c] += alpha * _RING_TINTS[k][c]

    # hexagonal aperture ring half way
    hx, hy = along_axis(0.5)
    dx, dy = xs - hx, ys - hy
    angle = np.arctan2(dy, dx)
    dist = np.sqrt(dx ** 2 + dy ** 2)
    edge = np.abs(np.mod(angle, np.pi / 3) - np.pi / 6)
    facet = np.clip(1 - edge / 0.2, 0, 1)
    off = np.abs(dist - 100 * scale)
    alpha = np.clip(1 - off / (15 * scale), 0, 1) ** 2 * facet * 0.3 * strength
    for c in range(3):
        flare[:, :, c] += alpha * _HEX_TINT[c]

    # four streaks through the source
    reach = min(w, h) * 0.4
    gain = strength * 0.3
    dx, dy = xs - src_x, ys - src_y
    dist = np.sqrt(dx ** 2 + dy ** 2)
    angle = np.arctan2(dy, dx)
    for axis in [0, np.pi / 2, np.pi, 3 * np.pi / 2]:
        delta = np.abs(np.mod(angle - axis + np.pi, 2 * np.pi) - np.pi)
        on = delta < 0.05
        falloff = np.exp(-dist / reach)
        for c in range(3):
            flare[:, :, c] += np.where(on, falloff * gain * _STREAK_TINT[c], 0)

    return np.clip(final + flare, 0, 1)


def apply_lens_flare(final_hw3: np.ndarray, disk_hw3: np.ndarray) -> np.ndarray:
    """(H, W, 3) frame + (H, W, 3) disk layer -> (H, W, 3) frame with the flare added."""
    final = np.ascontiguousarray(final_hw3.transpose(1, 0, 2))
    disk = np.ascontiguousarray(disk_hw3.transpose(1, 0, 2))
    out = _flare_wh(final, disk)
    return np.ascontiguousarray(out.transpose(1, 0, 2))
