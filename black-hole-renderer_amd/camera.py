"""Pin-hole camera of the reference (render.py:93-127) and the orbit path of its video driver."""
from __future__ import annotations

import math

import numpy as np


def build_camera(cam_pos, fov_deg: float, width: int, height: int):
    """Camera at ``cam_pos`` looking at the origin, +z up, square pixels.

    Returns ``(cam_pos, cam_right, cam_up, cam_forward, pixel_width, pixel_height)`` in f64,
    the tuple ``TaichiRenderer.render`` casts to f32 (render.py:3880-3891).  The image plane
    sits at unit distance, so its height is ``2 tan(fov/2)``.
    """
    eye = np.array(cam_pos, dtype=np.float64)
    forward = -eye / np.linalg.norm(eye)
    right = np.cross(forward, np.array([0.0, 0.0, 1.0]))
    length = np.linalg.norm(right)
    if length < 1e-6:          # looking straight down the z axis
        right = np.array([1.0, 0.0, 0.0])
    else:
        right /= length
    up = np.cross(right, forward)
    up /= np.linalg.norm(up)

    plane_h = 2.0 * np.tan(np.radians(fov_deg) / 2)
    plane_w = plane_h * (width / height)
    return eye, right, up, forward, plane_w / width, plane_h / height


def escape_radius(cam_pos, r_max: float) -> float:
    """r_escape = max(r_max, 2 |cam|)  (render.py:3883-3884)."""
    return max(r_max, float(np.linalg.norm(np.array(cam_pos, dtype=np.float64))) * 2)


def orbit_position(static_cam_pos, frame: int, n_frames: int, orbit_degrees: float = 360.0):
    """Camera of frame ``frame`` on the horizontal orbit of render_video (render.py:4408, 4440-4446)."""
    radius = float(np.linalg.norm(static_cam_pos))
    angle = np.radians(frame * (orbit_degrees / n_frames))
    return [radius * np.cos(angle), radius * np.sin(angle), static_cam_pos[2]]
