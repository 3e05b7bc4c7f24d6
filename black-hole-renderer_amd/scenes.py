"""Deterministic synthetic scenes shared by the parity tests, smoke() and bench.py.

No file or reference access: textures are analytic (SURVEY.md 8d "texture-free variant")
or seeded NumPy noise, so the same inputs exist in the build container and on the GPU box.
"""
import numpy as np

from .textures import compute_edge_alpha


def analytic_skybox(tex_h=256, tex_w=512):
    """Direction-to-RGB ramp with a few sharp features (so lensing errors show up)."""
    v, u = np.meshgrid(np.linspace(0, 1, tex_h, dtype=np.float32), np.linspace(0, 1, tex_w, dtype=np.float32),
                       indexing="ij")
    sky = np.stack([0.25 + 0.25 * np.sin(2 * np.pi * 3 * u) * np.sin(np.pi * v),
                    0.30 + 0.20 * np.cos(2 * np.pi * 5 * u),
                    0.20 + 0.30 * v], axis=-1).astype(np.float32)
    grid = ((np.floor(u * 64) + np.floor(v * 32)) % 2).astype(np.float32)
    sky *= (0.6 + 0.4 * grid)[..., None]
    return np.ascontiguousarray(sky)


def analytic_disk(n_r=128, n_phi=512, edge=True):
    """rgba = (0.8, 0.6, 0.4, (0.5 + 0.5 sin(16 phi) sin(8 pi v)) * edge_alpha(v)).  Like every disk
    texture the reference produces (load_disk_texture, the compose kernel) the alpha is softened to
    zero at both radial edges; ``edge=False`` keeps a hard-edged disk for stress tests."""
    v, phi = np.meshgrid(np.linspace(0, 1, n_r, dtype=np.float32),
                         np.linspace(0, 2 * np.pi, n_phi, endpoint=False, dtype=np.float32), indexing="ij")
    a = 0.5 + 0.5 * np.sin(16 * phi) * np.sin(8 * np.pi * v)
    if edge:
        a = a * compute_edge_alpha(n_r)[:, None]
    tex = np.stack([np.full_like(a, 0.8), np.full_like(a, 0.6), np.full_like(a, 0.4), a], axis=-1)
    return np.ascontiguousarray(tex.astype(np.float32))


def noisy_disk(n_r=128, n_phi=512, seed=7, edge=True):
    """Smooth-ish seeded RGBA texture with texel-scale detail (exercises the LOD path)."""
    rng = np.random.default_rng(seed)
    base = analytic_disk(n_r, n_phi, edge=edge)
    base[..., :3] *= 0.5 + 0.5 * rng.random((n_r, n_phi, 1), dtype=np.float32)
    base[..., 3] = np.clip(base[..., 3] * (0.6 + 0.4 * rng.random((n_r, n_phi), dtype=np.float32)), 0, 1)
    return np.ascontiguousarray(base)


def star_skybox(tex_h=256, tex_w=512, n=400, seed=3):
    """Dark sky with single-texel stars: the worst case for lensing sensitivity."""
    rng = np.random.default_rng(seed)
    sky = np.full((tex_h, tex_w, 3), 0.01, dtype=np.float32)
    ys, xs = rng.integers(0, tex_h, n), rng.integers(0, tex_w, n)
    sky[ys, xs] = rng.random((n, 3), dtype=np.float32)
    return sky


# name -> renderer kwargs + camera; sizes a single CPU thread can check in seconds
SCENES = {
    # BASELINE.json configs[1] camera (fhd default scene) at reduced resolution
    "default": dict(width=320, height=180, cam_pos=[6, 0, 0.5], fov=90, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0, anti_alias="disabled")),
    # tests/e2e_render.py:27-43 scene (configs[0])
    "e2e": dict(width=320, height=180, cam_pos=[6, 0, 0.5], fov=60, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15.0, anti_alias="disabled")),
    # configs[2]: tilt 25, lod_radius anti-aliasing
    "tilt_aa": dict(width=256, height=144, cam_pos=[6, 0, 0.5], fov=90, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=25.0, anti_alias="lod_radius",
        aa_strength=1.0)),
    # compare_aa.py:43 scene: far camera, strong minification
    "far_aa": dict(width=256, height=144, cam_pos=[-20, 0, 2], fov=60, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=20.0, anti_alias="lod_radius",
        aa_strength=1.5)),
    # configs[3] step size, camera off-axis, ragged size (not a multiple of the 8x8 tile)
    # camera on the polar axis: build_camera's fallback basis (render.py:108-111), disk seen face-on
    "polar": dict(width=96, height=64, cam_pos=[0, 0, 8], fov=70, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=6.0, disk_tilt=0.0, anti_alias="lod_radius")),
    # camera between the disk's inner and outer radius, just above the plane: several crossings per ray
    "inside": dict(width=120, height=80, cam_pos=[3.2, 0.5, 0.12], fov=100, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=9.0, disk_tilt=3.0, anti_alias="disabled")),
    # narrower than one bloom tap (R = int(0.02 W) = 0) and shorter than a tile
    "sliver": dict(width=40, height=3, cam_pos=[6, 0, 0.5], fov=90, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0, anti_alias="disabled")),
    "one_pixel": dict(width=1, height=1, cam_pos=[6, 0, 0.5], fov=20, kw=dict(
        step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0, anti_alias="lod_radius")),
    "fine_ragged": dict(width=203, height=117, cam_pos=[4, 3, 1.5], fov=75, kw=dict(
        step_size=0.05, r_max=10.0, r_disk_inner=2.0, r_disk_outer=8.0, disk_tilt=5.0, anti_alias="disabled")),
}
