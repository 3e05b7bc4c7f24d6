#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's importable NumPy code.

Run in the build container only (needs /root/reference; the GPU box never sees it):
    python tests/golden/make_golden.py

The reference's render.py imports taichi / imageio at module level (render.py:26-29); neither is
installed here and none of the functions called below touches them, so empty placeholder modules
are registered first.  No Taichi kernel is executed -- the device code cannot
run in this container; these vectors pin the host-side helpers, the NumPy twin of the compose +
mip kernels (the comparison tests/unit/test_gpu_texture_compose.py makes), the lifecycle
producer, the lens flare and disk_v2.  Only inputs and outputs are stored, never source text.
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ("taichi", "imageio", "imageio.v3"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["imageio"].v3 = sys.modules["imageio.v3"]
    sys.path.insert(0, REF)
    import render as ref  # noqa
    return ref


class _Field:
    """Minimal stand-in for a Taichi field: holds a NumPy array."""

    def __init__(self, arr=None):
        self.arr = arr

    def to_numpy(self):
        return self.arr

    def from_numpy(self, a):
        self.arr = np.array(a)


def main():
    ref = import_reference()
    g = {}

    # ---- camera (render.py:93-127) ------------------------------------------------
    cams = [([6, 0, 0.5], 90, 1920, 1080), ([6, 0, 0.5], 60, 320, 180), ([6, 0, 0.5], 90, 3840, 2160),
            ([6, 0, 0.5], 90, 7680, 4320), ([-20, 0, 2], 60, 640, 360), ([0, 0, 8], 75, 640, 360),
            ([4, 3, 1.5], 75, 203, 117), ([0.0, 6.0, 0.5], 90, 1280, 720)]
    cam_in, cam_out = [], []
    for pos, fov, w, h in cams:
        p, r, u, f, pw, ph = ref.build_camera(np.array(pos, dtype=np.float64), fov, w, h)
        cam_in.append(pos + [fov, w, h])
        cam_out.append(np.concatenate([p, r, u, f, [pw, ph]]))
    np.savez(os.path.join(OUT, "camera.npz"), inputs=np.array(cam_in, dtype=np.float64),
             outputs=np.array(cam_out, dtype=np.float64))

    # ---- texture helpers --------------------------------------------------------------
    res_in = [(320, 180, [6, 0, 0.5], 60, 2.0, 3.5), (640, 360, [6, 0, 0.5], 90, 2.0, 15.0),
              (1920, 1080, [6, 0, 0.5], 90, 2.0, 15.0), (3840, 2160, [6, 0, 0.5], 90, 2.0, 15.0),
              (7680, 4320, [6, 0, 0.5], 90, 2.0, 15.0), (640, 360, [20, 0, 2], 60, 2.0, 15.0)]
    res_out = [ref.compute_disk_texture_resolution(*a) for a in res_in]
    rng = np.random.default_rng(5)
    mip_base = rng.random((32, 64, 4), dtype=np.float32)
    mips = ref.generate_disk_mipmaps(mip_base, levels=4)
    T = np.array([1000, 1900, 2000, 3500, 5000, 6000, 6500, 6600, 6700, 9000, 15000, 40000], dtype=np.float64)
    np.savez(os.path.join(OUT, "texture_helpers.npz"),
             res_in=np.array([[a[0], a[1], *a[2], a[3], a[4], a[5]] for a in res_in], dtype=np.float64),
             res_out=np.array(res_out, dtype=np.int64),
             edge_128=ref.compute_edge_alpha(128), edge_17=ref.compute_edge_alpha(17),
             mip_base=mip_base, **{f"mip_{i}": m for i, m in enumerate(mips)},
             bb_T=T, bb_rgb=ref._blackbody_rgb(T))

    # ---- skybox (render.py:153-341) ---------------------------------------------------
    small = ref.generate_skybox(64, 32, seed=42, n_stars=10)
    big = ref.generate_skybox(2048, 1024, seed=42, n_stars=6000)
    srng = np.random.default_rng(0)
    ys, xs = srng.integers(0, 1024, 64), srng.integers(0, 2048, 64)
    np.savez(os.path.join(OUT, "skybox.npz"), small=small, big_sha256=hashlib.sha256(big.tobytes()).hexdigest(),
             big_ys=ys, big_xs=xs, big_samples=big[ys, xs], big_mean=big.mean(axis=(0, 1)),
             big_dtype=str(big.dtype))

    # ---- compose kernel + mip chain: NumPy twin (render.py:852-1021, 1113-1125) -------
    n_phi, n_r = 256, 128
    state = ref.build_disk_texture_rotating_state(n_phi=n_phi, n_r=n_r, seed=42, r_inner=2.0, r_outer=15.0,
                                                  enable_rt=True, generation_scale=1)
    comp = np.stack([state.temp_base, state.spiral, state.spiral_temp, state.turbulence, state.turb_temp,
                     state.arcs, state.arcs_temp, state.rt_spikes, state.rt_temp, state.hotspot,
                     state.hotspot_temp, state.az_hotspot, state.disturb_mod], axis=0).astype(np.float32)
    # statistics exactly as upload_parametric_state computes them (render.py:2361-2383)
    rt_weight = 0.20
    density = (0.15 + 0.10 * state.spiral + 0.30 * state.turbulence + 0.20 * state.hotspot + 0.30 * state.arcs
               + rt_weight * state.rt_spikes) * state.disturb_mod
    density *= state.edge[:, None]
    p98 = float(np.percentile(density, 98))
    ts = (state.spiral_temp + state.turb_temp + state.arcs_temp + state.rt_temp + state.hotspot_temp) * state.disturb_mod
    scale = float(np.percentile(ts[ts > 0], 95))
    tss = np.clip(ts / (scale + 1e-6) * 0.8, 0, 1.2)
    row_stats = np.stack([np.max(tss, axis=1), np.quantile(tss, 0.7, axis=1)], axis=1).astype(np.float32)
    texs = {f"tex_t{t}": ref._generate_disk_texture_rotating_from_state(state, t_offset=float(t))
            for t in (0, 5, 50, 180)}
    cmips = ref.generate_disk_mipmaps(texs["tex_t5"], levels=4)
    np.savez_compressed(os.path.join(OUT, "compose.npz"), comp=comp,
                        omega_rows=state.omega_rows, edge=state.edge, stats=np.array([p98, scale], dtype=np.float32),
                        row_stats=row_stats, color_temp=np.float32(state.color_temp),
                        **{k: v[::4] for k, v in texs.items() if k != "tex_t5"}, tex_t5=texs["tex_t5"],
                        **{f"mip5_{i}": m for i, m in enumerate(cmips) if i > 0})

    # ---- lifecycle: factories, rasterisation, statistics (render.py:624-792, 3564-3712, 4098-4123)
    ln_r, ln_phi, r_in, r_out = 48, 96, 2.0, 15.0
    r_norm_all = np.linspace(0, 1, ln_r)
    r_vals = r_in + (r_out - r_in) * r_norm_all
    omega_all = np.sqrt(0.5 / (r_vals ** 3 + 1e-6)).astype(np.float32)
    specs = {"filament": (ref._spawn_single_filament, 200, (15.0, 60.0), 0.0, 0.0, 142),
             "hotspot": (ref._spawn_single_hotspot, 30, (15.0, 30.0), 4.0, 4.0, 242),
             "rt_spike": (ref._spawn_single_rt_spike, 15, (15.0, 30.0), 3.0, 3.0, 342)}
    factories = {k: ref.EntityFactory(fn, target_count=n, lifetime_range=lr, fade_in=fi, fade_out=fo, n_r=ln_r,
                                      n_phi=ln_phi, r_norm_all=r_norm_all, omega_all=omega_all, seed=sd,
                                      entity_type=k)
                 for k, (fn, n, lr, fi, fo, sd) in specs.items()}
    for f in factories.values():
        f.seed_initial(now=0.0)

    fake = types.SimpleNamespace(_bg_n_r=ln_r, _bg_n_phi=ln_phi, _bg_omega_all_np=omega_all,
                                 _bg_r_norm_all=r_norm_all, _entity_staging_field=_Field(),
                                 _copy_entity_staging_to_comp=lambda a, b: None, _comp_field=None)
    life = {}

    def snapshot(tag, now):
        ref.TaichiRenderer.accumulate_entity_layer(fake, factories, now)
        life[f"staging_{tag}"] = fake._entity_staging_field.arr.copy()
        for k, f in factories.items():
            life[f"birth_{k}_{tag}"] = np.array([e.birth_time for e in f.entities])
            life[f"lifetime_{k}_{tag}"] = np.array([e.lifetime for e in f.entities])
            life[f"omega_{k}_{tag}"] = np.array([e.omega for e in f.entities])
            life[f"nrows_{k}_{tag}"] = np.array([len(e.row_indices) for e in f.entities])
            life[f"fade_noise_sum_{k}_{tag}"] = np.array([float(e.fade_noise.sum()) for e in f.entities])

    snapshot("t0", 0.0)
    # 120 video ticks at dt = 0.1 (render.py:4437-4457), then a later snapshot
    for fr in range(1, 121):
        for f in factories.values():
            f.tick(now=fr * 0.1, dt=0.1)
    snapshot("t12", 12.0)
    for fr in range(121, 1201):
        for f in factories.values():
            f.tick(now=fr * 0.1, dt=0.1)
    snapshot("t120", 120.0)

    # statistics on a synthetic 13-plane field
    crng = np.random.default_rng(11)
    scomp = crng.random((13, ln_r, ln_phi), dtype=np.float32)
    scomp[[2, 4, 6, 8, 10]] *= 0.1
    scomp[5:11] = life["staging_t12"]
    sfake = types.SimpleNamespace(_comp_field=_Field(scomp), _edge_field=_Field(ref.compute_edge_alpha(ln_r)),
                                  _param_enable_rt=1, _param_stats_field=_Field(), _param_row_stats_field=_Field())
    ref.TaichiRenderer.recompute_interactive_stats(sfake)
    np.savez_compressed(os.path.join(OUT, "lifecycle.npz"), n_r=ln_r, n_phi=ln_phi, r_inner=r_in, r_outer=r_out,
                        stats_comp=scomp, stats_out=sfake._param_stats_field.arr,
                        row_stats_out=sfake._param_row_stats_field.arr, **life)

    # ---- init_background_layer's RNG draws (render.py:3503-3511) -------------------
    az = []
    for seed in (42, 7, 123):
        r = np.random.default_rng(seed)
        az.append([seed, int(r.integers(2, 5)), float(r.uniform(2.0, 4.0))])
    # ---- lens flare (render.py:3925-4028), arrays are (W, H, 3) -------------------
    W, H = 64, 36
    xx, yy = np.mgrid[0:W, 0:H]
    disk = np.zeros((W, H, 3), dtype=np.float32)
    blob = np.exp(-((xx - 40) ** 2 + (yy - 14) ** 2) / 30.0).astype(np.float32)
    disk[..., 0], disk[..., 1], disk[..., 2] = blob, 0.7 * blob, 0.4 * blob
    final = np.clip(0.05 + disk, 0, 1).astype(np.float32)
    flare_out = ref.TaichiRenderer._apply_lens_flare(None, final.copy(), disk.copy())
    dark = ref.TaichiRenderer._apply_lens_flare(None, final.copy(), np.zeros_like(disk))
    np.savez_compressed(os.path.join(OUT, "misc.npz"), az_draws=np.array(az), flare_final=final, flare_disk=disk,
                        flare_out=flare_out, flare_dark_out=dark)

    # ---- orbit camera (render.py:4408, 4440-4446) ------------------------------------
    orb = []
    static = [6.0, 0.0, 0.5]
    radius = float(np.linalg.norm(static))
    for n_frames, deg in ((3600, 360.0), (8, 90.0), (10, -180.0)):
        step = deg / n_frames
        for fr in (0, 1, n_frames // 2, n_frames - 1):
            a = np.radians(fr * step)
            orb.append([n_frames, deg, fr, radius * np.cos(a), radius * np.sin(a), static[2]])
    np.savez(os.path.join(OUT, "orbit.npz"), rows=np.array(orb, dtype=np.float64))

    # ---- disk_v2 analytic model (numpy only, imported as shipped) -----------------------
    import disk_v2 as dv
    P = dv.DiskV2Params()
    r = np.linspace(1.5, 11.0, 64)
    Hh = dv.disk_half_thickness(r, P)
    zf = np.array([0.0, 0.25, -0.25, 1.0, -1.0, 1.1])
    rr = np.repeat(r[:, None], len(zf), axis=1)
    zz = zf[None, :] * np.asarray(Hh)[:, None]
    d2 = dict(r=r, zf=zf, H=Hh, smooth=dv.geometry.smoothstep(0.0, 1.0, np.linspace(-0.5, 1.5, 41)),
              mask_r=dv.disk_radial_mask(r, P), W_r=dv.disk_radial_weight(r, P),
              W_z=dv.disk_vertical_weight(rr, zz, P), mask_vol=dv.disk_volume_mask(rr, zz, P),
              omega=dv.angular_velocity_field(r, P), rho_mid=dv.midplane_density_field(r, P),
              T_mid=dv.midplane_temperature_field(r, P), rho=dv.density_field(rr, zz, P),
              T=dv.temperature_field(rr, zz, P),
              scalar_probe=np.array([dv.disk_half_thickness(3.0, P), dv.disk_radial_weight(2.0, P),
                                     dv.disk_radial_weight(10.0, P), dv.angular_velocity_field(2.0, P),
                                     dv.midplane_temperature_field(2.0, P), dv.density_field(4.0, 0.0, P)],
                                    dtype=np.float64))
    phi = np.linspace(0, 2 * np.pi, 96, endpoint=False)
    rg = np.linspace(1.5, 11.0, 48)
    rg2, pg2 = np.meshgrid(rg, phi, indexing="ij")
    d2["F_mode"] = dv.weak_mode_modulation(rg2, pg2, P)
    for seed in (7, 42, 123):
        d2[f"F_shear_{seed}"] = dv.shear_modulation(rg2, pg2, P, seed=seed)
        d2[f"F_hotspot_{seed}"] = dv.hotspot_modulation(rg2, pg2, P, seed=seed)
        d2[f"F_total_{seed}"] = dv.structure_modulation(rg2, pg2, P, seed=seed)
    d2["rg"], d2["phig"] = rg, phi
    np.savez_compressed(os.path.join(OUT, "disk_v2.npz"), **d2)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
