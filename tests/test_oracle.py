"""Pins the CPU oracle (oracle/bhr_oracle.c).  CPU only.

* compose + mip chain: against outputs of the reference's NumPy twin (tests/golden/compose.npz),
  with the tolerances the reference's own test uses (test_gpu_texture_compose.py:154-189, 229-262);
* noise / background: the reference's property tests restated (test_simplex_noise.py:30-146,
  test_background_layer.py:76-143);
* march: physics known-answer tests (the reference holds no numeric vector for it -- its
  e2e pin is an MD5 of float bytes, tests/e2e_baseline.txt).
"""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# --------------------------------------------------------------------------- compose + mips
@pytest.fixture(scope="module")
def compose_golden():
    return np.load(os.path.join(G, "compose.npz"))


@pytest.mark.parametrize("t", [0, 5, 50, 180])
def test_compose_matches_reference_numpy(oracle, compose_golden, t):
    d = compose_golden
    tex = oracle.compose_disk_texture(d["comp"], d["omega_rows"], d["edge"], d["stats"], d["row_stats"],
                                      float(t), 1, float(d["color_temp"]))
    want = d[f"tex_t{t}"]
    got = tex if t == 5 else tex[::4]
    assert np.max(np.abs(got - want)) < 1e-4


def test_mips_match_reference_numpy(oracle, compose_golden):
    d = compose_golden
    mips = oracle.build_mips_padded(d["tex_t5"])
    assert mips.shape[0] == 5
    np.testing.assert_array_equal(mips[0], d["tex_t5"])
    for lev in range(1, 5):
        want = d[f"mip5_{lev}"]
        h, w = want.shape[:2]
        assert np.max(np.abs(mips[lev, :h, :w] - want)) < 1e-3
        assert not mips[lev, h:].any() and not mips[lev, :, w:].any()   # padding stays zero


# --------------------------------------------------------------------------- noise properties
def test_simplex_range_continuity_and_fbm1(oracle):
    rng = np.random.default_rng(0)
    pts = (rng.random((20000, 3)) * 200 - 100).astype(np.float32)
    n = oracle.eval_noise(pts, "simplex")
    assert np.isfinite(n).all() and n.min() >= -1.0 - 1e-3 and n.max() <= 1.0 + 1e-3
    assert n.std() > 0.1
    eps = np.float32(1e-3)
    n2 = oracle.eval_noise(pts + eps, "simplex")
    assert np.max(np.abs(n2 - n)) < 0.05                     # continuity
    f1 = oracle.eval_noise(pts, "fbm", octaves=1, persistence=0.5, lacunarity=2.0)
    np.testing.assert_array_equal(f1, n)                     # fbm with one octave == simplex
    f4 = oracle.eval_noise(pts, "fbm", octaves=4, persistence=0.5, lacunarity=2.0)
    assert np.abs(f4).max() <= 1.875 + 1e-3


def test_simplex_seamless_in_phi(oracle):
    """(cos phi, sin phi) mapping: phi = 0 and 2 pi give identical noise (test_simplex_noise.py)."""
    z = np.linspace(0, 5, 64, dtype=np.float32)
    a = np.stack([np.full_like(z, np.cos(0.0) * 8), np.full_like(z, np.sin(0.0) * 8), z], axis=1)
    b = np.stack([np.full_like(z, np.float32(np.cos(2 * np.pi)) * 8), np.full_like(z, np.float32(np.sin(2 * np.pi)) * 8), z], axis=1)
    assert np.max(np.abs(oracle.eval_noise(a) - oracle.eval_noise(b))) < 1e-4


def test_background_component_ranges(oracle):
    """test_background_layer.py:76-143 restated."""
    n_r, n_phi = 32, 96
    comp = np.full((13, n_r, n_phi), -7.0, dtype=np.float32)
    oracle.generate_background(n_r, n_phi, 3, 2.5, 2.0, 15.0, 1.7, comp=comp)
    tb, turb, turb_t, az, dm = comp[0], comp[3], comp[4], comp[11], comp[12]
    assert tb.min() >= 0 and tb.max() <= 0.35
    assert (comp[1] == 0).all() and (comp[2] == 0).all()
    assert turb.min() >= 0 and turb.max() <= 1
    np.testing.assert_allclose(turb_t, 0.05 * turb, rtol=0, atol=1e-7)
    assert az.min() >= 0 and az.max() <= 1
    assert dm.min() >= 0.1 - 1e-7 and dm.max() <= 1
    assert (comp[5:11] == -7.0).all()                        # entity planes untouched
    assert turb.std() > 0.01 and dm.std() > 0.01
    # time evolution changes the field; same t reproduces it
    c2 = oracle.generate_background(n_r, n_phi, 3, 2.5, 2.0, 15.0, 1.7)
    np.testing.assert_array_equal(c2[3], turb)
    c3 = oracle.generate_background(n_r, n_phi, 3, 2.5, 2.0, 15.0, 9.0)
    assert np.abs(c3[3] - turb).max() > 1e-3


# --------------------------------------------------------------------------- march physics
def _renderer(oracle, w=64, h=64, **kw):
    sky = np.zeros((8, 16, 3), dtype=np.float32)
    sky[..., 0] = 1.0                                         # escaped rays are pure red
    tex = np.zeros((16, 32, 4), dtype=np.float32)             # transparent disk
    args = dict(step_size=0.05, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0)
    args.update(kw)
    return oracle.OracleRenderer(w, h, sky, tex, **args)


def test_capture_iff_impact_parameter_below_critical(oracle):
    """Shadow edge at b = 3 sqrt(3)/2 rs (SURVEY 9 KAT 2).  Camera far away, narrow fov: a pixel at
    angle a from the centre has impact parameter b = D sin a."""
    D, fov, n = 30.0, 20.0, 257
    R = _renderer(oracle, w=n, h=1, r_max=10.0)
    # one row through the centre: render a (n x 1)-pixel image with square pixels
    img, _ = R.march([D, 0.0, 0.0], fov)   # height 1: the pixel pitch is 2 tan(fov/2), one row through the centre
    captured = img[:, 0, 0] == 0.0
    pw = 2 * np.tan(np.radians(fov) / 2) / 1
    xs = (np.arange(n) + 0.5 - n / 2) * pw
    b = D * np.abs(xs) / np.sqrt(1 + xs ** 2)
    bc = 1.5 * np.sqrt(3.0)
    margin = 2 * D * pw
    assert captured[b < bc - margin].all()
    assert (~captured[(b > bc + margin)]).all()


def test_weak_field_deflection(oracle):
    """Deflection -> 2 rs / b for b >> rs (SURVEY 9 KAT 3).  A sky that encodes (phi/2pi, theta/pi)
    in (R, G) reads back the escape direction of a ray that passes the hole at impact parameter b."""
    th, tw = 512, 1024
    v, u = np.meshgrid(np.arange(th) / th, np.arange(tw) / tw, indexing="ij")
    sky = np.stack([u, v, np.zeros_like(u)], axis=-1).astype(np.float32)
    tex = np.zeros((16, 32, 4), dtype=np.float32)
    D = 400.0
    R = oracle.OracleRenderer(3, 3, sky, tex, step_size=0.05, r_max=10.0, r_disk_inner=2.0, r_disk_outer=3.0)
    fov = 2 * np.degrees(np.arctan(0.1))
    img, _ = R.march([D, 0.0, 0.0], fov)
    # camera on +x: forward = -x, right = +y.  Pixel (i=2, j=1) leaves along (-1, +pw, 0).
    pw = 2 * np.tan(np.radians(fov) / 2) / 3
    b = D * pw / np.sqrt(1 + pw ** 2)
    phi_in = np.pi - np.arctan(pw)
    phi_out = float(img[2, 1, 0]) * 2 * np.pi
    defl = phi_out - phi_in                                   # bent towards the hole: angle grows
    # finite source distance D and escape radius 2 D: deflection = (rs/b)(sqrt(1-(b/D)^2) + sqrt(1-(b/2D)^2))
    first = (1.0 / b) * (np.sqrt(1 - (b / D) ** 2) + np.sqrt(1 - (b / (2 * D)) ** 2))
    # the equation of motion is the exact Schwarzschild null geodesic, so the second-order term of
    # the bending angle, (15 pi / 16) (rs / b)^2, is there as well
    expected = first + (15 * np.pi / 16) * (1.0 / b) ** 2
    assert b > 20
    assert abs(defl - expected) / expected < 0.015, (defl, expected)
    assert abs(defl - first) / first > 0.03                   # ... and it is resolved
    assert abs(float(img[2, 1, 1]) * np.pi - np.pi / 2) < 1e-3   # the ray stays in the equatorial plane


def test_angular_momentum_conserved_and_step_counts(oracle):
    """L^2 = |x cross v|^2 is conserved by the central force; the RK4 march keeps it to ~1e-4."""
    R = _renderer(oracle, w=48, h=27)
    R.march([6, 0, 0.5], 90)
    steps = R.last_steps
    assert steps.min() >= 1 and steps.max() <= int(np.float32(12.041595) * 40 / 0.05) + 1
    assert R.last_total_steps == int(steps.sum())
    # default-view work model: ~72 steps/ray at 0.1 => ~144 at 0.05
    assert 120 < steps.mean() < 170


def test_transparent_disk_and_opaque_disk(oracle):
    """alpha = 0 leaves the sky untouched; alpha ~ 1 hides what is behind the disk."""
    sky = np.full((8, 16, 3), 0.5, dtype=np.float32)
    clear = np.zeros((16, 32, 4), dtype=np.float32)
    solid = np.ones((16, 32, 4), dtype=np.float32)
    kw = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0)
    a = oracle.OracleRenderer(64, 36, sky, clear, **kw)
    bgA, diskA = a.march([6, 0, 0.5], 90)
    assert not diskA.any()
    assert np.all((bgA == 0.0) | (np.abs(bgA - 0.5) < 1e-6))   # captured or (bilinear of a flat) sky
    b = oracle.OracleRenderer(64, 36, sky, solid, **kw)
    bgB, diskB = b.march([6, 0, 0.5], 90)
    hit = diskB.sum(axis=2) > 0
    assert hit.mean() > 0.3
    # 1 - (1 - 0.999)^6 == 1 in f32: fully opaque where hit
    assert np.all(bgB[hit] == 0.0)
    assert np.all(diskB <= 1.0) and np.all(diskB >= 0.0)


def test_render_composition_quirk(oracle):
    """render() = clip(bg + disk + blur) with the UN-scaled blur (render.py:3908-3918)."""
    from bhr_amd import scenes
    sky, tex = scenes.analytic_skybox(64, 128), scenes.analytic_disk(32, 64)
    R = oracle.OracleRenderer(64, 36, sky, tex, step_size=0.1, r_disk_outer=15.0)
    out, bg, disk, blur = R.render([6, 0, 0.5], 90, parts=True)
    np.testing.assert_array_equal(out, np.clip(bg + disk + blur, 0, 1).transpose(1, 0, 2))
    assert blur.max() > 0


def test_differentials_do_not_change_aa_off_pixels(oracle):
    """With anti_alias disabled the reference still integrates the differentials but never reads
    them (render.py:2957-2959): skipping them is pixel-identical."""
    from bhr_amd import scenes
    sky, tex = scenes.analytic_skybox(64, 128), scenes.noisy_disk(32, 64)
    R = oracle.OracleRenderer(48, 27, sky, tex, step_size=0.1, disk_tilt=15.0, anti_alias="disabled")
    a = R.render([6, 0, 0.5], 90, skip_differentials=False)
    b = R.render([6, 0, 0.5], 90, skip_differentials=True)
    np.testing.assert_array_equal(a, b)


# ---- Disk V2 restatement in the oracle, pinned by the reference package's own tables ----------------
def test_oracle_disk_v2_fields_match_reference_tables(oracle):
    import bhr_amd  # noqa: F401
    from bhr_amd import disk_v2 as dv
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "disk_v2.npz"))
    P = dv.DiskV2Params()
    cp = dv.pack_params(P)
    tol = dict(rtol=2e-14, atol=1e-15)
    r, zf = gold["r"], gold["zf"]
    rr = np.repeat(r[:, None], len(zf), axis=1)
    zz = zf[None, :] * gold["H"][:, None]
    ev = lambda f, *a, **k: oracle.dv2_eval(cp, f, *a, **k)
    np.testing.assert_allclose(ev(dv.F_H, r), gold["H"], **tol)
    np.testing.assert_array_equal(ev(dv.F_MASK_R, r) > 0.5, gold["mask_r"])
    np.testing.assert_allclose(ev(dv.F_W_R, r), gold["W_r"], **tol)
    np.testing.assert_allclose(ev(dv.F_W_Z, rr, zz), gold["W_z"], **tol)
    np.testing.assert_array_equal(ev(dv.F_MASK_VOL, rr, zz) > 0.5, gold["mask_vol"])
    np.testing.assert_allclose(ev(dv.F_OMEGA, r), gold["omega"], **tol)
    np.testing.assert_allclose(ev(dv.F_RHO_MID, r), gold["rho_mid"], **tol)
    np.testing.assert_allclose(ev(dv.F_T_MID, r), gold["T_mid"], **tol)
    np.testing.assert_allclose(ev(dv.F_RHO, rr, zz), gold["rho"], **tol)
    np.testing.assert_allclose(ev(dv.F_T, rr, zz), gold["T"], **tol)
    rg, pg = np.meshgrid(gold["rg"], gold["phig"], indexing="ij")
    np.testing.assert_allclose(ev(dv.F_MODE, rg, None, pg), gold["F_mode"], rtol=1e-13, atol=1e-14)
    for seed in (7, 42, 123):
        # the reference normalises each signed sum by its maximum over the evaluated array
        cs = dv.pack_params(P, None, shear_seed=seed, hotspot_seed=seed)
        sp = dv.DiskV2StructureParams()
        raw_s = oracle.dv2_eval(cs, dv.F_SHEAR, rg, None, pg)
        raw_h = oracle.dv2_eval(cs, dv.F_HOTSPOT, rg, None, pg)
        wr = oracle.dv2_eval(cs, dv.F_W_R, rg) > 0
        f_s = np.where(wr, 1.0 + sp.shear_strength * raw_s / np.abs(raw_s).max(), 1.0)
        f_h = np.where(wr, 1.0 + sp.hotspot_strength * raw_h / np.abs(raw_h).max(), 1.0)
        np.testing.assert_allclose(f_s, gold[f"F_shear_{seed}"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(f_h, gold[f"F_hotspot_{seed}"], rtol=1e-12, atol=1e-13)
        ct = dv.pack_params(P, None, shear_seed=seed, hotspot_seed=seed + 1)      # structure_modulation: seed, seed + 1
        m_s = np.abs(oracle.dv2_eval(ct, dv.F_SHEAR, rg, None, pg)).max()
        m_h = np.abs(oracle.dv2_eval(ct, dv.F_HOTSPOT, rg, None, pg)).max()
        np.testing.assert_allclose(oracle.dv2_eval(ct, dv.F_TOTAL, rg, None, pg, norm_shear=m_s, norm_hotspot=m_h),
                                   gold[f"F_total_{seed}"], rtol=1e-12, atol=1e-13)


# ---- the restated integrator against an independent one -------------------------------------------------
def test_rk4_converges_to_the_geodesic_at_fourth_order(oracle):
    """The escape directions of the binary64 build against SciPy's DOP853 on d2x/dl2 = -1.5 L^2 x / r^5 from the
    same initial rays: the error falls ~16x per halving of step_size (RK4 with a step proportional to it) and is
    below 1e-7 at step 0.025 -- the equation of motion, the RK4 tableau and the adaptive-step law are the
    reference's (render.py:2518-2524, 2858-2882), the yardstick is not."""
    from scipy.integrate import solve_ivp
    w, h, fov, cam = 8, 6, 40.0, [9.0, 2.0, 3.0]
    sky, tex = np.zeros((8, 16, 3), np.float32), np.zeros((8, 16, 4), np.float32)
    r_far = 400.0

    def directions(step):
        o = oracle.OracleRenderer(w, h, sky, tex, step_size=step, r_max=r_far, r_disk_inner=2.0, r_disk_outer=3.0, fast="f64")
        return o.escape_directions(cam, fov)

    # the same pixel -> ray construction in binary64 (render.py:2811-2822)
    p, right, up, fwd, pw, ph = oracle.build_camera(np.array(cam, dtype=np.float64), fov, w, h)
    p32 = p.astype(np.float32).astype(np.float64)
    r32, u32, f32 = (v.astype(np.float32).astype(np.float64) for v in (right, up, fwd))
    pw, ph = float(np.float32(pw)), float(np.float32(ph))
    tl = p32 + f32 - r32 * (pw * w / 2) + u32 * (ph * h / 2)
    want = np.zeros((w, h, 3))
    for i in range(w):
        for j in range(h):
            d0 = tl + (i + 0.5) * pw * r32 - (j + 0.5) * ph * u32 - p32
            d0 /= np.linalg.norm(d0)
            L2 = float(np.sum(np.cross(d0, p32) ** 2))

            def rhs(_, y):
                x = y[:3]
                return np.concatenate([y[3:], -1.5 * L2 * x / np.dot(x, x) ** 2.5])

            far = lambda _, y: np.dot(y[:3], y[:3]) - r_far ** 2
            far.terminal, far.direction = True, 1
            hole = lambda _, y: np.dot(y[:3], y[:3]) - 1.0          # r < rs: captured (render.py:2916)
            hole.terminal, hole.direction = True, -1
            sol = solve_ivp(rhs, (0.0, 5000.0), np.concatenate([p32, d0]), method="DOP853", rtol=1e-13, atol=1e-13,
                            events=(far, hole))
            assert sol.status == 1
            if len(sol.t_events[0]):                                # escaped
                want[i, j] = sol.y[3:, -1] / np.linalg.norm(sol.y[3:, -1])
    errs = []
    for step in (0.1, 0.05, 0.025):
        got = directions(step)
        esc = np.linalg.norm(want, axis=2) > 0
        assert 12 <= esc.sum() < w * h                              # the view has both kinds of rays
        np.testing.assert_array_equal(np.linalg.norm(got, axis=2) > 0, esc)     # same rays captured
        assert (np.abs(np.linalg.norm(got[esc], axis=1) - 1) < 1e-6).all()
        # at r ~ 400 the residual bending between the two stopping points is ~1e-9
        errs.append(float(np.abs(got - want).max()))
    assert errs[2] < 1e-7, errs
    assert errs[0] / errs[1] > 8 and errs[1] / errs[2] > 8, errs
