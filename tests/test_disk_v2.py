"""disk_v2 on the device against tables produced by the reference's NumPy package (tests/golden/disk_v2.npz),
plus the reference's own invariants restated (tests/unit/test_disk_v2_*.py)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- CPU: parameter validation and the random tables (no device call) ---------------------------------
def test_params_validation_matches_reference():
    import bhr_amd  # noqa: F401
    from bhr_amd.disk_v2 import DiskV2Params, DiskV2StructureParams
    p = DiskV2Params()
    assert (p.r_in, p.r_out, p.h0, p.beta_h, p.rho_power, p.temp_scale, p.omega_scale, p.edge_softness) == \
        (2.0, 10.0, 0.05, 0.05, 1.0, 1.0, 1.0, 0.1)
    s = DiskV2StructureParams()
    assert (s.mode1_strength, s.mode2_strength, s.shear_strength, s.shear_components, s.hotspot_strength,
            s.hotspot_count, s.hotspot_phi_sigma, s.hotspot_logr_sigma, s.hotspot_inner_bias) == \
        (0.03, 0.05, 0.22, 8, 0.16, 8, 0.18, 0.12, 2.0)
    for kw in (dict(r_in=0.0), dict(r_out=2.0), dict(h0=0.0), dict(rho_power=0.0), dict(temp_scale=0.0),
               dict(omega_scale=-1.0), dict(edge_softness=0.5), dict(edge_softness=-0.1)):
        with pytest.raises(ValueError):
            DiskV2Params(**kw)
    for kw in (dict(mode1_strength=-0.1), dict(mode1_strength=0.6, mode2_strength=0.4), dict(shear_strength=1.0),
               dict(shear_components=0), dict(hotspot_strength=1.0), dict(hotspot_count=0),
               dict(hotspot_phi_sigma=0.0), dict(hotspot_logr_sigma=0.0), dict(hotspot_inner_bias=0.0)):
        with pytest.raises(ValueError):
            DiskV2StructureParams(**kw)


def test_random_tables_are_seed_reproducible():
    import bhr_amd  # noqa: F401
    from bhr_amd import disk_v2 as dv
    sp = dv.DiskV2StructureParams()
    assert dv.shear_table(sp, 42) == dv.shear_table(sp, 42) != dv.shear_table(sp, 43)
    t = dv.shear_table(sp, 7)
    assert len(t) == 8 and all(2 <= a < 10 and 1 <= b < 6 and 0 <= c < 2 * np.pi for a, b, c in t)
    h = dv.hotspot_table(dv.DiskV2Params(), sp, 8)
    assert len(h) == 8 and all(0.6 <= w <= 1.0 and 0 <= lr <= np.log(5.0) for _, lr, w in h)
    with pytest.raises(ValueError):
        dv.smoothstep(1.0, 1.0, 0.5)


# ---- GPU ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "disk_v2.npz"))


@pytest.mark.gpu
def test_fields_match_reference_tables(gold, hip_lib):
    from bhr_amd import disk_v2 as dv
    P = dv.DiskV2Params()
    r, zf = gold["r"], gold["zf"]
    H = dv.disk_half_thickness(r, P)
    tol = dict(rtol=2e-14, atol=1e-15)
    np.testing.assert_allclose(H, gold["H"], **tol)
    rr = np.repeat(r[:, None], len(zf), axis=1)
    zz = zf[None, :] * gold["H"][:, None]
    np.testing.assert_array_equal(dv.disk_radial_mask(r, P), gold["mask_r"])
    np.testing.assert_allclose(dv.disk_radial_weight(r, P), gold["W_r"], **tol)
    np.testing.assert_allclose(dv.disk_vertical_weight(rr, zz, P), gold["W_z"], **tol)
    np.testing.assert_array_equal(dv.disk_volume_mask(rr, zz, P), gold["mask_vol"])
    np.testing.assert_allclose(dv.angular_velocity_field(r, P), gold["omega"], **tol)
    np.testing.assert_allclose(dv.midplane_density_field(r, P), gold["rho_mid"], **tol)
    np.testing.assert_allclose(dv.midplane_temperature_field(r, P), gold["T_mid"], **tol)
    np.testing.assert_allclose(dv.density_field(rr, zz, P), gold["rho"], **tol)
    np.testing.assert_allclose(dv.temperature_field(rr, zz, P), gold["T"], **tol)
    np.testing.assert_allclose(dv.smoothstep(0.0, 1.0, np.linspace(-0.5, 1.5, 41)), gold["smooth"], rtol=0, atol=0)
    probe = [dv.disk_half_thickness(3.0, P), dv.disk_radial_weight(2.0, P), dv.disk_radial_weight(10.0, P),
             dv.angular_velocity_field(2.0, P), dv.midplane_temperature_field(2.0, P), dv.density_field(4.0, 0.0, P)]
    assert all(isinstance(v, float) for v in probe)                       # scalars in, scalars out
    np.testing.assert_allclose(probe, gold["scalar_probe"], **tol)


@pytest.mark.gpu
def test_modulations_match_reference_tables(gold, hip_lib):
    from bhr_amd import disk_v2 as dv
    P = dv.DiskV2Params()
    rg, pg = np.meshgrid(gold["rg"], gold["phig"], indexing="ij")
    tol = dict(rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dv.weak_mode_modulation(rg, pg, P), gold["F_mode"], **tol)
    for seed in (7, 42, 123):
        np.testing.assert_allclose(dv.shear_modulation(rg, pg, P, seed=seed), gold[f"F_shear_{seed}"], **tol)
        np.testing.assert_allclose(dv.hotspot_modulation(rg, pg, P, seed=seed), gold[f"F_hotspot_{seed}"], **tol)
        np.testing.assert_allclose(dv.structure_modulation(rg, pg, P, seed=seed), gold[f"F_total_{seed}"], **tol)


@pytest.mark.gpu
def test_reference_invariants(hip_lib):
    """tests/unit/test_disk_v2_physical_fields.py / _structure_modulations.py restated."""
    from bhr_amd import disk_v2 as dv
    P = dv.DiskV2Params()
    r = np.linspace(2.0, 10.0, 257)
    om = dv.angular_velocity_field(r, P)
    assert np.all(np.diff(om) < 0)                                         # Omega decreases outwards
    assert dv.disk_radial_weight(2.0, P) == 0.0 and dv.disk_radial_weight(10.0, P) == 0.0   # exact boundaries
    assert dv.disk_radial_mask(2.0, P) and dv.disk_radial_mask(10.0, P) and not dv.disk_radial_mask(10.0001, P)
    T = dv.midplane_temperature_field(r, P)
    assert T[0] == 0.0 and r[np.argmax(T)] > P.r_in                        # peak outside r_in
    H = dv.disk_half_thickness(r, P)
    assert np.all(dv.density_field(r, 1.1 * H, P) == 0.0)                   # above the surface
    assert np.all(dv.density_field(r[1:-1], 0.0, P) > 0.0)
    rg, pg = np.meshgrid(np.linspace(1.0, 12.0, 40), np.linspace(0, 2 * np.pi, 64, endpoint=False), indexing="ij")
    F = dv.structure_modulation(rg, pg, P, seed=5)
    outside = (rg <= P.r_in) | (rg >= P.r_out)
    assert np.all(F[outside] == 1.0) and np.all(F > 0)                      # neutral outside, positive inside
    np.testing.assert_array_equal(F, dv.structure_modulation(rg, pg, P, seed=5))
    assert np.abs(F - dv.structure_modulation(rg, pg, P, seed=6)).max() > 1e-3


@pytest.mark.gpu
def test_fixed_normalisation_for_per_ray_use(hip_lib):
    """With the maxima of a reference grid passed in, a subset of the points evaluates to the same
    values as the full grid (the per-ray shading contract)."""
    from bhr_amd import disk_v2 as dv
    P = dv.DiskV2Params()
    cp = dv.pack_params(P, None, shear_seed=42, hotspot_seed=43)
    rg, pg = np.meshgrid(np.linspace(2.0, 10.0, 96), np.linspace(0, 2 * np.pi, 192, endpoint=False), indexing="ij")
    full, (m_sh, m_hs) = dv.evaluate(dv.F_TOTAL, cp, rg, phi=pg, return_max=True)
    assert m_sh > 0 and m_hs > 0
    sub = dv.evaluate(dv.F_TOTAL, cp, rg[10:20, 5:50], phi=pg[10:20, 5:50], norm_shear=m_sh, norm_hotspot=m_hs)
    np.testing.assert_allclose(sub, full[10:20, 5:50], rtol=1e-15, atol=0)


@pytest.mark.gpu
def test_march_with_analytic_disk_source(hip_lib):
    """bhr_set_disk_source(BHR_DISK_V2): the in-kernel binary64 model equals (i) its host twin at sample
    points and (ii) a render from a fine texture baked with that twin, up to texture interpolation."""
    from bhr_amd import HipRenderer, scenes
    from bhr_amd import disk_v2 as dv
    P = dv.DiskV2Params(r_in=2.0, r_out=10.0)
    kw = dict(step_size=0.1, r_disk_inner=2.0, r_disk_outer=10.0, disk_tilt=10.0)
    sky = scenes.analytic_skybox(128, 256)
    n_r, n_phi = 768, 3072
    r = HipRenderer(256, 144, sky, np.zeros((n_r, n_phi, 4), dtype=np.float32), **kw)
    r.use_disk_v2(P, seed=42)
    cp, m_sh, m_hs, t_peak = r._dv2
    img_model = r.render([6, 0, 1.5], 90, skip_bloom=True)
    assert img_model.max() > 0.05 and np.isfinite(img_model).all()
    # bake: texel (i, j) sits at r = r_in + (i / n_r) span, phi = 2 pi j / n_phi (the lookup of _sample_disk)
    rr = 2.0 + (np.arange(n_r) / n_r) * 8.0
    pp = 2 * np.pi * np.arange(n_phi) / n_phi
    rg, pg = np.meshgrid(rr, pp, indexing="ij")
    tex = dv.disk_rgba(rg, pg, cp, m_sh, m_hs, t_peak, ctx=r._ctx)
    assert tex.shape == (n_r, n_phi, 4) and 0.0 < tex[..., 3].mean() < 1.0
    r.use_disk_v2(None)
    r.update_disk_texture(tex)
    img_tex = r.render([6, 0, 1.5], 90, skip_bloom=True)
    d = np.abs(img_model - img_tex)
    assert np.sqrt(np.mean(d ** 2)) < 2e-3 and np.quantile(d, 0.999) < 2e-2, (np.sqrt(np.mean(d ** 2)), d.max())
    # rotation: frame != 0 advects the analytic pattern with the model's own Omega(r)
    r.use_disk_v2(P, seed=42)
    moved = r.render([6, 0, 1.5], 90, frame=40, skip_bloom=True)
    assert np.abs(moved - img_model).max() > 1e-3
    r.close()
