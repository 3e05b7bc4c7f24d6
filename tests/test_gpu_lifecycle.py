"""Device entity layer + statistics (include/bhr_lifecycle.h) against the reference-identical NumPy
implementations (bhr_amd.lifecycle, themselves pinned bit-for-bit by tests/golden/lifecycle.npz)."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu


def _renderer(n_r, n_phi, **kw):
    from bhr_amd import HipRenderer
    return HipRenderer(64, 36, scenes.analytic_skybox(32, 64), np.zeros((n_r, n_phi, 4), dtype=np.float32),
                       r_disk_inner=2.0, r_disk_outer=15.0, **kw)


@pytest.mark.parametrize("n_r,n_phi", [(48, 96), (128, 336), (416, 2912)])
def test_entity_layer_matches_host_rasteriser(hip_lib, n_r, n_phi):
    from bhr_amd.lifecycle import make_factories
    from lifecycle_checker import rasterize_entities
    r = _renderer(n_r, n_phi)
    r.init_background_layer(n_r, n_phi, seed=42)
    fac = make_factories(n_r, n_phi, 2.0, 15.0, seed=42)
    now = 0.0
    for step in range(0, 241, 60):                      # t = 0, 6, 12, 18, 24 s: births, deaths, fades
        while now < step * 0.1 - 1e-9:
            now += 0.1
            for f in fac.values():
                f.tick(now=now, dt=0.1)
        want = rasterize_entities(fac, now, n_r, n_phi, r._bg_omega_all_np, r._bg_r_norm_all)
        r.accumulate_entity_layer(fac, now)             # device path (default)
        got = r.read_comp()[5:11]
        # hotspots / RT spikes: f32 roll-scale-accumulate in the reference's order -> bit-exact
        np.testing.assert_array_equal(got[2:6], want[2:6])
        # filaments: binary64 Gaussian, ocml exp vs NumPy's exp (<= 1 ulp of f64 before the f32 rounding)
        np.testing.assert_allclose(got[0:2], want[0:2], rtol=0, atol=1e-7)
        assert (got[0:2] != want[0:2]).mean() < 1e-3
        assert want[0].max() > 0.1 and want[2].max() > 0.1 and want[4].max() > 0.1
    r.close()


def test_profile_pool_survives_turnover(hip_lib):
    """Many ticks: entities die and spawn, the pool is rebuilt; results stay equal to the host path."""
    from bhr_amd.lifecycle import make_factories
    from lifecycle_checker import rasterize_entities
    n_r, n_phi = 48, 96
    r = _renderer(n_r, n_phi)
    r.init_background_layer(n_r, n_phi, seed=42)
    fac = make_factories(n_r, n_phi, 2.0, 15.0, seed=42)
    for k in range(1, 1501):
        now = k * 0.1
        for f in fac.values():
            f.tick(now=now, dt=0.1)
        if k % 100 == 0:
            r.accumulate_entity_layer(fac, now)
            want = rasterize_entities(fac, now, n_r, n_phi, r._bg_omega_all_np, r._bg_r_norm_all)
            got = r.read_comp()[5:11]
            np.testing.assert_array_equal(got[2:6], want[2:6])
            np.testing.assert_allclose(got[0:2], want[0:2], rtol=0, atol=1e-7)
    r.close()


@pytest.mark.parametrize("n_r,n_phi", [(48, 96), (416, 2912)])
def test_entity_records_and_pair_tables_give_identical_layers(hip_lib, n_r, n_phi):
    """bhr_accumulate_population (the library evaluates fades and per-row scalars from entity records) against
    bhr_accumulate_entities fed with the NumPy-built pair tables: the same bits, through births, deaths, pool
    rebuilds and frames without any change of population."""
    from bhr_amd.lifecycle import make_factories
    r = _renderer(n_r, n_phi)
    r.init_background_layer(n_r, n_phi, seed=42)
    fac = make_factories(n_r, n_phi, 2.0, 15.0, seed=42)
    checked = 0
    for k in range(0, 1201):
        now = k * 0.1
        if k:
            for f in fac.values():
                f.tick(now=now, dt=0.1)
        if k % 150 == 0 or k in (1, 2, 3, 601, 602):
            r.accumulate_entity_layer(fac, now)
            got = r.read_comp()[5:11]
            r.accumulate_entity_layer(fac, now, pairs_on_host=True)
            want = r.read_comp()[5:11]
            np.testing.assert_array_equal(got, want, err_msg=f"tick {k}")
            assert want[0].max() > 0.05
            checked += 1
        elif k % 7 == 0:
            r.accumulate_entity_layer(fac, now)            # keeps the record cache and the staging ring turning
    assert checked >= 12 and r._population_tables.usable
    r.close()


def _use_host_lifecycle(r):
    """Checker configuration: the entity layer and the statistics from the reference-identical NumPy forms
    (tests/lifecycle_checker.py), uploaded through the C ABI -- what the device path is compared with."""
    import bhr_amd._lib as L
    from lifecycle_checker import compose_statistics, rasterize_entities

    def accumulate(factories, now):
        staging = rasterize_entities(factories, now, r._bg_n_r, r._bg_n_phi, r._bg_omega_all_np, r._bg_r_norm_all)
        L.check(r._lib.bhr_set_entity_staging(r._ctx, L.fptr(staging)))

    def stats():
        r._set_stats(*compose_statistics(r.read_comp(), r._edge_np, r._param_enable_rt))

    r.accumulate_entity_layer, r.recompute_interactive_stats = accumulate, stats


@pytest.mark.parametrize("n_r,n_phi", [(48, 96), (416, 2912)])
def test_statistics_match_numpy_exactly(hip_lib, n_r, n_phi):
    from bhr_amd.lifecycle import make_factories
    from lifecycle_checker import compose_statistics, rasterize_entities
    r = _renderer(n_r, n_phi)
    r.init_background_layer(n_r, n_phi, seed=42)
    fac = make_factories(n_r, n_phi, 2.0, 15.0, seed=42)
    r.generate_background(3.0)
    _use_host_lifecycle(r)                               # upload the host staging: identical comp on both sides
    r.accumulate_entity_layer(fac, 0.0)
    comp = r.read_comp()
    want = compose_statistics(comp, r._edge_np, 1)
    del r.accumulate_entity_layer, r.recompute_interactive_stats     # back to the product's device path
    r.recompute_interactive_stats()
    assert np.float32(want[0]) == r._stats_np[0] and np.float32(want[1]) == r._stats_np[1]
    np.testing.assert_array_equal(r._row_stats_np, want[2])
    # degenerate input: no positive structural temperature -> struct_scale falls back to 1.0 -> floor 0.01 rule
    import bhr_amd._lib as L
    z = np.zeros_like(comp)
    z[12] = 1.0
    L.check(r._lib.bhr_set_comp(r._ctx, L.fptr(z)))
    r.recompute_interactive_stats()
    w = compose_statistics(z, r._edge_np, 1)
    assert np.float32(w[0]) == r._stats_np[0] and np.float32(w[1]) == r._stats_np[1]
    np.testing.assert_array_equal(r._row_stats_np, w[2])
    r.close()


def test_device_and_host_lifecycle_render_the_same_frame(hip_lib):
    from bhr_amd import HipRenderer
    from bhr_amd.drivers import advance_lifecycle_frame, init_lifecycle_system
    imgs = []
    for dev in (True, False):
        r = HipRenderer(320, 180, scenes.analytic_skybox(64, 128), np.zeros((128, 336, 4), dtype=np.float32),
                        r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15.0)
        if not dev:
            _use_host_lifecycle(r)
        fac = init_lifecycle_system(r, 128, 336, seed=42)
        for k in range(1, 4):
            advance_lifecycle_frame(r, fac, t=0.1 * k, dt=0.1, recompute_stats=(k == 3))
        imgs.append(r.render([6, 0, 0.5], 60))
        r.close()
    assert np.abs(imgs[0] - imgs[1]).max() < 2e-5


# ---- the reference's test_lifecycle_perf.py restated: same 640x360 set-up, same budgets, same smoke asserts ----
@pytest.fixture(scope="module")
def sd_setup(hip_lib):
    from bhr_amd import drivers
    cam, fov = [6, 0, 0.5], 90
    r, use_lifecycle, n_r, n_phi = drivers.make_renderer(640, 360, cam, fov, n_stars=200, tex_w=256, tex_h=128)
    assert use_lifecycle
    fac = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
    yield r, fac, cam, fov
    r.close()


def _median_ms(renderer, fn, n=5):
    import time
    fn()
    renderer.sync()
    times = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        renderer.sync()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)) * 1e3


def test_reference_time_budgets(sd_setup):
    """test_lifecycle_perf.py:93-135: background < 500 ms, entity accumulation < 200 ms, compose + mipmaps
    < 50 ms, statistics < 100 ms, full texture frame < 800 ms (the reference's CPU-mode ceilings)."""
    r, fac, _, _ = sd_setup
    assert _median_ms(r, lambda: r.generate_background(t=1.0)) < 500
    assert _median_ms(r, lambda: r.accumulate_entity_layer(fac, now=1.0)) < 200
    assert _median_ms(r, lambda: r.compose_interactive_texture()) < 50
    assert _median_ms(r, lambda: r.recompute_interactive_stats()) < 100

    def full():
        r.generate_background(t=2.0)
        r.accumulate_entity_layer(fac, now=2.0)
        r.compose_interactive_texture()
    ms = _median_ms(r, full)
    assert ms < 800
    assert ms < 20, f"texture frame took {ms:.1f} ms; this build's own budget is 20 ms at 640x360"


def test_reference_texture_smoke_asserts(sd_setup):
    """test_lifecycle_perf.py:145-201: structure, time dependence, no NaN/Inf, the render is not black."""
    r, fac, cam, fov = sd_setup

    def texture(t):
        r.generate_background(t=t)
        r.accumulate_entity_layer(fac, now=t)
        r.recompute_interactive_stats()
        r.compose_interactive_texture()
        return r.disk_texture_field.to_numpy()

    tex = texture(0.0)
    assert tex[..., :3].std() > 0.01 and tex[..., 3].max() > 0.01 and tex[..., 3].std() > 0.001
    later = texture(5.0)
    assert np.abs(later - tex).mean() > 0.001
    assert np.isfinite(tex).all() and np.isfinite(later).all()
    img = r.render(cam, fov)
    assert img.shape == (360, 640, 3) and img.max() > 0.01 and np.isfinite(img).all()
