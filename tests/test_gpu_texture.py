"""GPU parity of the disk-texture pipeline (through the C ABI): simplex/FBM evaluation, the
background generator, the compose kernel and the mip chain against the oracle and against the
reference-NumPy golden vectors (same comparisons and tolerances as the reference's own
tests/unit/test_gpu_texture_compose.py: 1e-4 for compose, 1e-3 for mips)."""
import os
import types

import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip(hip_lib):
    from bhr_amd import HipRenderer
    r = HipRenderer(64, 36, scenes.analytic_skybox(32, 64), np.zeros((128, 256, 4), dtype=np.float32))
    yield r
    r.close()


def test_eval_noise_matches_oracle(hip, oracle):
    rng = np.random.default_rng(1)
    pts = (rng.random((100000, 3)) * 400 - 200).astype(np.float32)
    np.testing.assert_allclose(hip.eval_noise(pts, "simplex"), oracle.eval_noise(pts, "simplex"), atol=2e-6, rtol=0)
    for octaves, pers, lac in ((4, 0.5, 2.0), (5, 0.45, 2.0), (3, 0.35, 2.0), (1, 0.5, 2.0)):
        a = hip.eval_noise(pts, "fbm", octaves=octaves, persistence=pers, lacunarity=lac)
        b = oracle.eval_noise(pts, "fbm", octaves=octaves, persistence=pers, lacunarity=lac)
        np.testing.assert_allclose(a, b, atol=5e-6, rtol=0)
    # reference property tests (test_simplex_noise.py): range, fbm(1 octave) == simplex, empty input
    n = hip.eval_noise(pts, "simplex")
    assert n.min() >= -1.001 and n.max() <= 1.001
    np.testing.assert_array_equal(hip.eval_noise(pts, "fbm", octaves=1), n)
    assert hip.eval_noise(np.zeros((0, 3), dtype=np.float32)).shape == (0,)


@pytest.mark.parametrize("t", [0.0, 1.7, 36.0])
def test_background_matches_oracle(hip_lib, oracle, t):
    from bhr_amd import HipRenderer
    n_r, n_phi = 48, 160
    r = HipRenderer(64, 36, scenes.analytic_skybox(32, 64), np.zeros((n_r, n_phi, 4), dtype=np.float32),
                    r_disk_inner=2.0, r_disk_outer=15.0)
    r.init_background_layer(n_r, n_phi, seed=42)
    r.generate_background(t)
    got = r.read_comp()
    want = oracle.generate_background(n_r, n_phi, r._bg_az_freq, r._bg_az_shear, 2.0, 15.0, t)
    # sin/cos/pow differ by <= 2 ulp between ocml and glibc; at frequency 800 that moves the noise
    # argument by ~1e-4 and the noise by ~1e-3 on isolated texels: compare robustly
    for idx in (0, 1, 2, 3, 4, 11, 12):
        d = np.abs(got[idx] - want[idx])
        assert np.mean(d) < 2e-5, (idx, np.mean(d))
        assert np.quantile(d, 0.999) < 2e-3, (idx, np.quantile(d, 0.999))
    assert not got[5:11].any()                                 # entity planes untouched
    # the reference's range tests (test_background_layer.py:76-143)
    assert got[0].min() >= 0 and got[0].max() <= 0.35
    assert got[3].min() >= 0 and got[3].max() <= 1
    np.testing.assert_allclose(got[4], 0.05 * got[3], atol=1e-7, rtol=0)
    assert got[12].min() >= 0.1 - 1e-7 and got[12].max() <= 1
    r.close()


def test_background_and_compose_at_the_fhd_texture_size(hip_lib, oracle):
    """The fhd frame's own texture (416 x 2912, compute_disk_texture_resolution): background planes against the
    oracle with the same robust statistics as above, then the compose kernel + mip chain against the oracle's
    compose of the SAME planes (tolerances of tests/unit/test_gpu_texture_compose.py)."""
    from bhr_amd import HipRenderer
    from bhr_amd.textures import compute_disk_texture_resolution, compute_edge_alpha
    n_phi, n_r = compute_disk_texture_resolution(1920, 1080, [6, 0, 0.5], 90, 2.0, 15.0)
    assert (n_r, n_phi) == (416, 2912)
    r = HipRenderer(64, 36, scenes.analytic_skybox(32, 64), np.zeros((n_r, n_phi, 4), dtype=np.float32),
                    r_disk_inner=2.0, r_disk_outer=15.0)
    r.init_background_layer(n_r, n_phi, seed=42)
    r.generate_background(5.0)
    got = r.read_comp()
    want = oracle.generate_background(n_r, n_phi, r._bg_az_freq, r._bg_az_shear, 2.0, 15.0, 5.0)
    for idx in (0, 3, 4, 11, 12):
        d = np.abs(got[idx] - want[idx])
        assert np.mean(d) < 2e-5 and np.quantile(d, 0.999) < 2e-3, (idx, np.mean(d))
    assert got[0].min() >= 0 and got[0].max() <= 0.35 and got[3].max() <= 1 and got[12].min() >= 0.1 - 1e-6
    # compose the device's planes on both sides
    r.recompute_interactive_stats()
    r.compose_interactive_texture()
    tex = r.disk_texture_field.to_numpy()
    stats = r._param_stats_field.to_numpy()
    row_stats = r._param_row_stats_field.to_numpy()
    ref = oracle.compose_disk_texture(got, r._omega_rows_field.to_numpy(), r._edge_field.to_numpy(), stats, row_stats,
                                      t_offset=0.0)
    assert np.abs(tex - ref).max() < 1e-4
    assert tex[..., :3].std() > 0.01 and tex[..., 3].max() > 0.01 and np.isfinite(tex).all()
    mips = oracle.build_mips_padded(tex)
    for level in range(1, r.num_mip_levels):
        mh, mw = n_r >> level, n_phi >> level
        assert np.abs(r.read_mip_level(level) - mips[level, :mh, :mw]).max() < 1e-3
    r.close()


def test_generate_background_requires_init(hip):
    with pytest.raises(AssertionError):
        hip.generate_background(0.0)                           # "Must call init_background_layer() first"


def _state():
    d = np.load(os.path.join(G, "compose.npz"))
    c = d["comp"]
    st = types.SimpleNamespace(
        n_r=c.shape[1], n_phi=c.shape[2], enable_rt=True, color_temp=float(d["color_temp"]),
        omega_rows=d["omega_rows"], edge=d["edge"], temp_base=c[0], spiral=c[1], spiral_temp=c[2],
        turbulence=c[3], turb_temp=c[4], arcs=c[5], arcs_temp=c[6], rt_spikes=c[7], rt_temp=c[8], hotspot=c[9],
        hotspot_temp=c[10], az_hotspot=c[11], disturb_mod=c[12])
    return d, st


def test_compose_and_mips_match_reference_numpy(hip, oracle):
    """upload_parametric_state + update_disk_texture_gpu vs the reference's NumPy path
    (test_gpu_texture_compose.py:154-189, 229-262) and vs the oracle."""
    d, st = _state()
    hip.upload_parametric_state(st)
    # statistics computed on upload equal the reference's (test_statistics_match_cpu)
    np.testing.assert_allclose(hip._param_stats_field.to_numpy(), d["stats"], rtol=1e-6)
    np.testing.assert_allclose(hip._param_row_stats_field.to_numpy(), d["row_stats"], atol=1e-6)
    np.testing.assert_array_equal(hip._comp_field.to_numpy(), d["comp"])
    for t in (0, 5, 50, 180):
        hip.update_disk_texture_gpu(float(t))
        tex = hip.disk_texture_field.to_numpy()
        want = d[f"tex_t{t}"]
        got = tex if t == 5 else tex[::4]
        assert np.max(np.abs(got - want)) < 1e-4, t
        ora = oracle.compose_disk_texture(d["comp"], d["omega_rows"], d["edge"], d["stats"], d["row_stats"], float(t),
                                          1, float(d["color_temp"]))
        assert np.max(np.abs(tex - ora)) < 2e-6, t
    hip.update_disk_texture_gpu(5.0)
    mips = hip.disk_mips_field.to_numpy()
    assert mips.shape == (5, st.n_r, st.n_phi, 4) and hip.num_mip_levels == 5
    for lev in range(1, 5):
        want = d[f"mip5_{lev}"]
        h, w = want.shape[:2]
        assert np.max(np.abs(mips[lev, :h, :w] - want)) < 1e-3
        assert not mips[lev, h:].any() and not mips[lev, :, w:].any()
    omips = oracle.build_mips_padded(hip.disk_texture_field.to_numpy())
    np.testing.assert_array_equal(mips, omips)                 # same 2x2 order as the Taichi kernel: bit-exact


def test_update_disk_texture_and_size_mismatch(hip):
    tex = scenes.noisy_disk(128, 256)
    hip.update_disk_texture(tex)
    np.testing.assert_array_equal(hip.disk_texture_field.to_numpy(), tex)
    from bhr_amd.textures import generate_disk_mipmaps
    host = generate_disk_mipmaps(tex, levels=4)
    for lev in range(5):
        np.testing.assert_allclose(hip.read_mip_level(lev), host[lev], atol=1e-6, rtol=0)
    with pytest.raises(AssertionError):
        hip.update_disk_texture(np.zeros((64, 256, 4), dtype=np.float32))   # render.py:2299


def test_lifecycle_texture_end_to_end(hip_lib, oracle):
    """init_lifecycle_system + advance (render_image's texture path): the device texture equals the
    oracle's compose of the same components, statistics included."""
    from bhr_amd import HipRenderer
    from bhr_amd.drivers import init_lifecycle_system, advance_lifecycle_frame
    n_r, n_phi = 128, 336      # the e2e scene's texture
    r = HipRenderer(320, 180, scenes.analytic_skybox(32, 64), np.zeros((n_r, n_phi, 4), dtype=np.float32),
                    r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15.0)
    fac = init_lifecycle_system(r, n_r, n_phi, seed=42)
    advance_lifecycle_frame(r, fac, t=0.0, dt=0.0, recompute_stats=True)
    comp = r.read_comp()
    assert comp[5].max() > 0 and comp[9].max() > 0 and comp[7].max() > 0      # all three populations present
    want = oracle.compose_disk_texture(comp, r._omega_np, r._edge_np, r._stats_np, r._row_stats_np, 0.0, 1, 6000.0)
    tex = r.disk_texture_field.to_numpy()
    assert np.max(np.abs(tex - want)) < 2e-6
    assert 0.2 < tex[..., 3].mean() < 0.9 and tex[..., :3].max() <= 1.0
    img = r.render([6, 0, 0.5], 60)
    assert img.shape == (180, 320, 3) and img.max() > 0.01 and np.isfinite(img).all()   # test_lifecycle_perf.py:191-201
    r.close()


@pytest.mark.parametrize("tex_w,tex_h,n_stars", [(256, 128, 300), (2048, 1024, 6000)])
def test_skybox_glow_on_the_device(hip_lib, tex_w, tex_h, n_stars):
    """Host random part + device Milky-Way glow == the host generator (itself pinned to the reference by
    tests/golden: full 64x32 texture and the SHA-256 of the 2048x1024 default sky), up to the last bit of the
    f64 transcendentals."""
    from bhr_amd import HipRenderer
    from bhr_amd.skybox import generate_skybox
    want = generate_skybox(tex_w, tex_h, seed=42, n_stars=n_stars)
    base = generate_skybox(tex_w, tex_h, seed=42, n_stars=n_stars, glow=False)
    assert np.abs(want - np.clip(base, 0, 1)).max() > 0.05                    # the glow is a visible part of the sky
    r = HipRenderer(64, 36, base, np.zeros((32, 64, 4), np.float32))
    r.add_skybox_glow()
    got = r.read_skybox()
    assert got.min() >= 0.0 and got.max() <= 1.0
    assert np.abs(got - want).max() <= 6e-8
    assert (got != want).mean() < 1e-3                                        # and almost everywhere the same bits
    r.close()


@pytest.mark.parametrize("tex_h,tex_w,n_stars", [(1024, 2048, 6000), (64, 128, 40), (128, 512, 300)])
def test_procedural_skybox_on_the_device(hip_lib, tex_h, tex_w, n_stars):
    """bhr_skybox_build: nebula (Pillow's bilinear resize through u8) + star blobs (np.add.at's order) on the device
    are BIT-IDENTICAL to the host's NumPy / Pillow result; with the glow the texture is the reference's
    generate_skybox to the last bit of the f64 transcendentals (SHA-256-pinned host generator as the checker)."""
    from bhr_amd import HipRenderer
    from bhr_amd.skybox import generate_skybox
    r = HipRenderer(32, 18, np.zeros((tex_h, tex_w, 3), np.float32), np.zeros((16, 32, 4), np.float32))
    r.build_procedural_skybox(seed=42, n_stars=n_stars)
    got = r.read_skybox()
    want = generate_skybox(tex_w, tex_h, seed=42, n_stars=n_stars)
    assert np.abs(got - want).max() <= 6e-8 and (got != want).mean() < 2e-3
    # before the glow: exact
    from bhr_amd import _lib
    from bhr_amd.skybox import STAR_PATCH_R, pillow_bilinear_coeffs, sky_tables
    import ctypes as C
    t = sky_tables(tex_w, tex_h, 42, n_stars)
    kh, bh = pillow_bilinear_coeffs(t["coarse_u8"].shape[1], tex_w)
    kv, bv = pillow_bilinear_coeffs(t["coarse_u8"].shape[0], tex_h)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))   # noqa: E731
    coarse = np.ascontiguousarray(t["coarse_u8"])
    _lib.check(r._lib.bhr_skybox_build(r._ctx, tex_h, tex_w, coarse.ctypes.data_as(C.POINTER(C.c_uint8)), coarse.shape[0],
                                       coarse.shape[1], i32(kh), i32(bh), kh.shape[1], i32(kv), i32(bv), kv.shape[1],
                                       len(t["cx"]), _lib.fptr(t["cx"]), _lib.fptr(t["cy"]), _lib.fptr(t["colors"]),
                                       _lib.fptr(t["vals"]), STAR_PATCH_R))
    np.testing.assert_array_equal(r.read_skybox(), generate_skybox(tex_w, tex_h, seed=42, n_stars=n_stars, glow=False))
    r.close()
