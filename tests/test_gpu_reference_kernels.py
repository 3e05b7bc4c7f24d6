"""GPU parity against the reference's OWN kernel statements (through the C ABI).

The fixtures under tests/golden/{march_ref_*,bloom_ref,texture_ref}.npz hold what the unmodified
`@ti.kernel` / `@ti.func` function objects of /root/reference/render.py produce when run as plain Python
(tests/golden/make_kernel_golden.py, tests/golden/ti_shim.py): once with every operation rounded to
binary32 ("f32": IEEE arithmetic in the reference's order -- the closest available stand-in for its
`--device cpu` build) and once in binary64 ("f64": the rounding-free value of the same statements on the
same typed inputs).  No oracle is involved here: HIP output vs reference-statement output.

Bars.  BASELINE.json north_star: per-channel RMSE <= 1e-4 on the pre-tonemap float image.  Against the
f32 fixtures the strict kernel is held to RMSE <= 5e-6 and the step counters must be EQUAL (same ray paths);
against the f64 fixtures both kernels (strict and fast) are held to the north-star bar on the frame and its
layers, or -- where the binary32 evaluation of the reference's own statements is itself further than that from
their binary64 value (f32 rounding of a chaotic map next to the photon ring) -- to 1.5x that distance.
"""
import os

import numpy as np
import pytest

from test_reference_kernels import FLARE, GOLD, KW, MARCH, load_scene

pytestmark = pytest.mark.gpu

NORTH_STAR = 1e-4
STRICT_VS_F32 = 5e-6


def _rmse_c(a, b):
    return np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2, axis=(0, 1)))


def _hip_layers(name, g, sky, tex, **kw):
    from bhr_amd import HipRenderer, _lib
    hip = HipRenderer(int(g["width"]), int(g["height"]), sky, tex, lens_flare=(name in FLARE), **kw, **KW[name])
    final = hip.render(list(g["cam_pos"]), float(g["fov"]), frame=int(g["frame"]))
    lay = dict(final=final, bg=hip.read_layer(_lib.LAYER_BG), disk=hip.read_layer(_lib.LAYER_DISK),
               blur=hip.read_layer(_lib.LAYER_BLUR))
    c = hip.counters()
    hip.close()
    return lay, c


def _ref_layer(g, mode, k):
    a = g[f"{mode}_{k}"]
    return a if k == "final" else a.transpose(1, 0, 2)      # fields are (W, H, 3), render() returns (H, W, 3)


@pytest.mark.parametrize("name", MARCH)
def test_strict_kernel_vs_reference_statements_f32(name, hip_lib):
    g, sky, tex = load_scene(name)
    lay, c = _hip_layers(name, g, sky, tex)
    assert c["rays"] == int(g["width"]) * int(g["height"])
    assert c["ray_steps"] == int(g["f32_steps"].sum()), "ray paths differ from the reference's statements"
    for k in ("bg", "disk", "blur", "final"):
        ref = _ref_layer(g, "f32", k)
        assert lay[k].shape == ref.shape
        e = _rmse_c(lay[k], ref)
        assert (e <= STRICT_VS_F32).all(), f"{name}/{k}: per-channel RMSE {e}"
        assert np.abs(lay[k] - ref).max() <= 1e-4, f"{name}/{k}: max {np.abs(lay[k] - ref).max()}"


# f32 rounding of a chaotic map: the binary32 evaluation of the reference's own statements (f32 fixture)
# sits this far from their binary64 value (f64 fixture) on these 64x36 frames; the HIP kernels may not be
# further away than the bar, or than 1.5x the reference-f32 distance where that alone exceeds the bar.
def _f32_vs_f64(g, k):
    return float(_rmse_c(_ref_layer(g, "f32", k), _ref_layer(g, "f64", k)).max())


@pytest.mark.parametrize("math", ["strict", "fast"])
@pytest.mark.parametrize("name", MARCH)
def test_kernels_vs_reference_statements_f64(name, math, hip_lib):
    g, sky, tex = load_scene(name)
    lay, c = _hip_layers(name, g, sky, tex, math=math)
    steps64 = int(g["f64_steps"].sum())
    assert abs(c["ray_steps"] - steps64) <= 2e-4 * steps64
    for k in ("bg", "disk", "blur", "final"):
        e = float(_rmse_c(lay[k], _ref_layer(g, "f64", k)).max())
        # the plain north-star bar on every view and layer, for both arithmetics (round 2 allowed 1.5 x the reference's own
        # f32-vs-f64 distance where that exceeded it; it never does on these views: tools/fast_bars.py measures the fast
        # kernel at 3.8e-7 ... 4.9e-5 -- worst: `inside`, camera in the annulus -- against a reference f32-vs-f64 distance of
        # 3e-7 ... 3e-5)
        assert e <= NORTH_STAR, f"{name}/{math}/{k}: RMSE {e:.3g} > {NORTH_STAR:g} (reference f32 vs f64: {_f32_vs_f64(g, k):.3g})"


def test_reference_f32_vs_f64_distance_is_what_the_bar_assumes():
    """Documents the yardstick: the largest per-channel RMSE between the two evaluations of the reference."""
    worst = {n: max(_f32_vs_f64(np.load(os.path.join(GOLD, f"march_ref_{n}.npz")), k)
                    for k in ("bg", "disk", "blur", "final")) for n in MARCH}
    print("reference statements, f32 vs f64 RMSE:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-3


# --------------------------------------------------------------------------- bloom
@pytest.mark.parametrize("tag", ["wide", "narrow"])
def test_bloom_kernel_vs_reference_statements(tag, hip_lib):
    from bhr_amd import HipRenderer, _lib
    g = np.load(os.path.join(GOLD, "bloom_ref.npz"))
    layer = g[f"{tag}_layer"]                                  # (W, H, 3)
    W, H = layer.shape[:2]
    hip = HipRenderer(W, H, np.zeros((8, 16, 3), np.float32), np.zeros((16, 32, 4), np.float32))
    disk = np.ascontiguousarray(layer.transpose(1, 0, 2))
    hip.write_layer(_lib.LAYER_DISK, disk)
    hip.write_layer(_lib.LAYER_BG, np.zeros_like(disk))
    hip.bloom_only()
    blur = hip.read_layer(_lib.LAYER_BLUR)
    final = hip.read_layer(_lib.LAYER_FINAL)
    hip.close()
    for mode in ("f32", "f64"):
        ref = g[f"{tag}_{mode}_blur"].transpose(1, 0, 2)
        assert np.abs(blur - ref).max() <= 3e-6, (mode, np.abs(blur - ref).max())
    # render()'s combine (render.py:3918): clip(bg + disk + blur) with the UNSCALED blur
    np.testing.assert_allclose(final, np.clip(disk + g[f"{tag}_f32_blur"].transpose(1, 0, 2), 0, 1), atol=3e-6)


# --------------------------------------------------------------------------- noise / background / compose / mips
@pytest.fixture(scope="module")
def texg():
    return np.load(os.path.join(GOLD, "texture_ref.npz"))


def test_noise_vs_reference_statements(texg, hip_lib):
    from bhr_amd import HipRenderer
    hip = HipRenderer(32, 18, np.zeros((8, 16, 3), np.float32), np.zeros((16, 48, 4), np.float32))
    c = texg["noise_coords"]
    # the device evaluates the same f32 operations but may contract a*b+c into an FMA: <= a few ulp of O(1) sums
    assert np.abs(hip.eval_noise(c, "simplex") - texg["f32_simplex"]).max() <= 2e-6
    assert np.abs(hip.eval_noise(c, "fbm", octaves=4, persistence=0.5, lacunarity=2.0)
                  - texg["f32_fbm_4_05_2"]).max() <= 5e-6
    assert np.abs(hip.eval_noise(c[:300], "fbm", octaves=5, persistence=0.45, lacunarity=2.0)
                  - texg["f32_fbm_5_045_2"]).max() <= 5e-6
    hip.close()


@pytest.mark.parametrize("t", [0.0, 5.0, 36.5])
def test_background_vs_reference_statements(t, texg, hip_lib):
    from bhr_amd import HipRenderer
    ref = texg[f"f32_bg_t{t:g}"]
    n_r, n_phi = ref.shape[1:]
    hip = HipRenderer(32, 18, np.zeros((8, 16, 3), np.float32), np.zeros((n_r, n_phi, 4), np.float32),
                      r_disk_inner=2.0, r_disk_outer=15.0)
    hip.init_background_layer(n_r, n_phi, seed=42)
    assert (hip._bg_az_freq, hip._bg_az_shear) == (int(texg["bg_az"][0]), float(texg["bg_az"][1]))
    hip.generate_background(t)
    got = hip.read_comp()
    hip.close()
    # Round 3: the kernel evaluates its sin / cos / pow in binary64 and rounds once, as the reference's statements do under
    # the shim -- every plane within 1.2e-7 (1 ulp of the sums; 93-100 % of the texels bit-identical) at every t.  (Round 2,
    # ocml's f32 functions: an ulp of cos(phi) x 1600 in the noise coordinates, planes 3 and 12 only held to median 2e-5 /
    # q99 3e-3 -- the loosest bar of the suite.)
    for idx in (0, 3, 4, 11, 12):
        d = np.abs(got[idx] - ref[idx])
        assert d.max() <= 5e-7, (idx, d.max(), np.median(d))
    assert not got[1].any() and not got[2].any() and not got[5:11].any()


@pytest.mark.parametrize("t", [0.0, 12.5])
def test_compose_and_mips_vs_reference_statements(t, texg, hip_lib):
    from bhr_amd import HipRenderer, _lib
    comp = texg["compose_comp"]
    n_r, n_phi = comp.shape[1:]
    hip = HipRenderer(32, 18, np.zeros((8, 16, 3), np.float32), np.zeros((n_r, n_phi, 4), np.float32),
                      r_disk_inner=2.0, r_disk_outer=15.0)
    hip.init_background_layer(n_r, n_phi, seed=42)
    np.testing.assert_array_equal(hip._omega_rows_field.to_numpy(), texg["compose_omega"])
    np.testing.assert_array_equal(hip._edge_field.to_numpy(), texg["compose_edge"])
    _lib.check(hip._lib.bhr_set_comp(hip._ctx, _lib.fptr(np.ascontiguousarray(comp))))
    hip._set_stats(float(texg["compose_stats"][0]), float(texg["compose_stats"][1]), texg["compose_row_stats"])
    _lib.check(hip._lib.bhr_compose_texture(hip._ctx, float(t), 1, 6000.0))   # update_disk_texture_gpu / compose_interactive_texture
    tex = hip._read_disk_texture()
    mips = hip._read_mips_padded()
    hip.close()
    # same tolerances as the reference's own GPU-vs-NumPy test (tests/unit/test_gpu_texture_compose.py: 1e-4, 1e-3),
    # met with two orders of magnitude to spare against its kernel statements
    assert np.abs(tex - texg[f"f32_tex_t{t:g}"]).max() <= 2e-6
    ref_m = texg[f"f32_mips_t{t:g}"]
    assert mips.shape == ref_m.shape
    assert np.abs(mips - ref_m).max() <= 2e-6


# --------------------------------------------------------------------------- the reference's e2e frame (configs[0])
def test_e2e_frame_on_the_reference_s_texture(hip_lib):
    """HIP march + bloom + combine on the sky and the disk texture the reference's own pipeline produced, against
    the frame its render_image returned (tests/e2e_render.py scene, every kernel run as binary32 Python)."""
    from bhr_amd import HipRenderer
    from test_reference_kernels import E2E_KW, load_e2e
    g, sky = load_e2e()
    hip = HipRenderer(320, 180, sky, g["disk_tex"], **E2E_KW)
    out = hip.render([6, 0, 0.5], 60)
    hip.close()
    e = _rmse_c(out, g["final"])
    assert (e <= STRICT_VS_F32).all() and np.abs(out - g["final"]).max() <= 1e-4, (e, np.abs(out - g["final"]).max())


def test_e2e_whole_pipeline_vs_the_reference_s_render_image(hip_lib, capsys):
    """`render_image` end to end on the device -- skybox, entity lifecycle, background / compose / mip kernels, march,
    bloom -- against the reference's render_image output.  Measured (round 3): texture max 4.2e-7, compose statistics
    equal, frame per-channel RMSE 2.3-3.3e-7 -- 300x inside the north star's 1e-4.  (Round 2: the background generator's
    f32 sin / cos / pow moved texels by up to ~1e-4 through noise look-ups at frequencies up to 1600; frame 1.3-2.0e-6.)"""
    import hashlib
    from bhr_amd import HipRenderer, drivers
    from test_reference_kernels import E2E_KW, load_e2e
    g, sky = load_e2e()
    img = drivers.render_image(320, 180, [6, 0, 0.5], 60, n_stars=100, lens_flare=False, **E2E_KW)
    r = HipRenderer(320, 180, sky, np.zeros((128, 336, 4), np.float32), **E2E_KW)
    fac = drivers.init_lifecycle_system(r, 128, 336, seed=42)
    drivers.advance_lifecycle_frame(r, fac, t=0.0, dt=0.0, recompute_stats=True)
    tex = r.disk_texture_field.to_numpy()
    stats, row_stats = r._param_stats_field.to_numpy(), r._param_row_stats_field.to_numpy()
    r.close()
    dt = np.abs(tex - g["disk_tex"])
    e = _rmse_c(img, g["final"])
    with capsys.disabled():
        print(f"\n[e2e] md5 HIP {hashlib.md5(img.tobytes()).hexdigest()}  reference-statements f32 {g['md5']}  "
              f"tests/e2e_baseline.txt {g['baseline_md5']}")
        print(f"[e2e] texture: max {dt.max():.3g} median {np.median(dt):.3g} p99.9 {np.quantile(dt, 0.999):.3g}; "
              f"stats {stats} vs {g['stats']}; frame RMSE {e}, max {np.abs(img - g['final']).max():.3g}")
    np.testing.assert_allclose(stats, g["stats"], rtol=1e-5)
    np.testing.assert_allclose(row_stats, g["row_stats"], rtol=1e-4, atol=1e-6)
    # round 3 (background kernel's libm calls in binary64, rounded once): texture max 4.2e-7, median 0, statistics equal to
    # the printed digits, frame RMSE 2.3-3.3e-7.  (Round 2: texture max 9.8e-5, frame 1.3-2.0e-6.)  North star: 1e-4.
    assert np.median(dt) <= 1e-7 and dt.max() <= 5e-6, (np.median(dt), dt.max())
    assert (e <= 2e-6).all(), e


def test_video_loop_vs_the_reference_s_video_loop(hip_lib, capsys):
    """The video driver's loop on the device -- populations ticking, background / entity / compose passes at t > 0,
    orbit camera, march, bloom -- against frames 0, 2 and 7 of the reference's own loop (`_advance_lifecycle_frame` +
    `render`, render.py:4436-4453) on the same scene.  Same sources of difference as the still e2e frame: libm
    rounding in the background generator moves texels by ~1e-4, the frame inherits that through the bilinear look-ups."""
    from bhr_amd import HipRenderer, drivers
    from bhr_amd.camera import orbit_position
    from test_reference_kernels import E2E_KW, load_video
    g, sky = load_video()
    n_r, n_phi = (int(v) for v in g["tex_shape"])
    r = HipRenderer(320, 180, sky, np.zeros((n_r, n_phi, 4), np.float32), **E2E_KW)
    fac = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
    dt = float(g["speed"])
    stored = [int(f) for f in g["frames"]]
    lines = []
    for frame in range(max(stored) + 1):
        drivers.advance_lifecycle_frame(r, fac, frame * dt, dt, recompute_stats=(frame % 60 == 0), compose=frame in stored)
        if frame not in stored:
            continue
        cam = orbit_position([6.0, 0.0, 0.5], frame, int(g["n_frames"]), float(g["orbit_degrees"]))
        np.testing.assert_allclose(cam, g[f"cam_{frame}"], rtol=0, atol=1e-12)
        img = r.render(cam, 60)
        tex = r.disk_texture_field.to_numpy()
        alive = [len(fac[k].alive_entities) for k in ("filament", "hotspot", "rt_spike")]
        assert alive == list(g[f"alive_{frame}"]), (frame, alive)
        dtx = np.abs(tex - g[f"disk_tex_{frame}"])
        e = _rmse_c(img, g[f"final_{frame}"])
        lines.append(f"[video] frame {frame}: texture max {dtx.max():.3g} median {np.median(dtx):.3g}; frame RMSE {e}, "
                     f"max {np.abs(img - g[f'final_{frame}']).max():.3g}")
        assert np.median(dtx) <= 1e-7 and dtx.max() <= 5e-6, (frame, np.median(dtx), dtx.max())   # measured: max 4.2e-7 (round 2: 1.3e-4)
        assert (e <= 2e-6).all(), (frame, e)                                                        # measured 2.6-4.1e-7 (round 2: 1.3-2.4e-6)
    r.close()
    with capsys.disabled():
        print("\n" + "\n".join(lines))
