"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerance (BASELINE.json north_star): per-channel RMSE <= 1e-4 on the pre-tonemap float image.
The march is chaotic next to the photon ring: two correct f32 evaluations with different operation
orders differ by ~1e-4 RMSE on noisy textures (the strict f32 oracle itself is 0.6-1.7e-4 away from
the binary64 value of the same algorithm).  The default kernel therefore reproduces the reference's
operation order exactly; the opt-in fast kernel is bounded against the binary64 yardstick.
"""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4  # north-star bar


def _rmse(a, b):
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=(0, 1)))


def _make(scene, sky, tex, oracle, **hip_kw):
    from bhr_amd import HipRenderer
    s = scenes.SCENES[scene]
    hip = HipRenderer(s["width"], s["height"], sky, tex, **hip_kw, **s["kw"])
    ora = oracle.OracleRenderer(s["width"], s["height"], sky, tex, **s["kw"])
    return s, hip, ora


# The default ("strict") march evaluates the RK4 loop in the reference's operation order with IEEE
# sqrt/divide, so ray paths are bit-identical to the oracle and only the per-hit transcendentals
# (atan2, pow, exp, acos: ocml vs glibc, <= 2 ulp) differ: the north-star bar of 1e-4 is met with
# two orders of magnitude to spare and the assert uses the tighter figure.
STRICT_RMSE_TOL = 5e-6


@pytest.mark.parametrize("scene", list(scenes.SCENES))
@pytest.mark.parametrize("persistent", [False, True])
def test_render_matches_oracle(scene, persistent, oracle, hip_lib):
    from bhr_amd import _lib
    sky = scenes.analytic_skybox()
    tex = scenes.noisy_disk()
    s, hip, ora = _make(scene, sky, tex, oracle)
    assert hip.math == "strict"
    hip.render_async(s["cam_pos"], s["fov"], compaction=persistent)
    final = hip.read_layer(_lib.LAYER_FINAL)
    bg = hip.read_layer(_lib.LAYER_BG)
    disk = hip.read_layer(_lib.LAYER_DISK)
    blur = hip.read_layer(_lib.LAYER_BLUR)
    ref, rbg, rdisk, rblur = ora.render(s["cam_pos"], s["fov"], parts=True)
    rbg, rdisk, rblur = (x.transpose(1, 0, 2) for x in (rbg, rdisk, rblur))

    assert final.shape == ref.shape == (s["height"], s["width"], 3)
    assert np.isfinite(final).all()
    for name, a, b in (("bg", bg, rbg), ("disk", disk, rdisk), ("blur", blur, rblur), ("final", final, ref)):
        e = _rmse(a, b)
        assert (e <= STRICT_RMSE_TOL).all(), f"{scene}/{name}: per-channel RMSE {e} > {STRICT_RMSE_TOL}"
        assert np.abs(a - b).max() <= 1e-4, f"{scene}/{name}: max abs diff {np.abs(a - b).max()}"

    # ray-step accounting: the in-kernel counter equals the oracle's loop count exactly
    c = hip.counters()
    assert c["rays"] == s["width"] * s["height"]
    assert c["ray_steps"] == ora.last_total_steps
    hip.close()


@pytest.mark.parametrize("scene", list(scenes.SCENES))
def test_fast_math_stays_within_f32_rounding_noise(scene, oracle, hip_lib):
    """math="fast" re-orders the arithmetic (2-D orbital-plane state, v_rsq, fast-math), so it cannot
    be bit-identical; what it must not do is add error beyond f32 rounding noise.  Yardstick: the
    binary64 evaluation of the same algorithm (oracle f64 build).  The strict f32 oracle itself sits
    e_ref away from it; the fast kernel may be at most 3x as far, and on the scenes of the
    BASELINE configs it also meets the absolute 1e-4 bar against the f64 value."""
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    s, hip, ora = _make(scene, sky, tex, oracle, math="fast")
    out = hip.render(s["cam_pos"], s["fov"]).astype(np.float64)
    ref = ora.render(s["cam_pos"], s["fov"]).astype(np.float64)
    truth = oracle.OracleRenderer(s["width"], s["height"], sky, tex, fast="f64", **s["kw"]).render(
        s["cam_pos"], s["fov"]).astype(np.float64)
    e_ref = float(np.sqrt(np.mean((ref - truth) ** 2)))
    e_fast = float(np.sqrt(np.mean((out - truth) ** 2)))
    assert e_fast <= 3.0 * e_ref + 1e-5, (e_fast, e_ref)
    if scene != "fine_ragged":     # stress scene: e_ref itself is 1.7e-4 there
        assert e_fast <= 1.5e-4, e_fast
    c = hip.counters()
    assert abs(c["ray_steps"] - ora.last_total_steps) <= 1e-4 * ora.last_total_steps
    hip.close()


def test_star_field_sensitivity(oracle, hip_lib):
    """Single-texel stars + strong lensing: the bar still holds."""
    sky = scenes.star_skybox()
    tex = scenes.analytic_disk()
    s, hip, ora = _make("default", sky, tex, oracle)
    out = hip.render(s["cam_pos"], s["fov"])
    ref = ora.render(s["cam_pos"], s["fov"])
    assert (_rmse(out, ref) <= STRICT_RMSE_TOL).all(), _rmse(out, ref)
    hip.close()


def test_skip_flags(oracle, hip_lib):
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    s, hip, ora = _make("tilt_aa", sky, tex, oracle)
    for kw in (dict(skip_bloom=True), dict(skip_differentials=True), dict(skip_bloom=True, skip_differentials=True)):
        out = hip.render(s["cam_pos"], s["fov"], **kw)
        ref = ora.render(s["cam_pos"], s["fov"], **kw)
        assert (_rmse(out, ref) <= STRICT_RMSE_TOL).all(), (kw, _rmse(out, ref))
    hip.close()


def test_frame_rotation_offset(oracle, hip_lib):
    """frame != 0 -> t_offset = frame * disk_rotation_speed shifts the texture lookup (render.py:3897)."""
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    s, hip, ora = _make("default", sky, tex, oracle)
    out = hip.render(s["cam_pos"], s["fov"], frame=37)
    ref = ora.render(s["cam_pos"], s["fov"], frame=37)
    assert (_rmse(out, ref) <= STRICT_RMSE_TOL).all()
    assert _rmse(out, ora.render(s["cam_pos"], s["fov"], frame=0)).max() > 1e-3  # it really moved
    hip.close()


def test_bloom_isolated(oracle, hip_lib):
    """Bloom alone: feed the oracle's bloom the GPU's own disk layer -> tight tolerance."""
    from bhr_amd import _lib
    sky, tex = scenes.analytic_skybox(), scenes.analytic_disk()
    s, hip, ora = _make("default", sky, tex, oracle)
    hip.render_async(s["cam_pos"], s["fov"])
    disk = hip.read_layer(_lib.LAYER_DISK)
    blur = hip.read_layer(_lib.LAYER_BLUR)
    rblur, _ = ora.bloom(np.ascontiguousarray(disk.transpose(1, 0, 2)))
    np.testing.assert_allclose(blur, rblur.transpose(1, 0, 2), atol=2e-6, rtol=1e-5)
    hip.close()


@pytest.mark.parametrize("w,h", [(6500, 300), (3300, 200)])
def test_bloom_wide_frame_tall_tiles(w, h, oracle, hip_lib):
    """The V pass picks its tile height from R = int(0.02 W): 32 rows below R = 64 (every other test), 128 rows
    up to R = 127 (w = 3300), 256 rows above (w = 6500, the 8k case).  Wide strips with a synthetic disk layer
    through the standalone layer API, against the oracle's bloom; ragged in both directions."""
    from bhr_amd import HipRenderer, _lib
    rng = np.random.default_rng(11)
    disk = np.zeros((h, w, 3), np.float32)
    ys, xs = rng.integers(0, h, 4000), rng.integers(0, w, 4000)
    disk[ys, xs] = rng.random((4000, 3), dtype=np.float32)
    disk[100:140, 1500:2100] = 0.7
    hip = HipRenderer(w, h, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32))
    hip.write_layer(_lib.LAYER_DISK, disk)
    hip.write_layer(_lib.LAYER_BG, np.zeros_like(disk))
    hip.bloom_only()
    blur = hip.read_layer(_lib.LAYER_BLUR)
    ora = oracle.OracleRenderer(w, h, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), fast=True)
    rblur, _ = ora.bloom(np.ascontiguousarray(disk.transpose(1, 0, 2)))
    np.testing.assert_allclose(blur, rblur.transpose(1, 0, 2), atol=3e-6, rtol=2e-5)
    final = hip.read_layer(_lib.LAYER_FINAL)
    np.testing.assert_allclose(final, np.clip(disk + blur, 0, 1), atol=1e-6)
    hip.close()


def test_u8_quantisation(hip_lib):
    sky, tex = scenes.analytic_skybox(), scenes.analytic_disk()
    from bhr_amd import HipRenderer
    s = scenes.SCENES["default"]
    hip = HipRenderer(s["width"], s["height"], sky, tex, **s["kw"])
    img = hip.render(s["cam_pos"], s["fov"])
    u8 = hip.read_final_u8()
    np.testing.assert_array_equal(u8, (np.clip(img, 0, 1) * 255).astype(np.uint8))  # truncation, render.py:423
    hip.close()


def test_row_block_equals_full_frame(hip_lib):
    """A context restricted to rows [r0, r1) renders exactly those rows of the full frame
    (march is per pixel; bloom differs only through missing neighbours, so compare without it)."""
    from bhr_amd import HipRenderer, _lib
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    s = scenes.SCENES["fine_ragged"]
    full = HipRenderer(s["width"], s["height"], sky, tex, **s["kw"])
    ref = full.render(s["cam_pos"], s["fov"], skip_bloom=True)
    part = HipRenderer(s["width"], s["height"], sky, tex, rows=(40, 93), **s["kw"])
    out = part.render(s["cam_pos"], s["fov"], skip_bloom=True)
    assert out.shape == (53, s["width"], 3)
    np.testing.assert_array_equal(out, ref[40:93])
    full.close()
    part.close()


def test_group_render_tiles_match_single_context(hip_lib):
    """bhr_group_render with 4 row-block contexts (all on device 0 here) == one full-frame
    context, bloom halo exchange included."""
    import ctypes as C
    from bhr_amd import HipRenderer, _lib
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    s = scenes.SCENES["default"]
    H = s["height"]
    full = HipRenderer(s["width"], H, sky, tex, **s["kw"])
    ref = full.render(s["cam_pos"], s["fov"])
    cuts = [0, 50, 54, 120, H]   # the 4-row tile is thinner than the bloom radius (6): two-hop halo
    n = len(cuts) - 1
    tiles = [HipRenderer(s["width"], H, sky, tex, rows=(cuts[k], cuts[k + 1]), **s["kw"]) for k in range(n)]
    arr = (C.c_void_p * n)(*[t._ctx for t in tiles])
    out = np.empty((H, s["width"], 3), dtype=np.float32)
    cam = full.camera_uniforms(s["cam_pos"], s["fov"])
    _lib.check(hip_lib.bhr_group_render(arr, n, C.byref(cam), 0, _lib.fptr(out)))
    np.testing.assert_allclose(out, ref, atol=1e-6, rtol=0)
    for t in tiles + [full]:
        t.close()


def test_exact_arithmetic_selftest(hip_lib):
    """The strict march's hand-written sqrt / divide / divide-by-6 sequences equal hipcc's IEEE
    sqrtf and operator/ on the device: every f32 in [2^-80, 2^80) for sqrt, 1/x and x/6, 2^30 random
    operand pairs for the general divide."""
    from bhr_amd import HipRenderer
    s = scenes.SCENES["default"]
    hip = HipRenderer(64, 36, scenes.analytic_skybox(32, 64), scenes.analytic_disk(16, 32), **s["kw"])
    r = hip.selftest()
    assert r["checked"] > 5 * (160 << 23)
    assert (r["bad_sqrt"], r["bad_div"], r["bad_div6"]) == (0, 0, 0), r
    hip.close()


@pytest.mark.parametrize("cam,fov", [([8.0, 0.0, 0.0], 60), ([60.0, 0.0, 5.0], 30), ([1.6, 0.0, 0.2], 100), ([0.0, 0.0, 9.0], 70)])
def test_fast_march_in_the_rays_own_clock_radial_and_extreme_rays(cam, fov, hip_lib):
    """The fast march rescales every ray's affine parameter by tau = (1.5 L2)^(-1/2) (csrc/march.hip, "the ray's own clock").
    Odd frame sizes put a pixel exactly on the optical axis: for a camera looking at the hole that ray is radial, L2 = 0, and
    tau comes from the clamp; far cameras have large L2 (small tau), cameras inside the photon sphere the opposite.  The frame
    must be finite everywhere, the axis pixel must end as strict's does, step totals and layers must agree with the strict
    (reference-order) march within the fast arithmetic's noise."""
    from bhr_amd import HipRenderer, _lib
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    w, h = 129, 73
    kw = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
    out = {}
    for math in ("strict", "fast"):
        r = HipRenderer(w, h, sky, tex, math=math, **kw)
        r.render_async(cam, fov)
        out[math] = dict(final=r.read_layer(_lib.LAYER_FINAL), bg=r.read_layer(_lib.LAYER_BG), disk=r.read_layer(_lib.LAYER_DISK),
                         steps=r.counters()["ray_steps"])
        r.close()
    f, s = out["fast"], out["strict"]
    for k in ("final", "bg", "disk"):
        assert np.isfinite(f[k]).all(), k
    assert abs(f["steps"] - s["steps"]) <= 5e-3 * s["steps"], (f["steps"], s["steps"])
    cy, cx = h // 2, w // 2
    np.testing.assert_allclose(f["bg"][cy, cx], s["bg"][cy, cx], atol=1e-3)      # the axis ray: captured, or straight out
    np.testing.assert_allclose(f["disk"][cy, cx], s["disk"][cy, cx], atol=1e-3)
    # pixels whose ray grazes the photon ring amplify any rounding (the hybrid march keeps those strict): compare where the
    # two marches agree on what the ray did -- all but a handful of pixels -- and bound the share of the rest
    d = np.abs(f["final"] - s["final"]).max(axis=2)
    assert (d > 0.05).mean() <= 0.01, float((d > 0.05).mean())
    assert np.median(d) <= 1e-5, float(np.median(d))
