"""The oracle against values produced by the reference's OWN kernel statements (CPU, no GPU).

`tests/golden/make_kernel_golden.py` ran the unmodified `@ti.kernel` / `@ti.func` function objects of
/root/reference/render.py (march 2787-3018, bloom 3022-3114, device functions 2407-2637, noise 2642-2785,
background 3332-3451, compose 3169-3257, mips 3261-3283) as plain Python on a primitive-op stand-in for
the taichi module (`tests/golden/ti_shim.py`), once in binary64 and once with every operation rounded
to binary32.  These tests pin `oracle/bhr_oracle.c` -- a restatement written from a reading of the same
source -- to those values: a misreading of a formula, constant, branch, index or operation order shows
up here.  The GPU twin of this file (test_gpu_reference_kernels.py) compares the HIP kernels with the
same fixtures directly.

Tolerances.  March/bloom, f32 fixtures vs the strict f32 build: the same IEEE operations in the same
order -- step counts must be equal on every pixel, values differ only through libm in the per-hit shading
(glibc's f32 routines vs correctly rounded results; measured max 6e-6, RMSE <= 2.4e-7).  March/bloom, f64
fixtures vs the binary64 build: equal step counts, escape directions to 1e-15, layers to the f32 rounding of
the buffers (measured max 6e-6, RMSE <= 3.2e-7).  Noise / background / compose: the f32 comparison is the
sharp one (simplex and FBM bit-identical, compose 1 ulp); the binary64 build of the oracle keeps f32-valued
literals (1/3, 1/6, 0.6, pi) where the f64 fixtures use binary64 literals, and simplex noise with the
reference's 0.6 kernel radius is discontinuous across cell faces, so that pair is only bounded loosely.
"""
import hashlib
import os

import numpy as np
import pytest

from bhr_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MARCH = ["default", "e2e", "tilt_aa", "far_aa", "inside", "polar", "fine_rot"]
# the reference-constructor arguments of make_kernel_golden.MARCH_SCENES (kept beside the fixtures' metadata)
KW = {
    "default": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0, anti_alias="disabled"),
    "e2e": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15.0, anti_alias="disabled"),
    "tilt_aa": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=25.0,
                    anti_alias="lod_radius", aa_strength=1.0),
    "far_aa": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=20.0,
                   anti_alias="lod_radius", aa_strength=1.5),
    "inside": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=9.0, disk_tilt=3.0, anti_alias="disabled"),
    "polar": dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=6.0, disk_tilt=0.0, anti_alias="lod_radius"),
    "fine_rot": dict(step_size=0.05, r_max=10.0, r_disk_inner=2.0, r_disk_outer=8.0, disk_tilt=5.0,
                     anti_alias="disabled", disk_rotation_speed=0.1),
}
FLARE = {"tilt_aa"}


def load_scene(name):
    """-> (fixture, sky, tex): the inputs are recreated from seeds and checked against the stored hashes."""
    g = np.load(os.path.join(GOLD, f"march_ref_{name}.npz"))
    sky = scenes.analytic_skybox(*g["sky_shape"])
    tex = scenes.noisy_disk(*g["tex_shape"])
    assert hashlib.sha256(sky.tobytes()).hexdigest() == str(g["sky_sha256"]), "skybox input drifted"
    assert hashlib.sha256(tex.tobytes()).hexdigest() == str(g["tex_sha256"]), "disk texture input drifted"
    return g, sky, tex


def _rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def _oracle_layers(oracle, name, g, sky, tex, build):
    o = oracle.OracleRenderer(int(g["width"]), int(g["height"]), sky, tex, fast=build, **KW[name])
    final, bg, disk, blur = o.render(list(g["cam_pos"]), float(g["fov"]), frame=int(g["frame"]), parts=True)
    if name in FLARE:
        from oracle import flare_np
        final = flare_np.apply_lens_flare(final, disk.transpose(1, 0, 2))
    return o, dict(final=final, bg=bg, disk=disk, blur=blur)


@pytest.mark.parametrize("name", MARCH)
def test_oracle_f32_reproduces_reference_statements_in_binary32(name, oracle):
    g, sky, tex = load_scene(name)
    o, lay = _oracle_layers(oracle, name, g, sky, tex, False)
    # ray paths: every pixel takes exactly the reference's number of RK4 steps
    assert np.array_equal(o.last_steps, g["f32_steps"]), \
        f"{name}: {(o.last_steps != g['f32_steps']).sum()} pixels with a different step count"
    for k in ("bg", "disk", "blur", "final"):
        a, b = lay[k], g[f"f32_{k}"]
        assert a.shape == b.shape
        d = np.abs(a.astype(np.float64) - b)
        assert d.max() <= 1e-5 and _rmse(a, b) <= 5e-7, f"{name}/{k}: max {d.max():.3g} rmse {_rmse(a, b):.3g}"


@pytest.mark.parametrize("name", MARCH)
def test_oracle_f64_reproduces_reference_statements_in_binary64(name, oracle):
    g, sky, tex = load_scene(name)
    o, lay = _oracle_layers(oracle, name, g, sky, tex, "f64")
    assert np.array_equal(o.last_steps, g["f64_steps"])
    for k in ("bg", "disk", "blur", "final"):
        a, b = lay[k], g[f"f64_{k}"]
        d = np.abs(a.astype(np.float64) - b)
        assert d.max() <= 1e-5 and _rmse(a, b) <= 5e-7, f"{name}/{k}: max {d.max():.3g} rmse {_rmse(a, b):.3g}"
    # escape directions handed to _sample_skybox (render.py:2921, 3013)
    esc = o.escape_directions(list(g["cam_pos"]), float(g["fov"]))
    ref = g["f64_escape_dir"]
    seen = np.abs(ref).sum(axis=2) > 0
    assert np.array_equal(seen, np.abs(esc).sum(axis=2) > 0)
    assert np.abs(esc - ref)[seen].max() <= 1e-12


@pytest.mark.parametrize("name", MARCH)
def test_bloom_in_place_update_of_the_disk_layer(name, oracle):
    """render.py:3112-3114: the kernel leaves clamp(disk + 0.4 blur) in the layer (used by render_to_field)."""
    g, sky, tex = load_scene(name)
    o = oracle.OracleRenderer(int(g["width"]), int(g["height"]), sky, tex, **KW[name])
    blur, after = o.bloom(g["f32_disk"])
    assert np.abs(after - g["f32_disk_after_bloom"]).max() <= 2e-6
    assert np.abs(blur - g["f32_blur"]).max() <= 2e-6


@pytest.mark.parametrize("name", MARCH)
@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_g_factor_on_the_reference_s_own_calls(name, mode, oracle):
    """_apply_g_factor (2439-2516): Doppler + redshift, luminosity law, radial boost, Wien shift, tint."""
    g, _, _ = load_scene(name)
    calls = g[f"{mode}_gfactor_calls"].astype(np.float64)
    assert len(calls) > 50
    out = oracle.probe_g_factor(calls[:, :16], fast=("f64" if mode == "f64" else False))
    ref = calls[:, 16:19]
    rel = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    assert rel.max() <= (3e-6 if mode == "f32" else 1e-6), rel.max()
    assert ref.max() > 0.05  # the sample is not all dark


@pytest.mark.parametrize("name", ["tilt_aa", "far_aa", "polar"])
@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_lod_sampler_on_the_reference_s_own_calls(name, mode, oracle):
    """_sample_disk_mip (2600-2637) with the LOD values the reference's march computed (2961-2990)."""
    g, _, tex = load_scene(name)
    calls = g[f"{mode}_mip_calls"].astype(np.float64)
    assert len(calls) > 50
    mips = oracle.build_mips_padded(tex)
    out = oracle.probe_disk_mip(mips, calls[:, :6], fast=("f64" if mode == "f64" else False))
    # atan2f (glibc) vs the correctly rounded angle, times n_phi = 512 texels of a noisy texture
    assert np.abs(out - calls[:, 6:10]).max() <= 2e-5


def test_lod_values_cover_every_sampled_level():
    lods = np.concatenate([np.load(os.path.join(GOLD, f"march_ref_{n}.npz"))["f64_mip_calls"][:, 5]
                           for n in ("tilt_aa", "far_aa", "polar")])
    assert set(np.floor(lods).astype(int)) == {0, 1, 2, 3}, sorted(set(np.floor(lods).astype(int)))


@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_skybox_sampler_on_escape_directions(mode, oracle):
    g, sky, tex = load_scene("default")
    esc = g[f"{mode}_escape_dir"].astype(np.float64).reshape(-1, 3)
    bg = g[f"{mode}_bg"].astype(np.float64).reshape(-1, 3)
    disk = g[f"{mode}_disk"].reshape(-1, 3)
    clear = (np.abs(esc).sum(axis=1) > 0) & (disk.sum(axis=1) == 0)      # escaped without crossing the disk
    out = oracle.probe_skybox(sky, esc[clear], fast=("f64" if mode == "f64" else False))
    assert clear.sum() > 200
    assert np.abs(out - bg[clear]).max() <= 5e-6


# --------------------------------------------------------------------------- bloom
@pytest.mark.parametrize("tag", ["wide", "narrow"])
def test_bloom_kernel(tag, oracle):
    g = np.load(os.path.join(GOLD, "bloom_ref.npz"))
    layer = g[f"{tag}_layer"]
    W, H = layer.shape[:2]
    assert int(g[f"{tag}_radius"]) == int(W * 0.02)
    for mode, build, tol in (("f32", False, 2e-6), ("f64", "f64", 2e-7)):
        o = oracle.OracleRenderer(W, H, scenes.analytic_skybox(8, 16), scenes.analytic_disk(16, 32), fast=build)
        blur, after = o.bloom(layer)
        assert np.abs(blur - g[f"{tag}_{mode}_blur"]).max() <= tol
        assert np.abs(after - g[f"{tag}_{mode}_layer_after"]).max() <= tol
    assert g["wide_f64_blur"].max() > 0.05


# --------------------------------------------------------------------------- noise / background / compose / mips
@pytest.fixture(scope="module")
def texg():
    return np.load(os.path.join(GOLD, "texture_ref.npz"))


def test_simplex_and_fbm(texg, oracle):
    c = texg["noise_coords"]
    for key, kw in (("simplex", dict(mode="simplex")), ("fbm_4_05_2", dict(mode="fbm", octaves=4, persistence=0.5,
                                                                         lacunarity=2.0))):
        out = oracle.eval_noise(c, **kw)
        assert np.array_equal(out, texg[f"f32_{key}"]), key           # bit-identical: no libm call in the noise
        small = np.abs(c).max(axis=1) <= 4
        assert np.abs(out - texg[f"f64_{key}"])[small].max() <= 2e-5, key
    out = oracle.eval_noise(c[:300], mode="fbm", octaves=5, persistence=0.45, lacunarity=2.0)
    assert np.array_equal(out, texg["f32_fbm_5_045_2"])
    assert np.abs(texg["f64_simplex"]).max() > 0.5
    # lattice points: every corner kernel is evaluated at its edge
    lattice = np.all(c == np.round(c), axis=1)
    assert lattice.sum() == 60


# comp plane -> tolerance.  t = 0: sin/cos see exactly representable products, everything agrees to an ulp.
# t > 0: cosf/sinf (glibc) vs correctly rounded values differ in the last bit, and planes 3 / 12 scale the
# unit circle by up to 800 before the noise lookup (render.py:3406-3409, 3443-3446).
_BG_TOL = {0: 2e-7, 1: 0.0, 2: 0.0, 3: 6e-5, 4: 3e-6, 11: 1e-6, 12: 6e-5}


@pytest.mark.parametrize("t", [0.0, 5.0, 36.5])
def test_background_generator(t, texg, oracle):
    ref = texg[f"f32_bg_t{t:g}"]
    n_r, n_phi = ref.shape[1:]
    az_freq, az_shear = int(texg["bg_az"][0]), float(texg["bg_az"][1])
    assert (az_freq, round(az_shear, 6)) == (2, 2.877757)            # init_background_layer(seed=42) draws
    out = oracle.generate_background(n_r, n_phi, az_freq, az_shear, 2.0, 15.0, t)
    for idx, tol in _BG_TOL.items():
        d = np.abs(out[idx] - ref[idx])
        assert d.max() <= (tol if t > 0 else min(tol, 2e-7)), (t, idx, d.max())
    assert not out[5:11].any() and not ref[5:11].any()
    assert ref[3].std() > 0.05 and ref[12].std() > 0.05 and ref[11].std() > 0.05
    # f32 against the binary64 value of the same statements: loose (discontinuous noise, see module docstring)
    f64 = texg[f"f64_bg_t{t:g}"]
    assert np.median(np.abs(out[3] - f64[3])) <= 2e-5 and np.abs(out[0] - f64[0]).max() <= 1e-4


@pytest.mark.parametrize("t", [0.0, 12.5])
def test_compose_kernel_and_mips(t, texg, oracle):
    tex = oracle.compose_disk_texture(texg["compose_comp"], texg["compose_omega"], texg["compose_edge"],
                                      texg["compose_stats"], texg["compose_row_stats"], t_offset=t)
    assert np.abs(tex - texg[f"f32_tex_t{t:g}"]).max() <= 2.5e-7
    # f32 against binary64: sqrt(temp) near 0 and the tint's own jump at 6600 K amplify f32 rounding (4.2e-4 measured)
    assert np.abs(tex - texg[f"f64_tex_t{t:g}"]).max() <= 1e-3
    mips = oracle.build_mips_padded(tex)
    assert mips.shape == texg[f"f32_mips_t{t:g}"].shape
    assert np.abs(mips - texg[f"f32_mips_t{t:g}"]).max() <= 2.5e-7
    if t > 0:   # the roll moved columns (render.py:3202-3207)
        assert np.abs(texg["f32_tex_t12.5"] - texg["f32_tex_t0"]).max() > 0.05


@pytest.mark.parametrize("mode", ["f32", "f64"])
def test_tint_on_the_reference_s_own_calls(mode, texg, oracle):
    calls = texg[f"{mode}_tint_calls"].astype(np.float64)
    out = oracle.probe_tint(calls[:, 0], fast=("f64" if mode == "f64" else False))
    assert np.abs(out - calls[:, 1:4]).max() <= (2e-6 if mode == "f32" else 1e-6)
    assert calls[:, 0].min() < 6600 < calls[:, 0].max()   # both branches of the 66-hundred-kelvin split


# --------------------------------------------------------------------------- the reference's e2e frame
def load_e2e():
    """tests/golden/e2e_ref.npz: the reference's own `render_image` on its tests/e2e_render.py scene (320x180),
    every Taichi kernel of the pipeline executed as binary32 Python (make_e2e_golden.py).  -> fixture, sky."""
    import hashlib as _h
    from bhr_amd.skybox import generate_skybox
    g = np.load(os.path.join(GOLD, "e2e_ref.npz"))
    sky = generate_skybox(2048, 1024, seed=42, n_stars=100)
    assert _h.sha256(np.ascontiguousarray(sky).tobytes()).hexdigest() == str(g["sky_sha256"]), \
        "host skybox generator no longer reproduces the reference's sky"
    return g, sky


E2E_KW = dict(step_size=0.1, r_max=10, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15, anti_alias="disabled")


def test_e2e_frame_march_and_bloom_on_the_reference_s_texture(oracle):
    """The oracle's march + bloom on the texture and sky the reference's pipeline produced, against the frame the
    reference's render() returned.  (Its MD5 differs from tests/e2e_baseline.txt, as it must: that hash belongs to
    the author's LLVM fast-math build and carries no values; both are recorded in the fixture.)"""
    g, sky = load_e2e()
    assert str(g["md5"]) != str(g["baseline_md5"]) and len(str(g["baseline_md5"])) == 32
    o = oracle.OracleRenderer(320, 180, sky, g["disk_tex"], **E2E_KW)
    out = o.render([6, 0, 0.5], 60)
    d = np.abs(out.astype(np.float64) - g["final"])
    assert d.max() <= 2e-5 and np.sqrt((d ** 2).mean()) <= 5e-7, (d.max(), np.sqrt((d ** 2).mean()))
    assert g["final"].max() > 0.5 and g["disk_tex"][..., 3].max() > 0.5


# --------------------------------------------------------------------------- frames of the reference's video loop
def load_video():
    """tests/golden/video_ref.npz: frames of the reference's own video loop on the e2e scene (24-frame orbit; stored
    frames 0, 2 and 7), every kernel executed as binary32 Python (make_video_golden.py).  -> fixture, sky."""
    import hashlib as _h
    from bhr_amd.skybox import generate_skybox
    g = np.load(os.path.join(GOLD, "video_ref.npz"))
    sky = generate_skybox(2048, 1024, seed=42, n_stars=int(g["n_stars"]))
    assert _h.sha256(np.ascontiguousarray(sky).tobytes()).hexdigest() == str(g["sky_sha256"])
    return g, sky


def test_video_frames_march_and_bloom_on_the_reference_s_textures(oracle):
    """Every stored frame of the reference's video loop: the oracle's march + bloom from the orbit camera on the
    texture the reference's lifecycle / background / compose kernels produced for that frame."""
    g, sky = load_video()
    for f in g["frames"]:
        o = oracle.OracleRenderer(320, 180, sky, g[f"disk_tex_{f}"], **E2E_KW)
        out = o.render(list(g[f"cam_{f}"]), 60)
        d = np.abs(out.astype(np.float64) - g[f"final_{f}"])
        assert d.max() <= 5e-5 and np.sqrt((d ** 2).mean()) <= 1e-6, (int(f), d.max(), np.sqrt((d ** 2).mean()))
    frames = [int(f) for f in g["frames"]]
    assert np.abs(g[f"final_{frames[0]}"] - g[f"final_{frames[-1]}"]).mean() > 1e-3          # the camera really moved
    assert np.abs(g[f"disk_tex_{frames[0]}"] - g[f"disk_tex_{frames[1]}"]).max() > 1e-3      # and so did the disk
