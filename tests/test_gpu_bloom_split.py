"""The post-pass on the bf16 matrix cores (csrc/bloom.hip: f32 operands cut into three bf16 parts, six MFMA products per
chunk of 16 taps), which the fast and hybrid arithmetic use from bloom radius 64 up (4k, 8k): against the exact f32
kernels on the same layers, against the oracle's `_bloom_kernel` restatement (render.py:3022-3114), through properties at
8k, and that row blocks give the same bits as one context (chunks are aligned to global multiples of 16)."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
CAM, FOV = [6.0, 0.0, 0.5], 90.0


def _both_blooms(r, disk, bg):
    """(blur, final) of the layers through the bf16 kernels and through the exact f32 kernels of one context"""
    from bhr_amd import _lib
    out = {}
    for math in ("fast", "strict"):                        # the march of a frame picks the post-pass that follows it
        r.render_async(CAM, FOV, math=math, skip_bloom=True)
        r.write_layer(_lib.LAYER_DISK, disk)
        r.write_layer(_lib.LAYER_BG, bg)
        r.bloom_only()
        out[math] = (r.read_layer(_lib.LAYER_BLUR), r.read_layer(_lib.LAYER_FINAL))
    return out["fast"], out["strict"]


def test_4k_split_bloom_against_exact_kernels_and_oracle(oracle, hip_lib):
    from bhr_amd import HipRenderer, _lib
    W, H = 3840, 2160
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **KW)
    r.render_async(CAM, FOV)
    disk, bg = r.read_layer(_lib.LAYER_DISK), r.read_layer(_lib.LAYER_BG)
    assert disk.max() > 0.3
    (blur_s, final_s), (blur_x, final_x) = _both_blooms(r, disk, bg)
    r.close()
    assert not np.array_equal(blur_s, blur_x)              # two different kernels did run
    d = np.abs(blur_s - blur_x)
    e = np.sqrt(np.mean(d.astype(np.float64) ** 2, axis=(0, 1)))
    print(f"\n[bloom split] 4k blur layer, bf16 x 3 against exact f32: max {d.max():.3g}, per-channel RMSE {e}")
    assert d.max() <= 3e-6 and (e <= 5e-7).all(), (d.max(), e)
    assert np.abs(final_s - final_x).max() <= 3e-6
    ora = oracle.OracleRenderer(W, H, sky, tex, **KW)
    ref, _ = ora.bloom(disk.transpose(1, 0, 2))
    ref = ref.transpose(1, 0, 2)
    for name, got in (("bf16 x 3", blur_s), ("exact f32", blur_x)):
        dd = np.abs(got - ref)
        ee = np.sqrt(np.mean(dd.astype(np.float64) ** 2, axis=(0, 1)))
        print(f"[bloom split] 4k blur layer, {name} against the oracle: max {dd.max():.3g}, per-channel RMSE {ee}")
        assert dd.max() <= 5e-6 and (ee <= 1e-6).all(), (name, dd.max(), ee)


def test_8k_split_bloom_properties(hip_lib):
    """radius 153, two stacked tiles per wave: a constant layer is a fixed point (edges included), halving is exact,
    the operator commutes with the left-right flip up to rounding, outputs are convex combinations of inputs"""
    from bhr_amd import HipRenderer, _lib
    W, H = 7680, 4320
    r = HipRenderer(W, H, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), math="fast", frame_slots=1,
                    **dict(KW, step_size=0.5))
    r.render_async(CAM, FOV, skip_bloom=True)              # a fast frame: the bf16 post-pass is the context's
    r.write_layer(_lib.LAYER_BG, np.zeros((H, W, 3), np.float32))

    def bloom(x):
        r.write_layer(_lib.LAYER_DISK, x)
        r.bloom_only()
        return r.read_layer(_lib.LAYER_BLUR)

    const = np.empty((H, W, 3), np.float32)
    const[...] = np.array([0.25, 0.5, 0.125], np.float32)
    np.testing.assert_allclose(bloom(const), const, rtol=0, atol=3e-6)
    rng = np.random.default_rng(8)
    x = np.zeros((H, W, 3), np.float32)
    ys, xs = rng.integers(0, H, 5000), rng.integers(0, W, 5000)
    x[ys, xs] = rng.random((5000, 3), dtype=np.float32)
    x[H // 3:H // 3 + 80, W // 5:W // 5 + 600] = 0.6
    x[:3, :] = 0.9                                          # the image edges: skipped taps, per-pixel renormalisation
    x[:, -2:] = 0.7
    bx = bloom(x)
    assert bx.max() > 0.05 and np.isfinite(bx).all()
    np.testing.assert_array_equal(bloom(0.5 * x), 0.5 * bx)                             # exact: every part halves
    np.testing.assert_allclose(bloom(np.ascontiguousarray(x[:, ::-1])), bx[:, ::-1], rtol=2e-5, atol=3e-6)
    assert bx.max() <= x.max() + 3e-6 and bx.min() >= -1e-9
    r.close()


def test_split_bloom_row_blocks_equal_one_context_bit_for_bit(hip_lib):
    """4k fast frame in 5 uneven row blocks (cuts not multiples of 16 or 32) == one context, every bit of the gathered
    frame: an output's operands meet the same MFMA slots whatever the tiling"""
    from bhr_amd import HipRenderer, multigpu
    W, H = 3840, 2160
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    full = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **KW)
    ref = full.render(CAM, FOV)
    full.close()
    cuts = [0, 401, 918, 1247, 1795, H]
    tiles = [HipRenderer(W, H, sky, tex, rows=(cuts[k], cuts[k + 1]), math="fast", frame_slots=1, **KW) for k in range(5)]
    for sched in ("serial", "pipelined"):
        multigpu.group_render(tiles, CAM, FOV, gather="peer", schedule=sched)
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), ref)
    for t in tiles:
        t.close()


def test_fhd_fast_keeps_the_f32_post_pass_and_odd_widths_fall_back(hip_lib, monkeypatch):
    """A small frame (radius 6 < 64, like fhd's 38): the fast arithmetic keeps the f32 post-pass; a width that is not a multiple of 16 keeps it even
    when the bf16 pass is forced on by the environment switch (the H kernel's chunks must tile a row)."""
    from bhr_amd import HipRenderer, _lib
    s = scenes.SCENES["default"]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    for (w, h), force in (((s["width"], s["height"]), False), ((200, 120), True)):
        if force:
            monkeypatch.setenv("BHR_BLOOM_SPLIT", "1")
        a = HipRenderer(w, h, sky, tex, math="fast", **s["kw"])
        b = HipRenderer(w, h, sky, tex, math="strict", **s["kw"])
        a.render_async([6, 0, 0.5], 90)
        disk, bg = a.read_layer(_lib.LAYER_DISK), a.read_layer(_lib.LAYER_BG)
        monkeypatch.delenv("BHR_BLOOM_SPLIT", raising=False)
        b.render_async([6, 0, 0.5], 90, skip_bloom=True)
        b.write_layer(_lib.LAYER_DISK, disk)
        b.write_layer(_lib.LAYER_BG, bg)
        b.bloom_only()
        np.testing.assert_array_equal(a.read_layer(_lib.LAYER_BLUR), b.read_layer(_lib.LAYER_BLUR))
        a.close()
        b.close()


def test_split_bloom_is_exactly_scale_invariant_over_the_f32_range(hip_lib, monkeypatch):
    """Cutting an f32 value into three bf16 parts commutes with powers of two, and so does every product and sum of the
    kernels: bloom(2^k x) == 2^k bloom(x) bit for bit from 2^-60 to 2^40 (no part under- or overflows anywhere near the
    values a frame holds); a layer of HDR spikes on a faint floor stays within 3e-6 of the exact kernels RELATIVE to the
    local result."""
    from bhr_amd import HipRenderer, _lib
    monkeypatch.setenv("BHR_BLOOM_SPLIT", "1")
    W, H = 1280, 720
    r = HipRenderer(W, H, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), math="fast", frame_slots=1,
                    **dict(KW, step_size=0.5))
    r.render_async(CAM, FOV, skip_bloom=True)
    r.write_layer(_lib.LAYER_BG, np.zeros((H, W, 3), np.float32))

    def bloom(x):
        r.write_layer(_lib.LAYER_DISK, x)
        r.bloom_only()
        return r.read_layer(_lib.LAYER_BLUR)

    rng = np.random.default_rng(3)
    x = (rng.random((H, W, 3), dtype=np.float32) * 1e-3).astype(np.float32)
    ys, xs = rng.integers(0, H, 400), rng.integers(0, W, 400)
    x[ys, xs] = rng.random((400, 3), dtype=np.float32) * 50.0
    base = bloom(x)
    assert np.isfinite(base).all() and base.max() > 0.01
    for k in (-60, -20, 14, 40):
        sc = np.float32(2.0) ** k
        np.testing.assert_array_equal(bloom(x * sc), base * sc, err_msg=f"2^{k}")
    monkeypatch.setenv("BHR_BLOOM_SPLIT", "0")
    r.render_async(CAM, FOV, skip_bloom=True, math="strict")
    exact = bloom(x)
    r.close()
    assert not np.array_equal(exact, base)
    rel = np.abs(base - exact) / np.maximum(exact, 1e-12)
    assert rel.max() <= 3e-6, rel.max()


def test_fhd_fast_frames_take_the_bf16_v_pass_only(hip_lib):
    """1920x1080 (radius 38: between the V pass's threshold of 16 and the H pass's 64): a fast frame's post-pass differs
    from the exact kernels' by the bf16 V pass alone -- within 3e-6 -- and equals an f32 H pass followed by a forced bf16 V."""
    from bhr_amd import HipRenderer, _lib
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    r = HipRenderer(1920, 1080, sky, tex, math="fast", frame_slots=1, **KW)
    r.render_async(CAM, FOV)
    disk, bg = r.read_layer(_lib.LAYER_DISK), r.read_layer(_lib.LAYER_BG)
    (blur_s, final_s), (blur_x, final_x) = _both_blooms(r, disk, bg)
    r.close()
    d = np.abs(blur_s - blur_x)
    assert 0 < d.max() <= 3e-6 and np.abs(final_s - final_x).max() <= 3e-6, d.max()


@pytest.mark.parametrize("seed", range(12))
def test_random_row_blocks_and_sizes_split_bloom(seed, hip_lib, monkeypatch):
    """Small frames with the bf16 post-pass forced on (radius 6 ... 25, one and two tiles per wave), cut into 2-6 random
    row blocks, some thinner than the radius (multi-hop halo): the gathered frame of both schedules == one context bit for
    bit, and the one-context frame sits within 3e-6 of the exact f32 kernels."""
    from bhr_amd import HipRenderer, multigpu
    rng = np.random.default_rng(100 + seed)
    # widths that are multiples of 16 but not of 32 or 128 leave partial column strips / output tiles at the right edge
    W, H = [(640, 400), (320, 208), (960, 540), (1280, 720), (400, 205), (1008, 567)][seed % 6]      # heights: any
    monkeypatch.setenv("BHR_BLOOM_SPLIT", "1")
    if seed % 2:
        monkeypatch.setenv("BHR_BLOOM_H", "bf16x2")
        monkeypatch.setenv("BHR_BLOOM_V", "bf16x2")
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    kw = dict(KW, disk_tilt=float(rng.uniform(-30, 30)))
    cam = [float(rng.uniform(4, 9)), float(rng.uniform(-2, 2)), float(rng.uniform(-1.5, 1.5))]
    full = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **kw)
    ref = full.render(cam, FOV)
    monkeypatch.setenv("BHR_BLOOM_SPLIT", "0")
    monkeypatch.delenv("BHR_BLOOM_H", raising=False)
    monkeypatch.delenv("BHR_BLOOM_V", raising=False)
    exact = full.render(cam, FOV)
    full.close()
    assert not np.array_equal(ref, exact) and np.abs(ref - exact).max() <= 3e-6
    monkeypatch.setenv("BHR_BLOOM_SPLIT", "1")
    if seed % 2:
        monkeypatch.setenv("BHR_BLOOM_H", "bf16x2")
        monkeypatch.setenv("BHR_BLOOM_V", "bf16x2")
    n = int(rng.integers(2, 7))
    inner = np.sort(rng.choice(np.arange(3, H - 3), size=n - 1, replace=False))
    inner = [int(c) for c in inner]
    cuts = [0] + [c for k, c in enumerate(inner) if k == 0 or c - inner[k - 1] >= 3] + [H]
    tiles = [HipRenderer(W, H, sky, tex, rows=(cuts[k], cuts[k + 1]), math="fast", frame_slots=1, **kw) for k in range(len(cuts) - 1)]
    for sched in ("serial", "pipelined"):
        multigpu.group_render(tiles, cam, FOV, gather="peer", schedule=sched)
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), ref, err_msg=f"{W}x{H} cuts {cuts} {sched}")
    for t in tiles:
        t.close()
