"""The post-pass on the f16 matrix cores (csrc/bloom.hip: every f32 operand cut into two f16 halves, three MFMA products
per chunk of 16 taps; operands pre-cut and in fragment order -- the march writes the H pass's input, the H pass the V
pass's), which the fast and hybrid arithmetic use at every size: against the exact f32 kernels on the same layers, against
the oracle's `_bloom_kernel` restatement (render.py:3022-3114), through properties at 8k, that row blocks give the same
bits as one context (chunks are aligned to global multiples of 16; halo rows stored by the neighbours' H passes, frame
rows by the V passes, no copies), and that the V pass stores only what was asked for -- the rest on demand, same bits."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
CAM, FOV = [6.0, 0.0, 0.5], 90.0


def _both_blooms(r, disk, bg):
    """(blur, final) of the layers through the split-f16 kernels and through the exact f32 kernels of one context"""
    from bhr_amd import _lib
    out = {}
    for split in (1, 0):
        r.set_option("bloom_split", split)
        r.render_async(CAM, FOV, skip_bloom=True)
        r.write_layer(_lib.LAYER_DISK, disk)
        r.write_layer(_lib.LAYER_BG, bg)
        r.bloom_only()
        out[split] = (r.read_layer(_lib.LAYER_BLUR), r.read_layer(_lib.LAYER_FINAL))
    r.set_option("bloom_split", -1)
    return out[1], out[0]


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160)])
def test_split_bloom_against_exact_kernels_and_oracle(size, oracle, hip_lib):
    from bhr_amd import HipRenderer, _lib
    W, H = size
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, outputs="f32+blur", **KW)
    r.render_async(CAM, FOV)
    disk, bg = r.read_layer(_lib.LAYER_DISK), r.read_layer(_lib.LAYER_BG)
    blur_frame, final_frame = r.read_layer(_lib.LAYER_BLUR), r.read_layer(_lib.LAYER_FINAL)
    assert disk.max() > 0.3
    (blur_s, final_s), (blur_x, final_x) = _both_blooms(r, disk, bg)
    r.close()
    # the frame's own post-pass (H input written by the march kernel) == the stand-alone pass (H input packed from the layer)
    np.testing.assert_array_equal(blur_frame, blur_s)
    np.testing.assert_array_equal(final_frame, final_s)
    assert not np.array_equal(blur_s, blur_x)              # two different kernels did run
    d = np.abs(blur_s - blur_x)
    e = np.sqrt(np.mean(d.astype(np.float64) ** 2, axis=(0, 1)))
    print(f"\n[bloom split] {W}x{H} blur layer, f16 x 2 against exact f32: max {d.max():.3g}, per-channel RMSE {e}")
    assert d.max() <= 3e-6 and (e <= 5e-7).all(), (d.max(), e)
    assert np.abs(final_s - final_x).max() <= 3e-6
    ora = oracle.OracleRenderer(W, H, sky, tex, **KW)
    ref, _ = ora.bloom(disk.transpose(1, 0, 2))
    ref = ref.transpose(1, 0, 2)
    for name, got in (("f16 x 2", blur_s), ("exact f32", blur_x)):
        dd = np.abs(got - ref)
        ee = np.sqrt(np.mean(dd.astype(np.float64) ** 2, axis=(0, 1)))
        print(f"[bloom split] {W}x{H} blur layer, {name} against the oracle: max {dd.max():.3g}, per-channel RMSE {ee}")
        assert dd.max() <= 5e-6 and (ee <= 1e-6).all(), (name, dd.max(), ee)


def test_8k_split_bloom_properties(hip_lib):
    """radius 153, eight stacked tiles per wave: a constant layer is a fixed point (edges included), the operator commutes
    with the left-right flip up to rounding, outputs are convex combinations of inputs, a faint floor keeps its relative
    accuracy (every half stays a normal f16 down to 4e-9)"""
    from bhr_amd import HipRenderer, _lib
    W, H = 7680, 4320
    r = HipRenderer(W, H, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32), math="fast", frame_slots=1,
                    **dict(KW, step_size=0.5))
    r.render_async(CAM, FOV, skip_bloom=True)
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
    np.testing.assert_allclose(bloom(np.ascontiguousarray(x[:, ::-1])), bx[:, ::-1], rtol=2e-5, atol=3e-6)
    assert bx.max() <= x.max() + 3e-6 and bx.min() >= -1e-9
    # a frame of faint values only: 1e-6 .. 1e-4 (far below the 6e-5 where an unscaled f16 would go subnormal)
    faint = (rng.random((H, W, 3), dtype=np.float32) * 1e-4 + 1e-6).astype(np.float32)
    r.set_option("bloom_split", 0)
    exact = bloom(faint)
    r.set_option("bloom_split", 1)
    got = bloom(faint)
    r.close()
    rel = np.abs(got - exact) / exact
    assert rel.max() <= 2e-5, rel.max()


def test_split_bloom_row_blocks_equal_one_context_bit_for_bit(hip_lib):
    """4k fast frame in 5 uneven row blocks (cuts not multiples of 4, 16 or 32) == one context, every bit of the gathered
    frame in every gather mode: an output's operands meet the same MFMA slots whatever the tiling; the halo rows arrive by
    the neighbours' H passes, the frame rows by the V passes' epilogues"""
    from bhr_amd import HipRenderer, multigpu
    W, H = 3840, 2160
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    full = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **KW)
    ref = full.render(CAM, FOV)
    ref_u8 = full.read_final_u8()
    full.close()
    cuts = [0, 401, 918, 1247, 1795, H]
    tiles = [HipRenderer(W, H, sky, tex, rows=(cuts[k], cuts[k + 1]), math="fast", frame_slots=1, **KW) for k in range(5)]
    for sched in ("serial", "pipelined"):
        multigpu.group_render(tiles, CAM, FOV, gather="peer", schedule=sched)
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), ref)
        multigpu.group_render(tiles, CAM, FOV, gather="peer_u8", schedule=sched)
        np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), ref_u8)
        np.testing.assert_array_equal(multigpu.group_render(tiles, CAM, FOV, gather="host", schedule=sched), ref)
    # after a direct gather a tile's own buffers hold nothing of the frame: reading one produces it on demand
    multigpu.group_render(tiles, CAM, FOV, gather="peer_u8")
    from bhr_amd import _lib
    np.testing.assert_array_equal(tiles[2].read_layer(_lib.LAYER_FINAL), ref[cuts[2]:cuts[3]])
    for t in tiles:
        t.close()


def test_strict_frames_keep_the_exact_post_pass_and_the_switch_forces_either(hip_lib):
    """strict frames run the exact f32 kernels (bit-identical to round 3's), fast frames the split ones; the option forces
    either for any arithmetic"""
    from bhr_amd import HipRenderer, _lib
    s = scenes.SCENES["default"]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    for (w, h) in ((s["width"], s["height"]), (203, 121)):           # any width, any height
        a = HipRenderer(w, h, sky, tex, math="strict", frame_slots=1, **s["kw"])
        a.render_async([6, 0, 0.5], 90)
        disk, bg, blur_strict = a.read_layer(_lib.LAYER_DISK), a.read_layer(_lib.LAYER_BG), a.read_layer(_lib.LAYER_BLUR)
        (blur_s, _), (blur_x, _) = _both_blooms(a, disk, bg)
        np.testing.assert_array_equal(blur_strict, blur_x)
        assert not np.array_equal(blur_s, blur_x) and np.abs(blur_s - blur_x).max() <= 3e-6
        a.set_option("bloom_split", 1)
        a.render_async([6, 0, 0.5], 90)
        np.testing.assert_array_equal(a.read_layer(_lib.LAYER_BLUR), blur_s)      # a strict march feeding the split post-pass
        a.close()


def test_frames_store_what_was_asked_for_and_the_rest_on_demand(hip_lib):
    """outputs="u8": the V pass writes 3 bytes per pixel and nothing else; the f32 frame and the blur layer are produced by the
    call that reads them -- the same bits a context that stores everything holds.  With the lens flare the u8 rows follow the
    flared f32 frame."""
    from bhr_amd import HipRenderer, _lib
    W, H = 1280, 720
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(128, 512)
    for math in ("fast", "strict"):
        for flare in (False, True):
            kw = dict(KW, disk_tilt=20.0, lens_flare=flare)
            everything = HipRenderer(W, H, sky, tex, math=math, frame_slots=1, outputs="f32+blur+u8", **kw)
            everything.render_async(CAM, FOV)
            want = (everything.read_layer(_lib.LAYER_FINAL), everything.read_layer(_lib.LAYER_BLUR), everything.read_final_u8())
            everything.close()
            np.testing.assert_array_equal(want[2], (np.clip(want[0], 0, 1) * 255).astype(np.uint8))
            for outputs in ("u8", "f32", "blur"):
                lean = HipRenderer(W, H, sky, tex, math=math, frame_slots=2, outputs=outputs, **kw)
                for order in ((2, 0, 1), (1, 0, 2), (0, 2, 1)):
                    lean.render_async(CAM, FOV)
                    for what in order:
                        got = lean.read_final_u8() if what == 2 else lean.read_layer(_lib.LAYER_FINAL if what == 0 else _lib.LAYER_BLUR)
                        np.testing.assert_array_equal(got, want[what], err_msg=f"{math} flare={flare} outputs={outputs} order={order} what={what}")
                lean.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_row_blocks_and_sizes_split_bloom(seed, hip_lib):
    """Small frames through the split post-pass (radius 4 ... 25), cut into 2-6 random row blocks, some thinner than the
    radius (multi-hop halo), any width and height: the gathered frame of both schedules == one context bit for bit, and the
    one-context frame sits within 3e-6 of the exact f32 kernels."""
    from bhr_amd import HipRenderer, multigpu
    rng = np.random.default_rng(100 + seed)
    W, H = [(640, 400), (320, 208), (960, 540), (1280, 720), (403, 205), (1001, 567)][seed % 6]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    kw = dict(KW, disk_tilt=float(rng.uniform(-30, 30)))
    cam = [float(rng.uniform(4, 9)), float(rng.uniform(-2, 2)), float(rng.uniform(-1.5, 1.5))]
    full = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **kw)
    ref = full.render(cam, FOV)
    full.set_option("bloom_split", 0)
    exact = full.render(cam, FOV)
    full.close()
    assert not np.array_equal(ref, exact) and np.abs(ref - exact).max() <= 3e-6
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
