"""Device lens flare (csrc/flare.hip) against the reference's own output (tests/golden/misc.npz, made
from TaichiRenderer._apply_lens_flare) and against the NumPy restatement on rendered frames."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# The frame sums are accumulated in NumPy's own order (bit-identical, test_flare_sums_are_numpys); what
# is left is the last f64 bit of atan2 / exp between the device's and the host's libm, which can flip
# the f32 rounding of a single accumulate.
TOL = 6e-8


def _blank(w, h):
    from bhr_amd import HipRenderer
    return HipRenderer(w, h, np.zeros((16, 32, 3), np.float32), np.zeros((32, 64, 4), np.float32))


def test_flare_golden_vectors():
    from bhr_amd import _lib
    d = np.load(os.path.join(GOLD, "misc.npz"))
    final = np.ascontiguousarray(d["flare_final"].transpose(1, 0, 2))     # reference arrays are (W, H, 3)
    disk = np.ascontiguousarray(d["flare_disk"].transpose(1, 0, 2))
    h, w = final.shape[:2]
    r = _blank(w, h)
    r.write_layer(_lib.LAYER_FINAL, final)
    r.write_layer(_lib.LAYER_DISK, disk)
    r.apply_lens_flare()
    got = r.read_layer(_lib.LAYER_FINAL)
    want = d["flare_out"].transpose(1, 0, 2)
    assert np.abs(want - final).max() > 0.01
    assert np.abs(got - want).max() <= TOL
    # no disk light: the frame is returned untouched (render.py:3934-3935)
    r.write_layer(_lib.LAYER_FINAL, final)
    r.write_layer(_lib.LAYER_DISK, np.zeros_like(disk))
    r.apply_lens_flare()
    np.testing.assert_array_equal(r.read_layer(_lib.LAYER_FINAL), d["flare_dark_out"].transpose(1, 0, 2))
    r.close()


@pytest.mark.parametrize("shape", [(36, 64), (90, 160), (144, 256), (128, 64), (131, 517), (7, 5), (1080, 1920)])
def test_flare_sums_are_numpys(shape):
    """np.sum's chunked pairwise order reproduced on the device: equality, not closeness."""
    from bhr_amd import _lib
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    disk = (rng.random((h, w, 3), dtype=np.float32) ** 4).astype(np.float32)
    r = _blank(w, h)
    r.write_layer(_lib.LAYER_DISK, disk)
    got = r.lens_flare_sums()
    glow = np.max(np.ascontiguousarray(disk.transpose(1, 0, 2)), axis=2)       # the reference's (W, H) layout
    xs, ys = np.mgrid[0:w, 0:h]
    want = np.array([np.sum(glow), np.sum(xs * glow), np.sum(ys * glow)], dtype=np.float64)
    np.testing.assert_array_equal(got, want)
    r.close()


@pytest.mark.parametrize("level", [0.02, 0.9])     # 0.9: brightness ratio > 1, Python's min() picks the float 1.0
def test_flare_synthetic_layers(level):
    from bhr_amd import _lib
    from oracle.flare_np import apply_lens_flare
    rng = np.random.default_rng(5)
    h, w = 90, 160
    final = rng.random((h, w, 3), dtype=np.float32) * 0.3
    yy, xx = np.mgrid[0:h, 0:w]
    blob = np.exp(-(((xx - 110) / 30.0) ** 2 + ((yy - 30) / 12.0) ** 2)).astype(np.float32)
    disk = (np.maximum(blob, level)[..., None] * np.array([1.0, 0.8, 0.5], np.float32)).astype(np.float32)
    want = apply_lens_flare(final, disk)
    r = _blank(w, h)
    r.write_layer(_lib.LAYER_FINAL, final)
    r.write_layer(_lib.LAYER_DISK, disk)
    r.apply_lens_flare()
    got = r.read_layer(_lib.LAYER_FINAL)
    assert np.abs(want - final).max() > 0.01
    assert np.abs(got - want).max() <= TOL
    r.close()


def test_flare_in_render_and_row_blocks():
    """render(lens_flare=True) and the row-block path (partial sums folded across tiles) give the
    frame the NumPy effect gives on the same layers."""
    from bhr_amd import HipRenderer, _lib, scenes
    from bhr_amd.multigpu import group_render, row_blocks
    from oracle.flare_np import apply_lens_flare
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    kw = dict(step_size=0.1, disk_tilt=25.0, anti_alias="lod_radius")
    cam, fov, (w, h) = [6, 0, 0.5], 90, (256, 144)
    r = HipRenderer(w, h, sky, tex, **kw)
    r.render_async(cam, fov)
    base, disk = r.read_layer(_lib.LAYER_FINAL), r.read_layer(_lib.LAYER_DISK)
    want = apply_lens_flare(base, disk)
    assert np.abs(want - base).max() > 0.01
    r.lens_flare = True
    got = r.render(cam, fov)
    assert np.abs(got - want).max() <= TOL
    np.testing.assert_array_equal(r.read_final_u8(), (got * 255).astype(np.uint8))
    r.close()
    tiles = [HipRenderer(w, h, sky, tex, rows=rows, **kw) for rows in row_blocks(h, 3)]
    got = group_render(tiles, cam, fov, lens_flare=True)
    assert np.abs(got - want).max() <= TOL
    tiles[0].lens_flare = True
    with pytest.raises(ValueError):
        tiles[0].render(cam, fov)
    for t in tiles:
        t.close()
