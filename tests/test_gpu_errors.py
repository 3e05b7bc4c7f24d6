"""Error behaviour of the C ABI and of the Python mirror: the exceptions the reference raises in the same
situations (ValueError for bad arguments, AssertionError for call-order / size violations), and a
status code + message -- never a crash -- for everything a caller can get wrong at the C level."""
import ctypes as C

import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from bhr_amd import _lib
    base = dict(width=64, height=36, row0=0, row1=36, step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0,
                disk_tilt_deg=0.0, anti_alias=0, aa_strength=1.0, disk_rotation_speed=0.1, device=0, math_mode=1)
    base.update(kw)
    return _lib.Config(**base)


@pytest.mark.parametrize("bad", [dict(width=0), dict(height=-3), dict(row0=10, row1=5), dict(row1=99), dict(step_size=0.0),
                                 dict(step_size=-0.1), dict(r_disk_inner=5.0, r_disk_outer=5.0), dict(device=99),
                                 dict(math_mode=7)])
def test_create_rejects_bad_configs(bad, hip_lib):
    from bhr_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.bhr_create(C.byref(_cfg(**bad)), C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert len(lib.bhr_last_error()) > 10


def test_null_and_range_arguments(hip_lib):
    from bhr_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.bhr_create(C.byref(_cfg()), C.byref(ctx)) == 0
    null_f = C.POINTER(C.c_float)()
    cam = _lib.Camera()
    assert lib.bhr_render(None, C.byref(cam), 0) == _lib.BHR_ERR_INVALID
    assert lib.bhr_render(ctx, None, 0) == _lib.BHR_ERR_INVALID
    assert lib.bhr_set_skybox(ctx, null_f, 16, 32) == _lib.BHR_ERR_INVALID
    assert lib.bhr_set_skybox(ctx, (C.c_float * 12)(), 0, 4) == _lib.BHR_ERR_INVALID
    assert lib.bhr_read_layer(ctx, 17, (C.c_float * (64 * 36 * 3))()) == _lib.BHR_ERR_INVALID
    assert lib.bhr_read_layer(ctx, 0, null_f) == _lib.BHR_ERR_INVALID
    assert lib.bhr_write_layer(ctx, -1, (C.c_float * (64 * 36 * 3))()) == _lib.BHR_ERR_INVALID
    assert lib.bhr_get_disk_mip(ctx, 9, (C.c_float * 4)()) != 0
    assert lib.bhr_get_counters(ctx, None) == _lib.BHR_ERR_INVALID
    assert lib.bhr_group_render(None, 2, C.byref(cam), 0, null_f) == _lib.BHR_ERR_INVALID
    # rendering before any texture was set is a call-order violation, not a fault
    rc = lib.bhr_render(ctx, C.byref(cam), 0)
    assert rc in (_lib.BHR_ERR_STATE, _lib.BHR_ERR_INVALID) and b"" != lib.bhr_last_error()
    lib.bhr_destroy(ctx)
    lib.bhr_destroy(None)                     # destroying nothing is allowed


def test_python_mirror_raises_what_the_reference_raises(hip_lib):
    from bhr_amd import HipRenderer
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    r = HipRenderer(64, 36, sky, tex)
    with pytest.raises(AssertionError):       # render.py:2299: texture size mismatch
        r.update_disk_texture(np.zeros((tex.shape[0] // 2, tex.shape[1], 4), np.float32))
    with pytest.raises(AssertionError):       # render.py:3558: generate_background before init_background_layer
        r.generate_background(t=0.0)
    with pytest.raises(AssertionError):       # render.py:3802: update_disk_texture_gpu before upload_parametric_state
        r.update_disk_texture_gpu(0.0)
    with pytest.raises(ValueError):
        r.write_layer(0, np.zeros((3, 3, 3), np.float32))
    with pytest.raises(ValueError):
        HipRenderer(64, 36, sky, tex, math="quick")
    with pytest.raises(ValueError):
        HipRenderer(64, 36, sky, tex, anti_alias="msaa")
    r.close()
    r.close()                                 # idempotent


def test_group_render_rejects_gaps_and_overlaps(hip_lib):
    from bhr_amd import HipRenderer
    from bhr_amd.multigpu import group_render
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    a = HipRenderer(64, 36, sky, tex, rows=(0, 12))
    b = HipRenderer(64, 36, sky, tex, rows=(20, 36))          # rows 12..19 missing
    with pytest.raises(ValueError):
        group_render([a, b], [6, 0, 0.5], 90)
    c = HipRenderer(64, 36, sky, tex, rows=(12, 36))
    with pytest.raises(ValueError):
        group_render([c, a], [6, 0, 0.5], 90)                  # out of order
    out = group_render([a, c], [6, 0, 0.5], 90)
    assert out.shape == (36, 64, 3) and np.isfinite(out).all()
    for r in (a, b, c):
        r.close()
