"""BASELINE.json's north star names "mipmap levels staged through LDS": the opt-in BHR_MIP_LDS=1 variant of the fast
anti-aliased march (csrc/march.hip: march_tile_mipstaged_kernel) copies the coarse levels of the packed mip stack into
LDS per block and samples them from there.  Same texels, same arithmetic: the frames must be the bits of the plain kernel;
the launcher falls back where no level fits."""
import time

import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=25.0, anti_alias="lod_radius", aa_strength=2.0)


@pytest.mark.parametrize("tex_hw,want_from", [((128, 256), 2), ((128, 512), 3), ((256, 1024), -1)])
def test_lds_staged_mip_levels_give_the_same_bits(tex_hw, want_from, hip_lib, monkeypatch):
    from bhr_amd import HipRenderer, _lib
    W, H = 640, 360
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk(*tex_hw)
    cam, fov = [9.0, 1.0, 1.2], 70.0
    frames, ms = {}, {}
    for on in ("0", "1"):
        monkeypatch.setenv("BHR_MIP_LDS", on)
        r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, **KW)
        r.render_async(cam, fov)
        frames[on] = (r.read_layer(_lib.LAYER_DISK), r.read_layer(_lib.LAYER_BG), r.counters()["ray_steps"])
        level = r.mip_lds_level()
        assert level == (want_from if on == "1" else -1), (on, level)
        r.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            r.render_async(cam, fov, skip_bloom=True)
        r.sync()
        ms[on] = (time.perf_counter() - t0) / 20 * 1e3
        r.close()
    assert frames["0"][0].max() > 0.2                                   # the disk is in view ...
    np.testing.assert_array_equal(frames["1"][0], frames["0"][0])
    np.testing.assert_array_equal(frames["1"][1], frames["0"][1])
    assert frames["1"][2] == frames["0"][2]
    print(f"\n[mip lds] texture {tex_hw}: levels from {want_from} in LDS {ms['1']:.3f} ms, plain {ms['0']:.3f} ms per {W}x{H} frame")


def test_lds_staging_is_for_the_fast_anti_aliased_march_only(hip_lib, monkeypatch):
    """strict and hybrid frames, and frames without anti-aliasing, ignore the switch"""
    from bhr_amd import HipRenderer
    monkeypatch.setenv("BHR_MIP_LDS", "1")
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk(128, 256)
    for math, aa in (("strict", "lod_radius"), ("hybrid", "lod_radius"), ("fast", "disabled")):
        r = HipRenderer(320, 200, sky, tex, math=math, **dict(KW, anti_alias=aa))
        r.render_async([9.0, 1.0, 1.2], 70.0)
        assert r.mip_lds_level() == -1, (math, aa)
        r.close()
