"""Finite-thickness Disk V2 source of the march kernel (BHR_DISK_V2_VOLUME, docs/design_ad_v2.md 4.2-4.3).

The reference specifies this integrator but does not implement it (design_ad_v2.md 5.1 status table:
integrator.py "planned"), so there are no reference vectors: the kernel is checked against the oracle's
restatement of the same specification (whose Disk V2 fields ARE pinned by the reference's tables,
tests/test_oracle.py) and against the acceptance criteria the design lists for test_disk_v2_integrator.py /
test_disk_v2_advection.py: nothing off the disk, opacity grows with the path's optical depth, the grazing
gain thickens oblique views, more pieces per step converge, an axisymmetric disk does not change with time.
"""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

VIEWS = {
    "edge_on": dict(cam=[9.0, 0.0, 0.6], fov=70, tilt=0.0, frame=0),
    "tilted_later": dict(cam=[7.0, 2.0, 2.5], fov=80, tilt=12.0, frame=25),       # t_offset = 2.5: advected pattern
}
W, H = 192, 108


def _rmse(a, b):
    return np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=(0, 1)))


def _pair(oracle, view, math="strict", fast=False, **vol):
    from bhr_amd import HipRenderer, disk_v2 as dv
    P = dv.DiskV2Params()
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    kw = dict(step_size=0.1, r_disk_inner=P.r_in, r_disk_outer=P.r_out, disk_tilt=view["tilt"])
    hip = HipRenderer(W, H, sky, tex, math=math, **kw)
    hip.use_disk_v2(P, seed=42, volume=True, **vol)
    cp, m_s, m_h, t_peak = hip._dv2
    ora = oracle.OracleRenderer(W, H, sky, tex, fast=fast, **kw)
    ora.set_volume(cp, m_s, m_h, t_peak, vol.get("absorption", 4.0), vol.get("grazing_gain", 1.0), vol.get("substeps", 2))
    return hip, ora


@pytest.mark.parametrize("view", list(VIEWS))
def test_volume_matches_oracle_strict(view, oracle, hip_lib):
    from bhr_amd import _lib
    v = VIEWS[view]
    hip, ora = _pair(oracle, v, substeps=3)
    try:
        hip.render_async(v["cam"], v["fov"], frame=v["frame"], skip_bloom=True)
        bg, disk = hip.read_layer(_lib.LAYER_BG), hip.read_layer(_lib.LAYER_DISK)
        rbg, rdisk = (x.transpose(1, 0, 2) for x in ora.march(v["cam"], v["fov"], frame=v["frame"]))
    finally:
        ora.set_volume(None)
    assert disk.max() > 0.3 and (disk.sum(axis=2) > 0).mean() > 0.1          # the disk is really in view
    # identical ray paths (strict arithmetic); the model runs in binary64 on both sides, the per-sample
    # g-factor transcendentals differ by <= 2 ulp (ocml vs glibc) and accumulate over tens of samples
    for name, a, b in (("bg", bg, rbg), ("disk", disk, rdisk)):
        assert (_rmse(a, b) <= 1e-5).all(), f"{view}/{name}: RMSE {_rmse(a, b)}"
        assert np.abs(a - b).max() <= 2e-4, f"{view}/{name}: max {np.abs(a - b).max()}"
    assert hip.counters()["ray_steps"] == ora.last_total_steps
    hip.close()


def test_volume_fast_math_against_binary64(oracle, hip_lib):
    from bhr_amd import _lib
    v = VIEWS["edge_on"]
    hip, ora = _pair(oracle, v, math="fast", fast="f64")
    try:
        hip.render_async(v["cam"], v["fov"], skip_bloom=True)
        disk = hip.read_layer(_lib.LAYER_DISK)
        rdisk = ora.march(v["cam"], v["fov"])[1].transpose(1, 0, 2)
    finally:
        ora.set_volume(None)
    assert (_rmse(disk, rdisk) <= 1e-4).all(), _rmse(disk, rdisk)
    hip.close()


def _render(view, **vol):
    from bhr_amd import HipRenderer, _lib, disk_v2 as dv
    P = vol.pop("params", None) or dv.DiskV2Params()
    sp = vol.pop("structure", None)
    frame = vol.pop("frame", view["frame"])
    hip = HipRenderer(W, H, scenes.analytic_skybox(), scenes.noisy_disk(), step_size=0.1, r_disk_inner=P.r_in,
                      r_disk_outer=P.r_out, disk_tilt=view["tilt"])
    hip.use_disk_v2(P, sp, seed=42, volume=True, **vol)
    hip.render_async(view["cam"], view["fov"], frame=frame, skip_bloom=True)
    out = hip.read_layer(_lib.LAYER_BG), hip.read_layer(_lib.LAYER_DISK)
    hip.close()
    return out


def test_integrator_acceptance_criteria(hip_lib):
    from bhr_amd import HipRenderer, _lib
    v = VIEWS["edge_on"]
    # no absorption, no emission: the disk layer is exactly empty and the sky is untouched
    bg0, disk0 = _render(v, absorption=0.0)
    assert disk0.max() == 0.0
    plain = HipRenderer(W, H, scenes.analytic_skybox(), np.zeros((32, 64, 4), np.float32), step_size=0.1,
                        r_disk_inner=2.0, r_disk_outer=10.0)
    plain.render_async(v["cam"], v["fov"], skip_bloom=True)
    np.testing.assert_array_equal(bg0, plain.read_layer(_lib.LAYER_BG))
    plain.close()
    # the longer the optical path, the less sky comes through: transmission falls monotonically with Ca
    sky_through = [(_render(v, absorption=ca)[0]).sum() for ca in (0.5, 2.0, 8.0)]
    assert sky_through[0] > sky_through[1] > sky_through[2] and sky_through[0] < bg0.sum()
    # grazing-angle gain: an edge-on view gets more opaque, and the gain only ever adds opacity
    flat, gained = _render(v, grazing_gain=0.0)[0], _render(v, grazing_gain=2.0)[0]
    assert gained.sum() < 0.98 * flat.sum() and (gained <= flat + 2e-4).all()      # 2e-4: rays stop at opacity 0.9999
    # more pieces per step converge
    ref = _render(v, substeps=16)[1]
    err = [np.abs(_render(v, substeps=n)[1] - ref).mean() for n in (1, 4, 8)]
    assert err[0] > err[1] > err[2] and err[2] < 0.25 * err[0]


def test_advection_needs_structure(hip_lib):
    """phi_adv = phi + t Omega(r): an axisymmetric disk (no modes, shear or hotspots) looks the same at every
    time, the structured one does not."""
    from bhr_amd import disk_v2 as dv
    v = VIEWS["tilted_later"]
    flat = dv.DiskV2StructureParams(mode1_strength=0.0, mode2_strength=0.0, shear_strength=0.0, hotspot_strength=0.0)
    a, b = _render(v, structure=flat, frame=0)[1], _render(v, structure=flat, frame=40)[1]
    assert np.abs(a - b).max() <= 1e-6
    a, b = _render(v, frame=0)[1], _render(v, frame=40)[1]
    assert np.abs(a - b).mean() > 1e-3


def test_volume_option_validation(hip_lib):
    from bhr_amd import HipRenderer, disk_v2 as dv
    r = HipRenderer(32, 18, scenes.analytic_skybox(), scenes.noisy_disk())
    for bad in (dict(absorption=-1.0), dict(grazing_gain=-0.5), dict(substeps=0), dict(substeps=17)):
        with pytest.raises(ValueError):
            r.use_disk_v2(dv.DiskV2Params(), volume=True, **bad)
    r.use_disk_v2(dv.DiskV2Params(), volume=True)
    r.use_disk_v2(None)                       # back to the texture
    r.close()
