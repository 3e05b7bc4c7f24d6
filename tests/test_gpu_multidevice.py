"""Row-block tiles across devices (BASELINE.json configs[3]): bhr_group_render with the xGMI peer gather.

On the one-GPU test box the device-spanning test skips itself and the same code path is exercised with several
contexts on device 0 (same kernels, same hipMemcpyPeerAsync calls, source device == destination device); on a
multi-GPU node `test_group_render_across_two_devices` runs the halo exchange and the gather over a real link."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu


def _tiles(cuts, devices, lens_flare=False):
    from bhr_amd import HipRenderer
    s = scenes.SCENES["default"]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    tiles = [HipRenderer(s["width"], s["height"], sky, tex, rows=(cuts[k], cuts[k + 1]), device_index=devices[k],
                         **s["kw"]) for k in range(len(cuts) - 1)]
    full = HipRenderer(s["width"], s["height"], sky, tex, lens_flare=lens_flare, **s["kw"])
    return s, tiles, full


@pytest.mark.parametrize("flare", [False, True])
def test_peer_gather_equals_host_gather_equals_one_context(flare, hip_lib):
    from bhr_amd import multigpu
    s, tiles, full = _tiles([0, 50, 54, 120, 180], [0, 0, 0, 0], lens_flare=flare)
    ref = full.render(s["cam_pos"], s["fov"])
    host = multigpu.group_render(tiles, s["cam_pos"], s["fov"], lens_flare=flare, gather="host")
    assert multigpu.group_render(tiles, s["cam_pos"], s["fov"], lens_flare=flare, gather="peer") is None
    peer = multigpu.read_gathered(tiles)
    np.testing.assert_array_equal(peer, host)
    np.testing.assert_allclose(peer, ref, atol=1e-6, rtol=0)
    # the ray-step counters of the tiles add up to the frame's
    assert sum(t.counters()["ray_steps"] for t in tiles) == full.counters()["ray_steps"]
    for t in tiles + [full]:
        t.close()


def test_read_gathered_needs_a_gather(hip_lib):
    from bhr_amd import multigpu
    s, tiles, full = _tiles([0, 96, 180], [0, 0])
    with pytest.raises(AssertionError, match="BHR_GATHER_PEER"):   # call-order violation, as the reference asserts
        multigpu.read_gathered(tiles)
    for t in tiles + [full]:
        t.close()


def test_group_render_across_two_devices(hip_lib):
    """Halo exchange + gather between DISTINCT devices (skips where only one is visible)."""
    from bhr_amd import multigpu
    if int(hip_lib.bhr_device_count()) < 2:
        pytest.skip("needs two HIP devices")
    s, tiles, full = _tiles([0, 88, 180], [0, 1])
    ref = full.render(s["cam_pos"], s["fov"])
    multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer")
    np.testing.assert_allclose(multigpu.read_gathered(tiles), ref, atol=1e-6, rtol=0)
    np.testing.assert_allclose(multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host"), ref, atol=1e-6, rtol=0)
    for t in tiles + [full]:
        t.close()


def test_make_tiles_cost_balanced_blocks(hip_lib):
    """bench.py's row-block leg at a small size: blocks cover the frame, cuts fall on tile rows, the gathered frame
    equals the one-context frame of the same scene."""
    from bhr_amd import multigpu, workloads
    wl = dict(width=640, height=360, cam_pos=[6.0, 0.0, 0.5], fov=90.0, step_size=0.1, anti_alias="disabled", disk_tilt=0.0)
    tiles, blocks, _ = workloads.make_tiles(wl, [0, 0, 0], n_stars=200)
    assert blocks[0][0] == 0 and blocks[-1][1] == 360 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    assert all(b[0] % 8 == 0 for b in blocks)
    multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer")
    got = multigpu.read_gathered(tiles)
    steps = np.array([t.counters()["ray_steps"] for t in tiles], dtype=np.float64)
    assert steps.max() / steps.mean() <= 1.10, steps            # the cut is by cost: no tile marches 10 % more than the mean
    one, _, _, _ = workloads.make_scene(wl, n_stars=200)
    np.testing.assert_allclose(got, one.render(wl["cam_pos"], wl["fov"]), atol=1e-6, rtol=0)
    for t in tiles + [one]:
        t.close()


def test_threaded_submission_gives_the_same_frame(hip_lib, monkeypatch):
    """With distinct devices bhr_group_render submits each tile's march from its own host thread; forced here on one
    device (BHR_GROUP_THREADS=1) so that the path runs on the test box: the frame must not change."""
    from bhr_amd import multigpu
    s, tiles, full = _tiles([0, 50, 54, 120, 180], [0, 0, 0, 0])
    monkeypatch.setenv("BHR_GROUP_THREADS", "0")
    want = multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host")
    monkeypatch.setenv("BHR_GROUP_THREADS", "1")
    for _ in range(5):
        np.testing.assert_array_equal(multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host"), want)
        multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer")
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), want)
    for t in tiles + [full]:
        t.close()
