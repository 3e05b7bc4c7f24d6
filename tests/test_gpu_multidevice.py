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
    tiles[0].set_option("group_threads", 0)          # the group's switches are the first tile's (BHR_GROUP_THREADS at bhr_create)
    want = multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host")
    tiles[0].set_option("group_threads", 1)
    for _ in range(5):
        np.testing.assert_array_equal(multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host"), want)
        multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer")
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), want)
    for t in tiles + [full]:
        t.close()


@pytest.mark.parametrize("flare", [False, True])
def test_pipelined_schedule_equals_serial_bit_for_bit(flare, hip_lib):
    """The pipelined row-block schedule (halo bands marched first, halo pulls under the march of the middle rows, V pass
    and pushes in row chunks, csrc/group.hip) against the step-after-step one: same bytes, in every gather mode, with a
    tile thinner than the bloom radius (two-hop halo) and with the lens flare."""
    from bhr_amd import multigpu
    s, tiles, full = _tiles([0, 50, 54, 120, 180], [0, 0, 0, 0], lens_flare=flare)
    kw = dict(lens_flare=flare)
    want = multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host", schedule="serial", **kw)
    for _ in range(3):
        got = multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host", schedule="pipelined", **kw)
        np.testing.assert_array_equal(got, want)
    multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer", schedule="pipelined", **kw)
    np.testing.assert_array_equal(multigpu.read_gathered(tiles), want)
    want_u8 = (np.clip(want, 0, 1) * np.float32(255)).astype(np.uint8)      # save_image's truncation (render.py:423)
    for sched in ("pipelined", "serial"):
        multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer_u8", schedule=sched, **kw)
        np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), want_u8)
    np.testing.assert_allclose(want, full.render(s["cam_pos"], s["fov"]), atol=1e-6, rtol=0)
    for t in tiles + [full]:
        t.close()


def test_subset_render_leaves_the_frame_unchanged(hip_lib):
    """bhr_group_render_subset: one tile re-rendered alone (its neighbours resting) reproduces its rows of the frame."""
    from bhr_amd import multigpu
    s, tiles, full = _tiles([0, 48, 96, 136, 180], [0, 0, 0, 0])
    multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer")
    want = multigpu.read_gathered(tiles)
    multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer_u8")
    want_u8 = multigpu.read_gathered_u8(tiles)
    for k in range(4):
        live = [int(q == k) for q in range(4)]
        for sched in ("pipelined", "serial"):
            multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer", schedule=sched, live=live)
            np.testing.assert_array_equal(multigpu.read_gathered(tiles), want)
            multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="peer_u8", schedule=sched, live=live)
            np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), want_u8)
        assert tiles[k].counters()["frame_ms"] > 0
    with pytest.raises(ValueError, match="every tile live"):
        multigpu.group_render(tiles, s["cam_pos"], s["fov"], gather="host", live=[1, 0, 0, 0])
    for t in tiles + [full]:
        t.close()


def test_8k_frame_in_eight_blocks_equals_one_context(hip_lib):
    """BASELINE.json configs[3] at full size: the 7680x4320 step-0.05 frame cut into 8 cost-balanced row blocks (all on
    device 0 here), pipelined schedule, against the same frame from one context: every pixel of the gathered f32 frame to
    1e-6 (the bloom sums its taps tile by tile), the quantised gather equal to the truncation of the f32 gather, ray-step
    totals equal."""
    from bhr_amd import HipRenderer, multigpu
    W, H = 7680, 4320
    sky, tex = scenes.analytic_skybox(256, 512), scenes.noisy_disk(256, 1024)
    kw = dict(step_size=0.05, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
    cam, fov = [6.0, 0.0, 0.5], 90.0
    per_row, band_rows = multigpu.probe_row_costs(W, H, cam, fov, **kw)
    blocks = multigpu.balanced_row_blocks(H, 8, per_row, band_rows, fixed_cost_per_row=0.1 * float(per_row.mean()))
    full = HipRenderer(W, H, sky, tex, frame_slots=1, **kw)
    ref = full.render(cam, fov)
    ref_steps = full.counters()["ray_steps"]
    full.close()
    tiles = [HipRenderer(W, H, sky, tex, rows=b, frame_slots=1, **kw) for b in blocks]
    multigpu.group_render(tiles, cam, fov, gather="peer")
    got = multigpu.read_gathered(tiles)
    assert sum(t.counters()["ray_steps"] for t in tiles) == ref_steps
    d = np.abs(got - ref)
    assert d.max() <= 1e-6, d.max()
    multigpu.group_render(tiles, cam, fov, gather="peer_u8")
    np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), (np.clip(got, 0, 1) * np.float32(255)).astype(np.uint8))
    for t in tiles:
        t.close()


def test_bench_row_block_leg_in_a_child_process(hip_lib):
    """bench.py at N > 1 runs its auxiliary row-block leg (one process driving every device) in a child process, so that
    a fault in code no multi-GPU node has run yet cannot take the headline with it: the child's dict comes back whole,
    and a child that dies is reported, not raised."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    t = bench.tile_leg_in_child("fhd", 1, 4, "hybrid")
    assert t.get("n_gpus") == 1 and t["frames"] == 4 and t["value"] > 0 and len(t["row_blocks"]) == 1, t
    bad = bench.tile_leg_in_child("fhd", 3, 4, "hybrid")        # three devices on a one-GPU box: skipped, not an error
    assert "skipped" in bad or "error" in bad, bad


@pytest.mark.parametrize("math", ["hybrid", "fast"])
def test_group_frames_in_flight_equal_frames_waited_for(math, hip_lib):
    """BHR_GROUP_ASYNC (group_render(..., wait=False)): frames whose rows the kernels store themselves are submitted back to
    back, nothing on the host waits between them; a tile's H pass waits on the device for its neighbours' previous V passes
    before it stores into their halo rows.  Views alternate so that a halo row written too early, or a frame row of the wrong
    frame, would show: after every burst the gathered frame equals the waited-for frame of the burst's LAST view, and one
    context's; read_gathered_u8 alone (no group_sync) sees the whole last frame."""
    from bhr_amd import HipRenderer, multigpu
    W, H = 1920, 1080
    sky, tex = scenes.analytic_skybox(256, 512), scenes.noisy_disk(256, 1024)
    kw = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)
    views = [([6.0, 0.0, 0.5], 90.0), ([5.0, 2.0, 1.0], 80.0), ([-7.0, 1.0, 0.3], 70.0)]
    blocks = [(0, 312), (312, 544), (544, 800), (800, 1080)]
    tiles = [HipRenderer(W, H, sky, tex, rows=b, frame_slots=1, math=math, **kw) for b in blocks]
    want = []
    for cam, fov in views:
        multigpu.group_render(tiles, cam, fov, gather="peer_u8")
        want.append(multigpu.read_gathered_u8(tiles).copy())
    assert not np.array_equal(want[0], want[1])
    full = HipRenderer(W, H, sky, tex, frame_slots=1, math=math, **kw)
    full.render_async(*views[2])
    np.testing.assert_array_equal(want[2], full.read_final_u8())
    full.close()
    for burst in range(3):
        for k in range(7):
            cam, fov = views[(k + burst) % 3]
            multigpu.group_render(tiles, cam, fov, gather="peer_u8", wait=False)
        last = (6 + burst) % 3
        if burst == 1:
            multigpu.group_sync(tiles)
        np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), want[last])
    # a waited-for frame behind frames in flight
    multigpu.group_render(tiles, *views[0], gather="peer_u8", wait=False)
    multigpu.group_render(tiles, *views[1], gather="peer_u8")
    np.testing.assert_array_equal(multigpu.read_gathered_u8(tiles), want[1])
    for t in tiles:
        t.close()
