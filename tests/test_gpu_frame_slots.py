"""Two frames in flight per context (bhr_frame_slot, csrc/api.hip): successive bhr_render calls alternate between
two slots / streams that share the scene.  Overlap must never change a pixel: every frame of a two-slot context is
bit-identical to the same frame from a one-slot context, with scene updates, read-backs and stand-alone passes in
between; the per-frame counters stay exact."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu

CAMS = [([6, 0, 0.5], 90), ([5, 2, 1.0], 80), ([-7, 1, 0.3], 70), ([3.2, 0.5, 0.12], 100), ([0, 0, 8], 70)]


def _mk(slots, w=256, h=144, **kw):
    from bhr_amd import HipRenderer
    return HipRenderer(w, h, scenes.analytic_skybox(), scenes.noisy_disk(), frame_slots=slots, **kw)


def test_frames_in_flight_are_bit_identical_to_one_at_a_time(hip_lib):
    from bhr_amd import _lib
    one, two = _mk(1), _mk(2)
    assert (one.frame_slots, two.frame_slots) == (1, 2)
    want = [one.render(c, f) for c, f in CAMS]
    # queue all frames without reading anything back, then read the last one; then one by one
    for c, f in CAMS:
        two.render_async(c, f)
    np.testing.assert_array_equal(two.read_layer(_lib.LAYER_FINAL), want[-1])
    for (c, f), w in zip(CAMS, want):
        np.testing.assert_array_equal(two.render(c, f), w)
        for layer in (_lib.LAYER_BG, _lib.LAYER_DISK, _lib.LAYER_BLUR):
            one.render_async(c, f)
            np.testing.assert_array_equal(two.read_layer(layer), one.read_layer(layer))
    one.close()
    two.close()


def test_scene_updates_between_frames_in_flight(hip_lib):
    """A texture upload right after a render must not reach the frame still in flight, and must reach the next."""
    one, two = _mk(1), _mk(2)
    texs = [scenes.noisy_disk(seed=s) for s in (7, 8, 9, 10)]
    want = []
    for t in texs:
        one.update_disk_texture(t)
        want.append(one.render(CAMS[0][0], CAMS[0][1]))
    got = []
    for k, t in enumerate(texs):
        two.update_disk_texture(t)
        two.render_async(CAMS[0][0], CAMS[0][1])
        if k % 2:                                   # read some frames only after the next upload has been queued
            got.append(None)
        else:
            got.append(two.render(CAMS[0][0], CAMS[0][1]))
    for k, (g, w) in enumerate(zip(got, want)):
        if g is not None:
            np.testing.assert_array_equal(g, w, err_msg=f"frame {k}")
    assert np.abs(want[0] - want[1]).max() > 1e-3   # the textures really differ
    # final state: the last texture, whatever was in flight before
    np.testing.assert_array_equal(two.render(CAMS[0][0], CAMS[0][1]), want[-1])
    one.close()
    two.close()


def test_standalone_passes_and_flare_with_two_slots(hip_lib):
    from bhr_amd import _lib
    one, two = _mk(1, lens_flare=True), _mk(2, lens_flare=True)
    for c, f in CAMS[:3]:                      # the flare's scratch buffers are shared by the slots
        two.render_async(c, f)
    for c, f in CAMS[:3]:
        w = one.render(c, f)
    np.testing.assert_array_equal(two.read_layer(_lib.LAYER_FINAL), w)
    # write_layer + bloom_only act on the frame the reads see
    disk = two.read_layer(_lib.LAYER_DISK)
    two.write_layer(_lib.LAYER_DISK, 0.5 * disk)
    two.bloom_only()
    one.write_layer(_lib.LAYER_DISK, 0.5 * one.read_layer(_lib.LAYER_DISK))
    one.bloom_only()
    np.testing.assert_array_equal(two.read_layer(_lib.LAYER_FINAL), one.read_layer(_lib.LAYER_FINAL))
    np.testing.assert_array_equal(two.read_final_u8(), one.read_final_u8())
    one.close()
    two.close()


@pytest.mark.parametrize("slots", [1, 2])
def test_timing_ring_counts_every_frame_it_reports(slots, hip_lib):
    """the cells the bloom kernel pre-clears for the frames to come must not be reported.  After more
    frames than the ring holds, ray_steps_sum == frames_timed x ray_steps of the (identical) frame."""
    r = _mk(slots, w=96, h=64)
    cam, fov = CAMS[0]
    r.timing_reset()
    for _ in range(7):
        r.render_async(cam, fov)
    c = r.counters()
    assert c["frames_timed"] == 7 and c["ray_steps_sum"] == 7 * c["ray_steps"] and c["ray_steps"] > 0
    for _ in range(520):
        r.render_async(cam, fov)
    c = r.counters()
    assert 500 <= c["frames_timed"] <= 510
    assert c["ray_steps_sum"] == c["frames_timed"] * c["ray_steps"]
    assert c["march_ms_sum"] > 0 and c["bloom_ms_sum"] > 0
    r.close()


def test_persistent_schedule_and_row_costs_stay_exclusive(hip_lib):
    """Launches that use per-context scratch (the persistent schedule's work queue, the row-cost profile) run alone."""
    one, two = _mk(1), _mk(2)
    cam, fov = CAMS[0]
    w = one.render(cam, fov)
    for _ in range(3):
        two.render_async(cam, fov)
        two.render_async(cam, fov, compaction=True)
    np.testing.assert_array_equal(two.render(cam, fov), w)
    np.testing.assert_array_equal(two.row_costs(cam, fov), one.row_costs(cam, fov))
    one.close()
    two.close()


@pytest.mark.parametrize("slots", [1, 2])
def test_march_busy_time_is_the_union_of_the_march_intervals(slots, hip_lib):
    """bhr_counters.march_busy_ms: with one frame at a time it equals the sum of the march brackets; with two frames in
    flight it is at most that sum (shared time counted once) and at most the span of the timed frames."""
    r = _mk(slots, w=1920, h=1080)
    cam, fov = CAMS[0]
    for _ in range(5):
        r.render_async(cam, fov)
    r.timing_reset()
    for _ in range(40):
        r.render_async(cam, fov)
    c = r.counters()
    assert c["frames_timed"] == 40 and c["march_busy_ms"] > 0 and c["span_ms"] >= c["march_busy_ms"] * 0.999
    if slots == 1:
        assert abs(c["march_busy_ms"] - c["march_ms_sum"]) <= 1e-3 * c["march_ms_sum"] + 1e-3
    else:
        assert c["march_busy_ms"] <= c["march_ms_sum"] * 1.001
        assert c["march_busy_ms"] < 0.98 * c["march_ms_sum"]         # the launches really overlap
    r.close()


def test_slot_stream_calibration_is_invisible_in_the_frames(hip_lib):
    """A two-slot context times six candidate streams for slot 1 once eight two-slot frames have been asked for
    (csrc/api.hip: calibrate_slot_streams) -- ~500 extra renders of the ninth frame's view.  Nothing of it may show:
    every frame before, at and after the calibration equals the one-slot context's, the ray-step and frame counters count
    the caller's frames only, the report names the stream kept; option calibrate_streams 0 never calibrates."""
    from bhr_amd import _lib
    one, two, off = _mk(1, math="hybrid"), _mk(2, math="hybrid"), _mk(2, math="hybrid", options={"calibrate_streams": 0})
    assert not two.stream_calibration()["done"]
    two.timing_reset()
    steps = 0
    for k in range(14):
        c, f = CAMS[k % len(CAMS)]
        want = one.render(c, f)
        steps += one.counters()["ray_steps"]
        np.testing.assert_array_equal(two.render(c, f), want)
        np.testing.assert_array_equal(off.render(c, f), want)
        assert two.stream_calibration()["done"] == (k >= 8), k
    cal = two.stream_calibration()
    assert 0 <= cal["kept"] < 6 and max(cal["candidates_fps"]) > 0 and cal["candidates_fps"][cal["kept"]] > 0, cal
    assert not off.stream_calibration()["done"]
    c2 = two.counters()
    assert c2["frames_timed"] == 14 and c2["ray_steps_sum"] == steps, (c2["frames_timed"], c2["ray_steps_sum"], steps)
    m = two.stream_map()
    assert set(m) >= {"scene+slot0", "scene+slot1", "slot0+slot1"} and all(isinstance(v, bool) for v in m.values())
    for r in (one, two, off):
        r.close()
