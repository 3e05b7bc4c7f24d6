"""The video loop keeps two frames in flight and lets the next frame's texture passes run beside the current march
(component passes do not join the frames in flight, composition waits for the marches only, PNG work rides the frame's
own stream).  None of that may change a pixel: the files of the overlapped loop decode to the frames of a
loop that runs on ONE stream, one frame at a time, with the round-1 host pair tables and the host PNG encoder."""
import functools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(tmp, tag, n_frames, size, frame_slots, png_level, pairs_on_host, math=None):
    from bhr_amd import drivers
    w, h = size
    r, _, _, _ = drivers.make_renderer(w, h, [6, 0, 0.5], 90, n_stars=600, tex_w=512, tex_h=256, frame_slots=frame_slots, math=math)
    assert r.frame_slots == frame_slots
    if pairs_on_host:
        r.accumulate_entity_layer = functools.partial(r.accumulate_entity_layer, pairs_on_host=True)
    out = os.path.join(tmp, tag, "v.mp4")
    drivers.render_video(r, w, h, n_frames=n_frames, fps=30, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5],
                         orbit=True, assemble=False, png_level=png_level, video_stream="off")
    r.close()
    return drivers._frames_dir(out)


@pytest.mark.parametrize("size,n_frames,math", [((1920, 1080), 40, None), ((320, 180), 150, None),
                                                 ((1920, 1080), 40, "hybrid"), ((320, 180), 150, "hybrid")])
def test_overlapped_video_loop_writes_the_frames_of_the_serial_loop(tmp_path, size, n_frames, math, hip_lib):
    """math="hybrid": two tile lists per frame on two streams, per-slot cached lists, the second march streams -- the frames
    of two slots in flight must still be the frames of one"""
    from PIL import Image
    from bhr_amd.output import DEVICE
    serial = _run(str(tmp_path), "serial", n_frames, size, frame_slots=1, png_level=0, pairs_on_host=True, math=math)
    overlapped = _run(str(tmp_path), "overlapped", n_frames, size, frame_slots=2, png_level=DEVICE, pairs_on_host=False, math=math)
    differing = []
    for f in range(n_frames):
        a = np.asarray(Image.open(os.path.join(serial, f"frame_{f:04d}.png")).convert("RGB"))
        b = np.asarray(Image.open(os.path.join(overlapped, f"frame_{f:04d}.png")).convert("RGB"))
        if not np.array_equal(a, b):
            differing.append((f, int((a != b).any(axis=2).sum())))
    assert not differing, f"frames that differ (frame, pixels): {differing[:10]}"
    first = np.asarray(Image.open(os.path.join(serial, "frame_0000.png")).convert("RGB"))
    last = np.asarray(Image.open(os.path.join(serial, f"frame_{n_frames - 1:04d}.png")).convert("RGB"))
    assert first.max() > 100 and (first != last).mean() > 0.01      # the frames are real and the scene moves
