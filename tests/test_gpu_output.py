"""Frame sink: device-quantised frames reach PNG files unchanged while rendering continues."""
import os

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu


def test_sink_frames_equal_device_frames(tmp_path):
    from bhr_amd import HipRenderer, scenes
    from bhr_amd.output import FrameSink
    r = HipRenderer(320, 180, scenes.analytic_skybox(), scenes.noisy_disk(), disk_tilt=20.0)
    want = []
    with FrameSink(r, slots=3, workers=2, level=1) as sink:      # fewer slots than frames: submit must block, not drop
        for k in range(10):
            r.render_async([6, 0.3 * k, 0.5], 90, frame=k)
            sink.submit(str(tmp_path / f"f{k:02d}.png"))
        frames, nbytes = sink.drain()
        assert frames == 10 and nbytes > 10 * 100
        for k in range(10):                                        # same frames again, read back synchronously
            r.render_async([6, 0.3 * k, 0.5], 90, frame=k)
            want.append(r.read_final_u8())
    for k in range(10):
        got = np.array(Image.open(tmp_path / f"f{k:02d}.png"))
        np.testing.assert_array_equal(got, want[k])
    assert any((want[0] != want[9]).ravel())
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    r.close()


def test_sink_reports_write_errors(tmp_path):
    from bhr_amd import HipRenderer, scenes
    from bhr_amd.output import FrameSink
    r = HipRenderer(64, 36, scenes.analytic_skybox(), scenes.noisy_disk())
    sink = FrameSink(r, slots=2, workers=1)
    r.render_async([6, 0, 0.5], 90)
    sink.submit(str(tmp_path / "missing_dir" / "f.png"))
    with pytest.raises(ValueError):
        sink.drain()
    r.render_async([6, 0, 0.5], 90)                                 # the sink stays usable
    sink.submit(str(tmp_path / "ok.png"))
    assert sink.drain()[0] == 1
    sink.close()
    with pytest.raises(ValueError):
        FrameSink(r, slots=1000)
    late = FrameSink(r, slots=2, workers=1)
    r.render_async([6, 0, 0.5], 90)
    late.submit(str(tmp_path / "late.png"))
    r.close()                                                       # closing the renderer drains and frees its sinks
    assert os.path.isfile(tmp_path / "late.png") and not late._sink
    late.close()


# --------------------------------------------------------------------------- yuv420p stream (include/bhr_output.h)
def test_y4m_stream_matches_the_u8_frames(tmp_path):
    """Device RGB -> Y'CbCr 4:2:0 + ordered writer: the file parses as YUV4MPEG2, every plane equals the integer
    BT.601 conversion of the frame's own u8 quantisation (the bytes the PNG path would have stored), frames come out
    in submission order, and decoding back gives the u8 frame within BT.601 rounding on a flat-chroma check."""
    from bhr_amd import HipRenderer, scenes
    from bhr_amd.output import Y4MStream, read_y4m, rgb_to_yuv420
    s = scenes.SCENES["default"]
    r = HipRenderer(160, 90, scenes.analytic_skybox(), scenes.noisy_disk(), **s["kw"])
    cams = [[6, 0, 0.5], [5, 2, 1.0], [4, -3, 0.7], [6.5, 0.5, -0.4], [3.2, 0.5, 0.12]]
    path = str(tmp_path / "v" / "out.y4m")
    u8 = []
    with Y4MStream(r, path, fps=24, slots=2) as st:              # fewer slots than frames: submit has to wait
        for c in cams:
            r.render_async(c, 90)
            st.submit()
            u8.append(r.read_final_u8())
        frames, nbytes = st.drain()
    assert frames == len(cams) and nbytes == len(cams) * (6 + 160 * 90 * 3 // 2)
    head, planes = read_y4m(path)
    assert (head["width"], head["height"], head["fps"], head["chroma"]) == (160, 90, "24:1", "C420jpeg")
    assert "XCOLORRANGE=LIMITED" in head["tags"] and len(planes) == len(cams)
    for (y, cb, cr), img in zip(planes, u8):
        wy, wcb, wcr = rgb_to_yuv420(img)
        np.testing.assert_array_equal(y, wy)
        np.testing.assert_array_equal(cb, wcb)
        np.testing.assert_array_equal(cr, wcr)
        assert 16 <= y.min() and y.max() <= 235 and 16 <= cb.min() and cb.max() <= 240
    # decode (BT.601 limited range, chroma replicated) and compare with the u8 frame: luma-exact up to rounding
    y, cb, cr = (p.astype(np.float64) for p in planes[0])
    cbu, cru = np.repeat(np.repeat(cb, 2, 0), 2, 1) - 128, np.repeat(np.repeat(cr, 2, 0), 2, 1) - 128
    yy = 1.164383 * (y - 16)
    dec = np.stack([yy + 1.596027 * cru, yy - 0.391762 * cbu - 0.812968 * cru, yy + 2.017232 * cbu], axis=-1)
    luma = lambda a: 0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]   # noqa: E731
    assert np.abs(luma(dec) - luma(u8[0].astype(np.float64))).max() <= 2.5      # quantisation of Y (step 255/219) + chroma spill
    r.close()


def test_y4m_rejects_odd_sizes_and_bad_paths(tmp_path):
    from bhr_amd import HipRenderer, scenes
    from bhr_amd.output import Y4MStream
    r = HipRenderer(161, 90, scenes.analytic_skybox(32, 64), scenes.analytic_disk(16, 32))
    with pytest.raises(ValueError, match="even"):
        Y4MStream(r, str(tmp_path / "x.y4m"), fps=30)
    r.close()
    r = HipRenderer(160, 90, scenes.analytic_skybox(32, 64), scenes.analytic_disk(16, 32))
    with pytest.raises(ValueError, match="cannot open"):
        Y4MStream(r, str(tmp_path), fps=30)                       # a directory
    r.close()


def test_render_video_writes_the_stream(tmp_path):
    """render_video(video_stream="y4m"): the stream's frames are the PNG frames, converted."""
    from PIL import Image
    from bhr_amd import drivers
    from bhr_amd.output import read_y4m, rgb_to_yuv420
    out = str(tmp_path / "vid" / "v.mp4")
    r, _, _, _ = drivers.make_renderer(160, 90, [6, 0, 0.5], 90, n_stars=50, tex_w=256, tex_h=128)
    drivers.render_video(r, 160, 90, n_frames=5, fps=12, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5], orbit=True,
                         orbit_degrees=60.0, assemble=False, video_stream="y4m")
    r.close()
    head, planes = read_y4m(str(tmp_path / "vid" / "v.y4m"))
    assert head["fps"] == "12:1" and len(planes) == 5
    d = drivers._frames_dir(out)
    for k, (y, cb, cr) in enumerate(planes):
        png = np.array(Image.open(os.path.join(d, f"frame_{k:04d}.png")))
        wy, wcb, wcr = rgb_to_yuv420(png)
        np.testing.assert_array_equal(y, wy)
        np.testing.assert_array_equal(cb, wcb)
        np.testing.assert_array_equal(cr, wcr)


def test_render_video_pipes_the_stream_into_ffmpeg(tmp_path, monkeypatch):
    """video_stream="auto" with an `ffmpeg` on PATH: the frames reach the encoder's stdin as one YUV4MPEG2 stream while
    they render, and no PNG is read back.  The image has no ffmpeg, so a stand-in that stores its stdin in the output
    file plays the encoder (what is under test is the pipe, not libx264)."""
    import stat
    from bhr_amd import drivers
    from bhr_amd.output import read_y4m
    fake = tmp_path / "bin" / "ffmpeg"
    fake.parent.mkdir()
    fake.write_text('#!/bin/sh\nfor a in "$@"; do out="$a"; done\ncat > "$out"\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ["PATH"])
    out = str(tmp_path / "vid" / "v.mp4")
    r, _, _, _ = drivers.make_renderer(160, 90, [6, 0, 0.5], 90, n_stars=50, tex_w=256, tex_h=128)
    drivers.render_video(r, 160, 90, n_frames=7, fps=30, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5], orbit=True,
                         orbit_degrees=45.0, assemble=True, video_stream="auto")
    r.close()
    head, planes = read_y4m(out)                      # the stand-in wrote the stream it was given
    assert (head["width"], head["height"], head["fps"]) == (160, 90, "30:1") and len(planes) == 7
    assert all(p[0].max() > 40 for p in planes)
    d = drivers._frames_dir(out)
    assert sorted(f for f in os.listdir(d) if f.endswith(".png")) == [f"frame_{k:04d}.png" for k in range(7)]


def test_render_video_survives_an_encoder_that_dies(tmp_path, monkeypatch, capsys):
    """An `ffmpeg` that exits at once (e.g. built without libx264): every later write to the pipe fails with EPIPE.  The
    stream is dropped, the child reaped, and all PNG frames + progress.json are still written (advisor finding, round 2)."""
    import json, stat
    from bhr_amd import drivers
    fake = tmp_path / "bin" / "ffmpeg"
    fake.parent.mkdir()
    fake.write_text('#!/bin/sh\nexit 1\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + os.environ["PATH"])
    out = str(tmp_path / "vid" / "v.mp4")
    r, _, _, _ = drivers.make_renderer(160, 90, [6, 0, 0.5], 90, n_stars=50, tex_w=256, tex_h=128)
    drivers.render_video(r, 160, 90, n_frames=40, fps=30, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5], orbit=True,
                         orbit_degrees=45.0, assemble=True, video_stream="auto")
    r.close()
    d = drivers._frames_dir(out)
    assert sorted(f for f in os.listdir(d) if f.endswith(".png")) == [f"frame_{k:04d}.png" for k in range(40)]
    assert json.load(open(os.path.join(d, "progress.json")))["completed"] == list(range(40))
    text = capsys.readouterr().out
    assert "continuing with the PNG frames" in text or "falling back to the PNG frames" in text


def test_render_video_ends_with_an_mp4_even_without_an_encoder(tmp_path, monkeypatch):
    """No imageio / pyav and no ffmpeg (this image): render_video still leaves `output_path` behind -- the device-encoded
    PNG frames as the samples of an MP4 track (bhr_amd/mp4.py), every sample byte for byte a frame file, and the frame
    files decode to what the renderer quantised."""
    import io
    from PIL import Image
    from bhr_amd import drivers, mp4
    monkeypatch.setenv("PATH", str(tmp_path / "no_such_dir"))
    out = str(tmp_path / "vid" / "clip.mp4")
    r, _, _, _ = drivers.make_renderer(160, 96, [6, 0, 0.5], 90, n_stars=50, tex_w=256, tex_h=128)
    drivers.render_video(r, 160, 96, n_frames=12, fps=24, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5], orbit=True,
                         orbit_degrees=30.0, assemble=True, video_stream="auto")
    r.close()
    info = mp4.read_samples(out)
    assert (info["width"], info["height"], info["timescale"], info["duration"], info["object_type"]) == (160, 96, 24, 12, 0x6D)
    d = drivers._frames_dir(out)
    buf = open(out, "rb").read()
    for k, (off, size) in enumerate(info["samples"]):
        frame = open(os.path.join(d, f"frame_{k:04d}.png"), "rb").read()
        assert buf[off:off + size] == frame
        img = np.asarray(Image.open(io.BytesIO(frame)).convert("RGB"))
        assert img.shape == (96, 160, 3) and img.max() > 0
