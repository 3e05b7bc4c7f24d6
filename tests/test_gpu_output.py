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
