"""The library's PNG encoder (host code of libbhr_hip.so, no GPU needed): files decode, with an
independent decoder (PIL), to exactly the pixels handed in -- the property save_image relies on
(render.py:420-425)."""
import io
import os

import numpy as np
import pytest
from PIL import Image


def _decode(data: bytes) -> np.ndarray:
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


def _images():
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:90, 0:160]
    smooth = np.stack([xx * 255 // 159, yy * 255 // 89, (xx + yy) % 256], axis=2).astype(np.uint8)
    stars = np.zeros((64, 97, 3), np.uint8)
    stars[rng.integers(0, 64, 40), rng.integers(0, 97, 40)] = rng.integers(0, 256, (40, 3))
    return {"smooth": smooth, "noise": rng.integers(0, 256, (75, 131, 3), dtype=np.uint8), "stars": stars,
            "one_pixel": np.array([[[1, 2, 3]]], np.uint8), "one_row": rng.integers(0, 256, (1, 33, 3), dtype=np.uint8),
            "one_col": rng.integers(0, 256, (40, 1, 3), dtype=np.uint8), "black": np.zeros((48, 48, 3), np.uint8)}


@pytest.mark.parametrize("name", list(_images()))
@pytest.mark.parametrize("level,threads", [(0, 1), (1, 1), (6, 1), (9, 1), (1, 4), (6, 3)])
def test_png_round_trip(name, level, threads):
    from bhr_amd.output import png_encode
    img = _images()[name]
    data = png_encode(img, level=level, threads=threads)
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    np.testing.assert_array_equal(_decode(data), img)


def test_png_band_splice_large_and_compresses():
    """A frame-sized image in 8 parallel bands is one valid stream, and a smooth image compresses as
    well as PIL's encoder at the same level, within 25 %."""
    from bhr_amd.output import png_encode
    yy, xx = np.mgrid[0:720, 0:1280]
    img = np.stack([(xx // 5) % 256, (yy // 3) % 256, ((xx + 2 * yy) // 7) % 256], axis=2).astype(np.uint8)
    data = png_encode(img, level=6, threads=8)
    np.testing.assert_array_equal(_decode(data), img)
    ref = io.BytesIO()
    Image.fromarray(img).save(ref, format="PNG")
    assert len(data) < 1.25 * len(ref.getvalue())


def test_png_write_and_errors(tmp_path):
    from bhr_amd.output import png_write, png_encode, quantize
    frame = np.random.default_rng(1).random((36, 64, 3), dtype=np.float32) * 1.2 - 0.1
    path = tmp_path / "sub" / "f.png"
    png_write(str(path), quantize(frame))
    np.testing.assert_array_equal(np.array(Image.open(path)), (np.clip(frame, 0, 1) * 255).astype(np.uint8))
    assert not os.path.exists(str(path) + ".tmp")
    with pytest.raises(ValueError):
        png_encode(np.zeros((4, 4, 4), np.uint8))
    with pytest.raises(ValueError):
        png_encode(np.zeros((4, 4, 3), np.uint8), level=11)
    with pytest.raises(ValueError):
        png_write(str(tmp_path / "no_such_dir_file" / ".." / ".." / "x" / "\0bad.png"), np.zeros((4, 4, 3), np.uint8))
