"""The last-resort MP4 writer (bhr_amd/mp4.py): PNG frames as the samples of an 'mp4v' track with object type 0x6D.
No player exists in this image; the file is checked box by box with an independent walk and every sample is decoded."""
import io
import os
import struct

import numpy as np
import pytest


def _frames(tmp_path, n, w, h):
    from PIL import Image
    rng = np.random.default_rng(5)
    paths, arrays = [], []
    for k in range(n):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a[: h // 2] = k * 7 % 256                      # compressible half: sample sizes differ from frame to frame
        p = tmp_path / f"frame_{k:04d}.png"
        Image.fromarray(a).save(p)
        paths.append(str(p))
        arrays.append(a)
    return paths, arrays


def _boxes(buf, start, end, depth=0, out=None):
    """independent of mp4._walk: (depth, type, start, size) of every box, descending into the container types"""
    out = [] if out is None else out
    at = start
    while at < end:
        size, kind = struct.unpack_from(">I4s", buf, at)
        head = 8
        if size == 1:
            size, head = struct.unpack_from(">Q", buf, at + 8)[0], 16
        assert size >= head and at + size <= end, (kind, at, size, end)
        out.append((depth, kind, at, size))
        if kind in (b"moov", b"trak", b"mdia", b"minf", b"dinf", b"stbl"):
            _boxes(buf, at + head, at + size, depth + 1, out)
        at += size
    assert at == end
    return out


def test_png_mp4_structure_and_samples(tmp_path):
    from PIL import Image
    from bhr_amd import mp4
    paths, arrays = _frames(tmp_path, 7, 48, 32)
    out = str(tmp_path / "v.mp4")
    nbytes = mp4.write_png_mp4(paths, 30, out, 48, 32)
    buf = open(out, "rb").read()
    assert nbytes == len(buf) and not os.path.exists(out + ".part")
    boxes = _boxes(buf, 0, len(buf))
    kinds = [k for _, k, _, _ in boxes]
    assert kinds[:2] == [b"ftyp", b"moov"] and kinds[-1] == b"mdat"
    for need in (b"mvhd", b"trak", b"tkhd", b"mdia", b"mdhd", b"hdlr", b"minf", b"vmhd", b"dinf", b"dref", b"stbl", b"stsd", b"stts",
                 b"stsc", b"stsz", b"co64"):
        assert kinds.count(need) == 1, need
    assert buf[8:12] == b"isom"
    info = mp4.read_samples(out)
    assert (info["width"], info["height"], info["timescale"], info["duration"]) == (48, 32, 30, 7)
    assert info["codec"] == "mp4v" and info["object_type"] == 0x6D and info["file_size"] == len(buf)
    mdat = [b for b in boxes if b[1] == b"mdat"][0]
    assert len(info["samples"]) == 7
    at = mdat[2] + 16                                   # 64-bit mdat header
    for (off, size), p, a in zip(info["samples"], paths, arrays):
        assert off == at and size == os.path.getsize(p)   # samples back to back, in order, inside mdat
        at += size
        data = buf[off:off + size]
        assert data == open(p, "rb").read()
        np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), a)
    assert at == mdat[2] + mdat[3]
    # handler, sample entry and decoder configuration as ISO/IEC 14496-12 / -1 lay them out
    hdlr = [b for b in boxes if b[1] == b"hdlr"][0]
    assert buf[hdlr[2] + 16:hdlr[2] + 20] == b"vide"
    stsd = [b for b in boxes if b[1] == b"stsd"][0]
    entry = stsd[2] + 16
    assert buf[entry + 4:entry + 8] == b"mp4v" and struct.unpack_from(">H", buf, entry + 8 + 6)[0] == 1      # data reference 1
    assert struct.unpack_from(">H", buf, entry + 8 + 74)[0] == 24 and buf[entry + 8 + 78 + 4:entry + 8 + 82 + 4] == b"esds"
    esds = buf[entry + 8 + 78 + 12:entry + struct.unpack_from(">I", buf, entry)[0]]
    assert esds[0] == 0x03 and esds[8] == 0x04 and esds[13] == 0x6D and esds[14] == 0x11 and esds[-6] == 0x06 and esds[-1] == 0x02
    stts = [b for b in boxes if b[1] == b"stts"][0]
    assert struct.unpack_from(">III", buf, stts[2] + 12) == (1, 7, 1)


def test_png_mp4_rejects_what_is_not_png(tmp_path):
    from bhr_amd import mp4
    p = tmp_path / "frame_0000.png"
    p.write_bytes(b"not a png at all, just bytes")
    with pytest.raises(ValueError, match="not a PNG"):
        mp4.write_png_mp4([str(p)], 30, str(tmp_path / "v.mp4"), 8, 8)
    assert not os.path.exists(tmp_path / "v.mp4")
    with pytest.raises(ValueError, match="no frames"):
        mp4.write_png_mp4([], 30, str(tmp_path / "v.mp4"), 8, 8)


def test_assemble_video_writes_an_mp4_without_any_encoder(tmp_path, monkeypatch):
    """drivers.assemble_video in this image (no imageio / pyav, no ffmpeg): the MP4 holds the PNG frames"""
    from bhr_amd import drivers, mp4
    monkeypatch.setenv("PATH", str(tmp_path / "nothing_here"))
    paths, arrays = _frames(tmp_path, 4, 64, 40)
    out = str(tmp_path / "clip.mp4")
    assert drivers.assemble_video(str(tmp_path), 4, 24, out) is True
    info = mp4.read_samples(out)
    assert (info["width"], info["height"], info["timescale"], info["duration"]) == (64, 40, 24, 4)
    assert [s for _, s in info["samples"]] == [os.path.getsize(p) for p in paths]
    os.remove(paths[2])
    assert drivers.assemble_video(str(tmp_path), 4, 24, str(tmp_path / "clip2.mp4")) is False
