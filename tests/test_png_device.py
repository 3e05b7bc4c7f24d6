"""Device PNG encoder (csrc/png_device.hip, include/bhr_output.h).

CPU part: the code menu the kernels use is inspected through bhr_png_device_menu -- every entry is a complete
prefix code whose announced block header a stock inflater accepts (streams are assembled here in Python, bit by bit,
exactly as the kernels lay them out).  GPU part: whole files -- decoded with zlib and PIL they give back the pixels
of bhr_read_final_u8; every chunk CRC and the stream's Adler-32 hold; the frame sink writes the same files."""
import os
import struct
import zlib

import numpy as np
import pytest

from bhr_amd import scenes


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, nbits):
        self.acc |= (int(v) & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def align(self):
        if self.n:
            self.out.append(self.acc & 255)
            self.acc, self.n = 0, 0


def _deflate_with(entry, rows):
    """Raw deflate stream of `rows` (bytes objects), one dynamic block + empty stored block per row."""
    codes, hdr, hdr_bits = entry
    b = Bits()
    for k, row in enumerate(rows):
        for w in range((hdr_bits + 31) // 32):
            b.put(int(hdr[w]), min(32, hdr_bits - 32 * w))
        for s in list(row) + [256]:
            c = int(codes[s])
            b.put(c >> 4, c & 15)
        b.put(1 if k == len(rows) - 1 else 0, 3)
        b.align()
        b.out += b"\x00\x00\xff\xff"
    return bytes(b.out)


def test_menu_entries_are_complete_prefix_codes(hip_lib):
    from bhr_amd.output import png_device_menu
    menu = png_device_menu()
    assert len(menu) == 16
    for codes, hdr, hdr_bits in menu:
        lens = (codes & 15).astype(int)
        assert lens.min() >= 1 and lens.max() <= 15                      # every literal and end-of-block has a code
        assert sum(2.0 ** -l for l in lens) == 1.0                       # Kraft equality: complete
        rev = codes >> 4
        words = set()
        for s in range(257):                                             # prefix-free, as an LSB-first decoder reads them
            c = "".join(str((int(rev[s]) >> k) & 1) for k in range(lens[s]))
            words.add(c)
        assert len(words) == 257
        assert not any(a != b and b.startswith(a) for a in words for b in words if len(a) < len(b))
        assert 0 < hdr_bits <= 64 * 32
    assert int((menu[-1][0] & 15).max()) == 9 and int((menu[-1][0] & 15).min()) == 8    # the flat code bounds the worst case


def test_every_menu_header_inflates(hip_lib):
    from bhr_amd.output import png_device_menu
    rng = np.random.default_rng(5)
    for k, entry in enumerate(png_device_menu()):
        rows = []
        for n in (1, 7, 300):
            v = np.rint(rng.laplace(0, 1 + k, n)).astype(np.int64) & 255
            rows.append(bytes(v.astype(np.uint8)))
        rows.append(bytes(range(256)))                                    # every literal once
        raw = _deflate_with(entry, rows)
        d = zlib.decompressobj(-15)
        got = d.decompress(raw)
        assert d.eof and got == b"".join(rows), f"menu entry {k}"


def test_device_encoder_width_limit_is_beyond_every_baseline_size(hip_lib):
    from bhr_amd import _lib
    w = _lib.load().bhr_png_device_max_width()
    assert 2 * 7680 <= w < 20000                  # a scanline lives in LDS: ~17 000 pixels; 8k is 7680


def _parse_png(data):
    """-> (width, height, [IDAT payloads]); checks signature, chunk order and every CRC."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(data):
        n, typ = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        crc, = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert zlib.crc32(typ + body) == crc, f"CRC of {typ} chunk at {at}"
        chunks.append((typ, body))
        at += 12 + n
    assert at == len(data)
    assert chunks[0][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)
    assert all(t == b"IDAT" for t, _ in chunks[1:-1])
    return w, h, [b for t, b in chunks[1:-1]]


def _unfilter(raw, w, h):
    n = 3 * w
    out = np.zeros((h, n), np.uint8)
    prev = np.zeros(n, np.int32)
    types = []
    for r in range(h):
        line = np.frombuffer(raw, np.uint8, n + 1, r * (n + 1))
        f, x = int(line[0]), line[1:].astype(np.int32)
        types.append(f)
        cur = np.zeros(n, np.int32)
        if f == 0:
            cur = x
        elif f == 2:
            cur = (x + prev) & 255
        else:
            for i in range(n):
                a = cur[i - 3] if i >= 3 else 0
                b = prev[i]
                c = prev[i - 3] if i >= 3 else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    q = a + b - c
                    pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (x[i] + p) & 255
        out[r] = cur
        prev = cur
    return out.reshape(h, w, 3), types


def _frame(w, h, **kw):
    from bhr_amd import HipRenderer
    r = HipRenderer(w, h, scenes.analytic_skybox(), scenes.noisy_disk(), **kw)
    r.render_async([6, 0, 0.5], 90)
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(256, 144), (97, 33), (1, 1), (2, 5), (1920, 64), (3840, 16), (7680, 8)])
def test_device_png_decodes_to_the_quantised_frame(size, hip_lib):
    from PIL import Image
    import io
    from bhr_amd.output import png_encode_device
    w, h = size
    r = _frame(w, h)
    data = png_encode_device(r)
    want = r.read_final_u8()
    pw, ph, idat = _parse_png(data)
    assert (pw, ph) == (w, h) and len(idat) == h                          # one chunk per scanline
    raw = zlib.decompress(b"".join(idat))                                # checks the Adler-32
    assert len(raw) == h * (3 * w + 1)
    img, types = _unfilter(raw, w, h) if w * h <= 256 * 144 else (None, None)
    if img is not None:
        np.testing.assert_array_equal(img, want)
        assert set(types) <= {0, 1, 2, 3, 4}
    np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want)
    assert len(data) <= _lib_bound(w, h)
    r.close()


def _lib_bound(w, h):
    from bhr_amd import _lib
    return _lib.load().bhr_png_device_bound(w, h)


@pytest.mark.gpu
def test_device_png_of_extreme_frames(hip_lib):
    """Black, white, and incompressible noise through write_layer: filters and code choice at their extremes; the
    noise frame must stay inside the bound and close to the raw size (flat 8/9-bit code)."""
    from PIL import Image
    import io
    from bhr_amd import _lib
    from bhr_amd.output import png_encode_device
    w, h = 320, 90
    r = _frame(w, h)
    rng = np.random.default_rng(11)
    sizes = {}
    for name, frame in (("black", np.zeros((h, w, 3), np.float32)), ("white", np.ones((h, w, 3), np.float32)),
                        ("noise", rng.random((h, w, 3), dtype=np.float32)),
                        ("ramp", np.broadcast_to(np.linspace(0, 1, w, dtype=np.float32)[None, :, None], (h, w, 3)).copy())):
        r.write_layer(_lib.LAYER_FINAL, frame)
        data = png_encode_device(r)
        want = (np.clip(frame, 0, 1) * 255).astype(np.uint8)
        _parse_png(data)
        np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want, err_msg=name)
        sizes[name] = len(data)
    raw = h * (3 * w + 1)
    assert sizes["black"] < raw / 5 and sizes["white"] < raw / 5          # one bit per zero residual + header + framing
    assert raw <= sizes["noise"] <= raw * 1.02 + h * 80
    assert sizes["ramp"] < raw / 4
    r.close()


@pytest.mark.gpu
def test_device_png_size_against_zlib(hip_lib):
    """Huffman-only coding from a static menu: within 35 % of zlib level 1 on a rendered frame, far below raw."""
    from bhr_amd.output import png_encode, png_encode_device
    r = _frame(960, 540)
    dev = len(png_encode_device(r))
    host = len(png_encode(r.read_final_u8(), 1))
    print(f"[png] 960x540: device {dev} B, zlib level 1 {host} B, raw {960 * 540 * 3} B")
    assert dev < 1.35 * host and dev < 0.6 * 960 * 540 * 3
    r.close()


@pytest.mark.gpu
def test_sink_with_device_encoder_writes_the_same_files(tmp_path, hip_lib):
    from PIL import Image
    from bhr_amd.output import FrameSink, DEVICE, png_encode_device
    r = _frame(256, 144)
    cams = [([6, 0, 0.5], 90), ([5, 2, 1.0], 80), ([-7, 1, 0.3], 70), ([3.2, 0.5, 0.12], 100)] * 3
    want, direct = [], []
    with FrameSink(r, slots=3, workers=2, level=DEVICE) as sink:
        for k, (c, f) in enumerate(cams):
            r.render_async(c, f)
            sink.submit(str(tmp_path / f"f{k:02d}.png"))
            if k < 4:
                want.append(r.read_final_u8())
                direct.append(png_encode_device(r))
        frames, nbytes = sink.drain()
    assert frames == len(cams)
    total = 0
    for k in range(len(cams)):
        p = tmp_path / f"f{k:02d}.png"
        total += os.path.getsize(p)
        np.testing.assert_array_equal(np.asarray(Image.open(p).convert("RGB")), want[k % 4])
        assert p.read_bytes() == direct[k % 4]                            # deterministic: same frame, same bytes
    assert total == nbytes
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    r.close()


@pytest.mark.gpu
def test_device_png_of_a_row_block_context(hip_lib):
    """A tile context (rows 40..104 of a 144-row frame) encodes its own rows: a 64-row PNG."""
    from PIL import Image
    import io
    from bhr_amd import HipRenderer
    from bhr_amd.output import png_encode_device
    r = HipRenderer(256, 144, scenes.analytic_skybox(), scenes.noisy_disk(), rows=(40, 104))
    r.render_async([6, 0, 0.5], 90)
    img = np.asarray(Image.open(io.BytesIO(png_encode_device(r))).convert("RGB"))
    np.testing.assert_array_equal(img, r.read_final_u8())
    assert img.shape == (64, 256, 3)
    r.close()


@pytest.mark.gpu
def test_widest_frame_and_the_error_beyond_it(hip_lib):
    """The widest frame the encoder takes (150 KB of LDS per scanline) decodes correctly; one pixel more is refused at
    sink creation and at encode time with a message naming the host encoder."""
    from PIL import Image
    import io
    from bhr_amd import HipRenderer, _lib
    from bhr_amd.output import DEVICE, FrameSink, png_encode_device
    wmax = _lib.load().bhr_png_device_max_width()
    r = HipRenderer(wmax, 3, scenes.analytic_skybox(), scenes.noisy_disk())
    r.render_async([6, 0, 0.5], 90)
    np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(png_encode_device(r))).convert("RGB")), r.read_final_u8())
    r.close()
    r = HipRenderer(wmax + 1, 3, scenes.analytic_skybox(), scenes.noisy_disk())
    r.render_async([6, 0, 0.5], 90)
    with pytest.raises(ValueError, match="host encoder"):
        png_encode_device(r)
    with pytest.raises(ValueError, match="host encoder"):
        FrameSink(r, slots=2, workers=1, level=DEVICE)
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1920, 1080), (7680, 4320)])
def test_whole_frames_at_the_baseline_sizes(size, hip_lib):
    """fhd and 8k frames (analytic textures of bhr_amd.scenes): every scanline's chunk CRC, the Adler-32 of the stream and
    the decoded pixels; the file stays far below the raw size."""
    from PIL import Image
    import io
    from bhr_amd.output import png_encode_device
    Image.MAX_IMAGE_PIXELS = None
    w, h = size
    r = _frame(w, h, step_size=0.1 if w < 4000 else 0.05)
    data = png_encode_device(r)
    want = r.read_final_u8()
    pw, ph, idat = _parse_png(data)
    assert (pw, ph) == (w, h) and len(idat) == h
    assert len(zlib.decompress(b"".join(idat))) == h * (3 * w + 1)
    np.testing.assert_array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want)
    assert len(data) < 0.5 * want.nbytes
    r.close()
