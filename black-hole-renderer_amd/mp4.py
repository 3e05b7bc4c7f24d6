"""A minimal ISO base-media (MP4) writer for the video driver's last resort.

The reference assembles its MP4 from the PNG frames with libx264 through imageio + pyav (render.py:4497-4503).  Where
neither those modules nor an ``ffmpeg`` binary exist (this image), ``drivers.assemble_video`` still hands back an .mp4: the
PNG frames themselves, one per sample, in an 'mp4v' track whose decoder configuration carries object type 0x6D -- the
MP4 registration authority's code point for PNG, which ffmpeg / mpv / VLC decode.  Lossless, large (the frames are not
re-coded), and one ``ffmpeg -i in.mp4 -c:v libx264 -pix_fmt yuv420p out.mp4`` away from the reference's file.

Only what such a file needs: ftyp, moov (mvhd, one trak: tkhd, mdia: mdhd, hdlr, minf: vmhd, dinf/dref, stbl: stsd
[mp4v + esds], stts, stsc, stsz, co64), mdat with a 64-bit size.  ``read_samples`` walks the same boxes back (tests).
"""
import os
import struct
from typing import Iterable, List, Sequence, Tuple

PNG_OBJECT_TYPE = 0x6D          # ISO/IEC 14496-1 objectTypeIndication registered for PNG
PNG_MAGIC = b"\x89PNG\r\n\x1a\n"


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag: int, payload: bytes) -> bytes:
    """MPEG-4 descriptor: tag, length in 7-bit groups (four bytes, as every muxer writes it), payload."""
    n = len(payload)
    return bytes([tag, 0x80 | (n >> 21) & 0x7F, 0x80 | (n >> 14) & 0x7F, 0x80 | (n >> 7) & 0x7F, n & 0x7F]) + payload


MATRIX = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def _moov(width: int, height: int, fps: int, sizes: Sequence[int], first_offset: int) -> bytes:
    n = len(sizes)
    timescale, duration = int(fps), n                       # one tick per frame
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">IH", 0x10000, 0x0100) +
                 b"\0" * 10 + MATRIX + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\0" * 8 + struct.pack(">hhhH", 0, 0, 0, 0) +
                 MATRIX + struct.pack(">II", width << 16, height << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, b"\0" * 8)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    peak = max(sizes) if n else 0
    avg_bitrate = min(int(sum(sizes) * 8 * fps / max(n, 1)), 0xFFFFFFFF)
    dec = _descr(0x04, bytes([PNG_OBJECT_TYPE, (0x04 << 2) | 1]) + struct.pack(">I", min(peak, 0xFFFFFF))[1:] +
                 struct.pack(">II", min(peak * 8 * fps, 0xFFFFFFFF), avg_bitrate))
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) + dec + _descr(0x06, b"\x02")))
    name = b"PNG frames (bhr)"
    mp4v = _box(b"mp4v", b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HHIIIH", width, height, 0x480000, 0x480000, 0, 1) +
                bytes([len(name)]) + name + b"\0" * (31 - len(name)) + struct.pack(">Hh", 24, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + mp4v)
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, 1))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + struct.pack(f">{n}I", *sizes))
    offs, at = [], first_offset
    for s in sizes:
        offs.append(at)
        at += s
    co64 = _full(b"co64", 0, 0, struct.pack(">I", n) + struct.pack(f">{n}Q", *offs))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + co64)
    minf = _box(b"minf", vmhd + dinf + stbl)
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    return _box(b"moov", mvhd + _box(b"trak", tkhd + mdia))


def write_png_mp4(frame_paths: Sequence[str], fps: int, output_path: str, width: int, height: int) -> int:
    """Muxes the PNG files (in order, one sample each) into ``output_path``; returns the bytes written.  The sample
    table comes first (players start without reading the whole file), the frames are streamed through, never held."""
    sizes = [os.path.getsize(p) for p in frame_paths]
    if not sizes:
        raise ValueError("write_png_mp4: no frames")
    if any(s >= 1 << 32 for s in sizes):
        raise ValueError("write_png_mp4: a frame of 4 GiB or more does not fit a sample size")
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    moov_len = len(_moov(width, height, fps, sizes, 0))
    first = len(ftyp) + moov_len + 16                        # mdat header with a 64-bit size
    moov = _moov(width, height, fps, sizes, first)
    assert len(moov) == moov_len
    total = sum(sizes)
    tmp = output_path + ".part"
    with open(tmp, "wb") as out:
        out.write(ftyp)
        out.write(moov)
        out.write(struct.pack(">I4sQ", 1, b"mdat", 16 + total))
        for p, s in zip(frame_paths, sizes):
            with open(p, "rb") as f:
                head = f.read(8)
                if head != PNG_MAGIC:
                    raise ValueError(f"write_png_mp4: {p} is not a PNG file")
                out.write(head)
                left = s - 8
                while left > 0:
                    chunk = f.read(min(left, 1 << 22))
                    if not chunk:
                        raise IOError(f"write_png_mp4: {p} changed size while being read")
                    out.write(chunk)
                    left -= len(chunk)
    os.replace(tmp, output_path)
    return first + total


def png_size(path: str) -> Tuple[int, int]:
    """(width, height) from the IHDR chunk of a PNG file."""
    with open(path, "rb") as f:
        head = f.read(24)
    if head[:8] != PNG_MAGIC or head[12:16] != b"IHDR":
        raise ValueError(f"{path} is not a PNG file")
    return struct.unpack(">II", head[16:24])


# ---- reading it back (tests; also what a maintainer can point a debugger at) ----------------------------------------

def _walk(buf: bytes, start: int, end: int) -> Iterable[Tuple[bytes, int, int]]:
    at = start
    while at + 8 <= end:
        size, kind = struct.unpack_from(">I4s", buf, at)
        head = 8
        if size == 1:
            size = struct.unpack_from(">Q", buf, at + 8)[0]
            head = 16
        elif size == 0:
            size = end - at
        if size < head or at + size > end:
            raise ValueError(f"box {kind!r} at {at}: size {size} runs past its parent")
        yield kind, at + head, at + size
        at += size


def _find(buf: bytes, start: int, end: int, path: Sequence[bytes]) -> Tuple[int, int]:
    for kind, a, b in _walk(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, path[1:])
    raise ValueError(f"box {path[0]!r} not found")


def read_samples(path: str) -> dict:
    """{'width', 'height', 'timescale', 'duration', 'object_type', 'samples': [(offset, size)], 'file_size'} of a file
    written by write_png_mp4 (any single-track file with stsz + co64/stco and one sample per chunk)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(min(size, 64 << 20))                  # ftyp + moov sit in front of the frames
    top = {k: (a, b) for k, a, b in _walk(head, 0, min(len(head), size)) if k in (b"ftyp", b"moov")} if size <= len(head) else None
    if top is None:                                         # mdat runs past what was read: walk the leading boxes only
        top, at = {}, 0
        while at + 8 <= len(head):
            n, kind = struct.unpack_from(">I4s", head, at)
            hl = 8
            if n == 1:
                n, hl = struct.unpack_from(">Q", head, at + 8)[0], 16
            if kind in (b"ftyp", b"moov"):
                top[kind] = (at + hl, at + n)
            if kind == b"mdat":
                break
            at += n
    if b"ftyp" not in top or b"moov" not in top:
        raise ValueError("not an MP4 file written with the sample table in front")
    ma, mb = top[b"moov"]
    a, b = _find(head, ma, mb, [b"mvhd"])
    timescale, duration = struct.unpack_from(">II", head, a + 12)
    sa, sb = _find(head, ma, mb, [b"trak", b"mdia", b"minf", b"stbl"])
    a, b = _find(head, sa, sb, [b"stsd"])
    entry = a + 8                                           # version/flags + entry count, then the sample entry box
    esize, ekind = struct.unpack_from(">I4s", head, entry)
    width, height = struct.unpack_from(">HH", head, entry + 8 + 24)
    ea, eb = _find(head, entry + 8 + 78, entry + esize, [b"esds"])
    esds = head[ea + 4:eb]
    at = esds.index(b"\x04", 1 + 4 + 3)                     # DecoderConfigDescriptor behind the ES descriptor's header
    while esds[at + 1] & 0x80:
        at += 1
    object_type = esds[at + 2]
    a, b = _find(head, sa, sb, [b"stsz"])
    uniform, n = struct.unpack_from(">II", head, a + 4)
    sizes = [uniform] * n if uniform else list(struct.unpack_from(f">{n}I", head, a + 12))
    try:
        a, b = _find(head, sa, sb, [b"co64"])
        offs = list(struct.unpack_from(f">{n}Q", head, a + 8))
    except ValueError:
        a, b = _find(head, sa, sb, [b"stco"])
        offs = list(struct.unpack_from(f">{n}I", head, a + 8))
    return {"width": width, "height": height, "timescale": timescale, "duration": duration, "codec": ekind.decode(),
            "object_type": object_type, "samples": list(zip(offs, sizes)), "file_size": size}
