"""Frame output: PNG files and the pipelined frame sink (include/bhr_output.h).

Counterpart of save_image (render.py:420-425) and of the PIL thread pool in render_video
(render.py:4412-4413, 4458-4467).  Decoded pixels are the reference's
``(np.clip(frame, 0, 1) * 255).astype(np.uint8)``; the compressed bytes are this encoder's own.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

DEFAULT_LEVEL = 6          # PIL's default zlib level for PNG
VIDEO_LEVEL = 1            # frames that are re-encoded into an MP4 anyway
DEVICE = -1                # BHR_PNG_DEVICE: filter + Huffman-code the frame on the GPU (csrc/png_device.hip)


def _u8(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected an (H, W, 3) uint8 image, got {a.dtype} {a.shape}")
    return a


def png_encode(rgb_u8: np.ndarray, level: int = DEFAULT_LEVEL, threads: int = 1) -> bytes:
    """(H, W, 3) uint8 -> PNG file bytes."""
    a = _u8(rgb_u8)
    h, w = a.shape[:2]
    lib = _lib.load()
    cap = lib.bhr_png_bound(w, h)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_int64(0)
    p8 = C.POINTER(C.c_uint8)
    _lib.check(lib.bhr_png_encode(a.ctypes.data_as(p8), w, h, level, threads, out.ctypes.data_as(p8), cap, C.byref(n)))
    return out[:n.value].tobytes()


def png_write(path: str, rgb_u8: np.ndarray, level: int = DEFAULT_LEVEL, threads: int = 0) -> None:
    """Image.fromarray(rgb_u8).save(path) with row bands deflated on ``threads`` threads
    (0: one per 256 rows, at most the CPUs of this process)."""
    a = _u8(rgb_u8)
    h, w = a.shape[:2]
    if threads <= 0:
        threads = max(1, min(len(os.sched_getaffinity(0)), h // 256))
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    _lib.check(_lib.load().bhr_png_write(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, level, threads))


def png_encode_device(renderer) -> bytes:
    """PNG file bytes of the renderer's FINAL layer, filtered and entropy coded on the device (bhr_png_encode_device)."""
    lib = _lib.load()
    cap = lib.bhr_png_device_bound(renderer.width, renderer.rows)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_int64(0)
    _lib.check(lib.bhr_png_encode_device(renderer._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[:n.value].tobytes()


def png_device_menu():
    """The encoder's code menu: list of (codes[257] uint32 = reversed code << 4 | length, header words, header bits)."""
    lib = _lib.load()
    n = C.c_int32(0)
    U32P = C.POINTER(C.c_uint32)
    out, k = [], 0
    while True:
        codes, hdr, bits = np.zeros(257, np.uint32), np.zeros(64, np.uint32), C.c_uint32(0)
        _lib.check(lib.bhr_png_device_menu(k, codes.ctypes.data_as(U32P), hdr.ctypes.data_as(U32P), C.byref(bits), C.byref(n)))
        out.append((codes, hdr, int(bits.value)))
        k += 1
        if k >= n.value:
            return out


def quantize(image: np.ndarray) -> np.ndarray:
    """save_image's 8-bit conversion: truncation, not rounding (render.py:423)."""
    return (np.clip(image, 0, 1) * 255).astype(np.uint8)


class FrameSink:
    """Device frame -> PNG file without stalling the render stream.

    ``submit(path)`` quantises the renderer's FINAL layer on the device, starts the copy into a pinned
    host slot and returns; worker threads encode and write.  ``drain()`` waits for the files.
    ``level=DEVICE`` encodes on the GPU as well: the workers only fetch the finished bytes and write them."""

    def __init__(self, renderer, slots: int = 0, workers: int = 0, level: int = VIDEO_LEVEL):
        if workers <= 0:
            workers = max(1, min(16, len(os.sched_getaffinity(0)) - 1))
        if slots <= 0:
            slots = workers + 4        # every encoder busy plus a few frames of slack for the renderer
        self._lib = _lib.load()
        self._sink = C.c_void_p()
        self._renderer = renderer          # keeps the context alive
        _lib.check(self._lib.bhr_sink_create(renderer._ctx, slots, workers, level, C.byref(self._sink)))
        self.workers, self.slots, self.level = workers, slots, level
        import weakref
        if not hasattr(renderer, "_sinks"):
            renderer._sinks = []
        renderer._sinks.append(weakref.ref(self))

    def submit(self, path: str) -> None:
        _lib.check(self._lib.bhr_sink_submit(self._sink, os.fsencode(path)))

    def drain(self):
        """-> (frames written, bytes written) since creation."""
        frames, nbytes = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.bhr_sink_drain(self._sink, C.byref(frames), C.byref(nbytes)))
        return frames.value, nbytes.value

    def close(self) -> None:
        if self._sink:
            self._lib.bhr_sink_destroy(self._sink)
            self._sink = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rgb_to_yuv420(rgb_u8: np.ndarray):
    """The stream's colour conversion on the host (checker for the device kernel; integer BT.601 limited range,
    chroma from the rounded mean of each 2x2 block): (H, W, 3) uint8 -> (Y (H, W), Cb (H/2, W/2), Cr (H/2, W/2))."""
    a = _u8(rgb_u8).astype(np.int32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    h, w = r.shape
    m = (a.reshape(h // 2, 2, w // 2, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    r, g, b = m[..., 0], m[..., 1], m[..., 2]
    cb = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128
    cr = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128
    return y.astype(np.uint8), cb.astype(np.uint8), cr.astype(np.uint8)


def read_y4m(path: str):
    """Minimal YUV4MPEG2 reader (4:2:0): -> (header dict, list of (Y, Cb, Cr) uint8 planes)."""
    with open(path, "rb") as f:
        head = f.readline().decode("ascii").split()
        assert head[0] == "YUV4MPEG2", head
        tags = {t[0]: t[1:] for t in head[1:]}
        w, h = int(tags["W"]), int(tags["H"])
        frames = []
        while True:
            line = f.readline()
            if not line:
                break
            assert line.startswith(b"FRAME"), line[:20]
            buf = np.frombuffer(f.read(w * h * 3 // 2), dtype=np.uint8)
            assert buf.size == w * h * 3 // 2, "truncated frame"
            frames.append((buf[:w * h].reshape(h, w), buf[w * h:w * h * 5 // 4].reshape(h // 2, w // 2),
                           buf[w * h * 5 // 4:].reshape(h // 2, w // 2)))
    return {"width": w, "height": h, "fps": tags["F"], "chroma": [t for t in head[1:] if t.startswith("C")][0],
            "tags": head[1:]}, frames


class Y4MStream:
    """Device frames -> one YUV4MPEG2 (yuv420p) stream, in submission order, without a PNG detour
    (replaces the PNG -> imread -> libx264 assembly of render.py:4497-4503; include/bhr_output.h)."""

    def __init__(self, renderer, path: str, fps: int, slots: int = 8):
        self._lib = _lib.load()
        self._s = C.c_void_p()
        self._renderer = renderer
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        _lib.check(self._lib.bhr_y4m_open(renderer._ctx, os.fsencode(path), int(fps), 1, slots, C.byref(self._s)))
        import weakref
        if not hasattr(renderer, "_sinks"):
            renderer._sinks = []
        renderer._sinks.append(weakref.ref(self))

    def submit(self) -> None:
        _lib.check(self._lib.bhr_y4m_submit(self._s))

    def drain(self):
        frames, nbytes = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.bhr_y4m_drain(self._s, C.byref(frames), C.byref(nbytes)))
        return frames.value, nbytes.value

    def close(self) -> None:
        if self._s:
            self._lib.bhr_y4m_close(self._s)
            self._s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
