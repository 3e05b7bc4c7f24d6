"""MI355X-native Schwarzschild ray tracer behind the reference's renderer boundary.

Layout
  csrc/        hand-written HIP kernels (gfx950) + the C ABI of include/bhr.h
  lib/         libbhr_hip.so (built in-tree by build.py / __graft_entry__.build())
  _lib.py      ctypes binding of the C ABI; raises if the library is missing
  renderer.py  HipRenderer -- same Python surface as TaichiRenderer (render.py:2189-4028)
  camera.py, textures.py, skybox.py, lifecycle.py, flare.py   host helpers on the path
  drivers.py   render_image / render_video (render.py:4031-4076, 4356-4511)
  cli.py       the reference CLI flags (render.py:4518-4694) + --device hip / -r 8k / --gpus
"""
__all__ = ["HipRenderer", "build_library", "library_path"]


def __getattr__(name):
    if name == "HipRenderer":
        from .renderer import HipRenderer
        return HipRenderer
    if name in ("build_library", "library_path"):
        from . import build
        return getattr(build, name)
    raise AttributeError(name)
