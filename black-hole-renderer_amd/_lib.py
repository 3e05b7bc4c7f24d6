"""ctypes binding of include/bhr.h.

There is deliberately no fallback: if libbhr_hip.so is missing or cannot be
loaded this module raises, and every renderer entry point goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

from .build import library_path

BHR_OK = 0
BHR_ERR_INVALID, BHR_ERR_NO_DEVICE, BHR_ERR_HIP, BHR_ERR_STATE, BHR_ERR_NOMEM = -1, -2, -3, -4, -5

SKIP_DIFFERENTIALS, SKIP_BLOOM, PERSISTENT, FORCE_FAST, FORCE_STRICT, LENS_FLARE, ROW_COSTS, GATHER_PEER = 1, 2, 4, 8, 16, 32, 64, 128
FORCE_HYBRID, GATHER_U8, GROUP_SERIAL, GROUP_PIPELINED, GROUP_TIME_MARCH, GROUP_ASYNC = 256, 512, 1024, 2048, 4096, 8192
MATH_FAST, MATH_STRICT, MATH_HYBRID = 0, 1, 2
LAYER_FINAL, LAYER_BG, LAYER_DISK, LAYER_BLUR = 0, 1, 2, 3
OUTPUT_F32, OUTPUT_BLUR, OUTPUT_U8 = 1, 2, 4

# every symbol include/bhr.h declares (tests check the .so exports exactly these)
SYMBOLS = (
    "bhr_last_error", "bhr_abi_version", "bhr_device_count", "bhr_create", "bhr_destroy", "bhr_sync",
    "bhr_set_skybox", "bhr_skybox_add_glow", "bhr_skybox_build", "bhr_get_skybox", "bhr_set_disk_texture", "bhr_get_disk_texture", "bhr_get_disk_mip", "bhr_num_mip_levels",
    "bhr_bg_init", "bhr_generate_background", "bhr_set_entity_staging", "bhr_set_comp", "bhr_read_comp",
    "bhr_fill_comp_slice", "bhr_set_compose_stats", "bhr_compose_texture", "bhr_eval_noise", "bhr_render",
    "bhr_read_layer", "bhr_write_layer", "bhr_bloom", "bhr_set_outputs", "bhr_set_option", "bhr_debug_read", "bhr_lens_flare", "bhr_lens_flare_sums", "bhr_read_final_u8", "bhr_get_counters", "bhr_get_row_costs_split", "bhr_mip_lds_level", "bhr_hybrid_info", "bhr_hybrid_repairs", "bhr_timing_reset", "bhr_timing_dump", "bhr_get_row_costs", "bhr_selftest", "bhr_group_render", "bhr_group_render_subset", "bhr_group_sync", "bhr_read_gathered", "bhr_read_gathered_u8", "bhr_tile_export", "bhr_tile_connect", "bhr_tile_render", "bhr_disk_v2_eval", "bhr_set_disk_source", "bhr_set_disk_volume_options", "bhr_entity_profile_upload", "bhr_entity_profile_reset", "bhr_accumulate_entities", "bhr_accumulate_population",
    "bhr_stats_prepare", "bhr_stats_select", "bhr_stats_row_statistics",
    "bhr_png_bound", "bhr_png_encode", "bhr_png_write", "bhr_png_device_bound", "bhr_png_device_max_width", "bhr_png_encode_device", "bhr_png_device_menu",
    "bhr_sink_create", "bhr_sink_submit", "bhr_sink_drain",
    "bhr_sink_destroy", "bhr_y4m_open", "bhr_y4m_submit", "bhr_y4m_drain", "bhr_y4m_close",
)


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("row0", C.c_int32), ("row1", C.c_int32),
                ("step_size", C.c_float), ("r_max", C.c_float), ("r_disk_inner", C.c_float),
                ("r_disk_outer", C.c_float), ("disk_tilt_deg", C.c_float), ("anti_alias", C.c_int32),
                ("aa_strength", C.c_float), ("disk_rotation_speed", C.c_float), ("device", C.c_int32),
                ("math_mode", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("pixel_width", C.c_float), ("pixel_height", C.c_float),
                ("r_escape", C.c_float), ("t_offset", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("ray_steps", C.c_uint64), ("rays", C.c_uint64), ("march_ms", C.c_float), ("bloom_ms", C.c_float),
                ("frame_ms", C.c_float), ("background_ms", C.c_float), ("compose_ms", C.c_float),
                ("march_vgprs", C.c_int32), ("march_lds_bytes", C.c_int32), ("frames_timed", C.c_int32),
                ("march_ms_sum", C.c_float), ("bloom_ms_sum", C.c_float), ("ray_steps_sum", C.c_uint64),
                ("march_busy_ms", C.c_float), ("span_ms", C.c_float)]


TILE_SHM_WORDS = 8


class TileHandles(C.Structure):
    _fields_ = [("hblur", C.c_uint8 * 64), ("gather_f32", C.c_uint8 * 64), ("gather_u8", C.c_uint8 * 64),
                ("row0", C.c_int32), ("rows", C.c_int32), ("device", C.c_int32), ("has_gather_f32", C.c_int32),
                ("has_gather_u8", C.c_int32), ("reserved", C.c_int32 * 3)]


class BhrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libbhr_hip: {msg} (status {code})")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libbhr_hip.so and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise ImportError(
            f"{path} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(path)
    F, I32, P = C.POINTER(C.c_float), C.c_int32, C.c_void_p
    lib.bhr_last_error.restype = C.c_char_p
    lib.bhr_abi_version.restype = I32
    lib.bhr_device_count.restype = I32
    lib.bhr_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
    lib.bhr_destroy.argtypes = [P]
    lib.bhr_destroy.restype = None
    lib.bhr_sync.argtypes = [P]
    lib.bhr_set_skybox.argtypes = [P, F, I32, I32]
    lib.bhr_skybox_add_glow.argtypes = [P]
    lib.bhr_get_skybox.argtypes = [P, F]
    lib.bhr_set_disk_texture.argtypes = [P, F, I32, I32]
    lib.bhr_get_disk_texture.argtypes = [P, F]
    lib.bhr_get_disk_mip.argtypes = [P, I32, F]
    lib.bhr_num_mip_levels.argtypes = [P]
    lib.bhr_bg_init.argtypes = [P, I32, I32, I32, C.c_float, F, F]
    lib.bhr_generate_background.argtypes = [P, C.c_float]
    lib.bhr_set_entity_staging.argtypes = [P, F]
    lib.bhr_set_comp.argtypes = [P, F]
    lib.bhr_read_comp.argtypes = [P, F]
    lib.bhr_fill_comp_slice.argtypes = [P, I32, C.c_float]
    lib.bhr_set_compose_stats.argtypes = [P, C.c_float, C.c_float, F]
    lib.bhr_compose_texture.argtypes = [P, C.c_float, I32, C.c_float]
    lib.bhr_eval_noise.argtypes = [P, F, C.c_int64, I32, I32, C.c_float, C.c_float, F]
    lib.bhr_render.argtypes = [P, C.POINTER(Camera), C.c_uint32]
    lib.bhr_read_layer.argtypes = [P, I32, F]
    lib.bhr_read_gathered.argtypes = [P, F]
    PI32 = C.POINTER(C.c_int32)
    lib.bhr_skybox_build.argtypes = [P, I32, I32, C.POINTER(C.c_uint8), I32, I32, PI32, PI32, I32, PI32, PI32, I32, I32, F, F, F, F, I32]
    lib.bhr_write_layer.argtypes = [P, I32, F]
    lib.bhr_bloom.argtypes = [P]
    lib.bhr_set_outputs.argtypes = [P, C.c_uint32]
    lib.bhr_set_option.argtypes = [P, C.c_char_p, C.c_double]
    lib.bhr_debug_read.argtypes = [P, I32, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
    lib.bhr_lens_flare.argtypes = [P]
    lib.bhr_lens_flare_sums.argtypes = [P, C.POINTER(C.c_double)]
    lib.bhr_read_final_u8.argtypes = [P, C.POINTER(C.c_uint8)]
    lib.bhr_get_counters.argtypes = [P, C.POINTER(Counters)]
    lib.bhr_timing_reset.argtypes = [P]
    lib.bhr_timing_dump.argtypes = [P, F, I32]
    lib.bhr_hybrid_info.argtypes = [P, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    lib.bhr_get_row_costs.argtypes = [P, C.POINTER(C.c_uint64), I32]
    lib.bhr_hybrid_repairs.argtypes = [P, C.POINTER(I32)]
    lib.bhr_mip_lds_level.argtypes = [P]
    lib.bhr_mip_lds_level.restype = I32
    lib.bhr_get_row_costs_split.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), I32]
    lib.bhr_selftest.argtypes = [P, C.POINTER(C.c_uint64)]
    lib.bhr_group_render.argtypes = [C.POINTER(P), I32, C.POINTER(Camera), C.c_uint32, F]
    lib.bhr_group_render_subset.argtypes = [C.POINTER(P), I32, C.POINTER(Camera), C.c_uint32, F, C.POINTER(C.c_int32)]
    lib.bhr_group_sync.argtypes = [C.POINTER(P), I32]
    lib.bhr_read_gathered_u8.argtypes = [P, C.POINTER(C.c_uint8)]
    lib.bhr_tile_export.argtypes = [P, C.c_uint32, C.POINTER(TileHandles)]
    lib.bhr_tile_connect.argtypes = [P, I32, I32, C.POINTER(TileHandles), C.POINTER(C.c_uint64)]
    lib.bhr_tile_render.argtypes = [P, C.POINTER(Camera), C.c_uint32]
    U8, I64 = C.POINTER(C.c_uint8), C.c_int64
    lib.bhr_png_bound.argtypes = [I32, I32]
    lib.bhr_png_encode.argtypes = [U8, I32, I32, I32, I32, U8, I64, C.POINTER(I64)]
    lib.bhr_png_write.argtypes = [C.c_char_p, U8, I32, I32, I32, I32]
    lib.bhr_png_device_bound.argtypes = [I32, I32]
    lib.bhr_png_device_max_width.argtypes = []
    lib.bhr_png_encode_device.argtypes = [P, U8, I64, C.POINTER(I64)]
    U32P = C.POINTER(C.c_uint32)
    lib.bhr_png_device_menu.argtypes = [I32, U32P, U32P, U32P, C.POINTER(I32)]
    lib.bhr_sink_create.argtypes = [P, I32, I32, I32, C.POINTER(P)]
    lib.bhr_sink_submit.argtypes = [P, C.c_char_p]
    lib.bhr_sink_drain.argtypes = [P, C.POINTER(I64), C.POINTER(I64)]
    lib.bhr_sink_destroy.argtypes = [P]
    lib.bhr_y4m_open.argtypes = [P, C.c_char_p, I32, I32, I32, C.POINTER(P)]
    lib.bhr_y4m_submit.argtypes = [P]
    lib.bhr_y4m_drain.argtypes = [P, C.POINTER(I64), C.POINTER(I64)]
    lib.bhr_y4m_close.argtypes = [P]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("bhr_last_error", "bhr_destroy", "bhr_sink_destroy", "bhr_png_bound", "bhr_png_device_bound", "bhr_y4m_close"):
            fn.restype = I32
    lib.bhr_png_bound.restype = I64
    lib.bhr_png_device_bound.restype = I64
    lib.bhr_sink_destroy.restype = None
    lib.bhr_y4m_close.restype = None
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map a bhr_status to the exception the reference raises in the same situation."""
    if rc == BHR_OK:
        return
    msg = load().bhr_last_error().decode("utf-8", "replace")
    if rc == BHR_ERR_INVALID:
        raise ValueError(msg)
    if rc == BHR_ERR_STATE:
        raise AssertionError(msg)  # the reference asserts on call-order violations (render.py:3558, 3802)
    raise BhrError(rc, msg)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))
