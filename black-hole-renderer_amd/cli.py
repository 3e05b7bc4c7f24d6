"""Command line of the reference (render.py:4518-4694), same flags and defaults, on the HIP backend.

Additions: ``--device hip`` (the default and only backend; ``gpu`` is accepted as an alias, ``cpu`` is
refused -- this build has no CPU path), ``-r 8k``, ``--gpus N`` (row-block tiling of a still image
inside one process; for ``--video`` launch one process per GPU with torchrun and frames are sharded
round-robin by RANK / WORLD_SIZE), ``--disk_model`` (the analytic Disk V2 sources of the march kernel).
"""
from __future__ import annotations

import argparse
import math
import os

from .renderer import R_DISK_INNER_DEFAULT, R_DISK_OUTER_DEFAULT

DISK_GENERATION_SCALE_CHOICES = (1, 2, 4)
RESOLUTIONS = {"8k": (7680, 4320), "4k": (3840, 2160), "fhd": (1920, 1080), "hd": (1280, 720), "sd": (640, 360)}


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Schwarzschild black-hole ray tracer (MI355X / HIP)")
    p.add_argument("--pov", type=float, nargs=3, default=[6, 0, 0.5], metavar=("X", "Y", "Z"),
                   help="camera position (default: 6 0 0.5)")
    p.add_argument("--fov", type=float, default=90, help="field of view 0-180 deg (default: 90)")
    p.add_argument("--resolution", "-r", type=str, default="fhd", choices=list(RESOLUTIONS),
                   help="8k/4k/fhd/hd/sd (default: fhd)")
    p.add_argument("--texture", "-t", type=str, default=None, help="skybox texture path")
    p.add_argument("--output", "-o", type=str, default="output/blackhole.png",
                   help="output path (default: output/blackhole.png)")
    p.add_argument("--step_size", "-s", type=float, default=0.1, help="integration step (default: 0.1)")
    p.add_argument("--r_max", type=float, default=10, help="escape radius (default: 10)")
    p.add_argument("--n_stars", type=int, default=6000, help="stars of the procedural skybox (default: 6000)")
    p.add_argument("--disk_texture", type=str, default=None,
                   help="disk texture path (default: procedural; still images only)")
    p.add_argument("--disk_generation_scale", type=int, default=2, choices=DISK_GENERATION_SCALE_CHOICES,
                   help="[deprecated] ignored by the lifecycle system (default: 2)")
    p.add_argument("--force_regenerate_disk_texture", action="store_true",
                   help="[deprecated] the lifecycle system always regenerates")
    p.add_argument("--disk_inner_radius", "--ar1", dest="disk_inner_radius", type=float,
                   default=R_DISK_INNER_DEFAULT, help=f"disk inner radius (default: {R_DISK_INNER_DEFAULT})")
    p.add_argument("--disk_outer_radius", "--ar2", dest="disk_outer_radius", type=float,
                   default=R_DISK_OUTER_DEFAULT, help=f"disk outer radius (default: {R_DISK_OUTER_DEFAULT})")
    p.add_argument("--disk_tilt", type=float, default=0.0, help="disk tilt in degrees (default: 0)")
    p.add_argument("--lens_flare", action="store_true", help="enable the lens flare")
    p.add_argument("--anti_alias", type=str, default="disabled", choices=["disabled", "lod_radius"],
                   help="disabled | lod_radius (ray-differential mip LOD) (default: disabled)")
    p.add_argument("--aa_strength", type=float, default=1.0, help="LOD multiplier 0.5-2.0 (default: 1.0)")
    p.add_argument("--device", "-d", type=str, default="hip", choices=["hip", "gpu", "cpu"],
                   help="hip (MI355X). 'gpu' is an alias; 'cpu' is refused: there is no CPU path")
    p.add_argument("--disk_model", type=str, default="texture", choices=["texture", "v2", "v2_volume"],
                   help="disk source of still images: the lifecycle texture (default), the analytic Disk V2 model at "
                        "each plane crossing, or its finite-thickness emission-absorption integral")
    p.add_argument("--gpus", type=int, default=1, help="row-block tile a still image over N GPUs of this node")
    p.add_argument("--math", type=str, default="hybrid", choices=["hybrid", "strict", "fast"],
                   help="march arithmetic.  strict: the reference's operations one by one with exactly rounded sqrt / divide "
                        "(ray paths bit-identical to an IEEE f32 evaluation of the reference).  hybrid (default): strict on the "
                        "8x8 tiles whose rays pass near the photon sphere, the fast arithmetic elsewhere -- within 3e-5 RMSE of "
                        "strict, 1.7x faster; anti-aliased (--anti_alias lod_radius) and tilted views add guards to the fast tiles and "
                        "march the pixels they flag again with the strict arithmetic.  fast: v_rsq / v_rcp + fast-math "
                        "everywhere (the analogue of Taichi's fast_math=True)")
    p.add_argument("--video_stream", type=str, default="auto", choices=["auto", "y4m", "off"],
                   help="--video: also hand the frames to the encoder as a yuv420p stream converted on the device "
                        "(auto: pipe into ffmpeg when it is on PATH; y4m: write <output stem>.y4m; off: PNG frames only)")
    p.add_argument("--png_encoder", type=str, default="device", choices=["device", "host"],
                   help="--video: PNG frames are filtered and Huffman-coded on the GPU (device, default) or by zlib on "
                        "host threads (host: ~10 %% smaller files, ~50 ms of a core per fhd frame)")
    p.add_argument("--ignore_taichi_cache", action="store_true", help="accepted for compatibility; no effect")
    p.add_argument("--video", action="store_true", help="render frames and assemble a video")
    p.add_argument("--interactive", action="store_true", help="not available in this build (needs ti.GUI)")
    p.add_argument("--orbit", action="store_true", help="video: orbit the camera around the origin")
    p.add_argument("--orbit_degrees", type=float, default=360.0, help="total orbit angle (default: 360)")
    p.add_argument("--n_frames", type=int, default=3600, help="video frames (default: 3600)")
    p.add_argument("--fps", type=int, default=36, help="video frame rate (default: 36)")
    p.add_argument("--resume", action="store_true", help="video: resume from progress.json")
    p.add_argument("--disk_rotation_algorithm", type=str, default="baseline",
                   choices=["baseline", "parametric", "keyframes"], help="[deprecated] ignored")
    p.add_argument("--disk_rotation_speed", type=float, default=0.1, help="disk rotation speed (default: 0.1)")
    p.add_argument("--keyframes_count", type=int, default=10, help="[deprecated] ignored")
    return p.parse_args(argv)


def validate_args(args) -> None:
    """Same checks and messages as render.py:4586-4616, plus the backend restrictions."""
    if not (0 < args.fov < 180):
        raise ValueError(f"FOV must be between 0 and 180 degrees, got {args.fov}")
    if args.disk_inner_radius >= args.disk_outer_radius:
        raise ValueError(f"disk_inner_radius ({args.disk_inner_radius}) must be less than "
                         f"disk_outer_radius ({args.disk_outer_radius})")
    if args.step_size <= 0:
        raise ValueError(f"step_size must be positive, got {args.step_size}")
    if not (0.5 <= args.aa_strength <= 2.0):
        raise ValueError(f"aa_strength must be between 0.5 and 2.0, got {args.aa_strength}")
    if args.n_frames <= 0:
        raise ValueError(f"n_frames must be positive, got {args.n_frames}")
    if args.fps <= 0:
        raise ValueError(f"fps must be positive, got {args.fps}")
    if not math.isfinite(args.orbit_degrees):
        raise ValueError(f"orbit_degrees must be finite, got {args.orbit_degrees}")
    if args.disk_texture and (args.video or args.interactive):
        raise ValueError("--disk_texture only supports still images; video/interactive use the lifecycle system")
    if getattr(args, "disk_model", "texture") != "texture" and (args.video or args.disk_texture):
        raise ValueError("--disk_model v2/v2_volume renders still images and takes no --disk_texture")
    if getattr(args, "device", "hip") == "cpu":
        raise ValueError("--device cpu: this build renders on the MI355X only (no CPU path)")
    if getattr(args, "gpus", 1) < 1:
        raise ValueError(f"gpus must be >= 1, got {args.gpus}")
    if getattr(args, "interactive", False):
        raise ValueError("--interactive needs the Taichi GUI and is not part of this build")


def main(argv=None) -> int:
    args = parse_args(argv)
    validate_args(args)
    from . import drivers

    width, height = RESOLUTIONS[args.resolution]
    fov = args.fov % 180

    if args.video:
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if "BHR_FORCE_DEVICE" in os.environ:        # rehearsal: several ranks sharing one card
            local_rank = int(os.environ["BHR_FORCE_DEVICE"])
        elif world > 1:                             # a launcher that shows every rank only its own GPU: ordinal 0
            from . import _lib
            have = int(_lib.load().bhr_device_count())
            if have > 0 and local_rank >= have:
                local_rank %= have
        renderer, _, _, _ = drivers.make_renderer(
            width, height, args.pov, fov, args.step_size, args.texture, args.n_stars, 2048, 1024, args.r_max, None,
            args.disk_inner_radius, args.disk_outer_radius, args.disk_tilt, args.lens_flare, args.anti_alias,
            args.aa_strength, args.disk_rotation_speed, device_index=local_rank, math=args.math)
        print(f"Rendering video: {args.n_frames} frames at {width}x{height} (rank {rank}/{world})")
        drivers.render_video(renderer, width, height, n_frames=args.n_frames, fps=args.fps,
                             output_path=args.output, fov=fov, static_cam_pos=args.pov, orbit=args.orbit,
                             resume=args.resume, disk_rotation_speed=args.disk_rotation_speed,
                             orbit_degrees=args.orbit_degrees, rank=rank, world=world, video_stream=args.video_stream,
                             png_level=(drivers.DEVICE if args.png_encoder == "device" else drivers.VIDEO_LEVEL))
        if world > 1:
            from . import distributed as D
            dist = D.init("gloo")          # a barrier is all the ranks exchange: frames are independent
            dist.barrier()
            if rank == 0:
                done = D.merge_progress(drivers._frames_dir(args.output), world)
                if len(done) == args.n_frames:
                    drivers.assemble_video(drivers._frames_dir(args.output), args.n_frames, args.fps, args.output)
                else:
                    print(f"Warning: only {len(done)}/{args.n_frames} frames completed. Run again with --resume.")
            dist.barrier()
        renderer.close()
        return 0

    img = drivers.render_image(
        width=width, height=height, cam_pos=args.pov, fov=fov, step_size=args.step_size,
        skybox_path=args.texture, n_stars=args.n_stars, r_max=args.r_max, device="hip",
        disk_texture_path=args.disk_texture, r_disk_inner=args.disk_inner_radius,
        r_disk_outer=args.disk_outer_radius, disk_tilt=args.disk_tilt, lens_flare=args.lens_flare,
        anti_alias=args.anti_alias, aa_strength=args.aa_strength, disk_rotation_speed=args.disk_rotation_speed,
        gpus=args.gpus, disk_model=args.disk_model, math=args.math)
    drivers.save_image(img, args.output)
    return 0
