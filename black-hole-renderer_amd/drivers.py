"""Frame drivers on top of HipRenderer: single image and video (render.py:4031-4153, 4356-4511).

Kept from the reference: the lifecycle call sequence, the orbit camera, the resume format
(``.frames_<md5(output)[:16]>/progress.json`` = {params, completed}), PNG frames written by a
2-thread pool, MP4 assembly through imageio/pyav when those packages exist.
New: frames can be sharded round-robin over ranks (``rank``/``world``): every rank replays the
cheap CPU lifecycle for every frame index and renders only its own frames; normalisation
statistics are recomputed at every ``frame % 60 == 0`` by *frame index* (so the output does not
depend on the sharding), which also makes a resumed video identical to an uninterrupted one.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import time
from typing import List, Optional

import numpy as np

from .camera import orbit_position
from .lifecycle import make_factories
from .output import Y4MStream, FrameSink, VIDEO_LEVEL, DEVICE, png_write, quantize
from .renderer import HipRenderer, R_DISK_INNER_DEFAULT, R_DISK_OUTER_DEFAULT
from .skybox import load_or_generate_skybox
from .textures import compute_disk_texture_resolution, load_disk_texture


def save_image(image: np.ndarray, path: str) -> None:
    """float image -> 8-bit PNG with truncation, not rounding (render.py:420-425).  Encoded by the
    library (row bands deflated in parallel); non-PNG extensions go through PIL as in the reference."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if path.lower().endswith(".png"):
        png_write(path, quantize(image))
    else:
        from PIL import Image
        Image.fromarray(quantize(image)).save(path)
    print(f"Saved: {path}")


def init_lifecycle_system(renderer: HipRenderer, n_r: int, n_phi: int, seed: int = 42) -> dict:
    """Background parameters + the three pre-aged entity populations + a first composed texture
    (render.py:4079-4130)."""
    renderer.init_background_layer(n_r=n_r, n_phi=n_phi, seed=seed)
    factories = make_factories(n_r, n_phi, renderer.r_disk_inner, renderer.r_disk_outer, seed=seed)
    renderer.generate_background(t=0.0)
    renderer.accumulate_entity_layer(factories, now=0.0)
    renderer.recompute_interactive_stats()
    renderer.compose_interactive_texture()
    return factories


def advance_lifecycle_frame(renderer: HipRenderer, factories: dict, t: float, dt: float,
                            recompute_stats: bool = False, solo_idx: int = -1, compose: bool = True) -> None:
    """Tick the factories and rebuild the texture for time t (render.py:4133-4153).
    ``compose=False`` only advances the CPU state (used by ranks skipping a frame they do not render)."""
    for f in factories.values():
        f.tick(now=t, dt=dt)
    if not compose and not recompute_stats:
        return
    renderer.generate_background(t=t)
    renderer.accumulate_entity_layer(factories, now=t)
    if recompute_stats:
        renderer.recompute_interactive_stats()
    if compose:
        renderer.compose_interactive_texture(solo_idx=solo_idx)


def use_analytic_disk(renderer: HipRenderer, disk_model: str) -> bool:
    """--disk_model v2 / v2_volume: Disk V2 with the renderer's radii and default structure (seed 42)."""
    if disk_model == "texture":
        return False
    if disk_model not in ("v2", "v2_volume"):
        raise ValueError(f"unknown disk_model {disk_model!r}")
    from .disk_v2 import DiskV2Params
    renderer.use_disk_v2(DiskV2Params(r_in=renderer.r_disk_inner, r_out=renderer.r_disk_outer), seed=42,
                         volume=disk_model == "v2_volume")
    return True


def make_renderer(width, height, cam_pos, fov, step_size=0.1, skybox_path=None, n_stars=6000, tex_w=2048,
                  tex_h=1024, r_max=10.0, disk_texture_path=None, r_disk_inner=R_DISK_INNER_DEFAULT,
                  r_disk_outer=R_DISK_OUTER_DEFAULT, disk_tilt=0.0, lens_flare=False, anti_alias="disabled",
                  aa_strength=1.0, disk_rotation_speed=0.1, device_index=0, rows=None, frame_slots=None, math=None):
    """Renderer with a placeholder (or file) disk texture, as the reference's entry points build it
    (render.py:4044-4064, 4627-4644).  Returns (renderer, use_lifecycle, n_r, n_phi)."""
    # procedural sky: the host draws its random tables, the device rasterises them (nebula resize, star blobs in
    # NumPy's accumulation order, Milky-Way glow; skyglow.hip); an image file is loaded as it is
    procedural_sky = not (skybox_path and os.path.isfile(skybox_path))
    if procedural_sky:
        print(f"Texture not found: {skybox_path}, generating procedural skybox..." if skybox_path
              else "Generating procedural skybox...")
        skybox = np.zeros((tex_h, tex_w, 3), dtype=np.float32)          # fixes the size; built on the device below
    else:
        skybox, tex_h, tex_w = load_or_generate_skybox(skybox_path, tex_w, tex_h, n_stars)
    disk_tex = load_disk_texture(disk_texture_path)
    use_lifecycle = disk_tex is None
    if use_lifecycle:
        n_phi, n_r = compute_disk_texture_resolution(width, height, cam_pos, fov, r_disk_inner, r_disk_outer)
        disk_tex = np.zeros((n_r, n_phi, 4), dtype=np.float32)
    else:
        n_r, n_phi = disk_tex.shape[:2]
    renderer = HipRenderer(width, height, skybox, disk_tex, step_size=step_size, r_max=r_max,
                           r_disk_inner=r_disk_inner, r_disk_outer=r_disk_outer, disk_tilt=disk_tilt,
                           lens_flare=lens_flare, anti_alias=anti_alias, aa_strength=aa_strength,
                           disk_rotation_speed=disk_rotation_speed, device_index=device_index, rows=rows,
                           frame_slots=frame_slots, **({} if math is None else {"math": math}))
    if procedural_sky:
        renderer.build_procedural_skybox(seed=42, n_stars=n_stars)
    return renderer, use_lifecycle, n_r, n_phi


def render_image(width: int, height: int, cam_pos: List[float], fov: float, step_size: float,
                 skybox_path: Optional[str] = None, n_stars: int = 6000, tex_w: int = 2048, tex_h: int = 1024,
                 r_max: float = 10.0, device: str = "hip", disk_texture_path: Optional[str] = None,
                 r_disk_inner: float = R_DISK_INNER_DEFAULT, r_disk_outer: float = R_DISK_OUTER_DEFAULT,
                 disk_tilt: float = 0.0, lens_flare: bool = False, anti_alias: str = "disabled",
                 aa_strength: float = 1.0, disk_rotation_speed: float = 0.1, disk_generation_scale: int = 2,
                 force_regenerate_disk_texture: bool = False, ignore_taichi_cache: bool = False,
                 gpus: int = 1, disk_model: str = "texture", math: Optional[str] = None) -> np.ndarray:
    """One frame -> (H, W, 3) float32 (render.py:4031-4076).  ``gpus > 1`` tiles the frame in row
    blocks over that many devices of this node (bhr_group_render).  ``math``: march arithmetic
    ("strict" | "hybrid" | "fast"; None = HipRenderer's default, strict)."""
    if gpus > 1:
        from .multigpu import render_image_tiled
        return render_image_tiled(width, height, cam_pos, fov, gpus, step_size=step_size, skybox_path=skybox_path,
                                  n_stars=n_stars, tex_w=tex_w, tex_h=tex_h, r_max=r_max,
                                  disk_texture_path=disk_texture_path, r_disk_inner=r_disk_inner,
                                  r_disk_outer=r_disk_outer, disk_tilt=disk_tilt, lens_flare=lens_flare,
                                  anti_alias=anti_alias, aa_strength=aa_strength,
                                  disk_rotation_speed=disk_rotation_speed, disk_model=disk_model, math=math)
    renderer, use_lifecycle, n_r, n_phi = make_renderer(
        width, height, cam_pos, fov, step_size, skybox_path, n_stars, tex_w, tex_h, r_max, disk_texture_path,
        r_disk_inner, r_disk_outer, disk_tilt, lens_flare, anti_alias, aa_strength, disk_rotation_speed, math=math)
    if use_analytic_disk(renderer, disk_model):
        pass
    elif use_lifecycle:
        factories = init_lifecycle_system(renderer, n_r, n_phi, seed=42)
        advance_lifecycle_frame(renderer, factories, t=0.0, dt=0.0, recompute_stats=True)
    t0 = time.time()
    print(f"HIP: {width}x{height}, cam_pos={list(cam_pos)}, fov={fov}°, step_size={step_size}")
    img = renderer.render(cam_pos, fov, frame=0)
    c = renderer.counters()
    dt = time.time() - t0
    print(f"Done in {dt:.3f}s  (march {c['march_ms']:.2f} ms, bloom {c['bloom_ms']:.2f} ms, "
          f"{c['ray_steps'] / 1e6:.1f} Mray-steps, {c['ray_steps'] / max(c['march_ms'], 1e-6) / 1e3:.0f} Mray-steps/s)")
    renderer.close()
    return img


def _lib_max_png_width() -> int:
    from . import _lib
    return int(_lib.load().bhr_png_device_max_width())


def _frames_dir(output_path: str) -> str:
    name = ".frames_" + hashlib.md5(output_path.encode()).hexdigest()[:16]
    return os.path.join(os.path.dirname(output_path), name)


def assemble_video(temp_dir: str, n_frames: int, fps: int, output_path: str) -> bool:
    """PNG frames -> MP4 (render.py:4497-4503: libx264 through imageio's pyav plugin).  In order of preference: imageio +
    pyav as the reference; an ``ffmpeg`` binary on PATH (same codec, same pixel format); failing both, the frames
    themselves muxed into an MP4 as PNG-coded samples (mp4.write_png_mp4: lossless, plays in ffmpeg / mpv / VLC, and is one
    ffmpeg call away from the H.264 file).  Returns False only when a frame is missing."""
    frames = [os.path.join(temp_dir, f"frame_{frame:04d}.png") for frame in range(n_frames)]
    missing = [p for p in frames if not os.path.isfile(p)]
    if missing:
        print(f"{len(missing)} of {n_frames} frames are missing (first: {missing[0]}): no video assembled")
        return False
    try:
        import imageio.v3 as iio
        import av  # noqa: F401
    except ImportError:
        iio = None
    if iio is not None:
        writer = iio.imopen(output_path, "w", plugin="pyav")
        writer.init_video_stream("libx264", fps=fps)
        for p in frames:
            writer.write_frame(iio.imread(p))
        writer.close()
        print(f"Video saved: {output_path}")
        return True
    if shutil.which("ffmpeg"):
        import subprocess
        rc = subprocess.call(["ffmpeg", "-y", "-loglevel", "error", "-framerate", str(fps), "-i", os.path.join(temp_dir, "frame_%04d.png"),
                              "-c:v", "libx264", "-crf", "18", "-pix_fmt", "yuv420p", output_path])
        if rc == 0:
            print(f"Video saved: {output_path}")
            return True
        print(f"ffmpeg failed ({rc}) on the PNG frames; writing them into the MP4 as they are")
    from .mp4 import png_size, write_png_mp4
    w, h = png_size(frames[0])
    nbytes = write_png_mp4(frames, fps, output_path, w, h)
    print(f"Video saved: {output_path} ({n_frames} PNG-coded frames, {nbytes / 1e6:.0f} MB, lossless; no H.264 encoder in this "
          f"environment -- re-code with: ffmpeg -i {output_path} -c:v libx264 -crf 18 -pix_fmt yuv420p out.mp4)")
    return True


def _drop_stream(stream, encoder, err) -> None:
    """The yuv420p stream is an extra: when its consumer dies (an ffmpeg without libx264 exits at once and every later
    write fails with EPIPE) the stream is closed, the encoder reaped, and the render goes on with the PNG frames, from
    which assemble_video builds the MP4 at the end as the reference does (render.py:4497-4503).  Returns None."""
    print(f"Warning: video stream failed ({err}); continuing with the PNG frames")
    try:
        stream.close()
    except Exception:
        pass
    if encoder is not None:
        try:
            encoder.stdin.close()
        except Exception:
            pass
        try:
            encoder.kill()
        except Exception:
            pass
        try:
            encoder.wait(timeout=10)
        except Exception:
            pass
    return None


def render_video(renderer: HipRenderer, width: int, height: int, n_frames: int, fps: int, output_path: str,
                 fov: float, static_cam_pos: List[float], orbit: bool = False, resume: bool = False,
                 disk_rotation_speed: float = 0.1, orbit_degrees: float = 360.0, rank: int = 0, world: int = 1,
                 assemble: bool = True, png_level: int = DEVICE, sink_slots: int = 0, sink_workers: int = 0,
                 video_stream: str = "auto", stats: Optional[dict] = None, **_deprecated_kwargs) -> None:
    """N frames -> PNGs (+ MP4) (render.py:4356-4511).  Frame f is rendered by rank f % world.

    ``video_stream``: the reference assembles the MP4 by reading every PNG back (render.py:4497-4503).  Here a
    single-rank, non-resumed session can also hand the frames straight to the encoder as a yuv420p YUV4MPEG2
    stream converted on the device (output.Y4MStream): "auto" pipes it into ``ffmpeg -f yuv4mpegpipe`` when an
    ffmpeg binary is on PATH (MP4 written while the frames render; otherwise PNGs + assemble_video as before),
    "y4m" writes ``<output stem>.y4m`` beside the output, "off" never streams.  PNG frames and progress.json are
    written in every mode (resume format of the reference).

    ``png_level``: output.DEVICE (default) filters and Huffman-codes every frame on the GPU (csrc/png_device.hip; the
    sink's threads only fetch and write the finished files); 0..9 selects the host encoder at that zlib level
    (smaller files, ~50 ms of a host core per fhd frame at level 1)."""
    os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
    temp_dir = _frames_dir(output_path)
    submitted: List[int] = []
    progress_file = os.path.join(temp_dir, f"progress.json" if world == 1 else f"progress.rank{rank}.json")
    params = {"n_frames": n_frames, "fov": fov, "orbit": orbit, "disk_rotation_speed": disk_rotation_speed,
              "orbit_degrees": orbit_degrees}

    # Resume (render.py:4380-4434).  With several ranks the decision to start over is taken ONCE: every rank looks
    # at the same merged record of all ranks' progress files, only frame files and progress files are removed (never
    # the directory another rank may be creating or writing into), and each rank removes only what belongs to it.
    completed = set()
    os.makedirs(temp_dir, exist_ok=True)
    if resume:
        records = []
        for name in sorted(os.listdir(temp_dir)):
            if name == "progress.json" or (name.startswith("progress.rank") and name.endswith(".json")):
                try:
                    with open(os.path.join(temp_dir, name)) as f:
                        records.append(json.load(f))
                except (OSError, ValueError):
                    pass
        if records and any(r.get("params", {}) != params for r in records):
            print("Warning: parameters changed, starting over")
            for fr in range(rank, n_frames, world):                  # this rank's own frames
                try:
                    os.remove(os.path.join(temp_dir, f"frame_{fr:04d}.png"))
                except OSError:
                    pass
            # every stale record goes -- also the ones a run with another world size left under the other naming
            # (progress.json <-> progress.rank<r>.json), or every later --resume would see the mismatch again and start
            # over for ever.  Rank 0 removes the files no rank of THIS run owns; each rank removes its own.
            mine = os.path.basename(progress_file)
            current = {"progress.json"} if world == 1 else {f"progress.rank{r}.json" for r in range(world)}
            for name in sorted(os.listdir(temp_dir)):
                is_record = name == "progress.json" or (name.startswith("progress.rank") and name.endswith(".json"))
                if is_record and (name == mine or (rank == 0 and name not in current)):
                    try:
                        os.remove(os.path.join(temp_dir, name))
                    except OSError:
                        pass
        elif records:
            done = set()
            for r in records:                                        # a different world size last time: still counted
                done |= set(r.get("completed", []))
            completed = {f for f in done if os.path.isfile(os.path.join(temp_dir, f"frame_{f:04d}.png"))}
            print(f"Resuming: {len(completed)}/{n_frames} frames already rendered")

    total_t0 = time.time()
    rendered = 0
    # the reference saves through a 2-thread PIL pool (render.py:4412-4413); here the frame is quantised
    # on the device, copied into a pinned ring and encoded by worker threads while the next frames render
    if png_level == DEVICE and width > _lib_max_png_width():
        print(f"  frames wider than {_lib_max_png_width()} pixels are PNG-encoded on the host (zlib level {VIDEO_LEVEL})")
        png_level = VIDEO_LEVEL
    if png_level == DEVICE and sink_workers <= 0:
        sink_workers = 4                                 # copy + write only
    sink = FrameSink(renderer, slots=sink_slots, workers=sink_workers, level=png_level)
    if video_stream not in ("auto", "y4m", "off"):
        raise ValueError(f"video_stream must be 'auto', 'y4m' or 'off', got {video_stream!r}")
    stream = encoder = None
    streamable = world == 1 and not completed and width % 2 == 0 and height % 2 == 0
    if streamable and video_stream == "y4m":
        stream = Y4MStream(renderer, os.path.splitext(output_path)[0] + ".y4m", fps)
    elif streamable and video_stream == "auto" and assemble and shutil.which("ffmpeg"):
        import subprocess
        encoder = subprocess.Popen(["ffmpeg", "-y", "-loglevel", "error", "-f", "yuv4mpegpipe", "-i", "-", "-c:v", "libx264",
                                    "-pix_fmt", "yuv420p", output_path], stdin=subprocess.PIPE)
        stream = Y4MStream(renderer, f"/proc/self/fd/{encoder.stdin.fileno()}", fps)

    # the loop's consumers read the quantised rows (PNG sink) and, with a yuv420p stream, the f32 frame: the V pass of every
    # frame stores exactly those (12 bytes per pixel less to write without a stream, 24 with the blur layer nobody reads)
    renderer.set_outputs("u8" if stream is None else "f32+u8")
    n_r, n_phi = renderer.dtex_h, renderer.dtex_w
    factories = init_lifecycle_system(renderer, n_r, n_phi, seed=42)
    dt = disk_rotation_speed
    print(f"  lifecycle system ready (n_r={n_r}, n_phi={n_phi}), rank {rank}/{world}")
    t_loop0 = time.time()                               # ``stats`` (bench.py): the one-off set-up apart from the frame loop

    for frame in range(n_frames):
        t = frame * dt
        mine = frame % world == rank and frame not in completed
        # the factories advance on every frame index on every rank; statistics are a function of the
        # frame index (every 60th), texture composition only happens for frames rendered here
        advance_lifecycle_frame(renderer, factories, t, dt, recompute_stats=(frame % 60 == 0), compose=mine)
        if not mine:
            continue
        cam_pos = orbit_position(static_cam_pos, frame, n_frames, orbit_degrees) if orbit else static_cam_pos
        t0 = time.time()
        renderer.render_async(cam_pos, fov, frame=0)   # lens flare, when enabled, is applied on the device
        sink.submit(os.path.join(temp_dir, f"frame_{frame:04d}.png"))
        if stream is not None:
            try:
                stream.submit()
            except Exception as e:                      # the encoder went away (EPIPE): the PNG frames carry on
                stream, encoder = _drop_stream(stream, encoder, e), None
        elapsed = time.time() - t0
        rendered += 1
        submitted.append(frame)
        if rendered % 50 == 0 or frame >= n_frames - world:
            sink.drain()                                # progress.json only lists frames that are on disk
            completed.update(submitted)
            submitted.clear()
            with open(progress_file, "w") as f:
                json.dump({"params": params, "completed": sorted(completed)}, f)
        if rendered % 100 == 0 or frame == n_frames - 1:
            print(f"  frame {frame}/{n_frames} {elapsed * 1e3:.1f} ms, done {len(completed)}")

    frames_written, bytes_written = sink.drain()
    sink.close()
    if stats is not None:
        stats.update(setup_s=t_loop0 - total_t0, loop_s=time.time() - t_loop0, frames=rendered)
    streamed = False
    if stream is not None:
        try:
            n_streamed, _ = stream.drain()
        except Exception as e:
            stream, encoder = _drop_stream(stream, encoder, e), None
    if stream is not None:
        stream.close()
        if encoder is not None:
            encoder.stdin.close()
            streamed = encoder.wait() == 0 and n_streamed == n_frames
            print(f"Video saved: {output_path} (yuv420p stream, {n_streamed} frames)" if streamed
                  else "ffmpeg failed on the stream; falling back to the PNG frames")
        else:
            print(f"YUV4MPEG2 stream: {os.path.splitext(output_path)[0]}.y4m ({n_streamed} frames)")
    completed.update(submitted)
    with open(progress_file, "w") as f:
        json.dump({"params": params, "completed": sorted(completed)}, f)
    if rendered:
        print(f"Session rendered {rendered} frames in {time.time() - total_t0:.1f} s "
              f"({rendered / (time.time() - total_t0):.1f} fps incl. PNG encode {'on the device' if png_level == DEVICE else f'(zlib level {png_level})'}, "
              f"{bytes_written / max(frames_written, 1) / 1e6:.2f} MB/frame, {sink.workers} encoder threads)")
    if world > 1 or not assemble:
        return       # rank 0 assembles after a barrier (cli.py)
    if len(completed) < n_frames:
        print(f"Warning: only {len(completed)}/{n_frames} frames completed. Run again to resume.")
        return
    if not streamed:
        assemble_video(temp_dir, n_frames, fps, output_path)
