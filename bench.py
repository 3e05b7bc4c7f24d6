#!/usr/bin/env python3
"""bench.py -- Mray-steps/s and frames/s of the hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
under ``python -m torch.distributed.run --nproc-per-node N`` (one rank per GPU).  Rank 0 prints
ONE JSON line.

* step      one frame of the hot path: fused ray march + bloom H/V + final combine
            (TaichiRenderer.render(), render.py:3865-3923) with the scene resident in HBM; successive frames
            alternate between the context's two frame slots (two HIP streams), so the tail and the bloom of
            frame n run under the march of frame n + 1;
* workload  BASELINE.json configs[1]: fhd 1920x1080, default scene (pov 6 0 0.5, fov 90,
            step_size 0.1, disk 2-15, tilt 0, AA off), procedural disk texture + skybox;
* N > 1     the path shards by independent frames (configs[4], frames f % N == rank): every
            rank renders its own fhd frames, no data-path collective => "scaling": "weak";
            the row-block partition of ONE frame (configs[3], 8k in N blocks, halo exchange + gather with
            hipMemcpyPeerAsync) is timed at every N as well and reported as `tile_scaling` (strong scaling);
            `--workload 8k --gpus N` (or --strong) makes that leg the headline;
* value     total ray-steps of all ranks / max-over-ranks wall time of the K timed steps.
* beside it (never as `value`): `kernel_ms` / `roofline` / `roofline_valu` from isolated launches of the same scene,
            `other_math` (the opt-in fast arithmetic), `tile_scaling`, `video_loop` (N = 1: frames of the video driver's
            loop, configs[4], PNG files written) and `cpu_baseline` (N = 1: the CPU restatement on the host's threads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# SURVEY.md 8(d) / BASELINE.md 2: algorithmic cost of the dominant kernel (the march, AA off)
MARCH_BYTES_PER_PIXEL = 110.0     # 24 B framebuffer stores + 0.67*64 B disk texels + 0.90*48 B sky texels
MARCH_FLOP_PER_RAY_STEP = 205.0   # 190 flop + 7 sqrt + 8 div, each counted once
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector

WORKLOADS = {
    "fhd": dict(width=1920, height=1080, cam_pos=[6.0, 0.0, 0.5], fov=90.0, step_size=0.1, anti_alias="disabled",
                disk_tilt=0.0),
    "sd": dict(width=640, height=360, cam_pos=[6.0, 0.0, 0.5], fov=90.0, step_size=0.1, anti_alias="disabled",
               disk_tilt=0.0),
    "4k": dict(width=3840, height=2160, cam_pos=[6.0, 0.0, 0.5], fov=90.0, step_size=0.1, anti_alias="lod_radius",
               disk_tilt=25.0, lens_flare=True),
    "8k": dict(width=7680, height=4320, cam_pos=[6.0, 0.0, 0.5], fov=90.0, step_size=0.05, anti_alias="disabled",
               disk_tilt=0.0),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="fhd", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--persistent", action="store_true", help="A/B: persistent waves + queue refill instead of the tile schedule")
    ap.add_argument("--no-other-math", action="store_true", help="skip the informational A/B leg (profiling)")
    ap.add_argument("--no-config2", action="store_true", help="skip the informational 4k leg (BASELINE.json configs[2])")
    ap.add_argument("--frame-slots", type=int, default=None, choices=[1, 2],
                    help="frames in flight per context (default 2: successive frames overlap on two streams)")
    ap.add_argument("--strong", action="store_true",
                    help="headline = ONE frame of --workload in --gpus row blocks (strong scaling; default for --workload 8k with N > 1)")
    ap.add_argument("--tile-workload", default="8k", choices=sorted(WORKLOADS) + ["none"],
                    help="frame of the row-block (strong-scaling) leg reported beside the headline; 'none' skips it")
    ap.add_argument("--tile-leg-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: run tile_leg on this many devices, print its dict
    ap.add_argument("--tile-tail-tiles", type=int, default=8,
                    help="at N = 1: also time ONE tile of this many alone on the device (row-block leg's single-tile budget); 0 skips it")
    ap.add_argument("--video-frames", type=int, default=1800,
                    help="informational leg at N = 1: frames of the video driver's loop (configs[4]: orbit, lifecycle texture "
                         "every frame, PNG files written); 0 skips it")
    ap.add_argument("--math", default="hybrid", choices=["fast", "strict", "hybrid"],
                    help="march arithmetic of the headline (default hybrid: strict on the tiles whose rays pass near the photon "
                         "sphere, fast elsewhere; certified within the north-star tolerance by tests/test_gpu_hybrid.py).  The other "
                         "two are timed beside it as `other_math`")
    ap.add_argument("--spin-up-ms", type=float, default=300.0,
                    help="un-timed frames rendered for this long BEFORE the --warmup steps: the scene set-up leaves the GPU idle "
                         "and its clocks low; the march reaches its steady duration after ~25 ms of load and the last 3 % after "
                         "~0.3 s (tools/exp_bench_ramp.py; sweep 0 / 60 / 300 / 1000 ms in DESIGN section 5)")
    return ap.parse_args()


def cpu_baseline(wl, sky, tex):
    """Oracle (kind 'port': the reference's Taichi CPU path cannot run here -- taichi is not
    installed) on the host cores, -O3 -ffast-math + OpenMP, one full frame of the same workload."""
    from oracle import oracle as O
    lib = O.load(fast=True)
    cores = int(lib.oracle_num_threads())
    ora = O.OracleRenderer(wl["width"], wl["height"], sky, tex, step_size=wl["step_size"], r_max=10.0,
                           r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=wl["disk_tilt"],
                           anti_alias=wl["anti_alias"], fast=True)
    # the reference integrates the differentials even with AA off (render.py:3866, 4073): time what it does
    t0 = time.perf_counter()
    img, disk = ora.march(wl["cam_pos"], wl["fov"], skip_differentials=False, want_steps=False)
    t_march = time.perf_counter() - t0
    steps = ora.last_total_steps
    t0 = time.perf_counter()
    ora.bloom(disk)
    t_bloom = time.perf_counter() - t0
    t_frame = t_march + t_bloom
    # the same march with the differentials skipped (what the HIP path executes for AA off)
    t0 = time.perf_counter()
    ora.march(wl["cam_pos"], wl["fov"], skip_differentials=True, want_steps=False)
    t_skip = time.perf_counter() - t0
    # one thread on a band of rows through the middle of the frame (SURVEY 8d asks for both figures)
    h = wl["height"]
    band = (h // 2 - h // 32, h // 2 + h // 32)
    lib.oracle_set_num_threads(1)
    t0 = time.perf_counter()
    ora.march(wl["cam_pos"], wl["fov"], skip_differentials=False, want_steps=False, rows=band)
    t_one = time.perf_counter() - t0
    steps_one = ora.last_total_steps
    lib.oracle_set_num_threads(cores)
    return {"value": steps / t_frame / 1e6, "unit": "Mray-steps/s", "cores": cores, "kind": "port",
            "one_thread": {"value": steps_one / t_one / 1e6, "unit": "Mray-steps/s",
                           "sample": f"rows {band[0]}..{band[1]} of the same frame, {steps_one} ray-steps in {t_one:.2f}s"},
            "sample": f"1 full {wl['width']}x{wl['height']} frame (march with differentials as the reference "
                      f"executes it + bloom), {steps} ray-steps in {t_frame:.2f}s",
            "fps": 1.0 / t_frame,
            "march_only_skip_diff_mray_steps_per_s": steps / t_skip / 1e6}


def tile_leg(wl, n, frames, math=None, warmup=3):
    """BASELINE.json configs[3]: ONE frame of ``wl`` cut into n cost-balanced row blocks, block k on device k.  Rank 0
    drives all n devices from this process through bhr_group_render: every tile marches and H-blurs its rows
    concurrently, neighbours exchange the bloom halo rows and every tile pushes its final rows to device 0 with
    hipMemcpyPeerAsync (xGMI, no collective).  Strong scaling: the frame is fixed, n varies."""
    from bhr_amd import _lib, multigpu, workloads
    have = int(_lib.load().bhr_device_count())
    devices = list(range(n))
    if os.environ.get("BHR_TILE_DEVICES"):               # e.g. "0,0": rehearse two tiles on one card
        devices = [int(d) for d in os.environ["BHR_TILE_DEVICES"].split(",")]
    if len(devices) != n or max(devices) >= have:
        return {"skipped": f"this process sees {have} HIP device(s), the leg needs {n} (devices {devices})"}
    tiles, blocks, note = workloads.make_tiles(wl, devices, math=math)
    try:
        t_spin = time.perf_counter()                     # clocks back up after the tiles' scene set-up (un-timed)
        while (time.perf_counter() - t_spin) < 0.3:
            multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer_u8")
        for _ in range(max(warmup, 1)):
            multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer_u8")
        # frames back to back: submitted without a host wait in between (the tiles order themselves on the device: a tile's H
        # pass waits for its neighbours' previous V passes), one wait behind the last -- what a still-image batch or a tiled
        # video would do; `ms_per_frame_each_waited_for` is the same loop with the host waiting after every frame
        t0 = time.perf_counter()
        for _ in range(frames):
            multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer_u8", wait=False)
        multigpu.group_sync(tiles)
        el = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(frames):
            multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer_u8")   # synchronises every tile's stream
        el_sync = time.perf_counter() - t0
        multigpu.group_render(tiles, wl["cam_pos"], wl["fov"], gather="peer_u8", time_march=True)    # un-timed: per-tile march durations
        cs = [t.counters() for t in tiles]
        steps = sum(c["ray_steps"] for c in cs)
        return {"metric": "Mray-steps/s", "value": steps * frames / el / 1e6, "unit": "Mray-steps/s", "scaling": "strong",
                "n_gpus": n, "frames": frames, "ms_per_frame": el / frames * 1e3, "fps": frames / el,
                "ms_per_frame_each_waited_for": el_sync / frames * 1e3,
                "ray_steps_per_frame": int(steps),
                "workload": f"{wl['width']}x{wl['height']} default scene, step_size {wl['step_size']}, one frame in {n} row blocks",
                "row_blocks": [list(b) for b in blocks],
                "tile_ms": {"march": [round(c["march_ms"], 3) for c in cs], "frame": [round(c["frame_ms"], 3) for c in cs]},
                "exchange": "bloom halo rows stored by every tile's H pass straight into its neighbours' planes, quantised u8 rows (what save_image writes) "
                            "stored by its V pass straight into the frame buffer on device 0 (peer-mapped memory over xGMI, no copy stage, no collective)",
                "driven_by": "rank 0 drives all devices in one process (bhr_group_render); the other ranks wait at a host barrier",
                "scene": note}
    finally:
        for t in tiles:
            t.close()


def tile_leg_in_child(workload, n, frames, math, timeout_s=600):
    """tile_leg in a child process: at N > 1 the auxiliary row-block leg opens contexts on every device of the node from
    one process -- code that no run of this repository has yet executed on more than one GPU.  Should it die there (a
    fault is not an exception), the headline of the run survives; the child's dict, or the reason it has none, is
    reported as `tile_scaling`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--tile-leg-child", str(n), "--tile-workload", workload,
           "--steps", str(frames), "--math", math]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                            "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"the row-block leg's child process did not finish within {timeout_s} s"}
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"the row-block leg's child process ended with code {p.returncode}", "stderr_tail": p.stderr.decode(errors="replace")[-600:]}
    return json.loads(lines[-1])


def tile_leg_per_rank(wl, rank, world, local_rank, dist, frames, math=None, warmup=3):
    """The same row-block frame with one PROCESS per tile (multigpu.TileLink: HIP IPC memory handles for the neighbours'
    halo rows and rank 0's frame buffer, shared-memory counters for the pacing): what `--strong` runs when a rank sees
    only its own GPU.  Every rank calls this; rank 0 gets the result, the others None."""
    from bhr_amd import distributed as D, multigpu, workloads
    blocks = workloads.plan_blocks(wl, world, local_rank, math=math)
    import pickle
    blocks = pickle.loads(D.host_all_gather_bytes(pickle.dumps(blocks), dist)[0])        # the cut is refined by timings: every rank takes rank 0's
    tile, dims = workloads.make_tile(wl, blocks[rank], local_rank, math=math)
    shm = f"bhr_tiles_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    link = multigpu.TileLink(tile, rank, world, lambda b: D.host_all_gather_bytes(b, dist), shm, gather="peer_u8")
    try:
        for _ in range(max(warmup, 1) + 20):                         # + clocks back up after the scene set-up (a fixed count: every rank must make the same calls)
            link.render(wl["cam_pos"], wl["fov"])                    # returns on every rank when all rows have landed
        D.host_barrier(dist)
        t0 = time.perf_counter()
        for _ in range(frames):
            link.render(wl["cam_pos"], wl["fov"])
        el = time.perf_counter() - t0
        link.render(wl["cam_pos"], wl["fov"], time_march=True)       # un-timed: per-tile march durations
        c = tile.counters()
        cs = [pickle.loads(b) for b in D.host_all_gather_bytes(pickle.dumps((c["ray_steps"], c["march_ms"], c["frame_ms"], el)), dist)]
        if rank != 0:
            return None
        steps, el = sum(x[0] for x in cs), max(x[3] for x in cs)
        return {"metric": "Mray-steps/s", "value": steps * frames / el / 1e6, "unit": "Mray-steps/s", "scaling": "strong",
                "n_gpus": world, "frames": frames, "ms_per_frame": el / frames * 1e3, "fps": frames / el, "ray_steps_per_frame": int(steps),
                "workload": f"{wl['width']}x{wl['height']} default scene, step_size {wl['step_size']}, one frame in {world} row blocks",
                "row_blocks": [list(b) for b in blocks],
                "tile_ms": {"march": [round(x[1], 3) for x in cs], "frame": [round(x[2], 3) for x in cs]},
                "exchange": "bloom halo rows stored by the H pass into the neighbours' IPC-shared planes, quantised rows stored by the V pass into "
                            "rank 0's IPC-shared frame buffer (peer memory over xGMI, no copy stage, no collective)",
                "driven_by": "one process per tile (bhr_tile_render): pipelined schedule, shared-memory counters between the ranks",
                "scene": f"{world} row blocks {blocks} cut by cost, every rank its own scene copy (lifecycle disk texture {dims[0]}x{dims[1]})"}
    finally:
        link.close()
        tile.close()


def tile_tail_leg(wl, n_tiles=8, math=None, reps=8, verbose=False, schedules=("serial", "pipelined"), gathers=("peer_u8",)):
    """What ONE GPU can show of the N-GPU row-block leg: the frame of ``wl`` is cut into ``n_tiles`` cost-balanced row
    blocks, all on this device; after one full render every tile in turn is rendered ALONE (the other tiles keep their
    buffers: bhr_group_render_subset) and timed end to end -- first march launch .. its rows landed in the gather buffer
    on tile 0, halo pulls from the resting neighbours included -- against its march kernel alone.  On N real GPUs the
    tiles run at the same time, so the frame takes what the slowest tile takes: predicted efficiency = (one-GPU frame
    time) / (n_tiles x slowest tile).  Copies stay on this device (HBM, not xGMI): the 14 MB halo pull and the 12 MB u8
    push would take ~0.09 / ~0.08 ms on a 153 GB/s link, under the march / the next chunk's V pass in the pipelined
    schedule."""
    from bhr_amd import multigpu, workloads
    tiles, blocks, note = workloads.make_tiles(wl, [0] * n_tiles, math=math)
    cam, fov = wl["cam_pos"], wl["fov"]
    out = {"workload": f"{wl['width']}x{wl['height']} step_size {wl['step_size']}, {n_tiles} row blocks on one device",
           "row_blocks": [list(b) for b in blocks], "per_tile": [], "reps": reps}
    try:
        # the scene set-up of the tiles left the GPU idle for seconds: un-timed frames until the clocks are back
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) < 0.3:
            multigpu.group_render(tiles, cam, fov, gather=gathers[0], schedule=schedules[0])
        for sched in schedules:
            for g in gathers:
                for _ in range(2):
                    multigpu.group_render(tiles, cam, fov, gather=g, schedule=sched)
                t0 = time.perf_counter()
                for _ in range(max(reps // 2, 2)):
                    multigpu.group_render(tiles, cam, fov, gather=g, schedule=sched)
                out[f"all_tiles_one_device_ms_{sched}_{g}"] = (time.perf_counter() - t0) / max(reps // 2, 2) * 1e3
        # Per tile, INTERLEAVED: its march alone (own kernel bracket, nothing else on the device, no post-pass: the march kernel
        # then skips the packed copy it writes for the H pass) and the tile end to end under every schedule / gather, one of
        # each per repetition -- the chip's clock drifts by a few per cent over the seconds this leg runs, and timing all the
        # marches first made the later tiles' tails look 0.07 ms longer than they are.
        for k, t in enumerate(tiles):
            live = [1 if q == k else 0 for q in range(n_tiles)]
            combos = [(sched, g) for sched in schedules for g in gathers]
            for sched, g in combos:
                for _ in range(2):
                    multigpu.group_render(tiles, cam, fov, gather=g, schedule=sched, live=live)
            for _ in range(2):
                t.render_async(cam, fov, skip_bloom=True)
            alone, e2e, wall = [], {c: [] for c in combos}, {c: [] for c in combos}
            for _ in range(reps):
                t.render_async(cam, fov, skip_bloom=True)
                alone.append(t.counters()["march_ms"])
                for sched, g in combos:
                    t0 = time.perf_counter()
                    multigpu.group_render(tiles, cam, fov, gather=g, schedule=sched, live=live)
                    wall[(sched, g)].append((time.perf_counter() - t0) * 1e3)
                    e2e[(sched, g)].append(t.counters()["frame_ms"])
            row = {"tile": k, "rows": list(blocks[k]), "march_alone_ms": float(np.median(alone))}
            for sched, g in combos:
                row[f"e2e_ms_{sched}_{g}"] = float(np.median(e2e[(sched, g)]))
                row[f"wall_ms_{sched}_{g}"] = float(np.median(wall[(sched, g)]))
            # the march INSIDE the tile's frame (BHR_GROUP_TIME_MARCH brackets it with two more event records)
            for _ in range(2):
                multigpu.group_render(tiles, cam, fov, gather=gathers[0], schedule=schedules[0], live=live, time_march=True)
            row["march_in_frame_ms"] = t.counters()["march_ms"]
            out["per_tile"].append(row)
            if verbose:
                print(row, flush=True)
        key = f"e2e_ms_{schedules[0]}_{gathers[0]}"
        slow = max(out["per_tile"], key=lambda r: r[key])
        out["schedule"], out["gather"] = schedules[0], gathers[0]
        out["slowest_tile"] = slow["tile"]
        out["tile_ms"] = slow[key]
        out["tile_march_alone_ms"] = slow["march_alone_ms"]
        out["tile_tail_ms"] = slow[key] - slow["march_alone_ms"]
        out["tile_tail_frac"] = out["tile_tail_ms"] / slow[key]
        # what a tile's kernels store into peer memory on N devices (local HBM here): the rows each neighbour's V pass reaches
        # (bloom radius padded to 16-tap chunks, both f16 halves of three channels), and its own quantised rows
        W, R = wl["width"], int(wl["width"] * 0.02)
        rows = max(b[1] - b[0] for b in blocks)
        reach = 16 * ((R + 15) // 16) + 32
        out["peer_store_mb"] = {"halo_rows_per_neighbour": reach * W * 12 / 1e6, "u8_rows": rows * W * 3 / 1e6}
        out["scene"] = note
        return out
    finally:
        for t in tiles:
            t.close()


def video_leg(n_frames, math=None):
    """BASELINE.json configs[4] on this GPU: drivers.render_video at fhd (orbit camera, populations ticking, texture
    regenerated every frame, PNG frames encoded on the device and written to a temporary directory).  Informational:
    reported beside the headline, never as `value`."""
    import shutil, tempfile
    from bhr_amd import drivers
    tmp = tempfile.mkdtemp(prefix="bhr_bench_video_")
    try:
        r, _, _, _ = drivers.make_renderer(1920, 1080, [6, 0, 0.5], 90, n_stars=6000, math=math)
        t0 = time.perf_counter()
        st = {}
        drivers.render_video(r, 1920, 1080, n_frames=n_frames, fps=30, output_path=os.path.join(tmp, "v.mp4"), fov=90,
                             static_cam_pos=[6, 0, 0.5], orbit=True, assemble=False, video_stream="off", stats=st)
        dt = time.perf_counter() - t0
        files = [f for f in os.listdir(drivers._frames_dir(os.path.join(tmp, "v.mp4"))) if f.endswith(".png")]
        size = sum(os.path.getsize(os.path.join(drivers._frames_dir(os.path.join(tmp, "v.mp4")), f)) for f in files)
        r.close()
        return {"fps": n_frames / st["loop_s"], "fps_incl_setup": n_frames / dt, "setup_s": st["setup_s"], "loop_s": st["loop_s"],
                "frames": n_frames, "png_files": len(files), "mb_per_frame": size / max(len(files), 1) / 1e6,
                "png_encoder": "device", "march_math": r.math if hasattr(r, "math") else math,
                "what": "render_video at 1920x1080: every frame's populations, background / entity / compose passes, march, bloom "
                        "(u8 rows only), PNG file on disk; fps over the frame loop, the one-off resume scan + lifecycle init in setup_s"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config2_leg(math=None, frames=40):
    """BASELINE.json configs[2] on this GPU: 3840x2160, disk tilt 25 deg, --anti_alias lod_radius (the march integrates the two
    variational RK4s and samples the mip stack, render.py:2888-2911, 2961-2990), lens flare applied on the device inside the
    step (render.py:3925-4028).  fps with two frames in flight; kernel times from isolated launches (one frame slot)."""
    from bhr_amd import workloads
    wl = WORKLOADS["4k"]
    out = {"workload": "4k 3840x2160 default scene, disk_tilt 25, anti_alias lod_radius, lens_flare on (BASELINE.json configs[2])", "march_math": math}
    for slots in (1, 2):
        r, _, _, note = workloads.make_scene(wl, math=math, frame_slots=slots)
        try:
            t_spin = time.perf_counter()
            while (time.perf_counter() - t_spin) < 0.3:
                for _ in range(4):
                    r.render_async(wl["cam_pos"], wl["fov"], lens_flare=True)
                r.sync()
            r.timing_reset()
            r.sync()
            t0 = time.perf_counter()
            for _ in range(frames):
                r.render_async(wl["cam_pos"], wl["fov"], lens_flare=True)
            r.sync()
            el = time.perf_counter() - t0
            c = r.counters()
            if slots == 1:
                march_ms = c["march_ms_sum"] / max(c["frames_timed"], 1)
                steps = c["ray_steps_sum"] / max(c["frames_timed"], 1)
                out["kernel_ms"] = {"march": march_ms, "bloom_combine_flare": c["bloom_ms_sum"] / max(c["frames_timed"], 1),
                                    "frames_timed": c["frames_timed"], "how": "HIP events, isolated launches (one frame slot)"}
                out["fps_one_frame_at_a_time"] = frames / el
                out["ray_steps_per_frame"] = int(steps)
                out["scene"] = note
                # 625 flop per ray-step with the differentials as the reference writes them (three RK4s), ~330 with the
                # stage values shared between the main and the variational right-hand sides (what both builds execute)
                out["roofline_valu"] = {"bound": "valu_fp32", "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "as_the_reference_writes_it": {"flop_per_ray_step": 625.0, "achieved": 625.0 * steps / (march_ms * 1e-3) / 1e12,
                                                                       "frac": 625.0 * steps / (march_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS},
                                        "with_stage_reuse": {"flop_per_ray_step": 330.0, "achieved": 330.0 * steps / (march_ms * 1e-3) / 1e12,
                                                             "frac": 330.0 * steps / (march_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS}}
                if r.math == "hybrid":
                    hi = r.hybrid_info()
                    out["hybrid"] = {"strict_tiles": hi["strict_tiles"], "tiles": hi["tiles"], "repaired_pixels": hi["repaired_pixels"]}
            else:
                out["fps"] = frames / el
                out["ms_per_frame"] = el / frames * 1e3
                out["value"] = c["ray_steps_sum"] / max(c["frames_timed"], 1) * frames / el / 1e6
                out["unit"] = "Mray-steps/s"
        finally:
            r.close()
    return out


def main():
    args = parse()
    # Only the JSON line may reach stdout: RCCL prints a version banner there when NCCL_DEBUG is set, the scene
    # set-up prints progress.  Everything written to fd 1 before the result goes to stderr instead.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.tile_leg_child > 0:                          # internal (tile_leg_in_child): no process group, one JSON dict
        t = tile_leg(WORKLOADS[args.tile_workload], args.tile_leg_child, args.steps, math=args.math)
        os.write(real_stdout, (json.dumps(t) + "\n").encode())
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from bhr_amd import distributed as D
    # backend "nccl" is RCCL on ROCm.  BHR_DIST_BACKEND=gloo + BHR_FORCE_DEVICE=0 rehearse the N > 1 code
    # path with several ranks sharing one card (RCCL refuses duplicate GPUs).
    backend = os.environ.get("BHR_DIST_BACKEND", "nccl")
    if "BHR_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["BHR_FORCE_DEVICE"])
    else:
        local_rank = D.visible_device(local_rank)        # a launcher that shows every rank only its own GPU: ordinal 0
    # BHR_DIST_FORCE=1: initialise the process group even for one rank (rehearses RCCL init + all-reduce)
    dist = D.init(backend, local_rank) if (world > 1 or os.environ.get("BHR_DIST_FORCE") == "1") else None
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from bhr_amd import workloads
    wl = WORKLOADS[args.workload]
    if args.strong or (world > 1 and args.workload == "8k"):
        # configs[3] as the headline: the frame is fixed, the devices share it in row blocks
        D.host_barrier(dist)
        # one process drives all devices where it can see them; otherwise every rank renders its own tile
        from bhr_amd import _lib as L
        have = int(L.load().bhr_device_count())
        per_rank = world > 1 and (have < world or os.environ.get("BHR_STRONG_PER_RANK") == "1") and not os.environ.get("BHR_TILE_DEVICES")
        t = None
        if per_rank:
            t = tile_leg_per_rank(wl, rank, world, local_rank, dist, args.steps, math=args.math, warmup=args.warmup)
        elif rank == 0:
            t = tile_leg(wl, world, args.steps, math=args.math, warmup=args.warmup)
            if "skipped" in t:
                raise SystemExit("strong-scaling leg: " + t["skipped"])
        if rank == 0:
            slow = int(np.argmax(t["tile_ms"]["march"]))
            px = (t["row_blocks"][slow][1] - t["row_blocks"][slow][0]) * wl["width"]
            gbs = MARCH_BYTES_PER_PIXEL * px / (t["tile_ms"]["march"][slow] * 1e-3) / 1e9
            out = {"metric": "Mray-steps/s", "value": t["value"], "unit": "Mray-steps/s", "fps": t["fps"], "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": t["ms_per_frame"], "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": args.workload + " " + t["workload"], "scene": t["scene"], "row_blocks": t["row_blocks"],
                              "exchange": t["exchange"], "driven_by": t["driven_by"],
                              "ray_steps_per_frame": t["ray_steps_per_frame"]},
                   "kernel_ms": {"march_per_tile": t["tile_ms"]["march"], "frame_per_tile": t["tile_ms"]["frame"]},
                   "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "traffic": None, "kernel": f"march of the slowest tile (tile {slow})",
                                "algorithmic_bytes_per_launch": MARCH_BYTES_PER_PIXEL * px}}
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        D.host_barrier(dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    renderer, sky, tex, scene_note = workloads.make_scene(wl, device_index=local_rank, math=args.math,
                                                          frame_slots=args.frame_slots)

    stamps = {}

    def barrier(tag=None):
        ts = [time.perf_counter()]
        if dist is not None:
            dist.barrier()
        ts.append(time.perf_counter())
        renderer.sync()                 # the renderer's own streams (a stream wait + hipStreamSynchronize) ...
        ts.append(time.perf_counter())
        torch.cuda.synchronize()        # ... then torch's device-wide synchronise
        ts.append(time.perf_counter())
        if tag:
            stamps[tag] = ts

    # frames sharded round-robin over ranks (configs[4]); the camera is the static default pov
    compaction = args.persistent
    flare = bool(wl.get("lens_flare", False))       # configs[2]: on the device, inside the timed step
    # torch initialises its HIP context at the first torch.cuda call (hundreds of ms with the GPU idle): have that behind
    # us before the spin-up, not inside the barrier in front of the timed region (it cost the 20-step run of round 2 and
    # the first hybrid runs of this round 10-13 %: the timed frames started on a chip that had just idled)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    # Spin-up (un-timed, before the warm-up steps): after the scene set-up -- hundreds of ms of host work with an idle GPU
    # -- the shader clock is low and needs ~20 ms of load to come back; 5 warm-up frames are 4 ms.  Measured on this
    # scene (tools/exp_bench_ramp.py, profiles/r03_bench_ramp.md): the march takes 0.77 ms in the first timed frame of a
    # 20-step run straight after set-up, 0.74 in the tenth, 0.675 from ~25 ms on.
    n_spin = 0
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spin_up_ms:
        for _ in range(8):
            renderer.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
        renderer.sync()
        n_spin += 8
    for _ in range(args.warmup):
        renderer.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
    renderer.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        renderer.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
    barrier("close")
    elapsed = time.perf_counter() - t0
    tc = stamps["close"]
    region_ms = {"host_submit": (tc[0] - t0) * 1e3, "dist_barrier": (tc[1] - tc[0]) * 1e3,
                 "renderer_sync": (tc[2] - tc[1]) * 1e3, "torch_synchronize": (tc[3] - tc[2]) * 1e3}

    c = renderer.counters()
    steps_per_frame = c["ray_steps"]
    hybrid_note = None
    if renderer.math == "hybrid":
        try:
            hi = renderer.hybrid_info()
            hybrid_note = (f"strict on {hi['strict_tiles']} of {hi['tiles']} 8x8 tiles (impact parameter within "
                           f"[b_c - {hi['band_below']:g}, b_c + {hi['band_above']:g}] r_s of the photon sphere's 3 sqrt(3)/2), fast on the rest"
                           + (f"; guards on, {hi['repaired_pixels']} pixels marched again strict" if hi.get("repair_capacity") else ""))
        except Exception:      # no hybrid march has run on this context (a Disk V2 source or the persistent schedule: strict)
            hybrid_note = "no hybrid march on this context: this view ran strict (Disk V2 source or persistent schedule)"

    # same scene, same process, the other arithmetics: reported beside the headline, never as `value`
    n_other = 0 if args.no_other_math else max(args.steps // 4, 10)
    others = []
    for other in [m for m in ("strict", "hybrid", "fast") if m != renderer.math] if n_other else []:
        for _ in range(3):
            renderer.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, math=other, lens_flare=flare)
        renderer.timing_reset()
        renderer.sync()
        t1 = time.perf_counter()
        for _ in range(n_other):
            renderer.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, math=other, lens_flare=flare)
        renderer.sync()
        el_other = time.perf_counter() - t1
        co = renderer.counters()
        others.append({"math": other, "value": co["ray_steps"] * n_other / el_other / 1e6, "unit": "Mray-steps/s", "fps": n_other / el_other})
    # Per-kernel durations of ISOLATED launches (one frame slot, one stream): in the timed region above two frames
    # are in flight, their kernels overlap and a launch's own event bracket also covers time it shares with the
    # other frame's kernels.  The roofline below describes the kernel, so it is taken from launches that have the
    # chip to themselves: the same scene and view on a second context created with one frame slot, outside the
    # timed region.  (`rocprofv3 --kernel-trace` of `bench.py --frame-slots 1` is the matching profile.)
    iso = None
    if renderer.frame_slots != 1 and rank == 0:
        solo, _, _, _ = workloads.make_scene(wl, device_index=local_rank, math=args.math, frame_slots=1)
        n_iso = max(min(args.steps, 100), 10)
        t_spin = time.perf_counter()                   # the second scene set-up idled the chip again: spin up as above
        while (time.perf_counter() - t_spin) * 1e3 < args.spin_up_ms:
            for _ in range(8):
                solo.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
            solo.sync()
        for _ in range(10):
            solo.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
        solo.timing_reset()
        solo.sync()
        t2 = time.perf_counter()
        for _ in range(n_iso):
            solo.render_async(wl["cam_pos"], wl["fov"], compaction=compaction, lens_flare=flare)
        solo.sync()
        el_iso = time.perf_counter() - t2
        ci = solo.counters()
        iso = {"march_ms": ci["march_ms_sum"] / max(ci["frames_timed"], 1), "post_ms": ci["bloom_ms_sum"] / max(ci["frames_timed"], 1),
               "frames": ci["frames_timed"], "fps": n_iso / el_iso, "ray_steps": ci["ray_steps_sum"] / max(ci["frames_timed"], 1)}
        solo.close()
    # MAX over ranks of the time, SUM over ranks of the ray-steps each rank marched in the timed region
    elapsed, total_steps = D.aggregate_throughput(elapsed, float(c["ray_steps_sum"]) if c["frames_timed"] == args.steps
                                                  else float(steps_per_frame) * args.steps, dist, device=red_dev)

    # the row-block leg (configs[3]): reported beside the headline at every N, so that the per-N bench lines carry the
    # tile-scaling curve as well as the frame-sharding one.  Every rank's GPU is idle while rank 0 times it.
    tile = None
    if args.tile_workload != "none" and not args.no_other_math:
        D.host_barrier(dist)
        from bhr_amd import _lib as L
        have = int(L.load().bhr_device_count())
        if world > 1 and (have < world or os.environ.get("BHR_STRONG_PER_RANK") == "1") and not os.environ.get("BHR_TILE_DEVICES"):
            # every rank sees only its own GPU: one process per tile (a failure on one rank leaves the others waiting at
            # most for the library's own time-out)
            try:
                tile = tile_leg_per_rank(WORKLOADS[args.tile_workload], rank, world, local_rank, dist, max(args.steps // 10, 10), math=args.math)
            except Exception as e:
                tile = {"error": f"{type(e).__name__}: {e}"}
        elif rank == 0:
            try:
                if world > 1 and not os.environ.get("BHR_TILE_DEVICES"):
                    tile = tile_leg_in_child(args.tile_workload, world, max(args.steps // 10, 10), args.math)
                else:
                    tile = tile_leg(WORKLOADS[args.tile_workload], world, max(args.steps // 10, 10), math=args.math)
            except Exception as e:      # the headline stands on its own
                tile = {"error": f"{type(e).__name__}: {e}"}
        D.host_barrier(dist)

    if rank == 0:
        n_frames = c["frames_timed"]
        overlapped = {"march": c["march_ms_sum"] / max(n_frames, 1), "post": c["bloom_ms_sum"] / max(n_frames, 1),
                      "frames_timed": n_frames}
        if iso is not None:
            march_ms, bloom_ms, k_frames, k_steps = iso["march_ms"], iso["post_ms"], iso["frames"], iso["ray_steps"]
        else:   # one frame slot: the timed region's own launches are isolated
            march_ms, bloom_ms, k_frames = overlapped["march"], overlapped["post"], n_frames
            k_steps = c["ray_steps_sum"] / max(n_frames, 1)
        pixels = wl["width"] * wl["height"]
        alg_bytes = MARCH_BYTES_PER_PIXEL * pixels
        achieved_gbs = alg_bytes / (march_ms * 1e-3) / 1e9
        traffic = traffic_source = executed_flop = None
        tpath = os.path.join(ROOT, "profiles", "march_traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.workload}/{renderer.math}"
                traffic = tj.get(key, tj.get(args.workload) if renderer.math == "strict" else None)
                traffic_source = (tj.get("source", "") + f"; key {key}: " + json.dumps(tj.get(f"_{key}_detail", tj.get(f"_{args.workload}_detail")))) if traffic else None
                det = tj.get(f"_{key}_detail") or {}
                executed_flop = det.get("executed_fp32_flop_per_frame")
            except Exception:
                traffic = None
        valu_tflops = MARCH_FLOP_PER_RAY_STEP * k_steps / (march_ms * 1e-3) / 1e12
        try:
            stream_map = renderer.stream_map()      # after the timed region: which of the context's streams share a hardware queue
        except Exception as e:
            stream_map = {"error": str(e)}
        stream_cal = renderer.stream_calibration()
        out = {
            "metric": "Mray-steps/s", "value": total_steps / elapsed / 1e6, "unit": "Mray-steps/s",
            "fps": world * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload} {wl['width']}x{wl['height']} default scene, pov 6 0 0.5, "
                                   f"fov {wl['fov']:g}, step_size {wl['step_size']}, anti_alias {wl['anti_alias']}, "
                                   f"disk_tilt {wl['disk_tilt']:g}, lens_flare {'on' if flare else 'off'}",
                       "scene": scene_note, "frames_per_rank": args.steps,
                       "sharding": "independent frames per rank, no collective",
                       "march_schedule": "persistent+refill" if args.persistent else "tile",
                       "march_math": renderer.math, "march_math_note": hybrid_note, "frame_slots": renderer.frame_slots,
                       "spin_up": f"{n_spin} un-timed frames ({args.spin_up_ms:g} ms) before the {args.warmup} warm-up steps",

                       "ray_steps_per_frame": int(steps_per_frame), "steps_per_ray": steps_per_frame / pixels},
            "stream_map": stream_map, "stream_calibration": stream_cal,
            "kernel_ms": {"march": march_ms, "bloom_combine_flare" if flare else "bloom_and_combine": bloom_ms, "frames_timed": k_frames,
                          "march_vgprs": c["march_vgprs"],
                          "how": "HIP events on the launching stream, isolated launches (one frame slot)"
                                 + ("" if iso is None else f", {iso['fps']:.0f} fps one frame at a time")},
            "kernel_ms_in_timed_region": dict(overlapped, march_busy_ms=c["march_busy_ms"], span_ms=c["span_ms"],
                                              note="event brackets of overlapping launches (two frame slots): each covers time shared "
                                                   "with the other frame's kernels; march_busy_ms = union of the march intervals"),
            "timed_region_ms": dict(region_ms, total=elapsed * 1e3,
                                    note="host stamps: the K submissions, then the closing bracket's parts in the order they run"),
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "march", "algorithmic_bytes_per_launch": alg_bytes},
            # the same two ratios over the TIMED REGION itself: all march launches of the region / the time during which
            # at least one of them was running (union of their event brackets; launches of successive frames overlap)
            "roofline_in_timed_region": {
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": alg_bytes * n_frames / (c["march_busy_ms"] * 1e-3) / 1e9 if c["march_busy_ms"] > 0 else None,
                "frac": alg_bytes * n_frames / (c["march_busy_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if c["march_busy_ms"] > 0 else None,
                "valu_tflops": MARCH_FLOP_PER_RAY_STEP * c["ray_steps_sum"] / (c["march_busy_ms"] * 1e-3) / 1e12 if c["march_busy_ms"] > 0 else None,
                "valu_frac": MARCH_FLOP_PER_RAY_STEP * c["ray_steps_sum"] / (c["march_busy_ms"] * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS if c["march_busy_ms"] > 0 else None,
                "march_launches": n_frames, "march_busy_ms": c["march_busy_ms"]},
            # the march is not HBM bound: the governing ceiling is non-matrix FP32
            "roofline_valu": {"bound": "valu_fp32", "achieved": valu_tflops, "peak": VALU_FP32_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": valu_tflops / VALU_FP32_PEAK_TFLOPS,
                              "flop_per_ray_step": MARCH_FLOP_PER_RAY_STEP,
                              # what the kernels EXECUTE (the 205 above is the reference's 3-D formulation; the fast list marches in
                              # the 2-D orbital plane): 64 x (2 FMA + MUL + ADD wave-instructions) of the march kernels, from the
                              # PMC pass of the profile named in roofline.traffic_source, over this run's march time
                              "executed_flop_per_frame": executed_flop,
                              "executed_tflops": (executed_flop / (march_ms * 1e-3) / 1e12) if executed_flop else None,
                              "executed_frac": (executed_flop / (march_ms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS) if executed_flop else None},
        }
        if others:
            out["other_math"] = {"runs": others,
                                 "note": "informational A/B in the same process and scene; `value` above is --math " + renderer.math}
        if tile is not None:
            out["tile_scaling"] = tile
            # what ONE GPU can show of the 8-GPU leg: one tile of eight, alone on the device, timed end to end
            if world == 1 and "ms_per_frame" in tile and args.tile_tail_tiles > 1 and not args.no_other_math:
                try:
                    nt = args.tile_tail_tiles
                    tt = tile_tail_leg(WORKLOADS[args.tile_workload], nt, math=args.math, reps=6)
                    per = {sch: [round(r[f"e2e_ms_{sch}_peer_u8"], 3) for r in tt["per_tile"]] for sch in ("serial", "pipelined")}
                    tile["one_tile_alone"] = {"workload": tt["workload"], "row_blocks": tt["row_blocks"], "per_tile_ms": per,
                                              "per_tile_march_alone_ms": [round(r["march_alone_ms"], 3) for r in tt["per_tile"]],
                                              "slowest_tile": tt["slowest_tile"], "tile_ms": tt["tile_ms"], "tile_march_alone_ms": tt["tile_march_alone_ms"],
                                              "tile_tail_ms": tt["tile_tail_ms"], "tile_tail_frac": tt["tile_tail_frac"], "schedule": tt["schedule"],
                                              "peer_store_mb": tt["peer_store_mb"]}
                    tile["tile_tail_ms"] = tt["tile_tail_ms"]
                    # no copy stages any more: the H pass stores its halo rows into the neighbours' planes, the V pass its u8 rows into
                    # the frame buffer on tile 0's device -- on N devices those stores cross xGMI inside the kernels (14 + 12 MB per
                    # 8k tile), which one device cannot show
                    tile[f"predicted_efficiency_at_{nt}_gpus"] = {
                        "serial_schedule": tile["ms_per_frame"] / (nt * max(per["serial"])),
                        "pipelined_schedule": tile["ms_per_frame"] / (nt * max(per["pipelined"])),
                        "note": "PREDICTED, unmeasured on more than one device: one-GPU frame time / (tiles x slowest tile alone, first march launch "
                                ".. its V pass done, halo rows and frame rows stored by the kernels themselves); on this one device the "
                                "neighbours' planes and the frame buffer are local HBM, on N they are peer memory over xGMI"}
                except Exception as e:
                    tile["one_tile_alone"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_other_math and not args.no_config2:
            try:
                out["config2_4k"] = config2_leg(math=args.math)
            except Exception as e:      # the headline stands on its own
                out["config2_4k"] = {"error": f"{type(e).__name__}: {e}"}
        if args.video_frames > 0 and world == 1 and not args.no_other_math:
            try:
                out["video_loop"] = video_leg(args.video_frames, math=args.math)
            except Exception as e:      # the headline stands on its own
                out["video_loop"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, sky, tex)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    renderer.close()


if __name__ == "__main__":
    main()
