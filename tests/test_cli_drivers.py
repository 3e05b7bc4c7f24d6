"""CLI surface and drivers.  CPU part: flag names/defaults/validation of the reference
(render.py:4518-4616, tests/unit/test_orbit_degrees.py) and the multi-GPU partition helpers.
GPU part: render_image on the e2e scene, a short video with resume and frame sharding."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_defaults_match_reference():
    from bhr_amd import cli
    a = cli.parse_args([])
    assert a.pov == [6, 0, 0.5] and a.fov == 90 and a.resolution == "fhd"
    assert a.output == "output/blackhole.png" and a.step_size == 0.1 and a.r_max == 10 and a.n_stars == 6000
    assert a.disk_inner_radius == 2.0 and a.disk_outer_radius == 15.0 and a.disk_tilt == 0.0
    assert a.anti_alias == "disabled" and a.aa_strength == 1.0 and not a.lens_flare
    assert a.n_frames == 3600 and a.fps == 36 and a.orbit_degrees == 360.0 and a.disk_rotation_speed == 0.1
    assert a.device == "hip" and a.gpus == 1
    b = cli.parse_args(["--ar1", "3", "--ar2", "9", "-r", "8k", "-s", "0.05", "--anti_alias", "lod_radius",
                        "--video", "--orbit", "--n_frames", "12", "-o", "x/y.mp4", "--pov", "1", "2", "3"])
    assert (b.disk_inner_radius, b.disk_outer_radius, b.resolution, b.step_size) == (3.0, 9.0, "8k", 0.05)
    assert b.video and b.orbit and b.n_frames == 12 and b.pov == [1.0, 2.0, 3.0]
    assert cli.RESOLUTIONS["8k"] == (7680, 4320) and cli.RESOLUTIONS["fhd"] == (1920, 1080)
    # additions of this build: where the video frames are PNG-encoded, and the yuv420p stream
    assert a.png_encoder == "device" and a.video_stream == "auto"
    assert cli.parse_args(["--png_encoder", "host", "--video_stream", "y4m"]).png_encoder == "host"
    with pytest.raises(SystemExit):
        cli.parse_args(["--png_encoder", "zlib"])


@pytest.mark.parametrize("argv,msg", [
    (["--fov", "0"], "FOV"), (["--fov", "180"], "FOV"), (["--ar1", "5", "--ar2", "5"], "disk_inner_radius"),
    (["-s", "0"], "step_size"), (["--aa_strength", "0.4"], "aa_strength"), (["--aa_strength", "2.1"], "aa_strength"),
    (["--n_frames", "0"], "n_frames"), (["--fps", "0"], "fps"), (["--orbit_degrees", "inf"], "orbit_degrees"),
    (["--orbit_degrees", "nan"], "orbit_degrees"), (["--disk_texture", "x.png", "--video"], "disk_texture"),
    (["--device", "cpu"], "no CPU path"), (["--gpus", "0"], "gpus"), (["--interactive"], "interactive"),
])
def test_cli_validation_errors(argv, msg):
    from bhr_amd import cli
    with pytest.raises(ValueError, match=msg):
        cli.validate_args(cli.parse_args(argv))


def test_cli_negative_orbit_degrees_is_valid():
    from bhr_amd import cli
    cli.validate_args(cli.parse_args(["--orbit_degrees", "-90"]))


def test_row_blocks_and_frame_shards():
    from bhr_amd.multigpu import frames_of_rank, row_blocks
    assert row_blocks(4320, 8) == [(540 * k, 540 * (k + 1)) for k in range(8)]
    b = row_blocks(1080, 7)
    assert b[0][0] == 0 and b[-1][1] == 1080 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
    with pytest.raises(ValueError):
        row_blocks(4, 8)
    frames = sorted(f for r in range(8) for f in frames_of_rank(3600, r, 8))
    assert frames == list(range(3600))
    assert list(frames_of_rank(10, 3, 4)) == [3, 7]


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_render_image_e2e_scene(oracle, tmp_path):
    """tests/e2e_render.py:25-43 scene end to end (lifecycle texture + skybox + march + bloom) against
    the oracle fed with the same device-generated texture."""
    from bhr_amd import HipRenderer, drivers
    from bhr_amd.skybox import generate_skybox
    from bhr_amd.textures import compute_disk_texture_resolution
    kw = dict(step_size=0.1, r_max=10, r_disk_inner=2.0, r_disk_outer=3.5, disk_tilt=15, anti_alias="disabled")
    img = drivers.render_image(320, 180, [6, 0, 0.5], 60, n_stars=100, lens_flare=False, **kw)
    assert img.shape == (180, 320, 3) and img.dtype == np.float32
    # same pipeline by hand, then the oracle on the resulting texture
    n_phi, n_r = compute_disk_texture_resolution(320, 180, [6, 0, 0.5], 60, 2.0, 3.5)
    assert (n_r, n_phi) == (128, 336)
    sky = generate_skybox(2048, 1024, seed=42, n_stars=100)
    r = HipRenderer(320, 180, sky, np.zeros((n_r, n_phi, 4), dtype=np.float32), **kw)
    fac = drivers.init_lifecycle_system(r, n_r, n_phi, seed=42)
    drivers.advance_lifecycle_frame(r, fac, t=0.0, dt=0.0, recompute_stats=True)
    tex = r.disk_texture_field.to_numpy()
    np.testing.assert_array_equal(r.render([6, 0, 0.5], 60), img)          # deterministic end to end
    ref = oracle.OracleRenderer(320, 180, sky, tex, **kw).render([6, 0, 0.5], 60)
    rmse = np.sqrt(np.mean((img.astype(np.float64) - ref) ** 2, axis=(0, 1)))
    assert (rmse <= 5e-6).all(), rmse
    drivers.save_image(img, str(tmp_path / "o" / "e2e.png"))
    from PIL import Image
    png = np.array(Image.open(tmp_path / "o" / "e2e.png"))
    np.testing.assert_array_equal(png, (np.clip(img, 0, 1) * 255).astype(np.uint8))
    r.close()


@pytest.mark.gpu
def test_lens_flare_and_aa_image(oracle):
    from bhr_amd import HipRenderer, scenes
    from oracle.flare_np import apply_lens_flare
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    kw = dict(step_size=0.1, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=25.0, anti_alias="lod_radius")
    r = HipRenderer(256, 144, sky, tex, lens_flare=True, **kw)
    img = r.render([6, 0, 0.5], 90)
    o = oracle.OracleRenderer(256, 144, sky, tex, **kw)
    ref, _, rdisk, _ = o.render([6, 0, 0.5], 90, parts=True)
    want = apply_lens_flare(ref, np.ascontiguousarray(rdisk.transpose(1, 0, 2)))
    assert np.abs(img - want).max() < 2e-4 and np.abs(want - ref).max() > 0.01   # the flare is really there
    r.close()


@pytest.mark.gpu
def test_video_frames_resume_and_sharding(tmp_path):
    """render_video: PNG frames, progress.json format (render.py:4380-4403, 4469-4472), resume skips
    finished frames, and 2-way frame sharding produces the same frames as one rank."""
    from PIL import Image
    from bhr_amd import drivers

    def run(out, n_frames=6, rank=0, world=1, resume=False):
        r, _, _, _ = drivers.make_renderer(160, 90, [6, 0, 0.5], 90, n_stars=50, tex_w=256, tex_h=128)
        drivers.render_video(r, 160, 90, n_frames=n_frames, fps=4, output_path=out, fov=90, static_cam_pos=[6, 0, 0.5],
                             orbit=True, resume=resume, disk_rotation_speed=0.1, orbit_degrees=90.0, rank=rank,
                             world=world, assemble=False)
        r.close()
        return drivers._frames_dir(out)

    out1 = str(tmp_path / "a" / "v.mp4")
    d1 = run(out1)
    frames1 = [np.array(Image.open(os.path.join(d1, f"frame_{k:04d}.png"))) for k in range(6)]
    assert frames1[0].shape == (90, 160, 3) and frames1[0].max() > 10
    assert any((frames1[0] != frames1[5]).ravel())                          # the camera moved
    prog = json.load(open(os.path.join(d1, "progress.json")))
    assert sorted(prog["completed"]) == list(range(6))
    assert prog["params"] == {"n_frames": 6, "fov": 90, "orbit": True, "disk_rotation_speed": 0.1, "orbit_degrees": 90.0}

    # resume: drop two frames from the progress file, re-run, only those are re-rendered -- identically
    os.remove(os.path.join(d1, "frame_0003.png"))
    os.remove(os.path.join(d1, "frame_0005.png"))
    json.dump({"params": prog["params"], "completed": [0, 1, 2, 4]}, open(os.path.join(d1, "progress.json"), "w"))
    mtime = os.path.getmtime(os.path.join(d1, "frame_0001.png"))
    run(out1, resume=True)
    assert os.path.getmtime(os.path.join(d1, "frame_0001.png")) == mtime
    for k in (3, 5):
        np.testing.assert_array_equal(np.array(Image.open(os.path.join(d1, f"frame_{k:04d}.png"))), frames1[k])

    # two ranks, frames f % 2 == rank: same pixels as the single-rank run
    out2 = str(tmp_path / "b" / "v.mp4")
    d2 = run(out2, rank=0, world=2)
    run(out2, rank=1, world=2)
    for k in range(6):
        np.testing.assert_array_equal(np.array(Image.open(os.path.join(d2, f"frame_{k:04d}.png"))), frames1[k])

    # a changed parameter after a run with ANOTHER world size: the stale records of both namings go, so that the next
    # --resume really resumes instead of starting over for ever (advisor finding, round 2)
    assert sorted(f for f in os.listdir(d2) if f.startswith("progress")) == ["progress.rank0.json", "progress.rank1.json"]
    run(out2, n_frames=5, resume=True)                        # world 1, n_frames changed -> start over, once
    assert sorted(f for f in os.listdir(d2) if f.startswith("progress")) == ["progress.json"]
    mtime = os.path.getmtime(os.path.join(d2, "frame_0002.png"))
    run(out2, n_frames=5, resume=True)                        # nothing left to do: no frame is rendered again
    assert os.path.getmtime(os.path.join(d2, "frame_0002.png")) == mtime


def test_disk_model_flag():
    from bhr_amd import cli
    assert cli.parse_args([]).disk_model == "texture"
    a = cli.parse_args(["--disk_model", "v2_volume"])
    cli.validate_args(a)
    for bad in (["--disk_model", "v2", "--video"], ["--disk_model", "v2_volume", "--disk_texture", "x.png"]):
        with pytest.raises(ValueError):
            cli.validate_args(cli.parse_args(bad))


@pytest.mark.gpu
def test_cli_disk_models(tmp_path):
    """Still images with the analytic Disk V2 sources, single context and two row blocks."""
    from PIL import Image
    imgs = {}
    for model, gpus in (("v2", 1), ("v2_volume", 1), ("v2_volume", 2)):
        out = tmp_path / f"{model}_{gpus}.png"
        env = dict(os.environ, BHR_TILE_DEVICES="0,0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "render.py"), "-r", "sd", "--n_stars", "200", "-o", str(out),
                            "--pov", "9", "0", "1.2", "--ar1", "2", "--ar2", "10", "--disk_model", model, "--gpus", str(gpus)],
                           capture_output=True, text=True, cwd=ROOT, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        imgs[(model, gpus)] = np.array(Image.open(out)).astype(np.int32)
    assert imgs[("v2", 1)].max() > 100 and imgs[("v2_volume", 1)].max() > 100
    assert np.abs(imgs[("v2", 1)] - imgs[("v2_volume", 1)]).mean() > 1.0              # different disks
    assert np.abs(imgs[("v2_volume", 1)] - imgs[("v2_volume", 2)]).max() <= 1          # tiles == whole frame (8-bit)


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    out = tmp_path / "cli.png"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "render.py"), "-r", "sd", "--n_stars", "200", "-o", str(out),
                        "--disk_tilt", "10"], capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    from PIL import Image
    img = np.array(Image.open(out))
    assert img.shape == (360, 640, 3) and img.max() > 50


@pytest.mark.gpu
def test_cli_video_two_ranks_under_torchrun(tmp_path):
    """configs[4] as it is launched on a node: `torch.distributed.run ... render.py --video --orbit`, frames
    f % 2 == rank, a gloo barrier, rank 0 merges the progress files.  Both ranks share the card here
    (BHR_FORCE_DEVICE); the frames equal those of a single-process run."""
    import json
    from PIL import Image
    from bhr_amd import drivers
    common = ["-r", "sd", "--n_stars", "100", "--video", "--orbit", "--n_frames", "6", "--orbit_degrees", "60", "--fps", "6"]
    out2 = tmp_path / "two" / "v.mp4"
    env = dict(os.environ, BHR_FORCE_DEVICE="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "render.py"),
                        *common, "-o", str(out2)], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    d2 = drivers._frames_dir(str(out2))
    done = set()
    for r in range(2):
        done |= set(json.load(open(os.path.join(d2, f"progress.rank{r}.json")))["completed"])
    assert done == set(range(6))
    out1 = tmp_path / "one" / "v.mp4"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "render.py"), *common, "-o", str(out1)],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    d1 = drivers._frames_dir(str(out1))
    for k in range(6):
        a = np.array(Image.open(os.path.join(d1, f"frame_{k:04d}.png")))
        b = np.array(Image.open(os.path.join(d2, f"frame_{k:04d}.png")))
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """bench.py's N > 1 path (barrier, max-over-ranks time, summed ray-steps, one JSON line from rank 0) with two
    gloo ranks sharing the card; RCCL itself refuses two ranks on one device, its init is covered by the
    BHR_DIST_FORCE single-rank run in DESIGN.md."""
    import json
    env = dict(os.environ, BHR_DIST_BACKEND="gloo", BHR_FORCE_DEVICE="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "3", "--workload", "sd", "--no-other-math"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["metric"] == "Mray-steps/s"
    per_frame = d["config"]["ray_steps_per_frame"]
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 20 / (2 * 20 * per_frame) - 1) < 1e-6   # value = all ranks' steps / time
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"


@pytest.mark.gpu
@pytest.mark.parametrize("strong", [False, True])
def test_bench_row_block_leg_rehearsal(strong):
    """The strong-scaling leg of bench.py under torchrun: rank 0 drives one context per tile (both tiles on the one
    card here, BHR_TILE_DEVICES), the other rank waits at the host barrier; as `tile_scaling` beside the weak
    headline, and as the headline itself with --strong."""
    import json
    env = dict(os.environ, BHR_DIST_BACKEND="gloo", BHR_FORCE_DEVICE="0", BHR_TILE_DEVICES="0,0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29535" if strong else "29536", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "20", "--warmup", "3", "--workload", "sd", "--tile-workload", "sd", "--no-cpu-baseline"]
    p = subprocess.run(cmd + (["--strong"] if strong else []), capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    t = d if strong else d["tile_scaling"]
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if strong else "weak")
    assert t["scaling"] == "strong" and t["n_gpus"] == 2 and t["value"] > 0
    blocks = (d["config"] if strong else t)["row_blocks"]
    assert len(blocks) == 2 and blocks[0][0] == 0 and blocks[1][1] == 360 and blocks[0][1] == blocks[1][0]
    if strong:
        assert d["steps"] == 20 and abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / d["config"]["ray_steps_per_frame"] - 1) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("strong", [False, True])
def test_bench_row_block_leg_one_process_per_tile(strong):
    """The same leg when a rank sees only its own GPU (no BHR_TILE_DEVICES: rank 0 sees one device for two ranks): every
    rank renders its own tile through multigpu.TileLink (HIP IPC memory handles, shared-memory counters) -- bench.py
    --strong no longer exits there (round-2 review, item 5)."""
    import json
    env = dict(os.environ, BHR_DIST_BACKEND="gloo", BHR_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BHR_TILE_DEVICES", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29537" if strong else "29538", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "20", "--warmup", "3", "--workload", "sd", "--tile-workload", "sd", "--no-cpu-baseline"]
    p = subprocess.run(cmd + (["--strong"] if strong else []), capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    t = d if strong else d["tile_scaling"]
    assert "error" not in t, t
    assert t["scaling"] == "strong" and t["n_gpus"] == 2 and t["value"] > 0
    assert "one process per tile" in (d["config"]["driven_by"] if strong else t["driven_by"])
    blocks = (d["config"] if strong else t)["row_blocks"]
    assert len(blocks) == 2 and blocks[0][0] == 0 and blocks[1][1] == 360 and blocks[0][1] == blocks[1][0]


def test_balanced_row_blocks_properties():
    from bhr_amd.multigpu import balanced_row_blocks, row_blocks
    rng = np.random.default_rng(0)
    for height, n in ((4320, 8), (1080, 8), (360, 3), (64, 8), (1000, 7)):
        bands = -(-height // 8)
        cost = 1.0 + 3.0 * np.exp(-((np.arange(bands) - bands / 2) / (bands / 6)) ** 2) + 0.05 * rng.random(bands)
        blocks = balanced_row_blocks(height, n, cost, 8)
        assert blocks[0][0] == 0 and blocks[-1][1] == height and len(blocks) == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))                       # contiguous, ordered
        assert all(r0 % 8 == 0 for r0, _ in blocks) and all(r1 - r0 >= 8 for r0, r1 in blocks)
        per_row = np.repeat(cost / 8, 8)[:height]
        got = np.array([per_row[r0:r1].sum() for r0, r1 in blocks])
        even = np.array([per_row[r0:r1].sum() for r0, r1 in row_blocks(height, n)])
        if height >= 64 * n:
            assert got.max() <= even.max() + 1e-9 and got.max() / got.mean() < 1.10       # better than an even cut
    assert balanced_row_blocks(360, 3, np.zeros(45), 8) == row_blocks(360, 3)               # no information: even cut
    with pytest.raises(ValueError):
        balanced_row_blocks(40, 8, np.ones(5), 8)
    with pytest.raises(ValueError):
        balanced_row_blocks(360, 3, np.ones(10), 8)


@pytest.mark.gpu
def test_row_costs_and_balanced_tiles():
    """The per-band step profile adds up to the frame's ray-steps, does not depend on the textures, and row blocks
    cut by it render the same image as an even cut."""
    from bhr_amd import HipRenderer, scenes
    from bhr_amd.multigpu import balanced_row_blocks, group_render, probe_row_costs, row_blocks
    cam, fov, (w, h) = [6, 0, 0.5], 90, (256, 144)
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    r = HipRenderer(w, h, sky, tex)
    costs = r.row_costs(cam, fov)
    steps = r.counters()["ray_steps"]
    assert len(costs) == h // 8 and steps <= int(costs.sum()) <= 1.15 * steps     # ray-steps + 320 per shading pass
    assert costs.max() > 1.05 * costs.min()                          # not flat: the ring rows take more steps, the shadow fewer
    blank = HipRenderer(w, h, np.zeros((8, 16, 3), np.float32), np.zeros((32, 64, 4), np.float32))
    np.testing.assert_array_equal(blank.row_costs(cam, fov), costs)  # the profile is texture independent
    blank.close()
    r.render_async(cam, fov)
    with pytest.raises(AssertionError):                              # the last render carried no BHR_ROW_COSTS
        from bhr_amd import _lib
        import ctypes as C
        _lib.check(_lib.load().bhr_get_row_costs(r._ctx, (C.c_uint64 * len(costs))(), len(costs)))
    whole = r.render(cam, fov)
    r.close()
    per_row, band = probe_row_costs(w, h, cam, fov, scale=2)
    blocks = balanced_row_blocks(h, 4, per_row, band)
    assert blocks != row_blocks(h, 4) and all(r0 % 8 == 0 for r0, _ in blocks)
    tiles = [HipRenderer(w, h, sky, tex, rows=b) for b in blocks]
    img = group_render(tiles, cam, fov)
    assert np.abs(img - whole).max() <= 1e-6
    for t in tiles:
        t.close()
