"""Randomised parity: the strict march against the oracle over cameras, disks and step sizes far from the
BASELINE views -- inside the photon sphere, far away, steep tilts, coarse and fine steps, both AA modes.
The exact-rounding sequences of the strict build (sqrt / reciprocal / divide by Newton steps) are valid for
normal-range operands; this is where an operand outside that range would show up as a diverging ray."""
import numpy as np
import pytest

from bhr_amd import scenes

pytestmark = pytest.mark.gpu


def _cases(n=24, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        r = float(np.exp(rng.uniform(np.log(1.15), np.log(60.0))))
        th, ph = np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi)
        cam = [r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th)]
        r_in = float(rng.uniform(1.2, 4.0))
        out.append(dict(cam=cam, fov=float(rng.uniform(20, 150)), frame=int(rng.integers(0, 200)),
                        kw=dict(step_size=float(rng.choice([0.05, 0.1, 0.3])), r_max=float(rng.choice([10.0, 25.0])),
                                r_disk_inner=r_in, r_disk_outer=r_in + float(rng.uniform(0.5, 20.0)),
                                disk_tilt=float(rng.uniform(-80, 80)),
                                anti_alias=str(rng.choice(["disabled", "lod_radius"])),
                                aa_strength=float(rng.uniform(0.5, 2.0)))))
    return out


@pytest.mark.parametrize("k", range(24))
def test_random_view_matches_oracle(k, oracle, hip_lib):
    from bhr_amd import HipRenderer, _lib
    c = _cases()[k]
    w, h = 48, 32
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    hip = HipRenderer(w, h, sky, tex, **c["kw"])
    ora = oracle.OracleRenderer(w, h, sky, tex, **c["kw"])
    hip.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True)
    bg, disk = hip.read_layer(_lib.LAYER_BG), hip.read_layer(_lib.LAYER_DISK)
    rbg, rdisk = (x.transpose(1, 0, 2) for x in ora.march(c["cam"], c["fov"], frame=c["frame"]))
    assert np.isfinite(bg).all() and np.isfinite(disk).all()
    assert hip.counters()["ray_steps"] == ora.last_total_steps, f"case {k}: {c}"
    # identical ray paths; what differs is the last bits of the per-hit transcendentals (a handful of pixels)
    assert np.abs(bg - rbg).max() <= 2e-4 and np.abs(disk - rdisk).max() <= 2e-4, f"case {k}: {c}"
    assert np.sqrt(np.mean((disk - rdisk) ** 2)) <= 1e-5 and np.sqrt(np.mean((bg - rbg) ** 2)) <= 1e-5
    hip.close()


@pytest.mark.parametrize("k", range(24))
def test_random_view_hybrid_against_strict(k, hip_lib):
    """The same 24 views through math="hybrid" (guards as the library picks them: on for AA or a tilted disk) against the
    strict march of the same context, at a size where a frame has a few hundred 8x8 tiles: per-channel RMSE <= 6e-5 (north
    star 1e-4; the BASELINE views and fixtures hold 3e-5 in tests/test_gpu_hybrid.py -- these views include cameras 60 r_s
    away, whose rays take several hundred fast steps before they reach the hole), ray-step totals within 2e-4, and no
    colour flips -- the failure the guards exist for (a ray that hits the disk under one arithmetic and misses it under
    the other)."""
    from bhr_amd import HipRenderer, _lib
    c = _cases()[k]
    w, h = 192, 128
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    r = HipRenderer(w, h, sky, tex, math="hybrid", **c["kw"])
    lay, steps = {}, {}
    for math in ("hybrid", "strict"):
        r.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True, math=math)
        lay[math] = (r.read_layer(_lib.LAYER_BG), r.read_layer(_lib.LAYER_DISK))
        steps[math] = r.counters()["ray_steps"]
    r.close()
    assert abs(steps["hybrid"] - steps["strict"]) <= max(2e-4 * steps["strict"], 64), (k, steps)
    for name, a, b in (("bg", lay["hybrid"][0], lay["strict"][0]), ("disk", lay["hybrid"][1], lay["strict"][1])):
        assert np.isfinite(a).all()
        e = np.sqrt(np.mean((a.astype(np.float64) - b) ** 2, axis=(0, 1)))
        flips = int((np.abs(a - b).max(axis=2) > 0.05).sum())
        print(f"[hybrid fuzz] case {k} {name}: RMSE {e.max():.3g}, max {np.abs(a - b).max():.3g}")
        assert (e <= 6e-5).all() and flips == 0, f"case {k} {name}: RMSE {e}, {flips} pixels beyond 0.05: {c}"


@pytest.mark.parametrize("k,n,seed", [(215, 1500, 11), (724, 1500, 11), (786, 1500, 11), (1133, 1500, 11),
                                      (521, 3000, 23), (41, 3000, 23), (1368, 3000, 23), (218, 3000, 23), (595, 3000, 23),
                                      (1913, 3000, 23), (2979, 3000, 23), (2756, 3000, 23), (696, 1200, 5)])
def test_telephoto_views_hybrid_within_the_f32_noise_of_the_march(k, n, seed, oracle, hip_lib):
    """The worst views of tools/fuzz_hybrid.py's sweeps (1500 views of seed 11: the four beyond 1e-4; 3000 of seed 23: the
    eight worst, among them every one closer than 14 r_s): cameras far away behind a long lens, or a step of 0.3, where hybrid
    sits 1.2e-4 ... 3.1e-4 RMSE from strict with no pixel flipped and equal step totals.  Rays that start at r = 50 carry
    half an ulp of 50 per step for several hundred steps, and a coarse step turns an ulp of the hit point into more of
    the texture: ANY two f32 evaluation orders differ by that much on these views.  The yardstick is the strict march
    itself against the oracle's binary64 build (the reference's statements with binary64 intermediates): hybrid may be no
    further from strict than 1.5x strict is from binary64."""
    from bhr_amd import HipRenderer, _lib
    c = _cases(n, seed)[k]
    w, h = (512, 320) if seed == 5 else (192, 128)          # the seed-5 sweep ran at 512x320 (tools/fuzz_hybrid.py --size)
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    r = HipRenderer(w, h, sky, tex, math="hybrid", **c["kw"])
    lay = {}
    for math in ("hybrid", "strict"):
        r.render_async(c["cam"], c["fov"], frame=c["frame"], skip_bloom=True, math=math)
        lay[math] = r.read_layer(_lib.LAYER_DISK)
    r.close()
    ora = oracle.OracleRenderer(w, h, sky, tex, fast="f64", **c["kw"])
    _, d64 = ora.march(c["cam"], c["fov"], frame=c["frame"])
    d64 = d64.transpose(1, 0, 2)
    # up to two pixels of the 24 576 may sit on the other side of a faint disk-edge decision (none by more than 0.05: view
    # 1368 had one before the band widened with the step size -- 0.02 of a crossing at the inner edge that the fast arithmetic
    # did not register, nor does the oracle's own -ffast-math build; its binary64 build sides with strict --, which alone is 1.3e-4 of
    # RMSE at this frame size): the bound applies to the frame without its two worst pixels
    worst2 = np.argsort(np.abs(lay["hybrid"] - lay["strict"]).max(axis=2).ravel())[-2:]

    def rm(a, b):
        d2 = ((a.astype(np.float64) - b) ** 2).reshape(-1, 3)
        d2[worst2] = 0.0
        return float(np.sqrt(d2.mean(axis=0)).max())
    e_hs, e_s64, e_h64 = rm(lay["hybrid"], lay["strict"]), rm(lay["strict"], d64), rm(lay["hybrid"], d64)
    print(f"\n[telephoto] view {k}: hybrid-strict {e_hs:.3g}, strict-binary64 {e_s64:.3g}, hybrid-binary64 {e_h64:.3g}")
    dmax = np.abs(lay["hybrid"] - lay["strict"]).max(axis=2).ravel().copy()
    dmax[worst2] = 0.0
    assert int((dmax > 0.05).sum()) == 0       # view 696 (512x320) has ONE pixel where only the fast arithmetic registers a crossing
    assert e_hs <= max(6e-5, 1.5 * e_s64), (e_hs, e_s64)
    assert e_h64 <= max(6e-5, 1.5 * e_s64), (e_h64, e_s64)           # and no further from binary64 than strict is, x 1.5


@pytest.mark.parametrize("k", range(0, 24, 3))
def test_random_view_hybrid_row_blocks_equal_one_context(k, hip_lib):
    """Eight of the fuzzed views as hybrid frames cut into three uneven row blocks: every block classifies its own tiles
    (band, in-plane wedge), marches its two lists, repairs where the view asks for it, blooms over exchanged halo rows -- the
    gathered frame equals the frame of one context bit for bit, and the ray-step totals agree."""
    from bhr_amd import HipRenderer, multigpu
    c = _cases()[k]
    w, h = 320, 208
    rng = np.random.default_rng(900 + k)
    # cuts on multiples of 8, as balanced_row_blocks makes them: the strict / fast choice is per 8x8 tile of the block's own
    # tiling, and only then is that the tiling of the whole frame
    cuts = [0] + sorted(8 * int(v) for v in rng.choice(np.arange(1, h // 8 - 1), size=2, replace=False)) + [h]
    sky, tex = scenes.analytic_skybox(), scenes.noisy_disk()
    full = HipRenderer(w, h, sky, tex, math="hybrid", frame_slots=1, **c["kw"])
    ref = full.render(c["cam"], c["fov"], frame=c["frame"])
    steps = full.counters()["ray_steps"]
    full.close()
    tiles = [HipRenderer(w, h, sky, tex, rows=(cuts[q], cuts[q + 1]), math="hybrid", frame_slots=1, **c["kw"]) for q in range(3)]
    for sched in ("serial", "pipelined"):
        multigpu.group_render(tiles, c["cam"], c["fov"], frame=c["frame"], gather="peer", schedule=sched)
        np.testing.assert_array_equal(multigpu.read_gathered(tiles), ref, err_msg=f"case {k} cuts {cuts} {sched}: {c}")
    assert sum(t.counters()["ray_steps"] for t in tiles) == steps
    for t in tiles:
        t.close()


@pytest.mark.parametrize("k", range(1, 24, 4))
def test_random_view_row_costs_and_balanced_blocks(k, hip_lib):
    """The cost probes behind make_tiles on six of the fuzzed views, both arithmetics: profiles are finite and positive where
    the frame has rays, the hybrid split sums to the frame's step total, and the balanced blocks tile the frame on multiples
    of 8 rows."""
    from bhr_amd import multigpu
    c = _cases()[k]
    kw = {q: c["kw"][q] for q in ("step_size", "r_max", "r_disk_inner", "r_disk_outer", "disk_tilt")}
    W, H = 1280, 720
    for math in (None, "hybrid"):
        per_row, band = multigpu.probe_row_costs(W, H, c["cam"], c["fov"], math=math, **kw)
        assert per_row.shape == (H,) and np.isfinite(per_row).all() and (per_row >= 0).all() and per_row.sum() > 0, (k, math)
        blocks = multigpu.balanced_row_blocks(H, 5, per_row, band, fixed_cost_per_row=0.1 * float(per_row.mean()))
        assert blocks[0][0] == 0 and blocks[-1][1] == H and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert all(r0 % 8 == 0 and r1 > r0 for r0, r1 in blocks), blocks
        cost = np.array([per_row[r0:r1].sum() for r0, r1 in blocks])
        assert cost.max() <= 2.0 * cost.mean() + per_row.max() * 8, (k, math, blocks, cost)      # no block left with the lot
