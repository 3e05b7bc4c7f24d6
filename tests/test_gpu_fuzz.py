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
