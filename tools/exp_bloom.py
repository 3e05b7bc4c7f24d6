#!/usr/bin/env python3
"""Bloom H / V pass variants (csrc/bloom.hip: BHR_BLOOM_H, BHR_BLOOM_V) timed alone on a rendered disk layer, and checked
bit for bit against the round-2 kernels.  usage: python tools/exp_bloom.py [--sizes 8k,8ktile,4k,fhd] [--quick VARIANT]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bhr_amd import HipRenderer, _lib, scenes

SIZES = {"8k": (7680, 4320, None, 0.05), "4k": (3840, 2160, None, 0.1),
         "fhd": (1920, 1080, None, 0.1)}
H_VARIANTS = ["0", "2l", "mfma1", "mfma2", "mfma4", "bf16x1", "bf16x2"]          # "<NG>l": weights in LDS (VGPR operands)
V_VARIANTS = ["32x0", "16x2", "16x4", "16x4l", "mfma1", "mfma2", "mfma4", "bf16x1", "bf16x2"]


def time_pass(r, only, n):
    os.environ["BHR_BLOOM_ONLY"] = only
    for _ in range(3):
        r.bloom_only()
    r.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        r.bloom_only()
    r.sync()
    dt = (time.perf_counter() - t0) / n * 1e3
    os.environ.pop("BHR_BLOOM_ONLY", None)
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="8k,4k,fhd")
    ap.add_argument("--only", default=None, help="comma list of variants to run (both passes)")
    ap.add_argument("--quick", default=None, help="H,V variant pair only (profiling), e.g. 0,32x0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bloom_variants.json"))
    a = ap.parse_args()
    res = {}
    for name in a.sizes.split(","):
        W, H, rows, step = SIZES[name]
        sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
        for k in ("BHR_BLOOM_H", "BHR_BLOOM_V", "BHR_BLOOM_ONLY", "BHR_BLOOM_W"):
            os.environ.pop(k, None)
        r = HipRenderer(W, H, sky, tex, step_size=step, rows=rows, frame_slots=1)
        r.render_async([6, 0, 0.5], 90, skip_bloom=True)          # bg + disk layers; halo rows of a row block stay zero
        n = 20 if W > 4000 else 50
        if a.quick:
            hv, vv = a.quick.split(",")
            os.environ["BHR_BLOOM_W"] = "lds" if (hv.endswith("l") and not hv.startswith(("m", "b"))) or (vv.endswith("l") and not vv.startswith(("m", "b"))) else "sgpr"
            os.environ["BHR_BLOOM_H"], os.environ["BHR_BLOOM_V"] = (hv if hv.startswith(("m", "b")) else hv.rstrip("l")), (vv if vv.startswith(("m", "b")) else vv.rstrip("l"))
            print(name, a.quick, "H", round(time_pass(r, "h", n), 4), "V", round(time_pass(r, "v", n), 4), flush=True)
            r.close()
            continue
        r.bloom_only()
        base_blur, base_final = r.read_layer(_lib.LAYER_BLUR), r.read_layer(_lib.LAYER_FINAL)
        out = {"H": {}, "V": {}}
        hvs = [v for v in H_VARIANTS if not a.only or v in a.only.split(",")]
        vvs = [v for v in V_VARIANTS if not a.only or v in a.only.split(",")]
        for hv in hvs:
            os.environ["BHR_BLOOM_H"] = hv if hv.startswith(("m", "b")) else hv.rstrip("l")
            os.environ["BHR_BLOOM_W"] = "lds" if (hv.endswith("l") and not hv.startswith(("m", "b"))) else "sgpr"
            os.environ["BHR_BLOOM_V"] = "32x0"
            try:
                r.bloom_only()
            except ValueError as e:
                out["H"][hv] = {"error": str(e)[:200]}
                print(name, "H", hv, out["H"][hv], flush=True)
                continue
            blur = r.read_layer(_lib.LAYER_BLUR)
            same = bool(np.array_equal(blur, base_blur))
            out["H"][hv] = {"ms": time_pass(r, "h", n), "bit_identical": same, "max_diff_blur": float(np.abs(blur - base_blur).max())}
            print(name, "H", hv, out["H"][hv], flush=True)
        os.environ["BHR_BLOOM_H"] = "0"
        for vv in vvs:
            os.environ["BHR_BLOOM_V"] = vv if vv.startswith(("m", "b")) else vv.rstrip("l")
            os.environ["BHR_BLOOM_W"] = "lds" if (vv.endswith("l") and not vv.startswith(("m", "b"))) else "sgpr"
            try:
                r.bloom_only()
                blur, fin = r.read_layer(_lib.LAYER_BLUR), r.read_layer(_lib.LAYER_FINAL)
                same = bool(np.array_equal(blur, base_blur) and np.array_equal(fin, base_final))
                out["V"][vv] = {"ms": time_pass(r, "v", n), "bit_identical": same, "max_diff_blur": float(np.abs(blur - base_blur).max()),
                                "max_diff_final": float(np.abs(fin - base_final).max()), "blur_max": float(base_blur.max())}
            except Exception as e:
                out["V"][vv] = {"error": str(e)[:200]}
            print(name, "V", vv, out["V"][vv], flush=True)
        res[name] = out
        r.close()
    if not a.quick:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
