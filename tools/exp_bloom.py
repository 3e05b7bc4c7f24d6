"""Post-pass timing: H + V of the split-f16 and of the exact f32 kernels at the BASELINE sizes, whole frames and one
row block of eight of the 8k frame, with the V pass storing everything / the f32 frame / the u8 rows only.
usage: python tools/exp_bloom.py [fhd 4k 8k 8k_tile]      (HIP events of the library's own frame brackets, isolated launches)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bhr_amd import HipRenderer, scenes, _lib

SIZES = {"fhd": (1920, 1080, None), "4k": (3840, 2160, None), "8k": (7680, 4320, None), "8k_tile": (7680, 4320, (1890, 2430))}
KW = dict(step_size=0.1, r_max=10.0, r_disk_inner=2.0, r_disk_outer=15.0, disk_tilt=0.0)


def main():
    args = sys.argv[1:]
    only_split = [int(a.split("=")[1]) for a in args if a.startswith("split=")]
    only_out = [a.split("=")[1] for a in args if a.startswith("out=")]
    tiles = [int(a.split("=")[1]) for a in args if a.startswith("tiles=")] or [0]
    which = [a for a in args if "=" not in a] or ["fhd", "4k", "8k_tile", "8k"]
    sky, tex = scenes.analytic_skybox(128, 256), scenes.noisy_disk(256, 1024)
    out = {}
    for name in which:
        W, H, rows = SIZES[name]
        for split in (only_split or (1, 0)):
          for outputs in (only_out or ("u8", "f32", "f32+blur+u8")):
           for nt in (tiles if split else [0]):
                r = HipRenderer(W, H, sky, tex, math="fast", frame_slots=1, rows=rows, outputs=outputs, options={"bloom_split": split, "bloom_tiles": nt},
                                **dict(KW, step_size=0.3 if W > 4000 else 0.1))
                for _ in range(3):
                    r.render_async([6, 0, 0.5], 90)
                r.timing_reset()
                n = 20
                for _ in range(n):
                    r.render_async([6, 0, 0.5], 90)
                c = r.counters()
                ms = c["bloom_ms_sum"] / c["frames_timed"]
                px = W * ((rows[1] - rows[0]) if rows else H)
                key = f"{name}/{'split' if split else 'exact'}/{outputs}/t{nt}"
                out[key] = {"post_ms": round(ms, 4), "march_ms": round(c["march_ms_sum"] / c["frames_timed"], 4),
                            "GB_per_s_algorithmic_24B_per_px": round(px * 24 / ms / 1e6, 1)}
                print(key, out[key], flush=True)
                r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
