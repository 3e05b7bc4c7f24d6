"""One-off long form of tests/test_gpu_video_overlap.py: N fhd frames of the overlapped video loop against the
one-stream loop, every frame decoded and compared.  Usage: python tools/soak_overlap.py [n_frames] [math]"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
from test_gpu_video_overlap import _run
from bhr_amd.output import DEVICE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
math = sys.argv[2] if len(sys.argv) > 2 else None
tmp = tempfile.mkdtemp(prefix="bhr_overlap_")
a = _run(tmp, "serial", n, (1920, 1080), frame_slots=1, png_level=0, pairs_on_host=True, math=math)
b = _run(tmp, "overlapped", n, (1920, 1080), frame_slots=2, png_level=DEVICE, pairs_on_host=False, math=math)
diff = [f for f in range(n) if not np.array_equal(np.asarray(Image.open(os.path.join(a, f"frame_{f:04d}.png")).convert("RGB")),
                                                  np.asarray(Image.open(os.path.join(b, f"frame_{f:04d}.png")).convert("RGB")))]
print(f"{n} fhd frames (math {math}): {len(diff)} differ {diff[:10]}")
shutil.rmtree(tmp)
