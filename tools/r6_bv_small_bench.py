#!/usr/bin/env python
"""Small-M 1x1 GEMMs of the batch-2 step (16x16 / 32x32 levels): best tiled plan against bvgemm.hip's small-tile variant (32 rows x
2 waves x 32 columns, plan code 115) and its 64-row variant (111).  Set SD_TUNE=1 SD_BENCH_COLD=1 for cold operands (as in the step)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

rs = np.random.RandomState(0)
print("cold" if os.environ.get("SD_BENCH_COLD") else "warm", "operands; us per launch: best tiled plan | bvgemm 32x2x32 | bvgemm 64x8x32")
for cin, cout, hw in ((1280, 1280, 16), (5120, 1280, 16), (640, 640, 32), (2560, 640, 32), (1280, 1280, 8), (2560, 1280, 8), (320, 320, 64), (1280, 320, 64)):
    x = rs.randn(2, cin, hw, hw).astype(np.float16)
    w = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float16)
    res = rs.randn(2, cout, hw, hw).astype(np.float16)
    bias = np.zeros(cout, np.float32)
    m = 2 * hw * hw
    tiled = min(_lib.conv2d(x, w, bias, res, tile=c, iters=20)[1] for c in (0, 3, 23, 33, 63, 73, 83, 2, 62, 4, 64))
    b32 = min(_lib.conv2d(x, w, bias, res, tile=115, iters=20)[1] for _ in range(3))
    try:
        b64 = min(_lib.conv2d(x, w, bias, res, tile=111, iters=20)[1] for _ in range(3))
    except ValueError:
        b64 = float("nan")
    print(f"  {cin:5d}->{cout:5d} M={m:5d}: tiled {tiled * 1e3:6.1f} | bv32 {b32 * 1e3:6.1f} | bv64 {b64 * 1e3:6.1f}", flush=True)
