#!/usr/bin/env python
"""Fused q|k|v projection with LayerNorm fold: tiled kernels vs wsgemm.hip (C = 320) / bvgemm.hip (C >= 640), stand-alone, warm."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

rs = np.random.RandomState(0)
print("fused q|k|v + LayerNorm fold, us per launch (TFLOP/s): tiled | new kernel")
for batch, hw, c in ((2, 4096, 320), (16, 4096, 320), (2, 1024, 640), (16, 1024, 640), (16, 256, 1280)):
    x = rs.randn(batch * hw, c).astype(np.float16)
    w = (rs.randn(3 * c, c) / np.sqrt(c)).astype(np.float16)
    lw = (1 + 0.1 * rs.randn(c)).astype(np.float32)
    lb = (0.1 * rs.randn(c)).astype(np.float32)
    flop = 2.0 * batch * hw * c * 3 * c
    t = min(_lib.qkv_ln(x, lw, lb, w, batch, kernel=1, iters=20)[2] for _ in range(3))
    k = 2 if c == 320 else 3
    try:
        n = min(_lib.qkv_ln(x, lw, lb, w, batch, kernel=k, iters=20)[2] for _ in range(3))
    except ValueError as e:
        n = float("nan")
    print(f"  B={batch:2d} HW={hw:5d} C={c:5d}: tiled {t * 1e3:7.1f} ({flop / (t * 1e-3) / 1e12:5.0f}) | {'wsgemm' if k == 2 else 'bvgemm'} {n * 1e3:7.1f} ({flop / (n * 1e-3) / 1e12:5.0f})", flush=True)
