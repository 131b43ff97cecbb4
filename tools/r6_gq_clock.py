#!/usr/bin/env python
"""Phase clock of the one-launch SpatialTransformer head (gn_proj_qkv_kernel) at UNet batch 2 / 16: SD_TUNE=1 SD_GQ_CLOCK=1 python tools/r6_gq_clock.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-stable-diffusion_amd"))
from python_hip_stable_diffusion import _lib  # noqa: E402

rs = np.random.RandomState(0)
c = 320
for batch in (2, 16):
    x = rs.randn(batch, c, 64, 64).astype(np.float16)
    w = lambda n, k: (rs.randn(n, k) / np.sqrt(k)).astype(np.float16)
    v = lambda: (0.1 * rs.randn(c)).astype(np.float32)
    args = (x, w(c, c), 1 + v(), v(), w(c, c), v(), 1 + v(), v(), w(3 * c, c))
    for fused in (True, False):
        ms = min(_lib.gn_proj_qkv(*args, q_scale=0.18, fused=fused, iters=20)[4] for _ in range(3))
        print(f"batch {batch:2d} fused {int(fused)}: producer conv + head {ms * 1e3:7.1f} us", flush=True)
