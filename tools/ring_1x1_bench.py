#!/usr/bin/env python
"""1x1 GEMMs (K = C, only 5-20 K steps): 2-stage vs 3/4-stage LDS-DMA ring."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for cin, cout, h, cnt in [(320, 320, 64, 40), (640, 640, 32, 40), (1280, 1280, 16, 40), (1280, 320, 64, 5), (2560, 640, 32, 5), (5120, 1280, 16, 5), (1280, 1280, 8, 8)]:
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    res = {}
    for tile in (3, 23, 33, 4, 24, 2, 22, 1, 21):
        for sk in (1, 2, 4):
            try:
                _, ms = _lib.conv2d(x, w, b, None, tile=tile, splitk=sk, iters=30)
            except Exception:  # noqa: BLE001
                continue
            res[(tile, sk)] = ms * 1e3
    _, auto = _lib.conv2d(x, w, b, None, iters=30)
    best = sorted(res.items(), key=lambda kv: kv[1])[:8]
    print(f"k1 {cin}->{cout} @{h} x{cnt}: auto {auto*1e3:.2f} us | " + " ".join(f"t{t}k{k}={v:.2f}" for (t, k), v in best), flush=True)
