#!/usr/bin/env python
"""Ablation of the igemm kernel: full vs loads-only vs compute-only, per tile/staging variant."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for (k, s, cin, cout, h, up) in [(3, 1, 320, 320, 64, 0), (3, 1, 640, 640, 32, 1), (3, 1, 640, 640, 32, 0), (1, 1, 320, 320, 64, 0), (3, 1, 1280, 1280, 16, 0)]:
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    for tile in (1, 21, 2, 4, 24, 3, 23):
        row = []
        for mode in (0, 2, 3):
            _, ms = _lib.conv2d(x, w, b, None, stride=s, upsample=bool(up), tile=tile, splitk=1, force_generic=mode, iters=20)
            row.append(ms * 1e3)
        print(f"k{k} {cin}->{cout} @{h} up{up} tile {tile:2d}: full {row[0]:7.1f} us | loads-only {row[1]:7.1f} | compute-only {row[2]:7.1f}", flush=True)
