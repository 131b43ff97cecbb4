#!/usr/bin/env python
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for (k, s, cin, cout, h, up) in [(3, 1, 320, 320, 64, 0), (1, 1, 320, 320, 64, 0), (3, 1, 640, 640, 32, 0)]:
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    for tile in (1, 3):
        for mode, name in ((5, "full"), (7, "compute-only"), (7 + 8, "mfma-only"), (7 + 16, "ldsread-only"), (7 + 24, "barriers-only")):
            _, ms = _lib.conv2d(x, w, b, None, stride=s, upsample=bool(up), tile=tile, splitk=1, force_generic=mode, iters=20)
            print(f"k{k} {cin}->{cout} @{h} tile {tile} mode {name}: {ms*1e3:.1f} us", flush=True)
