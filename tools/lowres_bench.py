#!/usr/bin/env python
"""Weight-streaming 3x3 convs of the 16x16 / 8x8 levels: tile x staging x split-K sweep."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
shapes = [(1280, 1280, 8), (2560, 1280, 8), (1280, 1280, 16), (2560, 1280, 16), (640, 640, 32)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for cin, cout, h in shapes:
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    res = {}
    for tile in (3, 23, 33, 4, 24, 34, 1, 21):
        for sk in (1, 2, 4, 8, 16):
            try:
                _, ms = _lib.conv2d(x, w, b, None, tile=tile, splitk=sk, iters=20)
            except Exception as e:  # noqa: BLE001
                continue
            res[(tile, sk)] = ms * 1e3
    _, auto = _lib.conv2d(x, w, b, None, iters=20)
    best = sorted(res.items(), key=lambda kv: kv[1])[:10]
    mb = cout * cin * 9 * 2 / 1e6
    print(f"k3 {cin}->{cout} @{h}: auto {auto*1e3:.1f} us ({mb/(auto*1e3):.2f} TB/s weights) | " +
          " ".join(f"t{t}k{k}={v:.1f}" for (t, k), v in best), flush=True)
