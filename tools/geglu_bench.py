#!/usr/bin/env python
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
tot = 0
for (m, c, count) in [(8192, 320, 5), (2048, 640, 5), (512, 1280, 5), (128, 1280, 1)]:
    x = rs.randn(m, c).astype(np.float16)
    wt = (rs.randn(8 * c, c) / np.sqrt(c)).astype(np.float16)
    _, ms = _lib.geglu(x, wt, np.zeros(8 * c, np.float32), iters=20)
    tot += ms * count
    print(f"geglu {m}x{c}: {ms*1e3:.1f} us ({2.0*m*c*8*c/ms/1e9:.0f} TF)")
print(f"per step: {tot*1e3:.0f} us  [{os.environ.get('SD_MI355X_LIB','default')}]")
