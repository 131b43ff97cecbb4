#!/usr/bin/env python
"""GroupNorm variants (fused single launch vs slab 2-launch) over the UNet's GN shapes."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
        sys.path.insert(0, p)
    from python_hip_stable_diffusion import _lib
    rs = np.random.RandomState(0)
    out = {}
    for (c, hw) in [(320, 64), (640, 64), (960, 64), (320, 32), (640, 32), (960, 32), (1280, 32), (1920, 32), (640, 16), (1280, 16),
                    (1920, 16), (2560, 16), (1280, 8), (2560, 8)]:
        x = rs.randn(2, c, hw, hw).astype(np.float16)
        _, ms = _lib.groupnorm(x, np.ones(c, np.float32), np.zeros(c, np.float32), silu=True, iters=30)
        out[f"{c}@{hw}"] = ms * 1e3
    print(json.dumps(out))
else:
    res = {}
    for name, thr, wide in (("slab", "0", "0"), ("wide", "256", "1"), ("fused", "100000000", "0")):
        env = dict(os.environ, SD_GN_FUSED_MAX_HW=thr, SD_GN_WIDE=wide)
        r = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["slab"]:
        print(f"GN {k:10s}: slab pair {res['slab'][k]:6.1f} us   production (1024-thread one-launch at 32x32) {res['wide'][k]:6.1f} us   256-thread fused everywhere {res['fused'][k]:6.1f} us")
