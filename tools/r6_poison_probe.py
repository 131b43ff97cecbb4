#!/usr/bin/env python
"""Which operator reads LDS it never wrote?  Run each small-shape operator in a FRESH process right after tools/ubench/poison (LDS full of
NaN patterns) and report NaNs.  usage: python tools/r6_poison_probe.py [case]   (no argument: every case, each in its own process)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "ml-stable-diffusion_amd"))


def cases():
    from python_hip_stable_diffusion import _lib
    rs = np.random.RandomState(0)
    r = lambda *s: rs.randn(*s).astype(np.float16)
    c = {}
    for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
        c[f"attention {impl} d16 S16"] = lambda impl=impl: _lib.attention(impl, r(2, 32, 1, 16), r(2, 32, 1, 16), r(2, 32, 1, 16), 2, 16)[0]
        c[f"attention {impl} d16 S64"] = lambda impl=impl: _lib.attention(impl, r(2, 32, 1, 64), r(2, 32, 1, 64), r(2, 32, 1, 64), 2, 16)[0]
        c[f"attention {impl} d16 Sq16 Sk77"] = lambda impl=impl: _lib.attention(impl, r(2, 64, 1, 16), r(2, 64, 1, 77), r(2, 64, 1, 77), 4, 16)[0]
        c[f"attention {impl} d64 S64"] = lambda impl=impl: _lib.attention(impl, r(2, 128, 1, 64), r(2, 128, 1, 64), r(2, 128, 1, 64), 2, 64)[0]
    c["groupnorm C32 4x4"] = lambda: _lib.groupnorm(r(2, 32, 4, 4), np.ones(32, np.float32), np.zeros(32, np.float32), groups=32)[0]
    c["groupnorm C64 8x8"] = lambda: _lib.groupnorm(r(2, 64, 8, 8), np.ones(64, np.float32), np.zeros(64, np.float32), groups=32)[0]
    c["layernorm C32"] = lambda: _lib.layernorm(r(2, 32, 1, 16), np.ones(32, np.float32), np.zeros(32, np.float32))[0]
    c["conv3x3 32->64 4x4"] = lambda: _lib.conv2d(r(2, 32, 4, 4), r(64, 32, 3, 3) * 0.1, np.zeros(64, np.float32), None, stride=1)[0]
    c["conv3x3 32->32 8x8"] = lambda: _lib.conv2d(r(2, 32, 8, 8), r(32, 32, 3, 3) * 0.1, np.zeros(32, np.float32), None, stride=1)[0]
    c["conv1x1 64->64 4x4"] = lambda: _lib.conv2d(r(2, 64, 4, 4), r(64, 64, 1, 1) * 0.1, np.zeros(64, np.float32), None)[0]
    c["conv3x3 s2 32->32 8x8"] = lambda: _lib.conv2d(r(2, 32, 8, 8), r(32, 32, 3, 3) * 0.1, np.zeros(32, np.float32), None, stride=2)[0]
    return c


if len(sys.argv) > 1:
    out = cases()[sys.argv[1]]()
    print("NaN" if not np.isfinite(np.asarray(out, np.float32)).all() else "ok")
else:
    for name in cases():
        subprocess.run([os.path.join(ROOT, "tools", "ubench", "poison")], stdout=subprocess.DEVNULL)
        p = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True)
        last = (p.stdout.strip().splitlines() or ["?"])[-1]
        print(f"{name:40s} {last}  {'' if p.returncode == 0 else p.stderr.strip().splitlines()[-1][:150]}", flush=True)
