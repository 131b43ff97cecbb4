#!/usr/bin/env python
"""Run one conv shape repeatedly (for rocprofv3 --pmc).  usage: pmc_conv.py k stride cin cout h up tile iters"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

k, s, cin, cout, h, up, tile, iters = (int(v) for v in sys.argv[1:9])
rs = np.random.RandomState(0)
x = rs.randn(2, cin, h, h).astype(np.float16)
w = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float16)
_, ms = _lib.conv2d(x, w, np.zeros(cout, np.float32), None, stride=s, upsample=bool(up), tile=tile, iters=iters)
print(f"conv k{k} s{s} {cin}->{cout} @{h} up{up} tile{tile}: {ms:.4f} ms")
