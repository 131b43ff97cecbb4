#!/usr/bin/env python
"""attention8: the classic grid against the balanced form at the UNet's self-attention shapes (operator level, 20 launches each).
usage: python tools/r6_attn_sk_bench.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-stable-diffusion_amd"))
from python_hip_stable_diffusion import _lib  # noqa: E402

rs = np.random.RandomState(0)
print(f"{'B':>3} {'heads':>5} {'S':>5} {'variant':>12} {'us':>8} {'TFLOP/s':>8}")
for batch, heads, s in [(2, 5, 4096), (4, 5, 4096), (6, 5, 4096), (2, 10, 1024), (4, 10, 1024), (2, 20, 256), (16, 5, 4096)]:
    c = heads * 64
    q = (rs.randn(batch, c, 1, s)).astype(np.float16)
    k = (rs.randn(batch, c, 1, s)).astype(np.float16)
    v = (rs.randn(batch, c, 1, s)).astype(np.float16)
    flop = 4.0 * batch * c * s * s
    for name, variant in [("classic", 0), ("balanced", 100)]:
        try:
            best = min(_lib.attention("ORIGINAL", q, k, v, heads, 64, variant=variant, iters=20)[1] for _ in range(3))
        except Exception as e:  # the balanced form refuses shapes it does not split
            print(f"{batch:3d} {heads:5d} {s:5d} {name:>12}  refused ({str(e)[:60]})")
            continue
        print(f"{batch:3d} {heads:5d} {s:5d} {name:>12} {best * 1e3:8.1f} {flop / best / 1e9:8.0f}")
