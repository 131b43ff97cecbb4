#!/usr/bin/env python
"""Attention time vs number of workgroups (heads) at S = 4096, d = 64, batch 2: is a CU with two workgroups twice as slow?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for heads in (1, 2, 4, 5, 6, 8, 12, 16):
    q = rs.randn(2, heads * 64, 1, 4096).astype(np.float16)
    k = rs.randn(2, heads * 64, 1, 4096).astype(np.float16)
    v = rs.randn(2, heads * 64, 1, 4096).astype(np.float16)
    row = []
    for impl in ("ORIGINAL", "SPLIT_EINSUM"):
        _, ms = _lib.attention(impl, q, k, v, heads, 64, iters=20)
        row.append(f"{impl} {ms * 1e3:7.1f} us")
    print(f"heads {heads:2d}: {2 * heads * 32:4d} WGs ({2 * heads * 32 / 256:4.2f}/CU)  " + "  ".join(row), flush=True)
