#!/usr/bin/env python
"""Per-tile latency probe for the attention kernels: one workgroup, one per CU, two per CU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for (B, heads, sq, sk) in [(1, 1, 128, 4096), (1, 1, 128, 8192), (1, 8, 4096, 4096), (1, 16, 4096, 4096), (2, 16, 4096, 4096), (1, 1, 512, 4096), (1, 4, 4096, 4096)]:
    q = rs.randn(B, heads * 64, 1, sq).astype(np.float16)
    k = rs.randn(B, heads * 64, 1, sk).astype(np.float16)
    v = rs.randn(B, heads * 64, 1, sk).astype(np.float16)
    flop = 4.0 * B * heads * 64 * sq * sk
    row = []
    for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
        _, ms = _lib.attention(impl, q, k, v, heads, 64, iters=10)
        row.append(f"{impl} {ms*1e3:8.1f} us ({flop/ms/1e9:5.0f} TF, {ms*1e6/(sk/64):6.0f} ns/tile)")
    print(f"B{B} h{heads} {sq}x{sk}: " + "  ".join(row), flush=True)
