#!/usr/bin/env python
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
tot = {}
for (heads, sq, sk, count) in [(5, 4096, 4096, 5), (5, 4096, 77, 5), (10, 1024, 1024, 5), (10, 1024, 77, 5), (20, 256, 256, 5), (20, 256, 77, 5), (20, 64, 64, 1), (20, 64, 77, 1)]:
    q = rs.randn(2, heads * 64, 1, sq).astype(np.float16)
    k = rs.randn(2, heads * 64, 1, sk).astype(np.float16)
    v = rs.randn(2, heads * 64, 1, sk).astype(np.float16)
    flop = 4.0 * 2 * heads * 64 * sq * sk
    row = []
    for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
        _, ms = _lib.attention(impl, q, k, v, heads, 64, iters=20)
        tot[impl] = tot.get(impl, 0) + ms * count
        row.append(f"{impl} {ms*1e3:7.1f} us ({flop/ms/1e9:4.0f} TF)")
    print(f"attn h{heads} {sq}x{sk}: " + "  ".join(row), flush=True)
print({k: round(v, 3) for k, v in tot.items()})
