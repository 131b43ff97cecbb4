#!/usr/bin/env python
"""Round-4 attention A/B: the software-pipelined d = 64 kernel (attention8.hip) against the general kernels
(variant 1) on the UNet's self-attention shapes, stand-alone, plus a quick parity check against the oracle.
  python tools/r4_attn.py [check] [bench]          (SD_TUNE=1 SD_ATTN8_WAVES=4|8 forces the workgroup size)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

what = sys.argv[1:] or ["check", "bench"]
rs = np.random.RandomState(0)
if "check" in what:
    from oracle import attention_ref, psnr
    for (b, heads, sq, sk) in [(1, 2, 256, 256), (2, 5, 1024, 1024), (1, 2, 96, 448), (1, 3, 33, 64), (1, 1, 4096, 4096)]:
        q, k, v = (rs.randn(b, heads * 64, 1, n).astype(np.float16) for n in (sq, sk, sk))
        ref = attention_ref.original(q.astype(np.float32), k.astype(np.float32), v.astype(np.float32), heads, 64)
        for impl in ("ORIGINAL", "SPLIT_EINSUM"):
            row = []
            for variant in (0, 1):
                out, _ = _lib.attention(impl, q, k, v, heads, 64, variant=variant)
                err = np.abs(out.astype(np.float64) - ref).max()
                row.append(f"variant {variant}: PSNR {psnr.compute_psnr(out, ref):6.1f} dB max|err| {err:.2e}")
            print(f"check b{b} h{heads} {sq}x{sk} {impl}: " + "   ".join(row), flush=True)
if "bench" in what:
    for (heads, s, count) in [(5, 4096, 5), (10, 1024, 5), (20, 256, 5), (10, 9216, 0), (20, 2304, 0)]:
        q, k, v = (rs.randn(2, heads * 64, 1, s).astype(np.float16) for _ in range(3))
        flop = 4.0 * 2 * heads * 64 * s * s
        row = []
        for impl in ("ORIGINAL", "SPLIT_EINSUM"):
            for variant in (1, 0):
                _, ms = _lib.attention(impl, q, k, v, heads, 64, variant=variant, iters=20)
                row.append(f"{impl[:5]} v{variant} {ms * 1e3:7.1f} us ({flop / ms / 1e9:4.0f} TF)")
        print(f"attn h{heads} S={s}: " + "  ".join(row), flush=True)
