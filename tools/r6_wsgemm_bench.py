#!/usr/bin/env python
"""GEGLU projection of the 320-channel level (norm3 folded in): the tiled GEMM kernels (igemm.hip / gemm_pipe) against the
weight-stationary kernel of wsgemm.hip, stand-alone, back-to-back launches (operands warm).  usage: r6_wsgemm_bench.py [out.txt]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


say("GEGLU 320 -> 2560 with LayerNorm fold, fp16: tiled kernels (library plan of round 5) vs weight-stationary kernel; us, TFLOP/s of 2.M.K.N; peak 2500")
rs = np.random.RandomState(0)
for m in (2048, 8192, 16384, 32768, 65536):
    c, n2 = 320, 2560
    x = (rs.randn(m, c)).astype(np.float16)
    w = (rs.randn(n2, c) / np.sqrt(c)).astype(np.float16)
    bias = (0.1 * rs.randn(n2)).astype(np.float32)
    lw = (1 + 0.1 * rs.randn(c)).astype(np.float32)
    lb = (0.1 * rs.randn(c)).astype(np.float32)
    flop = 2.0 * m * c * n2
    row = f"   M={m:6d}:"
    for name, k in (("tiled", 1), ("weight-stationary", 2)):
        best = 1e9
        for _ in range(3):
            _, ms = _lib.geglu_ln(x, w, bias, lw, lb, kernel=k, iters=30)
            best = min(best, ms)
        row += f"  {name} {best * 1e3:7.1f} us {flop / (best * 1e-3) / 1e12:6.0f} TF"
    say(row)
say("ablation builds of the weight-stationary kernel at M = 65536 (41 row tiles per workgroup), us per launch:")
m, c, n2 = 65536, 320, 2560
x = (rs.randn(m, c)).astype(np.float16)
w = (rs.randn(n2, c) / np.sqrt(c)).astype(np.float16)
bias = (0.1 * rs.randn(n2)).astype(np.float32)
lw = (1 + 0.1 * rs.randn(c)).astype(np.float32)
lb = (0.1 * rs.randn(c)).astype(np.float32)
for abl, name in ((0, "full"), (1, "no MFMAs"), (2, "no erf-GELU (value * gate)"), (3, "no LayerNorm statistics pass"),
                  (4, "no DMA inside the loop"), (6, "no global stores"), (5, "with timestamps")):
    best = min(_lib.geglu_ln(x, w, bias, lw, lb, kernel=2 + 10 * abl, iters=20)[1] for _ in range(3))
    say(f"   {name:34s} {best * 1e3:7.1f} us  = {best * 1e3 / 41:5.2f} us per tile")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
