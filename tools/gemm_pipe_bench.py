#!/usr/bin/env python
"""Stand-alone times of the 1x1 GEMM shapes of the SD2.1-base step at UNet batch B for the plan codes of igemm.hip
(tile + 10 * staging; 6x / 7x = the software-pipelined gemm_pipe_kernel), back-to-back launches.
usage: gemm_pipe_bench.py [B]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [(320, 320, 64), (1280, 320, 64), (640, 640, 32), (2560, 640, 32), (1280, 1280, 16), (5120, 1280, 16), (320, 960, 64),
          (640, 1920, 32), (1280, 3840, 16), (320, 2560, 64), (1280, 10240, 16)]
if len(sys.argv) > 2 and sys.argv[2] == "sdxl":   # the 24x24 / 48x48 levels of SDXL-base at 768x768 (B = 2: 1152 / 4608 tokens)
    SHAPES = [(1280, 10240, 24), (1280, 3840, 24), (5120, 1280, 24), (1280, 1280, 24), (640, 5120, 48), (2560, 640, 48), (640, 1920, 48)]
CODES = [0, 9, 8, 1, 61, 2, 62, 4, 64, 3]
if os.environ.get("EXP_CODES"):   # extra plan codes (e.g. 81,82,84: the 2-stage ring)
    CODES = [0, 61, 62, 64] + [int(c) for c in os.environ["EXP_CODES"].split(",")]
rs = np.random.RandomState(0)
print(f"UNet batch {B}; columns: plan code -> us (TFLOP/s)")
for cin, cout, hw in SHAPES:
    x = rs.randn(B, cin, hw, hw).astype(np.float16)
    w = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float16)
    bias = np.zeros(cout, np.float32)
    res = rs.randn(B, cout, hw, hw).astype(np.float16)
    flop = 2.0 * B * hw * hw * cin * cout
    row = []
    ref = None
    for code in CODES:
        out, ms = _lib.conv2d(x, w, bias, res, tile=code, iters=30)
        if ref is None:
            ref = out.astype(np.float32)
        err = float(np.abs(out.astype(np.float32) - ref).max())
        row.append(f"{code}:{ms * 1e3:6.1f} ({flop / ms / 1e9:5.0f}){'!' if err > 0.05 else ''}")
    print(f"{cin:5d}->{cout:5d} @{hw:2d} M={B * hw * hw:6d}  " + "  ".join(row), flush=True)
