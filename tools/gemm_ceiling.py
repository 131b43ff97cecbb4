#!/usr/bin/env python
"""Where do the main loops top out?  The 1x1 GEMM kernels of this library (best plan code per shape out of igemm_kernel's
tiles, gemm_pipe_kernel's rings / tiles) and the K-split halo 3x3 conv against the vendor libraries
(hipBLASLt through torch F.linear, MIOpen through F.conv2d channels_last fp16 - yardsticks, never linked) at M = 128 ... 131 072 rows (round 5: the small-M shapes of the 8x8 / 16x16 / 32x32 levels too), back-to-back launches (operands L2 / Infinity-Cache warm: the ceiling, not the in-sequence time).
usage: gemm_ceiling.py [out.txt]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

dev = torch.device("cuda")
CODES = [0, 1, 61, 81, 2, 62, 82, 4, 64, 84]
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def torch_us(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


say("1x1 GEMM (bias + residual epilogue), fp16: best plan code of libsdmi355 vs hipBLASLt (F.linear, no epilogue), us and TFLOP/s; peak 2500")
rs = np.random.RandomState(0)
for cin, cout in ((320, 320), (640, 640), (1280, 1280), (1280, 320), (2560, 640), (5120, 1280), (320, 2560), (1280, 10240)):
    for m in (128, 512, 2048, 8192, 32768):   # (131 072 rows: round 3, profiles/r03_gemm_ceiling.txt)
        if m < 8192 and not ((m <= 512 and cin >= 1280) or (m == 2048 and cin in (640, 2560)) ):
            continue   # small M: only the shapes the 8x8 / 16x16 / 32x32 levels of the SD2.1 step have
        hw = int(round((m // 2) ** 0.5))
        if 2 * hw * hw != m:
            continue
        x = rs.randn(2, cin, hw, hw).astype(np.float16)
        w = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float16)
        res = rs.randn(2, cout, hw, hw).astype(np.float16)
        flop = 2.0 * m * cin * cout
        best = None
        for code in (CODES + [3, 23, 33, 43, 63, 73, 83, 24, 34] if m < 8192 else CODES):
            _, ms = _lib.conv2d(x, w, np.zeros(cout, np.float32), res, tile=code, iters=20)
            if best is None or ms < best[1]:
                best = (code, ms)
        xt = torch.randn(m, cin, device=dev, dtype=torch.float16)
        wt = torch.randn(cout, cin, device=dev, dtype=torch.float16)
        lib_us = torch_us(lambda: F.linear(xt, wt))
        say(f"  {cin:5d}->{cout:5d} M={m:6d}: ours {best[1] * 1e3:7.1f} us {flop / best[1] / 1e9:6.0f} TF = {flop / best[1] / 1e9 / 2500:.2f} (plan {best[0]:2d})"
            f" | hipBLASLt {lib_us:7.1f} us {flop / lib_us / 1e6:6.0f} TF")
        del xt, wt
say("3x3 conv stride 1 (bias + residual), fp16: conv3x3_halo_ks_kernel (plan from the table / heuristic) vs MIOpen (F.conv2d channels_last)")
for c, hw_list in ((320, (64, 128)), (640, (32, 64)), (1280, (8, 16, 32))):
    for hw in hw_list:
        m = 2 * hw * hw
        x = rs.randn(2, c, hw, hw).astype(np.float16)
        w = (rs.randn(c, c, 3, 3) / np.sqrt(9 * c)).astype(np.float16)
        res = rs.randn(2, c, hw, hw).astype(np.float16)
        flop = 2.0 * m * c * c * 9
        _, ms = _lib.conv2d(x, w, np.zeros(c, np.float32), res, iters=20)
        xt = torch.randn(2, c, hw, hw, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(c, c, 3, 3, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        lib_us = torch_us(lambda: F.conv2d(xt, wt, padding=1), iters=10)
        say(f"  {c:5d}->{c:5d} @{hw:3d}x{hw:<3d} M={m:6d}: ours {ms * 1e3:8.1f} us {flop / ms / 1e9:6.0f} TF = {flop / ms / 1e9 / 2500:.2f}"
            f" | MIOpen {lib_us:8.1f} us {flop / lib_us / 1e6:6.0f} TF")
        del xt, wt
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
