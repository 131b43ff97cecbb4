#!/usr/bin/env python
"""1x1 GEMMs with the weights global -> VGPR (bvgemm.hip, plan tile 11) against the library's tiled kernels (best plan code per
shape) and hipBLASLt (torch F.linear, yardstick only), stand-alone, back-to-back launches.  usage: r6_bvgemm_bench.py [out.txt]"""
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


def best(fn, n=3):
    return min(fn() for _ in range(n))


say("1x1 GEMM bias + residual, fp16; us (TFLOP/s): tiled = best of the igemm / gemm_pipe plan codes | bvgemm 64 rows x 8 waves x 32 columns | 128 x 8 x 32 | 128 x 4 x 64 (two workgroups per CU) | hipBLASLt (no epilogue)")
rs = np.random.RandomState(0)
CODES = [0, 1, 61, 81, 2, 62, 82, 4, 64, 84]
for cin, cout in ((1280, 1280), (5120, 1280), (1280, 2560), (640, 2560), (2560, 1280)):
    for m in (512, 2048, 4096, 8192, 16384, 32768):
        hw = int(round((m // 2) ** 0.5))
        if 2 * hw * hw != m:
            continue
        x = rs.randn(2, cin, hw, hw).astype(np.float16)
        w = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float16)
        res = rs.randn(2, cout, hw, hw).astype(np.float16)
        bias = np.zeros(cout, np.float32)
        flop = 2.0 * m * cin * cout
        tiled = min(_lib.conv2d(x, w, bias, res, tile=c, iters=20)[1] for c in (CODES + [3, 23, 33, 43, 63, 73, 83] if m < 8192 else CODES))
        b64 = best(lambda: _lib.conv2d(x, w, bias, res, tile=111, iters=20)[1])
        b128 = best(lambda: _lib.conv2d(x, w, bias, res, tile=112, iters=20)[1])
        b264 = best(lambda: _lib.conv2d(x, w, bias, res, tile=113, iters=20)[1])
        xt = torch.randn(m, cin, device=dev, dtype=torch.float16)
        wt = torch.randn(cout, cin, device=dev, dtype=torch.float16)
        lib_us = torch_us(lambda: F.linear(xt, wt))
        tf = lambda ms: flop / (ms * 1e-3) / 1e12  # noqa: E731
        say(f"  {cin:5d}->{cout:5d} M={m:6d}: tiled {tiled * 1e3:7.1f} ({tf(tiled):5.0f}) | bv64 {b64 * 1e3:7.1f} ({tf(b64):5.0f}) | bv128 {b128 * 1e3:7.1f} ({tf(b128):5.0f}) | bv128x64 {b264 * 1e3:7.1f} ({tf(b264):5.0f})"
            f" | hipBLASLt {lib_us:7.1f} ({flop / lib_us / 1e6:5.0f})")
say("GEGLU with LayerNorm fold: tiled | weight-stationary (K = 320 only) | bvgemm 64 | 128 rows; us (TFLOP/s)")
for c, n2 in ((320, 2560), (640, 5120), (1280, 10240)):
    for m in (512, 2048, 8192, 32768):
        if m * n2 > 2 ** 29:
            continue
        x = rs.randn(m, c).astype(np.float16)
        w = (rs.randn(n2, c) / np.sqrt(c)).astype(np.float16)
        bias = (0.1 * rs.randn(n2)).astype(np.float32)
        lw = (1 + 0.1 * rs.randn(c)).astype(np.float32)
        lb = (0.1 * rs.randn(c)).astype(np.float32)
        flop = 2.0 * m * c * n2
        row = f"  {c:5d}->{n2:5d} M={m:6d}:"
        for name, k in (("tiled", 1), ("ws", 2), ("bv64", 4), ("bv128", 5), ("bv128x64", 6)):
            if k == 2 and (c != 320 or m < 2048):
                row += "  ws       -      "
                continue
            ms = best(lambda: _lib.geglu_ln(x, w, bias, lw, lb, kernel=k, iters=20)[1])
            row += f"  {name} {ms * 1e3:7.1f} ({flop / (ms * 1e-3) / 1e12:5.0f})"
        say(row)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
