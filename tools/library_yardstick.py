#!/usr/bin/env python
"""Yardstick only (never linked into the product): what the vendor libraries reach on the hottest shapes of the
SD2.1-base step - hipBLASLt through torch.matmul for the 1x1 / GEGLU GEMMs, MIOpen through F.conv2d (channels_last
fp16) for the 3x3 convolutions - warm (back-to-back launches) and cold (a 512-MiB fill between launches, the state a
kernel finds inside the step).  SURVEY.md section 8(d): "report vs measured hipBLASLt peak".
usage: library_yardstick.py [out.json]"""
import json
import sys

import torch
import torch.nn.functional as F

dev = torch.device("cuda")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn, cold, iters=20):
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(iters):
        if cold:
            flush.fill_(i & 0xFF)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters * 1e3   # us


rows = []
GEMMS = [("gemm1x1 1280->1280 @16x16", 512, 1280, 1280), ("gemm1x1 640->640 @32x32", 2048, 640, 640),
         ("gemm1x1 320->320 @64x64", 8192, 320, 320), ("geglu 320->2560 @64x64 (GEMM part)", 8192, 2560, 320),
         ("geglu 640->5120 @32x32 (GEMM part)", 2048, 5120, 640), ("geglu 1280->10240 @16x16 (GEMM part)", 512, 10240, 1280),
         ("gemm1x1 5120->1280 @16x16", 512, 1280, 5120), ("gemm1x1 1280->320 @64x64", 8192, 320, 1280)]
for name, m, n, k in GEMMS:
    x = torch.randn(m, k, device=dev, dtype=torch.float16)
    w = torch.randn(n, k, device=dev, dtype=torch.float16)
    fn = lambda: F.linear(x, w)
    flop = 2.0 * m * n * k
    r = {"op": name, "M": m, "N": n, "K": k, "library": "hipBLASLt via torch F.linear"}
    for cold in (False, True):
        us = timed(fn, cold)
        r["cold_us" if cold else "warm_us"] = round(us, 2)
        r["cold_tflops" if cold else "warm_tflops"] = round(flop / us / 1e6, 1)
    rows.append(r)
CONVS = [("conv3x3 320->320 @64x64", 2, 320, 64, 320), ("conv3x3 640->640 @32x32", 2, 640, 32, 640),
         ("conv3x3 1280->1280 @16x16", 2, 1280, 16, 1280), ("conv3x3 1280->1280 @8x8", 2, 1280, 8, 1280)]
for name, b, cin, hw, cout in CONVS:
    x = torch.randn(b, cin, hw, hw, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    fn = lambda: F.conv2d(x, w, padding=1)
    flop = 2.0 * b * hw * hw * cout * cin * 9
    r = {"op": name, "M": b * hw * hw, "N": cout, "K": cin * 9, "library": "MIOpen via torch F.conv2d (channels_last fp16)"}
    for cold in (False, True):
        us = timed(fn, cold)
        r["cold_us" if cold else "warm_us"] = round(us, 2)
        r["cold_tflops" if cold else "warm_tflops"] = round(flop / us / 1e6, 1)
    rows.append(r)
for r in rows:
    print(f"{r['op'][:44]:44s} warm {r['warm_us']:8.1f} us {r['warm_tflops']:7.1f} TF | cold {r['cold_us']:8.1f} us {r['cold_tflops']:7.1f} TF")
if len(sys.argv) > 1:
    json.dump({"note": "vendor-library yardstick, not part of the product (libsdmi355 links libamdhip64 only)", "rows": rows},
              open(sys.argv[1], "w"), indent=1)
