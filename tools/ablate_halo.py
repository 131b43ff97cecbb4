#!/usr/bin/env python
"""What bounds the LDS-halo 3x3 conv?  (1) time vs number of workgroups at fixed per-workgroup work (Cout = 64 * n_tiles:
128 / 256 / 320 / 512 workgroups on 256 CUs) - the cost of the 320-on-256 quantisation; (2) ablation builds of the
BN = 64, 4-stage kernel (no MFMAs / no fragment reads / no weight DMA / no halo DMA)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)


def run(cin, cout, h, tile, mode=0, iters=30, splitk=1):
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    _, ms = _lib.conv2d(x, w, b, None, stride=1, upsample=False, tile=tile, splitk=splitk, force_generic=mode, iters=iters)
    return ms * 1e3


print("== workgroup-count scaling, halo BN=64 (tile 6), 320 input channels @64x64, batch 2: 64 m-tiles x Cout/64")
for st in (0, 2, 3):
    for cout in (64, 128, 192, 256, 320, 384, 512, 640, 1024):
        t = run(320, cout, 64, st * 10 + 6)
        wgs = 64 * cout // 64
        print(f"staging {st} Cout {cout:4d}: {wgs:4d} WGs ({wgs / 256:4.2f}/CU)  {t:7.1f} us   {2 * 8192 * cout * 2880 / t * 1e-6:6.0f} TF", flush=True)
print("== same, K-split pipelined halo (tile 7)")
for st in (0, 2, 3, 4):
    for cout in (64, 256, 320, 512, 640, 1024):
        t = run(320, cout, 64, st * 10 + 7)
        wgs = 64 * cout // 64
        print(f"staging {st} Cout {cout:4d}: {wgs:4d} WGs ({wgs / 256:4.2f}/CU)  {t:7.1f} us   {2 * 8192 * cout * 2880 / t * 1e-6:6.0f} TF", flush=True)
print("== ablations of the K-split pipelined halo, 4 stages (debug bits: 1 no MFMA, 2 no ds_read, 4 no W DMA, 8 no X DMA)")
for cout in (256, 320, 512):
    row = []
    for bits in (0, 1, 2, 3, 4, 8, 12, 13, 14, 15):
        t = run(320, cout, 64, 37, mode=(33 + bits) if bits else 0)
        row.append(f"{bits:2d}:{t:6.1f}")
    print(f"Cout {cout}: " + "  ".join(row), flush=True)
print("== same, halo BN=128 (tile 5)")
for st in (0, 2):
    for cout in (128, 256, 384, 512, 1024):
        t = run(320, cout, 64, st * 10 + 5)
        wgs = 64 * cout // 128
        print(f"staging {st} Cout {cout:4d}: {wgs:4d} WGs ({wgs / 256:4.2f}/CU)  {t:7.1f} us   {2 * 8192 * cout * 2880 / t * 1e-6:6.0f} TF", flush=True)
print("== same, igemm 128x128 (tile 1), 128x64 (tile 2), 64x64 (tile 3), 3-stage ring")
for tile in (21, 22, 23):
    for cout in (128, 256, 320, 512):
        t = run(320, cout, 64, tile)
        print(f"tile {tile} Cout {cout:4d}: {t:7.1f} us   {2 * 8192 * cout * 2880 / t * 1e-6:6.0f} TF", flush=True)
print("== ablations of halo<64, 4 stages> (debug bits: 1 no MFMA, 2 no ds_read, 4 no W DMA, 8 no X DMA)")
for cout in (256, 320, 512):
    row = []
    for bits in (0, 1, 2, 3, 4, 8, 12, 13, 14, 15):
        t = run(320, cout, 64, 36, mode=(33 + bits) if bits else 0)
        row.append(f"{bits:2d}:{t:6.1f}")
    print(f"Cout {cout}: " + "  ".join(row), flush=True)
print("== 640->640 @32 (16 m-tiles) and 1280->1280 @16 (4 m-tiles), halo BN=64 4-stage, split-K sweep")
for cin, h in ((640, 32), (1280, 16)):
    for sk in (1, 2, 4, 5, 10):
        for tile in (36, 26, 35, 37, 27):
            t = run(cin, cin, h, tile, splitk=sk)
            print(f"{cin}@{h} tile {tile} splitk {sk:2d}: {t:7.1f} us   {2 * 2 * h * h * cin * cin * 9 / t * 1e-6:6.0f} TF", flush=True)
