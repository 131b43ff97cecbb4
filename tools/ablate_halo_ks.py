#!/usr/bin/env python
"""K-split software-pipelined halo kernel (plan tile 7) (the halo kernel it replaced, tile 6, was removed in round 4: profiles/r02_ablate_halo_ks_cold.txt / _warm.txt hold the comparison): resnet conv shapes of
SD2.1-base at batch 2, ring depths, split-K, and the ablation builds (bit 1 no MFMA, 2 no ds_read, 4 no W DMA, 8 no X DMA)."""
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


print("== workgroup-count scaling @64x64, 320 input channels")
for tile in (7, 27, 37):
    row = []
    for cout in (64, 256, 320, 512):
        row.append(f"{cout}: {run(320, cout, 64, tile):6.1f}")
    print(f"tile {tile:2d}  " + "   ".join(row), flush=True)
print("== ablations of tile 37")
for cout in (256, 320):
    row = []
    for bits in (0, 1, 2, 3, 4, 8, 12, 13, 14, 15):
        t = run(320, cout, 64, 37, mode=(33 + bits) if bits else 0)
        row.append(f"{bits:2d}:{t:6.1f}")
    print(f"Cout {cout}: " + "  ".join(row), flush=True)
print("== layer shapes (us): tile/splitk")
for cin, cout, h in ((320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32), (960, 640, 32),
                     (320, 640, 32), (1280, 1280, 16), (2560, 1280, 16), (1920, 1280, 16), (640, 1280, 16)):
    row = []
    for tile, sk in ((27, 1), (37, 1), (37, 2), (37, 4), (37, 5), (47, 1), (47, 2)):
        if sk > cin // 64:
            continue
        t = run(cin, cout, h, tile, splitk=sk)
        row.append(f"{tile}/{sk}:{t:6.1f}")
    fl = 2 * 2 * h * h * cin * cout * 9
    print(f"{cin:4d}->{cout:4d}@{h:2d} ({fl * 1e-9:5.1f} GF)  " + "  ".join(row), flush=True)
