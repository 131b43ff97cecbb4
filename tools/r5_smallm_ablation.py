#!/usr/bin/env python
"""where do the small-M product kernels spend the time above the bare weight stream?
Cold launches (SD_TUNE=1 SD_BENCH_COLD=1: a 512-MiB fill in front of every timed launch, one event pair per launch) of
  conv3x3 1280->1280 @8x8 (M = 128) and @16x16 (M = 512), gemm1x1 1280->1280 M = 512, gemm1x1 5120->1280 M = 512
on the product plan, its ablation builds, the in-workgroup split-K (KG = 2) and the weight-streaming kernel (tile 9).
Times include the split-K combine launch where the plan has one (that is what the step pays)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
assert os.environ.get("SD_TUNE") and os.environ.get("SD_BENCH_COLD"), "run with SD_TUNE=1 SD_BENCH_COLD=1"
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
IT = 15


def conv(cin, cout, h, k, tile, splitk=0, mode=0):
    x = rs.randn(2, cin, h, h).astype(np.float16)
    w = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float16)
    b = np.zeros(cout, np.float32)
    r = rs.randn(2, cout, h, h).astype(np.float16)
    _, ms = _lib.conv2d(x, w, b, r, tile=tile, splitk=splitk, force_generic=mode, iters=IT)
    return ms * 1e3


def row(label, items):
    print(f"{label:44s} " + "  ".join(f"{n}:{t:6.1f}" for n, t in items), flush=True)


print("all times us per launch (+ combine launch), cold; wstream bound for 29.5 MB: 8.6-9.4 us (profiles/r04_ubench_weight_stream_cold.txt)")
for h, sk in ((8, 5), (16, 3)):
    m = 2 * h * h
    print(f"== conv3x3 1280->1280 @{h}x{h} (M = {m}); product plan in round 4: halo-ks 6-stage ring, split-K {sk}")
    row("halo-ks 6-stage, split-K 1/2/3/5/10", [(s, conv(1280, 1280, h, 3, 47, s)) for s in (1, 2, 3, 5, 10)])
    row("halo-ks 4-stage, split-K 1/2/3/5/10", [(s, conv(1280, 1280, h, 3, 37, s)) for s in (1, 2, 3, 5, 10)])
    names = {1: "no MFMA", 2: "no frag reads", 4: "no W DMA", 8: "no X DMA", 3: "no MFMA+reads", 12: "no DMA at all", 15: "nothing but barriers"}
    for s in (sk, 1):
        row(f"halo-ks 4-stage split-K {s}, ablations", [("full", conv(1280, 1280, h, 3, 37, s))] +
            [(names[b], conv(1280, 1280, h, 3, 37, s, mode=33 + b)) for b in (1, 2, 4, 8, 3, 12, 15)])
    row("weight-streaming kernel (tile 9): 8 / 4 waves", [("nw8", conv(1280, 1280, h, 3, 9)), ("nw4", conv(1280, 1280, h, 3, 49))])
print("== conv3x3 2560->1280 @8x8")
row("halo-ks 6-stage split-K 5/10 | wstream nw8 / nw4", [("sk5", conv(2560, 1280, 8, 3, 47, 5)), ("sk10", conv(2560, 1280, 8, 3, 47, 10)),
                                                            ("ws8", conv(2560, 1280, 8, 3, 9)), ("ws4", conv(2560, 1280, 8, 3, 49))])
for cin in (1280, 5120):
    print(f"== gemm1x1 {cin}->1280 M = 512")
    row("igemm 64x64: 2-stage / ring3 / ring4 / pipe d2,d3,d4", [(n, conv(cin, 1280, 16, 1, t, 1)) for n, t in
                                                                 (("2st", 3), ("r3", 23), ("r4", 33), ("p2", 83), ("p3", 63), ("p4", 73))])
    row("igemm 64x64 ring4 split-K 1/2/4 (+combine)", [(s, conv(cin, 1280, 16, 1, 33, s)) for s in (1, 2, 4)])
    row("in-workgroup split-K (KG=2) ring3 / ring4", [("r3", conv(cin, 1280, 16, 1, 123, 1)), ("r4", conv(cin, 1280, 16, 1, 133, 1))])
    row("other tiles ring3: 128x128 / 128x64 / 64x128", [(n, conv(cin, 1280, 16, 1, t, 1)) for n, t in (("128x128", 21), ("128x64", 22), ("64x128", 24))])
    # ablation builds of the 2-stage igemm kernel: force_generic = debug + 1
    names = {4: "full+ts", 5: "loads+barriers only", 6: "compute only", 14: "no LDS reads", 22: "no MFMA", 30: "no reads, no MFMA"}
    row("igemm 64x64 2-stage, ablations", [(names[d], conv(cin, 1280, 16, 1, 3, 1, mode=d + 1)) for d in (4, 5, 6, 14, 22, 30)])
    row("gemm_pipe 64x64 d2: full / no W DMA / no X DMA", [("full", conv(cin, 1280, 16, 1, 83, 1)), ("noW", conv(cin, 1280, 16, 1, 83, 1, mode=66)),
                                                          ("noX", conv(cin, 1280, 16, 1, 83, 1, mode=67))])
    row("wstream 1x1 (tile 9) nw8 / nw4", [("nw8", conv(cin, 1280, 16, 1, 9)), ("nw4", conv(cin, 1280, 16, 1, 49))])
print("== mid-size 1x1 GEMMs (M = 8192 / 2048): how much of the pipelined kernel is LDS fill?  gemm_pipe d2: full / no W DMA / no X DMA")
for cin, cout, h in ((320, 320, 64), (1280, 320, 64), (640, 640, 32), (2560, 640, 32), (320, 1280, 64), (320, 2560, 64)):
    for tname, t in (("64x64", 83), ("128x128", 81), ("128x64", 82), ("64x128", 84)):
        row(f"gemm1x1 {cin}->{cout} @{h}x{h} {tname}", [("full", conv(cin, cout, h, 1, t, 1)), ("noW", conv(cin, cout, h, 1, t, 1, mode=66)),
                                                       ("noX", conv(cin, cout, h, 1, t, 1, mode=67))])
