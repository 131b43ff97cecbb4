#!/usr/bin/env python
"""Per-kernel micro-benchmarks through the C ABI (HIP-event timing inside the library):
every heavy conv / GEMM shape of the SD2.1-base UNet under each tile /
split-K configuration, the attention shapes under the three schedules, and the norms.
Writes gpurun_out/microbench.json (per-shape best config -> tuning table + per-kernel roofline)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import _lib  # noqa: E402

B = 2
# (k, stride, Cin, Cout, H, W, upsample, count)   H,W = input spatial (pre-upsample)
CONVS = [
    (3, 1, 320, 320, 64, 64, 0, 7), (3, 1, 640, 320, 64, 64, 0, 2), (3, 1, 960, 320, 64, 64, 0, 1),
    (3, 1, 640, 640, 32, 32, 1, 1), (3, 1, 640, 640, 32, 32, 0, 6), (3, 1, 320, 640, 32, 32, 0, 1),
    (3, 1, 1280, 640, 32, 32, 0, 1), (3, 1, 1920, 640, 32, 32, 0, 1), (3, 1, 960, 640, 32, 32, 0, 1),
    (3, 1, 1280, 1280, 16, 16, 1, 1), (3, 1, 1280, 1280, 16, 16, 0, 7), (3, 1, 2560, 1280, 16, 16, 0, 2),
    (3, 1, 1920, 1280, 16, 16, 0, 1), (3, 1, 640, 1280, 16, 16, 0, 1),
    (3, 1, 1280, 1280, 8, 8, 0, 11), (3, 1, 2560, 1280, 8, 8, 0, 3), (3, 1, 1280, 1280, 8, 8, 1, 1),
    (3, 2, 320, 320, 64, 64, 0, 1), (3, 2, 640, 640, 32, 32, 0, 1), (3, 2, 1280, 1280, 16, 16, 0, 1),
    (1, 1, 320, 320, 64, 64, 0, 40), (1, 1, 1280, 320, 64, 64, 0, 5), (1, 1, 640, 320, 64, 64, 0, 2),
    (1, 1, 640, 640, 32, 32, 0, 40), (1, 1, 2560, 640, 32, 32, 0, 5),
    (1, 1, 1280, 1280, 16, 16, 0, 40), (1, 1, 5120, 1280, 16, 16, 0, 5),
    (1, 1, 1280, 1280, 8, 8, 0, 8), (1, 1, 5120, 1280, 8, 8, 0, 1), (1, 1, 2560, 1280, 8, 8, 0, 3),
]
GEGLU = [(8192, 320, 5), (2048, 640, 5), (512, 1280, 5), (128, 1280, 1)]
ATTN = [(5, 4096, 4096, 5), (5, 4096, 77, 5), (10, 1024, 1024, 5), (10, 1024, 77, 5), (20, 256, 256, 5),
        (20, 256, 77, 5), (20, 64, 64, 1), (20, 64, 77, 1)]


def main():
    out = {"conv": [], "geglu": [], "attention": [], "norm": []}
    rs = np.random.RandomState(0)
    t_start = time.time()
    for (k, s, cin, cout, h, w, up, count) in CONVS:
        x = rs.randn(B, cin, h, w).astype(np.float16)
        wt = (rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k)).astype(np.float16)
        bias = np.zeros(cout, np.float32)
        ho = (h * (2 if up else 1) + 2 * (k // 2) - k) // s + 1
        flop = 2.0 * B * ho * ho * cout * cin * k * k
        m = B * ho * ho
        res = {}
        splitks = [0, 1, 2] if m >= 2048 else [1, 2, 4, 8, 16]
        tiles = (1, 2, 3, 4, 7) if (k == 3 and s == 1 and not up and w >= 8) else (1, 2, 3, 4)
        if m <= 2048 and k == 3:
            tiles = tiles + (21, 23, 24, 33)      # LDS-DMA rings (3 / 4 stages) for the weight-streaming levels
        for tile in tiles:
            for sk in splitks:
                try:
                    _, ms = _lib.conv2d(x, wt, bias, None, stride=s, upsample=bool(up), tile=tile, splitk=sk, iters=10)
                    res[f"t{tile}_k{sk}"] = ms
                except Exception as e:  # noqa: BLE001
                    res[f"t{tile}_k{sk}"] = str(e)
        _, ms_auto = _lib.conv2d(x, wt, bias, None, stride=s, upsample=bool(up), iters=10)
        good = {kk: v for kk, v in res.items() if isinstance(v, float)}
        best = min(good, key=good.get)
        out["conv"].append(dict(k=k, stride=s, cin=cin, cout=cout, h=h, w=w, up=up, count=count, M=m, gflop=flop / 1e9,
                                auto_ms=ms_auto, auto_tflops=flop / ms_auto / 1e9, best=best, best_ms=good[best],
                                best_tflops=flop / good[best] / 1e9, all=res))
        print(f"conv k{k}s{s} {cin}->{cout} @{h}x{w} up{up}: auto {ms_auto:.4f} ms ({flop / ms_auto / 1e9:.0f} TF) "
              f"best {best} {good[best]:.4f} ms ({flop / good[best] / 1e9:.0f} TF)", flush=True)
    for (m, c, count) in GEGLU:
        x = rs.randn(m, c).astype(np.float16)
        wt = (rs.randn(8 * c, c) / np.sqrt(c)).astype(np.float16)
        _, ms = _lib.geglu(x, wt, np.zeros(8 * c, np.float32), iters=10)
        flop = 2.0 * m * c * 8 * c
        out["geglu"].append(dict(M=m, C=c, count=count, ms=ms, tflops=flop / ms / 1e9, gflop=flop / 1e9))
        print(f"geglu {m}x{c}: {ms:.4f} ms ({flop / ms / 1e9:.0f} TF)", flush=True)
    for (heads, sq, sk, count) in ATTN:
        q = rs.randn(B, heads * 64, 1, sq).astype(np.float16)
        kk = rs.randn(B, heads * 64, 1, sk).astype(np.float16)
        v = rs.randn(B, heads * 64, 1, sk).astype(np.float16)
        flop = 4.0 * B * heads * 64 * sq * sk
        row = dict(heads=heads, sq=sq, sk=sk, count=count, gflop=flop / 1e9)
        for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
            _, ms = _lib.attention(impl, q, kk, v, heads, 64, iters=10)
            row[impl] = ms
            row[impl + "_tflops"] = flop / ms / 1e9
        out["attention"].append(row)
        print(f"attn h{heads} {sq}x{sk}: " + " ".join(f"{i} {row[i]:.4f} ms ({row[i + '_tflops']:.0f} TF)"
                                                      for i in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2")), flush=True)
    for (c, hw) in [(320, 64), (640, 32), (1280, 16), (1280, 8), (960, 64), (2560, 16)]:
        x = rs.randn(B, c, hw, hw).astype(np.float16)
        _, ms = _lib.groupnorm(x, np.ones(c, np.float32), np.zeros(c, np.float32), silu=True, iters=20)
        gb = 2 * x.nbytes / 1e9
        out["norm"].append(dict(kind="groupnorm_silu", C=c, HW=hw, ms=ms, gbps=gb / (ms * 1e-3)))
        print(f"groupnorm {c}@{hw}: {ms:.4f} ms ({gb / (ms * 1e-3):.0f} GB/s alg.)", flush=True)
    for (c, s) in [(320, 4096), (640, 1024), (1280, 256)]:
        x = rs.randn(B, c, 1, s).astype(np.float16)
        _, ms = _lib.layernorm(x, np.ones(c, np.float32), np.zeros(c, np.float32), iters=20)
        gb = 2 * x.nbytes / 1e9
        out["norm"].append(dict(kind="layernorm", C=c, S=s, ms=ms, gbps=gb / (ms * 1e-3)))
        print(f"layernorm {c}x{s}: {ms:.4f} ms ({gb / (ms * 1e-3):.0f} GB/s alg.)", flush=True)
    # tuned table for csrc/tuned_convs.inc: {ksize, stride, up, ctot, n, m, tile, splitk}
    lines = []
    for r in out["conv"]:
        t, k_ = r["best"].split("_")
        if r["best_ms"] < 0.97 * r["auto_ms"]:
            lines.append("{%d, %d, %d, %d, %d, %d, %d, %d},  // %.4f -> %.4f ms" % (
                r["k"], r["stride"], 2 if r["up"] else 1, r["cin"], r["cout"], r["M"], int(t[1:]), int(k_[1:]),
                r["auto_ms"], r["best_ms"]))
    with open(os.path.join(ROOT, "gpurun_out", "tuned_convs.inc"), "w") as f:
        f.write("\n".join(lines) + "\n")
    # roll-up: predicted step time from the per-kernel bests
    conv_auto = sum(r["auto_ms"] * r["count"] for r in out["conv"])
    conv_best = sum(r["best_ms"] * r["count"] for r in out["conv"])
    geglu = sum(r["ms"] * r["count"] for r in out["geglu"])
    attn = {i: sum(r[i] * r["count"] for r in out["attention"]) for i in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2")}
    out["rollup_ms"] = dict(conv_auto=conv_auto, conv_best=conv_best, geglu=geglu, attention=attn)
    out["seconds"] = time.time() - t_start
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["rollup_ms"]))


if __name__ == "__main__":
    main()
