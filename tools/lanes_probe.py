#!/usr/bin/env python
"""Does the step gain from running the CFG halves as two concurrent launch chains?  Three measurements of the
SD2.1-base UNet forward (graph replay): one batch-2 handle, one batch-1 handle alone, two batch-1 handles replayed
concurrently from two host threads on their own streams (wall clock per pair of forwards)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint  # noqa: E402

MODEL = "stabilityai/stable-diffusion-2-1-base"
LAT = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)


def make(b):
    m = HipModel(MODEL, ck, batch=b, latent_height=LAT, latent_width=LAT, attention_implementation="ORIGINAL")
    x = np.random.RandomState(1).randn(b, 4, LAT, LAT).astype(np.float16)
    e = np.random.RandomState(2).randn(b, 1024, 1, 77).astype(np.float16)
    m(sample=x, timestep=np.full((b,), 951, np.float16), encoder_hidden_states=e)
    return m


m2 = make(2)
print(f"batch 2, one chain: {m2.time_forward(3, 30):.3f} ms", flush=True)
a, b = make(1), make(1)
print(f"batch 1 alone:      {a.time_forward(3, 30):.3f} ms", flush=True)
for n in (2, 3, 4):
    ms = [make(1) for _ in range(n - 2)] + [a, b] if n > 2 else [a, b]
    res = [0.0] * n
    bar = threading.Barrier(n + 1)

    def run(i):
        bar.wait()
        res[i] = ms[i].time_forward(3, 60)

    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = (time.perf_counter() - t0) / 63 * 1e3
    print(f"{n} x batch 1 concurrently: per-chain event time {['%.3f' % r for r in res]} ms, wall per round {wall:.3f} ms "
          f"-> {wall / n * 2:.3f} ms per CFG pair", flush=True)
