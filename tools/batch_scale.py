#!/usr/bin/env python
"""UNet-forward time of SD2.1-base against the UNet batch (graph replay, ORIGINAL attention, 64x64 latents): one
launch chain of batch B, and two concurrent chains of batch B/2 each on their own handles / streams (two host
threads).  Prints ms per forward, samples/s and the fraction of the 2.5 PFLOP/s MFMA roof (804.3 GFLOP per sample).
usage: batch_scale.py [out.json] [max_batch]"""
import json
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
OUT = sys.argv[1] if len(sys.argv) > 1 else None
MAXB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
FLOP = 804.3e9


def make(b):
    m = HipModel(MODEL, ck, batch=b, attention_implementation="ORIGINAL")
    x = np.random.RandomState(1).randn(b, 4, 64, 64).astype(np.float16)
    e = np.random.RandomState(2).randn(b, 1024, 1, 77).astype(np.float16)
    m(sample=x, timestep=np.full((b,), 951, np.float16), encoder_hidden_states=e)
    return m


def pair(ms, iters):
    n = len(ms)
    bar = threading.Barrier(n + 1)

    def run(i):
        ms[i].time_forward(2, 1)
        bar.wait()
        ms[i].time_forward(0, iters)

    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return (time.perf_counter() - t0) / iters * 1e3


rows = []
prev = None
b = 1
while b <= MAXB:
    iters = max(8, 80 // b)
    a = make(b)
    one = a.time_forward(2, iters)
    row = {"batch": b, "one_chain_ms": round(one, 3), "samples_per_s": round(b / one * 1e3, 1),
           "mfma_frac": round(b * FLOP / (one * 1e-3) / 2.5e15, 4)}
    if prev is not None:
        c = make(b // 2)
        two = pair([prev, c], iters)
        row.update({"two_chains_of_half_ms": round(two, 3), "two_chains_mfma_frac": round(b * FLOP / (two * 1e-3) / 2.5e15, 4)})
        c.close()
        prev.close()
    rows.append(row)
    print(json.dumps(row), flush=True)
    prev = a
    b *= 2
if OUT:
    json.dump(rows, open(OUT, "w"), indent=1)
