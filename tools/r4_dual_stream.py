#!/usr/bin/env python
"""Round-4 experiment: is the batch-2 step better run as two concurrent batch-1 steps (one HIP stream per CFG half)?
Two batch-1 SD2.1-base handles (guidance 1 -> one sample per step), each looping on its own stream from its own host
thread, against the batch-2 handle.  Only a measurement: the two handles do not share weights here.
  python tools/r4_dual_stream.py [steps]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint, schedulers  # noqa: E402
from python_hip_stable_diffusion.hip_model import UNET_CONFIGS, normalize_unet_config  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ucfg = normalize_unet_config(dict(UNET_CONFIGS["stabilityai/stable-diffusion-2-1-base"]))
ckpt = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(ucfg), seed=0)
sch = schedulers.DDIMScheduler()
sch.set_timesteps(steps)
ts, coef, hist = sch.device_tables()
rs = np.random.RandomState(0)


def make(batch):
    m = HipModel(ucfg, ckpt, batch=batch, latent_height=64, latent_width=64, device=0)
    ehs = rs.randn(batch, ucfg["cross_attention_dim"], 1, 77).astype(np.float16)
    lat = rs.randn(max(1, batch // 2) if batch > 1 else 1, 4, 64, 64).astype(np.float32)
    g = 7.5 if batch > 1 else 1.0
    return lambda: m.denoise_loop(lat, ts, coef, g, history=hist, encoder_hidden_states=ehs)


def wall(fns, reps=3):
    best = 1e9
    for _ in range(reps):
        th = [threading.Thread(target=f) for f in fns]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3 / steps


b2 = make(2)
b1a, b1b = make(1), make(1)
for f in (b2, b1a, b1b):
    f()
print(f"batch-2 loop            {wall([b2]):.3f} ms/step", flush=True)
print(f"one batch-1 loop        {wall([b1a]):.3f} ms/step", flush=True)
print(f"two batch-1 concurrent  {wall([b1a, b1b]):.3f} ms/step (both halves of one CFG step)", flush=True)
print(f"batch-2 + batch-2       {wall([b2, make(2)]):.3f} ms/step for two prompts", flush=True)
