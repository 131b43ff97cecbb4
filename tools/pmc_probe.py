#!/usr/bin/env python
"""Eager (no HIP graph) denoising steps for rocprofv3 --pmc passes.
usage: pmc_probe.py [mini|sd21] [steps]   (SD_LOG_CONVS=1 prints every conv plan before its launch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint, schedulers  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "mini"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if which == "mini":
    cfg = dict(sample_size=16, block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
    hw, ctx = 16, 128
else:
    cfg, hw, ctx = "stabilityai/stable-diffusion-2-1-base", 64, 1024
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(cfg), seed=0)
m = HipModel(cfg, ck, batch=2, attention_implementation=os.environ.get("SD_ATTN", "ORIGINAL"), use_graph=False)
sch = schedulers.DDIMScheduler()
sch.set_timesteps(steps)
ts, coef, hist = sch.device_tables()
lat = np.random.RandomState(93).randn(1, 4, hw, hw).astype(np.float32)
ehs = np.random.RandomState(94).randn(2, ctx, 1, 77).astype(np.float16)
out, ms = m.denoise_loop(lat, ts, coef, 7.5, history=hist, encoder_hidden_states=ehs)
print("pmc_probe", which, "ms/step", [round(float(v), 3) for v in ms], "finite", bool(np.isfinite(out).all()), flush=True)
