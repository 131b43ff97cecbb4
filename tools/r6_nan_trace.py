#!/usr/bin/env python
"""tiny UNet, eager, one forward: tools/ubench/poison && SD_TUNE=1 SD_NAN_TRACE=1 python tools/r6_nan_trace.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import unet_ref  # noqa: E402  (test infrastructure: configs only)
from python_hip_stable_diffusion import HipModel  # noqa: E402
from python_hip_stable_diffusion.checkpoint import random_checkpoint  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = unet_ref.CONFIGS[name]
sd = random_checkpoint(unet_ref.unet_param_shapes(cfg), seed=0)
m = HipModel(cfg, sd, batch=2, attention_implementation="ORIGINAL", use_graph=False)
sh = {k: v["shape"] for k, v in m.expected_inputs.items()}
rs = np.random.RandomState(1)
kw = {k: rs.randn(*s).astype(np.float16) for k, s in sh.items()}
kw["timestep"] = np.full(sh["timestep"], 500, np.float16)
y = m(**kw)["noise_pred"]
print("finite output:", bool(np.isfinite(y).all()))
