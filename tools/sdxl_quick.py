#!/usr/bin/env python
"""SDXL-base UNet forward at 96x96 latents (BASELINE config 4), graph replay, both GPU-relevant attention schedules."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint
MODEL = "stabilityai/stable-diffusion-xl-base-1.0"
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=1)
m = HipModel(MODEL, ck, batch=2, latent_height=96, latent_width=96, attention_implementation="ORIGINAL")
rs = np.random.RandomState(0)
kw = {}
for k, v in m.expected_inputs.items():
    a = rs.randn(*v["shape"]).astype(np.float32)
    if k == "timestep":
        a = np.full(v["shape"], 500.0, np.float32)
    kw[k] = a.astype(np.float16)
row = {"config": "SDXL-base UNet 768x768 (final plan table)"}
for impl in ("ORIGINAL", "SPLIT_EINSUM"):
    m.set_attention_implementation(impl)
    y = m(**kw)["noise_pred"]
    assert np.isfinite(y).all()
    row[impl + "_ms"] = round(m.time_forward(2, 10), 3)
print(json.dumps(row))
