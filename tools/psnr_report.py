#!/usr/bin/env python
"""PSNR of the HIP UNet vs the reference goldens (tracks numerical headroom across kernel changes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from oracle import psnr, unet_ref, weights
from python_hip_stable_diffusion import HipModel
for name in ("mini", "sd21-base"):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"unet_{name}_golden.npz")))
    cfg = unet_ref.CONFIGS[name]
    sd = weights.make_state_dict(unet_ref.unet_param_shapes(cfg), seed=int(g["seed"]), dtype=np.float16)
    m = HipModel(cfg, sd, batch=2, attention_implementation="ORIGINAL")
    del sd
    kw = dict(sample=g["sample"].astype(np.float16), timestep=g["timestep"].astype(np.float16),
              encoder_hidden_states=g["encoder_hidden_states"].astype(np.float16))
    for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
        m.set_attention_implementation(impl)
        y = m(**kw)["noise_pred"]
        print(f"{name} {impl}: PSNR {psnr.compute_psnr(y, g['noise_pred']):.2f} dB, max|err| {np.abs(y - g['noise_pred']).max():.2e}, "
              f"{m.time_forward(2, 10):.3f} ms/forward", flush=True)
    m.close()
