#!/usr/bin/env python
"""UNet-forward times of the other BASELINE configurations (random-init weights of the real
architectures, HIP-event time per graph-replayed forward, CFG batch 2):
SD2.1-base at 512^2 and 768^2 latents, SDXL-base 768^2, SD1.5 control-UNet + ControlNet 512^2."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint  # noqa: E402
from python_hip_stable_diffusion.hip_model import normalize_unet_config  # noqa: E402

rs = np.random.RandomState(0)
out = []


def run(name, model_id, lat, kind="unet", impl="SPLIT_EINSUM", extra=None):
    cfg = normalize_unet_config(model_id)
    shapes = checkpoint.controlnet_param_shapes(cfg) if kind == "controlnet" else checkpoint.unet_param_shapes(cfg)
    ck = checkpoint.random_checkpoint(shapes, seed=1)
    nparam = sum(int(np.prod(s)) for s in shapes.values())
    m = HipModel(cfg if extra else model_id, ck, kind=kind, batch=2, latent_height=lat, latent_width=lat,
                 attention_implementation=impl)
    del ck
    kw = {}
    for k, v in m.expected_inputs.items():
        a = rs.randn(*v["shape"]).astype(np.float32)
        if k == "timestep":
            a = np.full(v["shape"], 500.0, np.float32)
        if k.startswith("additional_residual"):
            a *= 0.1
        kw[k] = a.astype(np.float16)
    y = m(**kw)
    assert all(np.isfinite(v).all() for v in y.values())
    row = {"config": name, "params_M": round(nparam / 1e6, 1), "latents": lat, "hbm_GB": round(m.device_bytes / 1e9, 2)}
    for i in (["ORIGINAL", "SPLIT_EINSUM"] if kind == "unet" else [impl]):
        m.set_attention_implementation(i)
        m(**kw)
        row[i + "_ms"] = round(m.time_forward(2, 10), 3)
    m.close()
    out.append(row)
    print(json.dumps(row), flush=True)


run("SD2.1-base UNet 512x512", "stabilityai/stable-diffusion-2-1-base", 64)
run("SD2.1-base UNet 768x768", "stabilityai/stable-diffusion-2-1-base", 96)
if "--quick" not in sys.argv:
    from python_hip_stable_diffusion.hip_model import UNET_CONFIGS  # noqa: E402
    ctrl = dict(UNET_CONFIGS["runwayml/stable-diffusion-v1-5"], support_controlnet=True)
    run("SD1.5 control-UNet 512x512 (13 residual inputs)", ctrl, 64, extra=True)
    run("SD1.5 ControlNet 512x512", UNET_CONFIGS["runwayml/stable-diffusion-v1-5"], 64, kind="controlnet", impl="ORIGINAL", extra=True)
    run("SDXL-base UNet 768x768", "stabilityai/stable-diffusion-xl-base-1.0", 96)
    run("SDXL-refiner UNet 768x768", "stabilityai/stable-diffusion-xl-refiner-1.0", 96)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "model_bench.json"), "w"), indent=1)
