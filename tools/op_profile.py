#!/usr/bin/env python
"""Per-op timing of one SD2.1-base UNet forward (sd_unet_profile: HIP events around every launch-list entry,
eager launches) -> JSON + a table aggregated by op family.
usage: op_profile.py [out.json] [batch] [attention] [model: sd21 | sdxl | sdxl-refiner | sd15] [latent]"""
import json
import os
import re
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
impl = sys.argv[3] if len(sys.argv) > 3 else "ORIGINAL"
WHICH = sys.argv[4] if len(sys.argv) > 4 else "sd21"
MODEL = {"sd21": "stabilityai/stable-diffusion-2-1-base", "sdxl": "stabilityai/stable-diffusion-xl-base-1.0",
         "sdxl-refiner": "stabilityai/stable-diffusion-xl-refiner-1.0", "sd15": "runwayml/stable-diffusion-v1-5"}[WHICH]
HW = int(sys.argv[5]) if len(sys.argv) > 5 else (96 if WHICH.startswith("sdxl") else 64)
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
m = HipModel(MODEL, ck, batch=B, latent_height=HW, latent_width=HW, attention_implementation=impl)
del ck
kw = {}
for k, v in m.expected_inputs.items():
    kw[k] = np.random.RandomState(len(k)).randn(*v["shape"]).astype(np.float16)
kw["timestep"] = np.full((B,), 951, np.float16)
if "time_ids" in kw:
    kw["time_ids"] = np.tile(np.asarray([HW * 8, HW * 8, 0, 0, HW * 8, HW * 8][:kw["time_ids"].shape[1]], np.float16), (B, 1))
m(**kw)
graph_ms = m.time_forward(3, 20)
ops = m.profile(iters=9)
total = sum(o[2] for o in ops)
print(f"graph replay {graph_ms:.3f} ms; {len(ops)} ops, eager per-op sum {total:.3f} ms")
fam = defaultdict(lambda: [0, 0.0, 0.0])
for lbl, fl, ms in ops:
    key = re.sub(r" #[\d,]+$", "", lbl)
    key = re.sub(r" (down_blocks|up_blocks|mid_block|conv_in|conv_out|time_embedding|conv_norm_out)\S*$", "", key)
    f = fam[key]
    f[0] += 1
    f[1] += ms
    f[2] += fl
print(f"{'family':78s} {'n':>3s} {'ms':>8s} {'us/op':>7s} {'TF':>6s} {'%':>5s}")
for k, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{k[:78]:78s} {n:3d} {ms:8.3f} {ms / n * 1e3:7.1f} {tf:6.0f} {100 * ms / total:5.1f}")
if out:
    json.dump({"graph_ms": graph_ms, "batch": B, "attention": impl, "ops": [{"label": l, "flop": f, "ms": t} for l, f, t in ops]},
              open(out, "w"), indent=0)
