#!/usr/bin/env python
"""Step time of the SD2.1-base loop with and without torch's bundled HIP runtime in the process
(SD_MI355X_NO_TORCH=1 -> the library binds /opt/rocm's libamdhip64 instead)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint, schedulers
MODEL = "stabilityai/stable-diffusion-2-1-base"
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
m = HipModel(MODEL, ck, batch=2, attention_implementation=os.environ.get("ATT", "ORIGINAL"), device=0)
ehs = np.random.RandomState(94).randn(2, 1024, 1, 77).astype(np.float16)
lat = np.random.RandomState(93).randn(1, 4, 64, 64).astype(np.float32)
sch = schedulers.DDIMScheduler()
for n in (3, 10, 10):
    sch.set_timesteps(n)
    ts, coef, hist = sch.device_tables()
    t0 = time.perf_counter()
    out, ev = m.denoise_loop(lat, ts, coef, 7.5, history=hist, encoder_hidden_states=ehs)
    dt = time.perf_counter() - t0
print(f"torch loaded: {'torch' in sys.modules}; wall {dt/n*1e3:.3f} ms/step, event median {float(np.median(ev)):.3f} ms/step")
import ctypes
print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][:1])
