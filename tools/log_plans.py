import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, checkpoint
MODEL = "stabilityai/stable-diffusion-2-1-base"
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
m = HipModel(MODEL, ck, batch=2, use_graph=False)
rs = np.random.RandomState(0)
m(sample=rs.randn(2, 4, 64, 64).astype(np.float16), timestep=np.array([951, 951], np.float16),
  encoder_hidden_states=rs.randn(2, 1024, 1, 77).astype(np.float16))
