import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "ml-stable-diffusion_amd"))
from python_hip_stable_diffusion import _lib
from oracle import attention_ref, psnr
rs = np.random.RandomState(0)
b, heads, sq, sk, variant, iters = [int(x) for x in sys.argv[1:7]]
q, k, v = (rs.randn(b, heads * 64, 1, n).astype(np.float16) for n in (sq, sk, sk))
ref = attention_ref.original(q.astype(np.float32), k.astype(np.float32), v.astype(np.float32), heads, 64)
for rep in range(3):
    out, ms = _lib.attention("ORIGINAL", q, k, v, heads, 64, variant=variant, iters=iters)
print("ok", sys.argv[1:], "psnr %%.1f ms %%.4f" %% (psnr.compute_psnr(out, ref), ms), flush=True)
""" % (ROOT, ROOT)
for args in [(1, 1, 32, 64, 0, 1), (1, 2, 256, 256, 0, 1), (1, 2, 256, 256, 0, 20), (2, 5, 4096, 4096, 0, 1), (2, 5, 4096, 4096, 0, 20)]:
    for extra in ({}, {"AMD_SERIALIZE_KERNEL": "3"}, {"HIP_LAUNCH_BLOCKING": "1"}, {"SD_ATTN8_WAVES": "8"}, {"SD_ATTN8_WAVES": "4"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", CODE] + [str(a) for a in args], env=env, capture_output=True, text=True)
        print(args, extra, (r.stdout.strip().splitlines() or ["-"])[-1], "|", (r.stderr.strip().splitlines() or ["-"])[-1][:120], flush=True)
