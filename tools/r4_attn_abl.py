#!/usr/bin/env python
"""Ablation of attention8.hip on the self-attention shapes: python tools/r4_attn_abl.py  (one process per setting)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "ml-stable-diffusion_amd"))
from python_hip_stable_diffusion import _lib
rs = np.random.RandomState(0)
for heads, s in ((5, 4096), (10, 1024), (20, 256)):
    q, k, v = (rs.randn(2, heads * 64, 1, s).astype(np.float16) for _ in range(3))
    for variant in (0, 1):
        _, ms = _lib.attention("ORIGINAL", q, k, v, heads, 64, variant=variant, iters=30)
        print(f"S={s} v{variant}: {ms * 1e3:6.1f} us", end="  ")
print()
""" % (ROOT, ROOT)
NAMES = {0: "full", 1: "no exp", 2: "no V reads", 4: "no K reads", 6: "no K/V reads", 8: "no MFMA", 17: "no exp, no row max",
         23: "no exp / max / reads (MFMA + DMA + barriers)", 31: "DMA + barriers only", 100: "ring depth 3", 101: "ring depth 6"}
RUNS = [(0, {}), (0, {"SD_VT_PAD": "0"}), (0, {"SD_ATTN8_STAGGER": "0"}), (0, {"SD_VT_PAD": "0", "SD_ATTN8_STAGGER": "0"}),
        (0, {"SD_ATTN8_WAVES": "8"}), (0, {"SD_ATTN8_WAVES": "4"})]
RUNS += [(a, {"SD_ATTN8_WAVES": "8"}) for a in (1, 2, 4, 6, 8, 17, 23, 31, 100, 101)]
RUNS += [(31, {"SD_VT_PAD": "0", "SD_ATTN8_STAGGER": "0"})]
for abl, extra in RUNS:
    env = dict(os.environ)
    env["SD_TUNE"] = "1"   # the library reads its A/B switches only under SD_TUNE
    env.update(extra)
    if abl:
        env["SD_ATTN8_ABL"] = str(abl)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"abl {abl:3d} {NAMES[abl]:46s} {str(extra):52s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
