#!/usr/bin/env python
"""Race guard for the LDS-DMA ring kernels (counted vmcnt + raw s_barrier): on the FULL SD2.1-base handle, replays of the
captured step graph must be bit-identical to eager launches of the same launch list
  1. 200 times in each attention mode with the compiled-in plan table,
  2. for every ring depth (2 / 3 / 4 / 6 / 8 stages) and split-K factor of the K-split halo conv kernel and of the four
     GEMM tiles (plus the software-pipelined GEMM kernel's 3- / 4-stage rings and its 256x128 tile), forced onto every layer shape that admits them (sd_tune_set_candidate), 25 replays each,
  3. for the 20-step device-resident loop (graph per step vs eager steps).
A dropped round-2 kernel returned wrong results ONLY under graph replay; this is the test that would have caught it
on the shapes that matter.  Needs SD_TUNE=1 in the environment (debug ABI + workspaces sized for every split-K).
usage: SD_TUNE=1 python tools/replay_guard.py [replays_default] [replays_forced] [quick]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

assert os.environ.get("SD_TUNE"), "run with SD_TUNE=1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, _lib, checkpoint, schedulers  # noqa: E402

N_DEFAULT = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N_FORCED = int(sys.argv[2]) if len(sys.argv) > 2 else 25
QUICK = len(sys.argv) > 3
MODEL = "stabilityai/stable-diffusion-2-1-base"
IMPLS = ["ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"]
lib = _lib.lib()
t0 = time.time()
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
g = HipModel(MODEL, ck, batch=2, attention_implementation="ORIGINAL", use_graph=True)
e = HipModel(MODEL, ck, batch=2, attention_implementation="ORIGINAL", use_graph=False)
del ck
kw = dict(sample=np.random.RandomState(1).randn(2, 4, 64, 64).astype(np.float16), timestep=np.full((2,), 951, np.float16),
          encoder_hidden_states=np.random.RandomState(2).randn(2, 1024, 1, 77).astype(np.float16))
failures = []
report = {"default": {}, "forced": []}


def drop_graphs():
    n = C.c_int(0)
    _lib.check(lib.sd_tune_set_plan_table(None, g._h, C.byref(n)))


def compare(tag, replays):
    ref = e(**kw)["noise_pred"]
    assert np.isfinite(ref).all(), tag
    if not np.array_equal(e(**kw)["noise_pred"], ref):
        failures.append(f"{tag}: two eager runs differ")
    bad = 0
    for _ in range(replays):
        y = g(**kw)["noise_pred"]
        if not np.array_equal(y, ref):
            bad += 1
    if bad:
        failures.append(f"{tag}: {bad} of {replays} graph replays differ from the eager launches")
    return bad


for impl in IMPLS:
    g.set_attention_implementation(impl)
    e.set_attention_implementation(impl)
    report["default"][impl] = compare(f"default table, {impl}", N_DEFAULT)
print(f"default table: {report['default']} mismatching replays of {N_DEFAULT} ({time.time() - t0:.0f} s)", flush=True)

g.set_attention_implementation("ORIGINAL")
e.set_attention_implementation("ORIGINAL")
cands = [(7, s, k) for s in (0, 2, 3, 4, 5) for k in (1, 2, 4)]
cands += [(t, s, k) for t in (1, 2, 3, 4) for s in (0, 2, 3, 4, 5, 6, 7) for k in (1, 2, 4)]
cands += [(8, 0, k) for k in (1, 2, 4)] + [(9, 0, k) for k in (1, 2, 4)]
if QUICK:
    cands = cands[::7]
for tile, staging, splitk in cands:
    _lib.check(lib.sd_tune_set_candidate(tile, staging, splitk))
    drop_graphs()
    bad = compare(f"forced plan tile {tile} staging {staging} split-K {splitk}", N_FORCED)
    report["forced"].append([tile, staging, splitk, bad])
_lib.check(lib.sd_tune_set_candidate(0, 0, 0))
drop_graphs()
print(f"forced plans: {len(cands)} candidates x {N_FORCED} replays, {sum(1 for r in report['forced'] if r[3])} with mismatches "
      f"({time.time() - t0:.0f} s)", flush=True)

sch = schedulers.DDIMScheduler()
sch.set_timesteps(20)
ts, coef, hist = sch.device_tables()
lat = np.random.RandomState(93).randn(1, 4, 64, 64).astype(np.float32)
a, _ = g.denoise_loop(lat, ts, coef, 7.5, history=hist, encoder_hidden_states=kw["encoder_hidden_states"])
b, _ = e.denoise_loop(lat, ts, coef, 7.5, history=hist, encoder_hidden_states=kw["encoder_hidden_states"])
report["loop20_equal"] = bool(np.array_equal(a, b))
if not report["loop20_equal"]:
    failures.append("20-step loop: graph-per-step latents differ from eager steps")
report["failures"] = failures
print("REPLAY_GUARD " + json.dumps(report), flush=True)
sys.exit(1 if failures else 0)
