#!/usr/bin/env python
"""Measure the plan table of the conv / GEMM kernels IN SEQUENCE: every candidate (tile, LDS-DMA ring depth, split-K)
is forced on all layers at once (sd_tune_set_candidate) and one profiled eager forward of the real SD2.1-base UNet
(sd_unet_profile) times it on every layer shape with the caches as cold as they are inside the step.
usage: SD_TUNE=1 python tools/tune_plans.py <out_table.inc> [<out_report.json>] [batch] [model] [latent_size]
       model: sd21 (default) | sdxl (SDXL-base UNet, text_time inputs) | sd15 (SD1.5 control-UNet + ControlNet shapes)
The table lists, per shape key, the best candidate when it beats the current plan by more than 3 %."""
import json
import os
import sys
from collections import defaultdict

import numpy as np

assert os.environ.get("SD_TUNE"), "run with SD_TUNE=1 (sizes the split-K workspace for every candidate)"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, _lib, checkpoint  # noqa: E402

out_table = sys.argv[1]
out_report = sys.argv[2] if len(sys.argv) > 2 else None
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
WHICH = sys.argv[4] if len(sys.argv) > 4 else "sd21"
MODEL = {"sd21": "stabilityai/stable-diffusion-2-1-base", "sdxl": "stabilityai/stable-diffusion-xl-base-1.0",
         "sd15": "runwayml/stable-diffusion-v1-5"}[WHICH]
HW = int(sys.argv[5]) if len(sys.argv) > 5 else (96 if WHICH == "sdxl" else 64)
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
m = HipModel(MODEL, ck, batch=B, latent_height=HW, latent_width=HW, attention_implementation="ORIGINAL", use_graph=False)
ctx = m.expected_inputs["encoder_hidden_states"]["shape"][1]
x = np.random.RandomState(1).randn(B, 4, HW, HW).astype(np.float16)
e = np.random.RandomState(2).randn(B, ctx, 1, 77).astype(np.float16)
kw = dict(sample=x, timestep=np.full((B,), 951, np.float16), encoder_hidden_states=e)
if "time_ids" in m.expected_inputs:
    kw["time_ids"] = np.tile(np.array([[HW * 8, HW * 8, 0, 0, HW * 8, HW * 8]], np.float16), (B, 1))
    kw["text_embeds"] = np.random.RandomState(3).randn(*m.expected_inputs["text_embeds"]["shape"]).astype(np.float16)
ref = m(**kw)["noise_pred"]
lib = _lib.lib()


def measure(iters=5):
    per = defaultdict(float)
    for lbl, fl, ms in m.profile(iters=iters):
        if "#" in lbl:
            per[lbl.split("#")[1]] += ms
    return per


def set_candidate(t, st, sk):
    _lib.check(lib.sd_tune_set_candidate(t, st, sk))


set_candidate(0, 0, 0)
base = measure(7)
cands = [(t, st, sk) for t in (1, 2, 3, 4) for st in (0, 2, 3, 4) for sk in (1, 2, 4, 8, 16)]
cands += [(5, st, sk) for st in (0, 2, 3) for sk in (1, 2, 4, 8, 16)]          # LDS-halo 3x3 kernel, BN = 128
cands += [(6, st, sk) for st in (0, 2, 3, 4, 5) for sk in (1, 2, 4, 8, 16)]    # ... BN = 64
cands += [(7, st, sk) for st in (2, 3, 4) for sk in (1, 2, 3, 4, 5, 8, 10)]    # ... K-split waves, software-pipelined
if os.environ.get("TUNE_STAGINGS"):                                            # e.g. TUNE_STAGINGS=6,7,8 with TUNE_TILES=1,2,3,4,7
    keep_st = {int(t) for t in os.environ["TUNE_STAGINGS"].split(",")}
    cands = [c for c in cands if c[1] in keep_st or c[0] == 7]
if os.environ.get("TUNE_TILES"):                                               # e.g. TUNE_TILES=7: only the new kernel
    keep = {int(t) for t in os.environ["TUNE_TILES"].split(",")}
    cands = [c for c in cands if c[0] in keep]
results = {}
for c in cands:
    set_candidate(*c)
    y = m(**kw)["noise_pred"]            # every candidate must still compute the same network
    err = float(np.abs(y - ref).max())
    assert err < 0.05 * float(np.abs(ref).max()), (c, err)
    results[c] = measure(3)
set_candidate(0, 0, 0)
lines, report = [], {}
total_base = total_best = 0.0
for key, b_ms in sorted(base.items(), key=lambda kv: -kv[1]):
    best_c, best_ms = None, b_ms
    for c, per in results.items():
        if per[key] < best_ms:
            best_c, best_ms = c, per[key]
    total_base += b_ms
    if best_c is not None and best_ms < 0.97 * b_ms:
        # confirm with a second, longer measurement of the winner
        set_candidate(*best_c)
        again = measure(7)[key]
        set_candidate(0, 0, 0)
        if again < 0.97 * b_ms:
            lines.append("{%s, %d, %d, %d},  // %.1f -> %.1f us per step (in sequence)" % (key.replace(",", ", "), *best_c, b_ms * 1e3, again * 1e3))
            report[key] = {"base_ms": b_ms, "best_ms": again, "plan": best_c}
            total_best += again
            continue
    total_best += b_ms
    report[key] = {"base_ms": b_ms, "best_ms": b_ms, "plan": None}
with open(out_table, "w") as f:
    f.write("// Plan table: {kind, ksize, stride, up, Ctot, N, M, tile, staging, splitk}  (kind / staging: igemm.hip choose_plan,\n"
            "// launch_tile).  Measured IN SEQUENCE on MI355X by tools/tune_plans.py (per-op HIP events of the eager\n"
            "// CFG-batch-%d step of %s at %dx%d latents, caches as cold as in the step): entries beat the previous plan by > 3 %%.\n"
            % (B, WHICH, HW, HW))
    f.write("\n".join(lines) + "\n")
print(f"conv/GEMM ops per step: {total_base:.3f} ms with the current plans -> {total_best:.3f} ms with {len(lines)} new entries")
if out_report:
    json.dump({"total_base_ms": total_base, "total_best_ms": total_best, "per_key": report,
               "all": {",".join(map(str, c)): dict(per) for c, per in results.items()}}, open(out_report, "w"))
