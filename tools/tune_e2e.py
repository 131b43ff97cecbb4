#!/usr/bin/env python
"""End-to-end plan tuner: a (layer shape -> plan) candidate is accepted only if the GRAPH REPLAY time of the whole UNet step
drops with it.  Per-kernel timings - stand-alone or in sequence - mispredict what a plan change does to the step (round 2:
nine 1x1 plans that each won 20-40 % in the per-op profile made the step 0.11 ms SLOWER together), because a kernel's
duration depends on what its neighbours leave in the caches and on how its tail overlaps the next launch.
usage: SD_TUNE=1 python tools/tune_e2e.py <candidates.json> <out_table.inc> [<out_report.json>] [model] [latent]
candidates.json: {"<kind,ksize,stride,up,Ctot,N,M>": [[tile, staging, splitk], ...], ...} in the order to try them
(tools/shortlist_plans.py builds it from a tools/tune_plans.py report)."""
import ctypes as C
import json
import os
import sys

import numpy as np

assert os.environ.get("SD_TUNE"), "run with SD_TUNE=1 (sizes the split-K workspace for every candidate)"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    sys.path.insert(0, p)
from python_hip_stable_diffusion import HipModel, _lib, checkpoint  # noqa: E402

cands = json.load(open(sys.argv[1]))
out_table = sys.argv[2]
out_report = sys.argv[3] if len(sys.argv) > 3 else None
WHICH = sys.argv[4] if len(sys.argv) > 4 else "sd21"
MODEL = {"sd21": "stabilityai/stable-diffusion-2-1-base", "sdxl": "stabilityai/stable-diffusion-xl-base-1.0",
         "sdxl-refiner": "stabilityai/stable-diffusion-xl-refiner-1.0", "sd15": "runwayml/stable-diffusion-v1-5"}[WHICH]
HW = int(sys.argv[5]) if len(sys.argv) > 5 else (96 if WHICH.startswith("sdxl") else 64)
B = 2
ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(MODEL), seed=0)
m = HipModel(MODEL, ck, batch=B, latent_height=HW, latent_width=HW, attention_implementation="ORIGINAL", use_graph=True)
ctx = m.expected_inputs["encoder_hidden_states"]["shape"][1]
kw = dict(sample=np.random.RandomState(1).randn(B, 4, HW, HW).astype(np.float16), timestep=np.full((B,), 951, np.float16),
          encoder_hidden_states=np.random.RandomState(2).randn(B, ctx, 1, 77).astype(np.float16))
if "time_ids" in m.expected_inputs:
    nid = m.expected_inputs["time_ids"]["shape"][1]
    kw["time_ids"] = np.tile(np.array([[HW * 8, HW * 8, 0, 0, HW * 8, HW * 8][:nid]], np.float16), (B, 1))
    kw["text_embeds"] = np.random.RandomState(3).randn(*m.expected_inputs["text_embeds"]["shape"]).astype(np.float16)
ref = m(**kw)["noise_pred"]
lib = _lib.lib()
handle = m._h if hasattr(m, "_h") else m.handle


def set_table(rows):
    text = "\n".join("{%s, %d, %d, %d}" % (k.replace(",", ", "), *p) for k, p in rows.items())
    n = C.c_int(0)
    _lib.check(lib.sd_tune_set_plan_table(text.encode() if rows else None, handle, C.byref(n)))
    assert n.value == len(rows), (n.value, len(rows))


def step_ms(reps=3, iters=12):
    ts = []
    for _ in range(reps):
        ms = C.c_float(0)
        _lib.check(lib.sd_unet_time_forward(handle, 2, iters, C.byref(ms)))
        ts.append(ms.value)
    return float(np.median(ts))


accepted = {}
set_table(accepted)
best = step_ms(5)
base = best
log = []
print(f"step with the compiled-in table: {base:.4f} ms", flush=True)
for key, plans in cands.items():
    for plan in plans:
        trial = dict(accepted)
        trial[key] = tuple(plan)
        set_table(trial)
        y = m(**kw)["noise_pred"]                # every candidate must compute the same network (also re-captures the graph)
        err = float(np.abs(y - ref).max()) if np.isfinite(y).all() else float("inf")
        if not err < 0.02 * float(np.abs(ref).max()):
            print(f"WRONG RESULT  {key} {plan}: max |y - ref| = {err}", flush=True)
            log.append({"key": key, "plan": plan, "wrong": True, "max_err": err})
            continue
        t = step_ms()
        if t < best - 0.003:
            t2 = step_ms(5)                      # confirm
            y = m(**kw)["noise_pred"]            # and the network must still be the same function
            err = float(np.abs(y - ref).max())
            ok = t2 < best - 0.003 and np.isfinite(y).all() and err < 0.02 * float(np.abs(ref).max())
            log.append({"key": key, "plan": plan, "ms": t2, "best_before": best, "accepted": bool(ok), "max_err": err})
            if ok:
                print(f"{key:32s} {str(plan):14s} {best:.4f} -> {t2:.4f} ms", flush=True)
                accepted, best = trial, t2
        else:
            log.append({"key": key, "plan": plan, "ms": t, "best_before": best, "accepted": False})
set_table(accepted)
final = step_ms(5)
with open(out_table, "w") as f:
    f.write("// Plan table entries accepted END TO END by tools/tune_e2e.py: each lowered the graph-replay time of the whole\n"
            f"// CFG-batch-2 step of {WHICH} at {HW}x{HW} latents ({base:.3f} -> {final:.3f} ms on the tuning box).\n")
    for k, p in accepted.items():
        f.write("{%s, %d, %d, %d},\n" % (k.replace(",", ", "), *p))
print(f"step: {base:.4f} -> {final:.4f} ms with {len(accepted)} entries")
if out_report:
    json.dump({"base_ms": base, "final_ms": final, "accepted": {k: list(v) for k, v in accepted.items()}, "log": log}, open(out_report, "w"))
