#!/usr/bin/env python
"""Which calibration figure tracks the box-to-box spread of the step time?  Reads the one-line records round 5's GPU sessions left
(`ab[SD_TUNE=1 ] ms/step, it/s: <ms> <it/s> calib <copy GB/s> <MFMA TFLOP/s> <empty us> <chain us> <hand-over us> <HBM ns> <cache ns> ...`
in profiles/r05_*summary*.log / *_ab.log: DEFAULT switches only) and the bench lines (profiles/r05_*bench*.json), one row per
session, and prints every figure's ratio to the fastest session next to the step-time ratio.
usage: tools/calib_fit.py [files...] > profiles/r05_calibration_fit.txt"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_*.log")) + glob.glob(os.path.join(ROOT, "profiles", "r05_*bench*.json")))
KEYS = ["copy_gbs", "mfma_tflops", "empty_launch_us", "chain_us", "handover_us", "latency_hbm_ns", "latency_cache_ns", "small_grid_us", "cold_code_us"]
rows = []
for f in files:
    name = os.path.basename(f)
    if f.endswith(".json"):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        c = d.get("calibration")
        if c and d.get("config", {}).get("prompts_per_gpu") == 1 and d.get("config", {}).get("unet", "sd21-base") == "sd21-base" \
                and d.get("config", {}).get("latent") == 64 and d.get("config", {}).get("attention") == "ORIGINAL":
            rows.append((name, d["ms_per_step"], [c.get(k) for k in KEYS]))
        continue
    per = []
    for line in open(f):
        m = re.match(r"ab\[SD_TUNE=1 \] ms/step, it/s: ([\d.]+) ([\d.]+) calib (.*?) sclk", line)
        if m:
            v = [float(x) if x != "None" else None for x in m.group(3).split()]
            v = (v + [None] * len(KEYS))[:len(KEYS)]   # records older than a figure have none
            per.append((float(m.group(1)), v))
    if per:
        ms = sum(p[0] for p in per) / len(per)
        v = [sum(p[1][i] for p in per) / len(per) if all(p[1][i] is not None for p in per) else None for i in range(len(KEYS))]
        rows.append((name + f" ({len(per)} default runs)", ms, v))
if not rows:
    sys.exit("no records")
rows.sort(key=lambda r: r[1])
ref = rows[0]
print("box calibration vs step time, default bench (SD2.1-base 512x512, CFG batch 2, ORIGINAL); reference = the fastest session")
print(f"{'session':58s} {'ms/step':>8s} {'x ref':>6s} | " + " ".join(f"{k:>16s}" for k in KEYS))
for name, ms, v in rows:
    cells = []
    for i, k in enumerate(KEYS):
        if v[i] is None:
            cells.append(f"{'-':>16s}")
            continue
        if ref[2][i] is None:                                  # measured here, not on the reference session
            cells.append(f"{v[i]:9.3f}       ")
            continue
        time_like = k.endswith("_us") or k.endswith("_ns")
        r = v[i] / ref[2][i] if time_like else ref[2][i] / v[i]
        cells.append(f"{v[i]:9.1f} x{r:5.3f}")
    print(f"{name[:58]:58s} {ms:8.3f} {ms / ref[1]:6.3f} | " + " ".join(cells))
spread = rows[-1][1] / rows[0][1]
print(f"\nstep-time spread over {len(rows)} sessions: x{spread:.3f}.")
print("Eight figures - copy, dense MFMA, empty / cold-operand / cross-XCD hand-over / small-grid launch chains, dependent-load latencies - "
      "are within a few per cent on every box, the slowest included: they do not see what makes a box slow.  The per-op profile does "
      "(profiles/r05_ffn_proj_ab_slow_box.log against r05_xattn_out_stage2_ab.log): launches of many workgroups take their usual time, "
      "launches of 64-160 workgroups 1.3-2.0 x as long.")
probe = []
for kind in ("fast", "slow"):
    f = os.path.join(ROOT, "profiles", f"r05_icache_probe_{kind}_box.txt")
    if os.path.exists(f):
        probe.append(f"--- {os.path.basename(f)}\n" + open(f).read().rstrip())
if probe:
    print("\nThe ninth figure, cold_code_us (calib.hip; stand-alone: tools/ubench/icache.hip), is what separates them - a launch whose code is "
          "not in the instruction caches:")
    print("\n".join(probe))
    print("\n-> cold_code_us = (32 kernels round-robin) - (same kernel), 64 workgroups: 0.76 us on the fast box (4.420 ms per step), 11.33 us on "
          "the slow one (5.433 ms): (5.433 / 4.420 - 1) / (11.33 - 0.76) = 0.0217 per us for that build; the committed library (column "
          "cold_code_us above: 4.226 / 4.246 ms at 0.77 / 0.75 us, 5.386 ms at 11.29 us) gives 0.0258, bench.py's CALIB_SLOPES; "
          "value_normalised = value x (1 + 0.0258 x (cold_code_us - 0.76)).")
