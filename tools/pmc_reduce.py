#!/usr/bin/env python
"""Reduce rocprofv3 --pmc passes of tools/pmc_probe.py (rocpd sqlite) to per-step numbers.
usage: pmc_reduce.py <out.json> <db> [<db> ...]
For every counter found: the sum over the kernels of ONE denoising step (from one loop_prep_kernel to the next,
median over the traced steps) and the same sum per kernel family.  HBM traffic follows
MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB, collected in separate passes; on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled; WRITE_SIZE is taken as reported
(uncalibrated).  Both count memory-side (fabric) requests, Infinity-Cache hits included.
MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel duration x clock), with the clock taken
from GRBM_GUI_ACTIVE / duration when that counter is in the same pass, else 2.4 GHz."""
import hashlib
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

out_path, dbs = sys.argv[1], sys.argv[2:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"_ZN2sd12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", n)
    return m.group(1) if m else n.replace("sd::", "")


result = {"counters": {}, "families": {}}
for path in dbs:
    db = sqlite3.connect(path)
    rows = db.execute("select name, dispatch_id, start, end, counter_name, counter_value from pmc_events order by start").fetchall()
    by_disp = defaultdict(dict)
    meta = {}
    for name, disp, s, e, cn, cv in rows:
        by_disp[disp][cn] = by_disp[disp].get(cn, 0.0) + cv
        meta[disp] = (name, s, e)
    order = sorted(meta, key=lambda d: meta[d][1])
    starts = [i for i, d in enumerate(order) if "loop_prep" in meta[d][0]]
    if len(starts) < 3:
        raise SystemExit(f"{path}: fewer than 3 steps traced")
    a, b = starts[-2], starts[-1]          # the last complete step
    counters = sorted({cn for d in order[a:b] for cn in by_disp[d]})
    for cn in counters:
        per_step = []
        for s0, s1 in zip(starts[:-1], starts[1:]):
            per_step.append(sum(by_disp[d].get(cn, 0.0) for d in order[s0:s1]))
        per_step.sort()
        result["counters"][cn] = {"per_step_median": per_step[len(per_step) // 2], "steps": len(per_step),
                                  "kernels_per_step": b - a}
    fam = defaultdict(lambda: defaultdict(float))
    for d in order[a:b]:
        name, s, e = meta[d]
        f = fam[short(name)]
        f["calls"] += 1
        f["duration_ns"] += e - s
        for cn, cv in by_disp[d].items():
            f[cn] += cv
    for k, f in fam.items():
        result["families"].setdefault(k, {}).update(f)

c = result["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rd = 2.0 * c["FETCH_SIZE"]["per_step_median"] * 1024
    wr = c["WRITE_SIZE"]["per_step_median"] * 1024
    result.update({
        "read_bytes_per_step": rd, "write_bytes_per_step": wr, "bytes_per_step": rd + wr,
        "algorithmic_min_bytes": 1.732e9 + 2.4e9,
        "note": "memory-side (fabric) bytes incl. Infinity-Cache hits; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies "
                "128-B read requests at 64 B), WRITE_SIZE as reported; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                "separate passes over eager launches of the SD2.1-base CFG-batch-2 step (tools/pmc_probe.py); algorithmic "
                "minimum = 1.732 GB of weights + ~2.4 GB of unfused activations (SURVEY.md section 8d)"})
for k, f in result["families"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in f and f.get("duration_ns"):
        clk = f["GRBM_GUI_ACTIVE"] / f["duration_ns"] if f.get("GRBM_GUI_ACTIVE") else 2.4
        f["clock_ghz_from_grbm"] = clk if f.get("GRBM_GUI_ACTIVE") else None
        f["mfma_busy_frac_at_2p4ghz"] = f["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * f["duration_ns"] * 2.4)
# identity of what was profiled (bench.py reports the traffic only for the same library and workload)
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-stable-diffusion_amd", "lib", "libsdmi355.so")
with open(_lib, "rb") as _f:
    result["build_id"] = hashlib.sha256(_f.read()).hexdigest()[:16]
result.update({"model": "sd21-base", "latent": 64, "prompts_per_gpu": 1, "attention": os.environ.get("SD_ATTN", "ORIGINAL")})
json.dump(result, open(out_path, "w"), indent=1)
for k in ("read_bytes_per_step", "write_bytes_per_step", "bytes_per_step"):
    if k in result:
        print(k, f"{result[k] / 1e9:.3f} GB")
for k, f in sorted(result["families"].items(), key=lambda kv: -kv[1].get("duration_ns", 0))[:14]:
    extra = f" mfma_busy {100 * f['mfma_busy_frac_at_2p4ghz']:.1f}%" if "mfma_busy_frac_at_2p4ghz" in f else ""
    print(f"{k[:56]:56s} calls {int(f['calls']):4d} {f['duration_ns'] / 1e3:9.1f} us{extra}")
