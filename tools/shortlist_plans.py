#!/usr/bin/env python
"""Shortlist for tools/tune_e2e.py from a tools/tune_plans.py report: per layer shape the `top` candidates with the best
per-op in-sequence times (only those at least 2 % better than the current plan), shapes ordered by time spent."""
import json
import sys

rep = json.load(open(sys.argv[1]))
top = int(sys.argv[3]) if len(sys.argv) > 3 else 4
out = {}
for key, info in sorted(rep["per_key"].items(), key=lambda kv: -kv[1]["base_ms"]):
    kind, ks, st, up, ctot, n, mm = map(int, key.split(","))
    if ctot % 64:
        continue
    scored = []
    for c, per in rep["all"].items():
        tile, stg, sk = map(int, c.split(","))
        if key not in per:
            continue
        if tile == 7 and not (ks == 3 and st == 1):
            continue                      # not legal there: the candidate silently ran the table's plan
        if tile not in (1, 2, 3, 4, 7):
            continue                      # 5 / 6 / 8 / 9: kernels removed in round 4
        if stg >= 6 and tile < 5 and not (ks == 1):
            continue
        if kind != 0 and sk != 1:
            continue                      # LayerNorm-folded / GEGLU / q|k|v GEMMs cannot split K
        if kind == 2 and tile not in (1, 4):
            continue
        scored.append((per[key], [tile, stg, sk]))
    scored.sort()
    keep = [p for t, p in scored if t < 0.98 * info["base_ms"]][:top]
    if keep:
        out[key] = keep
json.dump(out, open(sys.argv[2], "w"), indent=0)
print(len(out), "shapes,", sum(len(v) for v in out.values()), "candidates")
