#!/usr/bin/env python
"""HBM traffic of ONE denoising step from rocprofv3 --pmc passes (rocpd databases).
usage: tools/step_traffic.py <fetch_db> <write_db> [out.json]
A step = the kernels from one loop_prep_kernel to the next.  FETCH_SIZE / WRITE_SIZE are in KB;
per MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads, so the read figure is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import sqlite3
import sys


def per_step(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, counter_name, counter_value from pmc_events order by start").fetchall()
    rows = [r for r in rows if r[2] == counter]
    starts = [i for i, r in enumerate(rows) if "loop_prep" in r[0]]
    if len(starts) < 3:
        raise SystemExit(f"{path}: fewer than 3 steps traced")
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        steps.append((b - a, sum(r[3] for r in rows[a:b])))
    return steps


fetch = per_step(sys.argv[1], "FETCH_SIZE")
write = per_step(sys.argv[2], "WRITE_SIZE")
f_kb = sorted(s[1] for s in fetch)[len(fetch) // 2]
w_kb = sorted(s[1] for s in write)[len(write) // 2]
out = {"kernels_per_step": fetch[0][0], "steps_traced": len(fetch),
       "FETCH_SIZE_KB_per_step": f_kb, "WRITE_SIZE_KB_per_step": w_kb,
       "read_bytes_corrected": 2.0 * f_kb * 1024, "write_bytes": w_kb * 1024,
       "traffic_bytes_per_step": 2.0 * f_kb * 1024 + w_kb * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); "
                     "WRITE_SIZE as reported; memory-side requests include Infinity-Cache hits",
       "collection": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace, eager launches"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
