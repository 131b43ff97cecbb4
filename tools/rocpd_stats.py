#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg, like --stats.
usage: tools/rocpd_stats.py gpurun_out/prof/bench_results.db [out.csv]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, (end - start) from kernels").fetchall()
agg = {}
for name, dur in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += dur
    a[2] = min(a[2], dur)
    a[3] = max(a[3], dur)
total = sum(a[1] for a in agg.values())
lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,pct"]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{name}",{a[0]},{a[1]},{a[1] / a[0]:.0f},{a[2]},{a[3]},{100.0 * a[1] / total:.2f}')
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
