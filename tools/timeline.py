#!/usr/bin/env python
"""Timeline of one denoising step from a rocprofv3 rocpd database: per-kernel busy time, the
gaps between consecutive kernels, and the per-kernel-type breakdown of the step.
usage: tools/timeline.py <bench_results.db> [step_index_from_end]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"_ZN2sd12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", n)
    return m.group(1) if m else n.replace("sd::", "")


# a step starts at loop_prep_kernel
starts = [i for i, r in enumerate(rows) if "loop_prep" in r[0]]
i0, i1 = starts[-back - 1], starts[-back]
step = rows[i0:i1]
t0, t1 = step[0][1], rows[i1][1]
busy = sum(e - s for _, s, e in step)
gaps = [step[i + 1][1] - step[i][2] for i in range(len(step) - 1)] + [t1 - step[-1][2]]
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, "
      f"gaps {sum(gaps) / 1e3:.1f} us (mean {sum(gaps) / len(gaps):.0f} ns, max {max(gaps)} ns)")
agg = defaultdict(lambda: [0, 0, 0])
for (n, s, e), g in zip(step, gaps):
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
    a[2] += g
print(f"{'kernel':58s} {'calls':>5s} {'busy us':>9s} {'avg us':>7s} {'gap-after us':>12s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:58s} {a[0]:5d} {a[1] / 1e3:9.1f} {a[1] / a[0] / 1e3:7.2f} {a[2] / 1e3:12.1f}")
if len(sys.argv) > 3:
    for (n, s, e), g in zip(step, gaps):
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.2f} gap {g / 1e3:6.2f}  {short(n)}")
