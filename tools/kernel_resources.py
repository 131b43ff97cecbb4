#!/usr/bin/env python
"""Per-kernel VGPR / scratch / occupancy table of one HIP source (compile-only: runs without a GPU).
usage: tools/kernel_resources.py <file.hip> [regex] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-DNDEBUG",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: \s*([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("sd::(anonymous namespace)::", "").replace("(IgemmArgs)", "")
    if pat and not pat.search(n):
        continue
    print(f"vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>3} spill {r.get('VGPRs Spill', '?'):>3} scratch {r.get('ScratchSize', '?'):>5} "
          f"occ {r.get('Occupancy', '?'):>2} lds {r.get('LDS Size', '?'):>6}  {n}")
