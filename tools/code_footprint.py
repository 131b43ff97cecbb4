#!/usr/bin/env python
"""Code bytes of every kernel the step launches, next to its launches per step (LAB_NOTES Finding 14: on the slow boxes of the pool a
launch whose code is not in the instruction caches costs ~0.37 us per KB fetched from L2 - 11 us for 30 KB - against 0.03 us per KB
on the fast ones; the step launches ~30 different kernels one after the other, so almost every launch starts cold).
Compile-only (hipcc cross-compiles without a GPU): each source of csrc/ is compiled to a device object, the FUNC symbol sizes are
read with llvm-readelf and joined with a step timeline (tools/timeline.py output).
usage: tools/code_footprint.py [profiles/r05_final_step_timeline_original.txt] > profiles/r05_code_footprint.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-stable-diffusion_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
         "attention8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-fno-slp-vectorize"],
         "wsgemm.hip": ["-fno-slp-vectorize"], "bvgemm.hip": ["-fno-slp-vectorize"]}
timeline = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_final_step_timeline_original.txt")

sizes = {}
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        obj = os.path.join(tmp, src + ".o")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DNDEBUG", "--cuda-device-only", "--no-gpu-bundle-output", "-c", src, "-o", obj] + EXTRA.get(src, []),
                       check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        out = subprocess.run([READELF, "-sW", obj], capture_output=True, text=True, check=True).stdout
        names, rows = [], []
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC" and f[2].isdigit() and int(f[2]) > 0:
                rows.append((f[7], int(f[2])))
        dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
        for (mangled, size), name in zip(rows, dem):
            if "groupnorm_fused_kernel" in mangled and "Lb1EEEv" in mangled:
                continue   # <.., RES = true>: round 5's register-resident form, launched only under SD_TUNE=1 SD_GN_RESIDENT=1
            m = re.match(r"_ZN2sd12_GLOBAL__N_1(\d+)", name)   # c++filt gives up on _Float16 parameters: the plain name from the mangling
            if m:
                name = name[m.end():m.end() + int(m.group(1))]
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(.*$", "", name).replace("sd::", "")
            sizes[name] = max(size, sizes.get(name, 0))

step = []
for line in open(timeline):
    m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s*$", line)
    if m and not line.startswith("kernel"):
        step.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))))
if not step:
    sys.exit("no kernel rows in " + timeline)

print(f"code bytes per kernel of the step ({os.path.relpath(timeline, ROOT)}), gfx950 device code, compile-only")
print(f"{'kernel':58s} {'launches':>8s} {'avg us':>8s} {'code KB':>8s} {'launches x KB':>14s}")
tot_l = tot_kb = 0
missing, table = [], []
for name, calls, busy, avg in step:
    size = sizes.get(name)
    if size is None:   # the timeline drops the template arguments of some kernels: the largest instantiation of that name
        # (groupnorm_fused_kernel<.., true> is the register-resident A/B variant, launched only under SD_TUNE=1 SD_GN_RESIDENT=1)
        cands = [v for k, v in sizes.items() if k.split("<")[0] == name and not (name == "groupnorm_fused_kernel" and k.endswith("true>"))]
        size = max(cands) if cands else None
    if size is None:
        missing.append(name)
        continue
    table.append((name, calls, avg, size / 1024.0))
for name, calls, avg, kb in sorted(table, key=lambda r: -r[1] * r[3]):
    tot_l += calls
    tot_kb += calls * kb
    print(f"{name[:58]:58s} {calls:8d} {avg:8.2f} {kb:8.1f} {calls * kb:14.1f}")
print(f"{'total':58s} {tot_l:8d} {'':8s} {'':8s} {tot_kb:14.1f}")
print(f"\n{len(step)} kernels, {tot_l} launches, {tot_kb / 1024:.2f} MB of code launched per step (upper bound of what is fetched: a launch runs one path "
      "through its kernel).  At the slow boxes' ~0.37 us per KB of cold code (tools/ubench/icache.hip: +11.3 us for 30 KB, code in L2) that "
      f"bound is {tot_kb * 0.37 / 1e3:.2f} ms per step against {tot_kb * 0.026 / 1e3:.2f} ms on the fast boxes (+0.8 us for 30 KB); the measured gap between the two kinds of box is "
      "0.9-1.15 ms.  The kernels at the top of this table are where a smaller instruction footprint (less unrolling, fewer template "
      "variants per step) buys the most on a slow box.")
if missing:
    print("not found in the objects (name mismatch): " + ", ".join(missing))
