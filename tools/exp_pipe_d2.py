#!/usr/bin/env python
"""Experiment build (NOT the product library): the software-pipelined 1x1 GEMM kernel with a 2-STAGE ring, whose 64-KB LDS
footprint lets two workgroups share a CU on the 128 x 128 tile (three on 128 x 64) - the question Finding 5 of DESIGN.md leaves
open: every one-workgroup-per-CU GEMM variant gets ~25 GB/s of LDS fill per CU, the two-workgroup halo conv ~50.
Copies csrc/ to a scratch directory, patches igemm.hip there (plan code staging 8 -> gemm_pipe_kernel<..., D = 2, ...> with
__launch_bounds__(256, 2)), and builds ml-stable-diffusion_amd/lib_exp/libsdmi355.so.  Use it through SD_MI355X_LIB=<that file>
(tools/gemm_pipe_bench.py with EXP_CODES=81,82,84).  The product source is not touched: its binary stays byte-identical."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ml-stable-diffusion_amd", "csrc")
OUT = os.path.join(ROOT, "ml-stable-diffusion_amd", "lib_exp")
WORK = "/tmp/sd_exp_pipe_d2"
shutil.rmtree(WORK, ignore_errors=True)
os.makedirs(os.path.join(WORK, "ml-stable-diffusion_amd"))
shutil.copytree(SRC, os.path.join(WORK, "ml-stable-diffusion_amd", "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(WORK, "include"))
p = os.path.join(WORK, "ml-stable-diffusion_amd", "csrc", "igemm.hip")
s = open(p).read()
old = "template <int BM, int BN, int WGM, int WGN, int D, bool LNF>\n__global__ __launch_bounds__(256) void gemm_pipe_kernel(IgemmArgs a) {"
assert old in s
s = s.replace(old, "template <int BM, int BN, int WGM, int WGN, int D, bool LNF>\n__global__ __launch_bounds__(256, D == 2 ? 2 : 1) void "
                   "gemm_pipe_kernel(IgemmArgs a) {")
old = "  if ((staging == 6 || staging == 7) && gemm_pipe_ok(a)) {"
assert old in s
s = s.replace(old, "  if (staging == 8 && gemm_pipe_ok(a)) {\n    launch_pipe<BM, BN, WGM, WGN, 2, LNF>(a, s);\n    return;\n  }\n" + old)
open(p, "w").write(s)
subprocess.run(["make", "-C", os.path.join(WORK, "ml-stable-diffusion_amd", "csrc"), "-j", "8"], check=True, stdout=subprocess.DEVNULL)
os.makedirs(OUT, exist_ok=True)
shutil.copy(os.path.join(WORK, "ml-stable-diffusion_amd", "lib", "libsdmi355.so"), os.path.join(OUT, "libsdmi355.so"))
print("built", os.path.join(OUT, "libsdmi355.so"))
