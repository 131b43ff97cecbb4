"""CPU suite: invariants of the compiled gfx950 code that the performance of the hot path depends on (hipcc cross-compiles
without a GPU): no kernel spills to scratch, and the kernels that are meant to run two workgroups per CU fit 256 VGPRs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-stable-diffusion_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def kernel_resources(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DNDEBUG", "--cuda-device-only", "-S", src, "-o", str(out)],
                   check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    res = {}
    text = out.read_text()
    kernel_resources.last_asm = text
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        res[m.group(1)] = (int(re.search(r"next_free_vgpr (\d+)", body).group(1)),
                           int(re.search(r"private_segment_fixed_size (\d+)", body).group(1)))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["igemm.hip", "attention.hip", "norm.hip", "misc.hip"])
def test_no_kernel_spills_and_two_workgroups_per_cu_where_planned(src, tmp_path):
    res = kernel_resources(os.path.join(CSRC, src), tmp_path)
    assert res, "no kernels found"
    spilled = {k: v for k, v in res.items() if v[1] != 0}
    assert not spilled, f"kernels with scratch (register spills): {spilled}"
    if src == "igemm.hip":
        # the K-split halo kernel with rings of <= 4 stages (78 KB of LDS) runs two workgroups per CU: <= 256 VGPRs;
        # so do the 64x64 / 64x128 / 128x64 GEMM tiles with 2-4 stages
        for name, (vgpr, _) in res.items():
            m = re.search(r"conv3x3_halo_ks_kernelILi(\d+)ELi0E", name)
            if m and int(m.group(1)) <= 4:
                assert vgpr <= 256, (name, vgpr)
            if "igemm_kernelILi64ELi64E" in name:
                assert vgpr <= 256, (name, vgpr)


def test_plan_table_rows_are_well_formed_and_unique():
    """csrc/tuned_convs.inc: every row is {kind, ksize, stride, up, Ctot, N, M, tile, staging, splitk} with legal codes, no
    shape key twice (the first match wins in choose_plan, a duplicate would be dead), the K-split halo kernel (tile 7) only on
    3x3 / stride-1 shapes, the software-pipelined GEMM kernel (staging 6 / 7 / 8) only on 1x1 / stride-1 shapes, split-K only where the epilogue kind can split."""
    rows = []
    for line in open(os.path.join(CSRC, "tuned_convs.inc")):
        if line.startswith("{"):
            v = [int(x) for x in re.findall(r"-?\d+", line.split("}")[0])]
            assert len(v) == 10, line
            rows.append(v)
    assert len(rows) >= 40
    # tile 9 (round 5: the weight-streaming kernel, wstream.hip) is only taken when the handle holds the pre-tiled weights;
    # choose_plan skips such a row otherwise, so a row of another tile for the same shape may FOLLOW it as the fallback
    keys = [tuple(r[:7]) for r in rows if r[7] != 9]
    assert len(set(keys)) == len(keys), [k for k in keys if keys.count(k) > 1]
    keys9 = [tuple(r[:7]) for r in rows if r[7] == 9]
    assert len(set(keys9)) == len(keys9)
    for k9 in keys9:
        first9 = next(i for i, r in enumerate(rows) if tuple(r[:7]) == k9 and r[7] == 9)
        assert all(i > first9 for i, r in enumerate(rows) if tuple(r[:7]) == k9 and r[7] != 9), "the fallback row must follow the tile-9 row"
    for kind, ks, st, up, ctot, n, m, tile, staging, sk in rows:
        assert kind in (0, 1, 2, 3) and ks in (1, 3) and st in (1, 2) and up in (1, 2)
        assert ctot % 64 == 0 and n % 4 == 0 and m > 0
        assert tile in (1, 2, 3, 4, 7, 9) and 0 <= staging <= 8 and 0 <= sk <= 16   # tiles 5 / 6 / 8: kernels removed in round 4
        if tile == 9:
            assert kind == 0 and st == 1 and m <= 512 and sk in (0, 1)
        if tile == 7:
            assert ks == 3 and st == 1
        if staging in (6, 7, 8):   # the software-pipelined GEMM kernel: 1x1 / stride 1 only
            assert ks == 1 and st == 1 and up == 1 and tile in (1, 2, 3, 4)
        if kind != 0:
            assert sk in (0, 1) and tile in (1, 2, 3, 4)
        if kind == 2:
            assert tile in (1, 4)
