"""CPU suite: the host-side bookkeeping of bench.py - kernel-family attribution of the per-op profile, the dominant
kernel by TIME share, and the rule that `roofline.traffic` is only reported from a PMC file measured on this very library
and workload (a committed file from another build must not pass as this run's traffic)."""
import argparse
import json
import os

import bench
from conftest import ROOT

OPS = [
    ("conv3x3 320->320 @64x64 M=8192 K=2880 down_blocks.0.resnets.0.conv1 #0,3,1,1,320,320,8192", 15.1e9, 0.028),
    ("conv3x3 4->320 @64x64 M=8192 K=36 conv_in #0,3,1,1,4,320,8192", 0.2e9, 0.009),
    ("gemm1x1 320->320 @64x64 M=8192 K=320 x.proj_in #0,1,1,1,320,320,8192", 1.7e9, 0.012),
    ("geglu1x1+ln 320->2560 @64x64 M=8192 K=320 x.ff.net.0.proj #2,1,1,1,320,2560,8192", 13.4e9, 0.048),
    ("attention h=5 d=64 Sq=4096 Sk=4096", 43e9, 0.112),
    ("xattn+ln q-proj 320->320 + attention h=5 d=64 Sq=4096 Sk=77 x.attn2", 2.1e9, 0.022),
    ("groupnorm C=320 @64x64 x.norm1", 0.0, 0.010),
    ("conv3x3 small-N conv_out -> fp32 NCHW", 0.19e9, 0.035),
    ("boundary: timestep f16->f32, sample NCHW->NHWC", 0.0, 0.011),
]


def test_kernel_families_and_dominant_kernel_by_time_share():
    fam = {bench.op_family(lbl) for lbl, _, _ in OPS}
    assert len(fam) == 6
    assert bench.op_family(OPS[1][0]).startswith("other") and bench.op_family(OPS[7][0]).startswith("other")
    r = bench.kernel_families(OPS, graph_ms=0.25)
    rows = r["kernel_families"]
    assert [x["family"] for x in rows][0].startswith("self-attention")            # 0.112 ms: the largest TIME share
    assert r["dominant_kernel"]["kernel"] == rows[0]["family"] and abs(sum(x["share"] for x in rows) - 1.0) < 1e-3
    att = rows[0]
    assert att["launches"] == 1 and abs(att["achieved"] - 43e9 / 0.112e-3 / 1e12) < 0.1 and abs(att["frac"] - att["achieved"] / 2500) < 1e-3
    gemm = next(x for x in rows if x["family"].startswith("1x1 GEMMs"))
    assert gemm["launches"] == 2 and abs(gemm["ms"] - 0.060) < 1e-6
    assert r["step_ops"] == len(OPS)


def test_traffic_is_reported_only_for_the_profiled_build_and_workload():
    path = os.path.join(ROOT, "profiles", bench.HBM_TRAFFIC_FILE)
    if not os.path.exists(path):   # no PMC pass committed (yet) for this round's library: nothing may be reported
        args = argparse.Namespace(model="sd21-base", prompts_per_gpu=1, attention="ORIGINAL")
        got = bench.hbm_traffic(5.0, args, 64)
        assert got["traffic"] is None and got["traffic_source"] is None
        return
    t = json.load(open(path))
    for key in ("build_id", "model", "latent", "prompts_per_gpu", "attention", "bytes_per_step", "read_bytes_per_step"):
        assert key in t, key
    args = argparse.Namespace(model=t["model"], prompts_per_gpu=t["prompts_per_gpu"], attention=t["attention"])
    got = bench.hbm_traffic(5.0, args, t["latent"])
    if bench.build_id() == t["build_id"]:            # the committed file belongs to the committed source
        assert got["traffic"] == t["bytes_per_step"] and 0.1 < got["traffic_detail"]["hbm_frac"] < 1.0
        assert "replayed from profiles/" in got["traffic_source"] and "NOT measured by this run" in got["traffic_source"]
    else:
        assert got["traffic"] is None
    for other in (argparse.Namespace(model="sdxl-base", prompts_per_gpu=1, attention="ORIGINAL"),
                  argparse.Namespace(model=t["model"], prompts_per_gpu=2, attention=t["attention"]),
                  argparse.Namespace(model=t["model"], prompts_per_gpu=1, attention="SPLIT_EINSUM")):
        miss = bench.hbm_traffic(5.0, other, t["latent"])
        assert miss["traffic"] is None and "another build / workload" in miss["traffic_detail"]["note"]
    assert bench.hbm_traffic(5.0, args, 96)["traffic"] is None


def test_bench_models_cover_the_baseline_configs():
    assert set(bench.MODELS) == {"sd21-base", "sdxl-base", "sdxl-refiner", "sd15-control"}
    assert bench.MODELS["sd21-base"]["latent"] == 64 and bench.MODELS["sdxl-base"]["latent"] == 96
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "SD2.1-base 512" in base["configs"][1] and "config 2" in bench.MODELS["sd21-base"]["config"]
