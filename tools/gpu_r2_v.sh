#!/bin/bash
# session V: 1024-thread one-launch GroupNorm at the 32x32 level - tests, stand-alone sweep, A/B on the step; SPLIT / V2 benches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "groupnorm or golden_all_attention or full_sd21 or graph_replay" > $OUT/v_tests.txt 2>&1; tail -2 $OUT/v_tests.txt
timeout 120 python tools/gn_sweep.py 2>&1 | grep -v amdgpu.ids | tail -16
for i in 1 2; do
SD_GN_WIDE=0 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
done
timeout 300 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM > $OUT/bench_v_split.log 2>/dev/null; tail -n 1 $OUT/bench_v_split.log | cut -c1-200
timeout 300 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM_V2 --prompts-per-gpu 2 > $OUT/bench_v_v2_b4.log 2>/dev/null; tail -n 1 $OUT/bench_v_v2_b4.log | cut -c1-200
