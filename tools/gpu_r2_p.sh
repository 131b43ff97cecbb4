#!/bin/bash
# session P: one-launch GroupNorm (grid barrier) - correctness, determinism tests, A/B on the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "groupnorm or golden_all_attention or graph_replay or batch_four" > $OUT/p_tests.txt 2>&1; tail -3 $OUT/p_tests.txt
timeout 120 python tools/gn_sweep.py 2>&1 | grep -v amdgpu.ids | tail -25
SD_GN_ONELAUNCH=0 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-330
timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-330
