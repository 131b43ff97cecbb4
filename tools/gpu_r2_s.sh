#!/bin/bash
# session S: software prefetch of the next conv's weights - parity, then A/B on the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "golden_all_attention or graph_replay or full_sd21" > $OUT/s_tests.txt 2>&1; tail -2 $OUT/s_tests.txt
for i in 1 2; do
SD_PREFETCH=0 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
done
