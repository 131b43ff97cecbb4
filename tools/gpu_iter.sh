#!/bin/bash
# one development iteration: conv/op parity, UNet/pipeline parity, bench (two attention schedules)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -u -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 300 -k "${1:-not attention}" 2>&1 | tail -n 5
python -u -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -n 12
python -u bench.py --steps 10 --warmup 2 --cpu-steps 0 > $OUT/bench_orig.log 2>&1; tail -n 1 $OUT/bench_orig.log | cut -c1-330
python -u bench.py --steps 10 --warmup 2 --cpu-steps 0 --attention SPLIT_EINSUM > $OUT/bench_split.log 2>&1; tail -n 1 $OUT/bench_split.log | cut -c1-330
