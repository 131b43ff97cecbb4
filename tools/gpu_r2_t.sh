#!/bin/bash
# session T: time-embedding table in the device loop - loop parity tests, then A/B on the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu -k "loop or pipeline or prompt or scheduler or refiner or controlnet or batch" > $OUT/t_tests.txt 2>&1; tail -3 $OUT/t_tests.txt
for i in 1 2; do
SD_NO_TEMB_TABLE=1 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
done
