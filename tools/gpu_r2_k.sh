#!/bin/bash
# session K: correctness + timing of the software-pipelined GEMM kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "gemm1x1_pipelined or pipelined_gemm_plan or geglu" > $OUT/k_tests.txt 2>&1
tail -5 $OUT/k_tests.txt
timeout 300 python tools/gemm_ks_bench.py 2>&1 | grep -v "amdgpu.ids" > $OUT/gemm_ks_warm.txt; cat $OUT/gemm_ks_warm.txt
SD_BENCH_COLD=1 timeout 300 python tools/gemm_ks_bench.py 2>&1 | grep -v "amdgpu.ids" > $OUT/gemm_ks_cold.txt; cat $OUT/gemm_ks_cold.txt
