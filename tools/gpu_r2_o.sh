#!/bin/bash
# session O: attention with two workgroups per CU (<= 256 VGPRs), with and without the software-pipelined score computation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $OUT/o_tests.txt 2>&1; tail -2 $OUT/o_tests.txt
SD_ATTN_PIPE=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $OUT/o_tests_pipe.txt 2>&1; tail -2 $OUT/o_tests_pipe.txt
echo "== plain (2 workgroups per CU)"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_plain.txt
echo "== pipelined"; SD_ATTN_PIPE=1 timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_pipe.txt
timeout 300 python bench.py --cpu-steps 0 --repeats 3 2>/dev/null | cut -c1-260
SD_ATTN_PIPE=1 timeout 300 python bench.py --cpu-steps 0 --repeats 3 2>/dev/null | cut -c1-260
