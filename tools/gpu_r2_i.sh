#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo or splitk_is_complete" > gpurun_out/i_tests.txt 2>&1
tail -3 gpurun_out/i_tests.txt
timeout 300 python tools/ablate_halo_ks.py 2>&1 | grep -v "sd prof\|amdgpu.ids" > gpurun_out/ablate_ks_warm.txt
SD_BENCH_COLD=1 timeout 300 python tools/ablate_halo_ks.py 2>&1 | grep -v "sd prof\|amdgpu.ids" > gpurun_out/ablate_ks_cold.txt
cat gpurun_out/ablate_ks_warm.txt
