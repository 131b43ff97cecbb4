#!/bin/bash
# session W: SDXL-base 96x96 - ring plans of the 1x1 GEMM shapes SD2.1 does not have: per-op shortlist, end-to-end acceptance
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SD_TUNE=1 TUNE_TILES=1,2,3,4 TUNE_STAGINGS=0,2,3 timeout 900 python tools/tune_plans.py $OUT/tuned_sdxl2.inc $OUT/tune_sdxl2_report.json 2 sdxl 96 > $OUT/tune_sdxl2.log 2>&1; tail -n 1 $OUT/tune_sdxl2.log
python tools/shortlist_plans.py $OUT/tune_sdxl2_report.json $OUT/shortlist_sdxl2.json 3
SD_TUNE=1 timeout 900 python tools/tune_e2e.py $OUT/shortlist_sdxl2.json $OUT/tuned_e2e_sdxl2.inc $OUT/tune_e2e_sdxl2_report.json sdxl 96 > $OUT/tune_e2e_sdxl2.log 2>&1
grep -v amdgpu.ids $OUT/tune_e2e_sdxl2.log | tail -n 40
