#!/bin/bash
# session Q: plan tuning for the other BASELINE configs (SDXL-base at 96x96 latents, SD1.5 at 64x64): per-op shortlist of the
# K-split halo kernel candidates, then end-to-end acceptance
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "sdxl 96" "sd15 64"; do
  set -- $cfg; M=$1; HW=$2
  SD_TUNE=1 TUNE_TILES=7 timeout 900 python tools/tune_plans.py $OUT/tuned_$M.inc $OUT/tune_${M}_report.json 2 $M $HW > $OUT/tune_$M.log 2>&1; tail -n 1 $OUT/tune_$M.log
  python tools/shortlist_plans.py $OUT/tune_${M}_report.json $OUT/shortlist_$M.json 3
  SD_TUNE=1 timeout 900 python tools/tune_e2e.py $OUT/shortlist_$M.json $OUT/tuned_e2e_$M.inc $OUT/tune_e2e_${M}_report.json $M $HW > $OUT/tune_e2e_$M.log 2>&1
  grep -v amdgpu.ids $OUT/tune_e2e_$M.log | tail -n 25
done
