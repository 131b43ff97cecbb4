#!/bin/bash
# round 2, GPU session C: operator tests of the new epilogue / deep rings, in-sequence plan tuning, validation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -m gpu -x -q > $OUT/pytest_c.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_c.log
tail -n 6 $OUT/pytest_c.log | cut -c1-300
timeout 300 python tools/op_profile.py $OUT/op_profile_c0.json 2 ORIGINAL > $OUT/op_profile_c0.txt 2>&1; head -n 3 $OUT/op_profile_c0.txt
SD_TUNE=1 timeout 900 python tools/tune_plans.py $OUT/tuned_convs_r2.inc $OUT/tune_report_r2.json 2 > $OUT/tune_c.log 2>&1; echo "tune rc=$?" | tee -a $OUT/summary_c.log
tail -n 3 $OUT/tune_c.log; wc -l $OUT/tuned_convs_r2.inc
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 300 python tools/op_profile.py $OUT/op_profile_c1.json 2 ORIGINAL > $OUT/op_profile_c1.txt 2>&1; head -n 30 $OUT/op_profile_c1.txt
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 600 python bench.py --cpu-steps 0 > $OUT/bench_c1.log 2> $OUT/bench_c1.err; echo "bench rc=$?" | tee -a $OUT/summary_c.log
tail -n 1 $OUT/bench_c1.log | cut -c1-400
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest_c1.log 2>&1; echo "pytest(tuned) rc=$?" | tee -a $OUT/summary_c.log
tail -n 4 $OUT/pytest_c1.log | cut -c1-300
