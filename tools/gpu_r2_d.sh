#!/bin/bash
# round 2, GPU session D: halo weight ring tests, full in-sequence tune incl. halo rings, validation, full suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "halo or conv2d" > $OUT/pytest_d0.log 2>&1; echo "pytest ops rc=$?" | tee $OUT/summary_d.log
tail -n 4 $OUT/pytest_d0.log | cut -c1-300
SD_TUNE=1 timeout 900 python tools/tune_plans.py $OUT/tuned_convs_r2.inc $OUT/tune_report_r2.json 2 > $OUT/tune_d.log 2>&1; echo "tune rc=$?" | tee -a $OUT/summary_d.log
tail -n 3 $OUT/tune_d.log; wc -l $OUT/tuned_convs_r2.inc
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 300 python tools/op_profile.py $OUT/op_profile_d1.json 2 ORIGINAL > $OUT/op_profile_d1.txt 2>&1; head -n 45 $OUT/op_profile_d1.txt
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 600 python bench.py --cpu-steps 0 > $OUT/bench_d1.log 2> $OUT/bench_d1.err; echo "bench rc=$?" | tee -a $OUT/summary_d.log
tail -n 1 $OUT/bench_d1.log | cut -c1-400
SD_PLAN_TABLE=$OUT/tuned_convs_r2.inc timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_d1.log 2>&1; echo "pytest(all, tuned) rc=$?" | tee -a $OUT/summary_d.log
tail -n 4 $OUT/pytest_d1.log | cut -c1-300
