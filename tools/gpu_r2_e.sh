#!/bin/bash
# round 2, GPU session E: software-pipelined ring variants (tests + in-sequence tune), vendor-library yardstick
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "every_tile" > $OUT/pytest_e0.log 2>&1; echo "pytest ops rc=$?" | tee $OUT/summary_e.log
tail -n 3 $OUT/pytest_e0.log | cut -c1-300
timeout 300 python tools/library_yardstick.py $OUT/r02_library_yardstick.json > $OUT/yardstick.log 2>&1; cat $OUT/yardstick.log | tail -n 14
timeout 300 python tools/op_profile.py $OUT/op_profile_e0.json 2 ORIGINAL > $OUT/op_profile_e0.txt 2>&1; head -n 3 $OUT/op_profile_e0.txt
SD_TUNE=1 timeout 1200 python tools/tune_plans.py $OUT/tuned_convs_r2e.inc $OUT/tune_report_r2e.json 2 > $OUT/tune_e.log 2>&1; echo "tune rc=$?" | tee -a $OUT/summary_e.log
tail -n 3 $OUT/tune_e.log; cat $OUT/tuned_convs_r2e.inc | head -40
SD_PLAN_TABLE=$OUT/tuned_convs_r2e.inc timeout 300 python tools/op_profile.py $OUT/op_profile_e1.json 2 ORIGINAL > $OUT/op_profile_e1.txt 2>&1; head -n 3 $OUT/op_profile_e1.txt
SD_PLAN_TABLE=$OUT/tuned_convs_r2e.inc timeout 600 python bench.py --cpu-steps 0 > $OUT/bench_e1.log 2> $OUT/bench_e1.err; echo "bench rc=$?" | tee -a $OUT/summary_e.log
tail -n 1 $OUT/bench_e1.log | cut -c1-300
