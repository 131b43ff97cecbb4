#!/bin/bash
# session L: halo-ks on upsample / 8-wide images (tests), then in-sequence tuning with the pipelined GEMM + halo-ks candidates
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo" > $OUT/l_tests.txt 2>&1; tail -3 $OUT/l_tests.txt
SD_TUNE=1 TUNE_TILES=1,2,3,4,7 TUNE_STAGINGS=6,7,8 timeout 900 python tools/tune_plans.py $OUT/tuned_l.inc $OUT/tune_l_report.json > $OUT/tune_l.log 2>&1; tail -n 2 $OUT/tune_l.log
cat $OUT/tuned_l.inc
timeout 300 python bench.py --cpu-steps 0 --repeats 3 > $OUT/bench_l_before.log 2>/dev/null; tail -n 1 $OUT/bench_l_before.log | cut -c1-250
SD_PLAN_TABLE=$OUT/tuned_l.inc timeout 300 python bench.py --cpu-steps 0 --repeats 3 > $OUT/bench_l_after.log 2>/dev/null; tail -n 1 $OUT/bench_l_after.log | cut -c1-250
SD_PLAN_TABLE=$OUT/tuned_l.inc timeout 300 python tools/op_profile.py $OUT/op_profile_l.json 2 ORIGINAL > $OUT/op_profile_l.txt 2>&1; head -n 45 $OUT/op_profile_l.txt
