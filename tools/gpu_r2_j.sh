#!/bin/bash
# session J: in-sequence tuning of the K-split pipelined halo kernel (plan tile 7), then bench + per-op profile with the new table
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SD_TUNE=1 TUNE_TILES=7 timeout 600 python tools/tune_plans.py $OUT/tuned_ks.inc $OUT/tune_ks_report.json > $OUT/tune_ks.log 2>&1; tail -n 2 $OUT/tune_ks.log
cat $OUT/tuned_ks.inc
timeout 300 python bench.py --cpu-steps 0 --repeats 3 > $OUT/bench_j_before.log 2>/dev/null; tail -n 1 $OUT/bench_j_before.log | cut -c1-250
SD_PLAN_TABLE=$OUT/tuned_ks.inc timeout 300 python bench.py --cpu-steps 0 --repeats 3 > $OUT/bench_j_after.log 2>/dev/null; tail -n 1 $OUT/bench_j_after.log | cut -c1-250
SD_PLAN_TABLE=$OUT/tuned_ks.inc timeout 300 python tools/op_profile.py $OUT/op_profile_j.json 2 ORIGINAL > $OUT/op_profile_j.txt 2>&1; head -n 30 $OUT/op_profile_j.txt
