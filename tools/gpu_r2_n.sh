#!/bin/bash
# session N: end-to-end plan tuning (graph replay of the whole step decides)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SD_TUNE=1 timeout 900 python tools/tune_e2e.py tools/tables/shortlist_2.json $OUT/tuned_e2e.inc $OUT/tune_e2e_report.json > $OUT/tune_e2e.log 2>&1
grep -v amdgpu.ids $OUT/tune_e2e.log | tail -n 60
cat $OUT/tuned_e2e.inc
timeout 300 python bench.py --cpu-steps 0 --repeats 5 > $OUT/bench_n_before.log 2>/dev/null; tail -n 1 $OUT/bench_n_before.log | cut -c1-330
SD_PLAN_TABLE=$OUT/tuned_e2e.inc timeout 300 python bench.py --cpu-steps 0 --repeats 5 > $OUT/bench_n_after.log 2>$OUT/bench_n_after.err; tail -n 1 $OUT/bench_n_after.log | cut -c1-330; tail -n 2 $OUT/bench_n_after.err
