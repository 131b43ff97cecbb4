#!/bin/bash
# session X: final state - full GPU suite, smoke, A/B of the residual prefetch, bench.py default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box_x.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_x.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_x.log
tail -n 3 $OUT/pytest_x.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_x.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary_x.log
SD_RES_PREFETCH=0 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
SD_RES_PREFETCH=0 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 600 python bench.py > $OUT/bench_x.log 2> $OUT/bench_x.err; echo "bench rc=$?" | tee -a $OUT/summary_x.log
tail -n 1 $OUT/bench_x.log | cut -c1-400
