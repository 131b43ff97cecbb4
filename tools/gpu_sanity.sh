#!/bin/bash
# what the driver runs at round end, nothing more: GPU tests, smoke, default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -u -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -n 3
python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3
python -u bench.py > $OUT/bench_sanity.log 2>&1; tail -n 1 $OUT/bench_sanity.log | cut -c1-400
