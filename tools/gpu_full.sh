#!/bin/bash
# what the driver does at round end: every GPU test, smoke, bench (+ optional profile)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2; echo "=== $name" | tee -a $OUT/summary.log; timeout $to "$@" > $OUT/$name.log 2>&1; echo "exit $? : $(tail -n 3 $OUT/$name.log | tr '\n' '|' | cut -c1-700)" | tee -a $OUT/summary.log; }
: > $OUT/summary.log
run pytest_gpu 1800 python -u -m pytest tests -m gpu -q --timeout 900 -x
run smoke 600 python -u -c "import __graft_entry__ as g; g.smoke()"
run bench 1200 python -u bench.py
if [ "${1:-}" = "prof" ]; then
  cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o bench -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-graph > /root/repo/$OUT/rocprof.log 2>&1
fi
