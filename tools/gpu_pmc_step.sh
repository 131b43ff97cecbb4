#!/bin/bash
# HBM bytes per denoising step: two rocprofv3 --pmc passes (counters only + kernel trace)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$C
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o r -- python /root/repo/bench.py --steps 4 --warmup 1 --cpu-steps 0 --no-graph > $OUT/pmc_$C.log 2>&1)
  tail -n 2 $OUT/pmc_$C.log | cut -c1-200
done
python tools/step_traffic.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -n 1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -n 1) $OUT/step_traffic.json
