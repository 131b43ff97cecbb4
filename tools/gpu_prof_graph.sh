#!/bin/bash
# rocprofv3 kernel trace of the GRAPH-mode step loop for two attention schedules
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for impl in ORIGINAL SPLIT_EINSUM; do
  rm -rf $OUT/prof_$impl
  (cd /tmp && rocprofv3 --kernel-trace -d /root/repo/$OUT/prof_$impl -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --attention $impl > /root/repo/$OUT/rocprof_$impl.log 2>&1)
  tail -n 1 $OUT/rocprof_$impl.log | cut -c1-300
  find $OUT/prof_$impl -name "*.db" | head -n 2
done
