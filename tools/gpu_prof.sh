#!/bin/bash
# bench (ORIGINAL + SPLIT_EINSUM) and a rocprofv3 kernel-trace of the eager step -> per-kernel CSV
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -u bench.py --steps 10 --warmup 2 --cpu-steps 0 > $OUT/bench_orig.log 2>&1; tail -n 1 $OUT/bench_orig.log | cut -c1-400
python -u bench.py --steps 10 --warmup 2 --cpu-steps 0 --attention SPLIT_EINSUM > $OUT/bench_split.log 2>&1; tail -n 1 $OUT/bench_split.log | cut -c1-400
rm -rf $OUT/prof
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o bench -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-graph > /root/repo/$OUT/rocprof.log 2>&1
cd /root/repo
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -n 1) > $OUT/kernel_stats.csv 2> $OUT/rocpd_stats.err || tail -n 3 $OUT/rocpd_stats.err
head -n 40 $OUT/kernel_stats.csv
