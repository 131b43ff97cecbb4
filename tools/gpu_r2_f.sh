#!/bin/bash
# round 2, GPU session F: in-kernel split-K combine + VAE encoder: tests, profile, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest_f.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_f.log
tail -n 5 $OUT/pytest_f.log | cut -c1-300
timeout 300 python tools/op_profile.py $OUT/op_profile_f.json 2 ORIGINAL > $OUT/op_profile_f.txt 2>&1; head -n 40 $OUT/op_profile_f.txt
timeout 600 python bench.py --cpu-steps 0 > $OUT/bench_f.log 2> $OUT/bench_f.err; echo "bench rc=$?" | tee -a $OUT/summary_f.log
tail -n 1 $OUT/bench_f.log | cut -c1-300
rm -rf $OUT/prof_f
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_f -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_f.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary_f.log
DB=$(find $OUT/prof_f -name "*.db" | head -n 1)
[ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_f.txt 2>&1 && head -n 30 $OUT/step_timeline_f.txt
