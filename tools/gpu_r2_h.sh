#!/bin/bash
# round 2, GPU session H: record run - full GPU suite, smoke, bench (default flags), kernel trace, per-op profile
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box_h.log; nproc >> $OUT/box_h.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_h.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_h.log
tail -n 4 $OUT/pytest_h.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_h.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary_h.log
timeout 600 python bench.py > $OUT/bench_h.log 2> $OUT/bench_h.err; echo "bench rc=$?" | tee -a $OUT/summary_h.log
tail -n 1 $OUT/bench_h.log | cut -c1-600
timeout 600 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM > $OUT/bench_h_split.log 2>/dev/null; tail -n 1 $OUT/bench_h_split.log | cut -c1-200
timeout 600 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM_V2 --prompts-per-gpu 2 > $OUT/bench_h_v2_b4.log 2>/dev/null; tail -n 1 $OUT/bench_h_v2_b4.log | cut -c1-200
rm -rf $OUT/prof_h
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_h -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_h.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary_h.log
DB=$(find $OUT/prof_h -name "*.db" | head -n 1)
[ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_h.txt 2>&1 && head -n 34 $OUT/step_timeline_h.txt
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_h.csv > /dev/null 2>&1
timeout 300 python tools/op_profile.py $OUT/op_profile_h.json 2 ORIGINAL > $OUT/op_profile_h.txt 2>&1; head -n 3 $OUT/op_profile_h.txt
