#!/bin/bash
# session U: validation of the last changes (time-embedding table pass, conv_in on the MFMA, halo-ks default): full GPU suite,
# smoke, bench (+ A/B of the table pass), per-op profile, kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box_u.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_u.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_u.log
tail -n 3 $OUT/pytest_u.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_u.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary_u.log
SD_NO_TEMB_TABLE=1 timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | cut -c1-300
timeout 600 python bench.py > $OUT/bench_u.log 2> $OUT/bench_u.err; echo "bench rc=$?" | tee -a $OUT/summary_u.log
tail -n 1 $OUT/bench_u.log | cut -c1-400
timeout 300 python tools/op_profile.py $OUT/op_profile_u.json 2 ORIGINAL > $OUT/op_profile_u.txt 2>&1; head -n 3 $OUT/op_profile_u.txt
rm -rf $OUT/prof_u
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_u -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_u.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary_u.log
DB=$(find $OUT/prof_u -name "*.db" | head -n 1)
[ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_u.txt 2>&1 && head -n 12 $OUT/step_timeline_u.txt
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_u.csv > /dev/null 2>&1
rm -rf $OUT/prof_u
