#!/bin/bash
# round 2, GPU session A: full GPU test suite, smoke, bench, per-op profile, rocprofv3 kernel trace, PMC probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box.log; nproc >> $OUT/box.log
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.log
tail -n 5 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.log
timeout 600 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.log
tail -n 1 $OUT/bench.log | cut -c1-1500
# kernel trace of the graph-mode loop (no counters)
rm -rf $OUT/prof_a
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_a -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_a.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.log
DB=$(find $OUT/prof_a -name "*.db" | head -n 1)
[ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_a.txt 2>&1 && head -n 40 $OUT/step_timeline_a.txt
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_a.csv > /dev/null 2>&1
# PMC probe: does counter collection survive eager steps of the mini / full model?
for W in mini sd21; do
  rm -rf $OUT/pmc_probe_$W
  (cd /tmp && SD_LOG_CONVS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_probe_$W -o r -- python /root/repo/tools/pmc_probe.py $W 3 > $OUT/pmc_probe_$W.log 2>&1); echo "pmc_probe $W rc=$?" | tee -a $OUT/summary.log
  grep -E "pmc_probe|SIGSEGV|Aborted" $OUT/pmc_probe_$W.log | head -n 3
  grep "sd conv" $OUT/pmc_probe_$W.log | tail -n 2
done
rm -rf $OUT/prof_a/*/*.pftrace 2>/dev/null
du -sh $OUT | tail -n 1
