#!/bin/bash
# One GPU-box session, parameterised (replaces the per-session scripts of rounds 1-2):
#   tools/gpu_session.sh <tag> <step> [<step> ...]
# steps: new (round-3 kernel tests)  tests (full -m gpu suite)  smoke  bench  ab:<ENV=VAL> (bench with an env switch)
#        profile (per-op table)  trace (rocprofv3 kernel trace of the bench)  py:<script and args> (python tools/<script>)
# Logs go to gpurun_out/<name>_<tag>.*; a one-line verdict per step is echoed (what gpurun shows at the end).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=$1; shift
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box_$TAG.log
: > $OUT/summary_$TAG.log
note() { echo "$*" | tee -a $OUT/summary_$TAG.log; }
i=0
for step in "$@"; do
  i=$((i + 1))
  case "$step" in
    new)
      timeout 900 python -m pytest tests/test_round3_gpu.py -m gpu -q -x > $OUT/pytest_new_$TAG.log 2>&1; note "new-tests rc=$? $(tail -n 1 $OUT/pytest_new_$TAG.log | cut -c1-200)"
      grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_new_$TAG.log | head -n 12 | cut -c1-300 ;;
    tests)
      timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_$TAG.log 2>&1; note "pytest rc=$? $(tail -n 1 $OUT/pytest_$TAG.log | cut -c1-200)"
      grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head -n 20 | cut -c1-300 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; note "smoke rc=$? $(tail -n 1 $OUT/smoke_$TAG.log | cut -c1-120)" ;;
    bench)
      timeout 900 python bench.py > $OUT/bench_$TAG.log 2> $OUT/bench_$TAG.err; note "bench rc=$?"
      tail -n 1 $OUT/bench_$TAG.log | cut -c1-600 ;;
    ab:*)
      kv=${step#ab:}
      r=$(env ${kv//,/ } timeout 300 python bench.py --cpu-steps 0 --repeats 5 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1)
      note "ab[$kv] ms/step, it/s: $r" ;;
    profile)
      timeout 300 python tools/op_profile.py $OUT/op_profile_$TAG.json 2 ORIGINAL > $OUT/op_profile_$TAG.txt 2>&1; note "profile rc=$?"; head -n 28 $OUT/op_profile_$TAG.txt | cut -c1-150 ;;
    trace)
      rm -rf $OUT/prof_$TAG
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_$TAG.log 2>&1); note "rocprof rc=$?"
      DB=$(find $OUT/prof_$TAG -name "*.db" | head -n 1)
      [ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_$TAG.txt 2>&1 && head -n 30 $OUT/step_timeline_$TAG.txt | cut -c1-160
      [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_$TAG.csv > /dev/null 2>&1
      rm -rf $OUT/prof_$TAG ;;
    py:*)
      cmd=${step#py:}
      timeout 900 python tools/$cmd > $OUT/py_${TAG}_$i.log 2>&1; note "py[$cmd] rc=$?"; tail -n 40 $OUT/py_${TAG}_$i.log | cut -c1-200 ;;
    *) note "unknown step $step" ;;
  esac
done
