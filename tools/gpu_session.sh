#!/bin/bash
# One GPU-box session, parameterised (replaces the per-session scripts of rounds 1-2):
#   tools/gpu_session.sh <tag> <step> [<step> ...]
# steps: epytest:<ENV=VAL,..>@<selection>  eprofile:<ENV=VAL,..>  new (round-3 kernel tests)  tests (full -m gpu suite)  smoke  bench  bench:<flags>  ab:<ENV=VAL> (bench with an env switch)  pmc
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
      kv=${step#ab:}; fl=""
      case "$kv" in *@*) fl=${kv#*@}; kv=${kv%%@*} ;; esac      # "ab:ENV=VAL@--prompts-per-gpu 8": extra bench flags behind @
      r=$(env ${kv//,/ } timeout 300 python bench.py --cpu-steps 0 --sustain-s 0 --repeats 5 $fl 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=(d.get('gpu_state') or {}).get('after_timed_region') or {}; c=d.get('calibration') or {}; print(d['ms_per_step'], d['value'], 'calib', c.get('copy_gbs'), c.get('mfma_tflops'), c.get('empty_launch_us'), c.get('chain_us'), c.get('handover_us'), c.get('latency_hbm_ns'), c.get('latency_cache_ns'), c.get('small_grid_us'), c.get('cold_code_us'), 'sclk', g.get('sclk clock speed:'), 'W', g.get('Current Socket Graphics Package Power (W)'))" 2>&1)
      note "ab[$kv $fl] ms/step, it/s: $r" ;;
    profile)
      timeout 300 python tools/op_profile.py $OUT/op_profile_$TAG.json 2 ORIGINAL > $OUT/op_profile_$TAG.txt 2>&1; note "profile rc=$?"; head -n 28 $OUT/op_profile_$TAG.txt | cut -c1-150 ;;
    trace)
      rm -rf $OUT/prof_$TAG
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --sustain-s 0 --repeats 1 > $OUT/rocprof_$TAG.log 2>&1); note "rocprof rc=$?"
      DB=$(find $OUT/prof_$TAG -name "*.db" | head -n 1)
      [ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_$TAG.txt 2>&1 && head -n 30 $OUT/step_timeline_$TAG.txt | cut -c1-160
      [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_$TAG.csv > /dev/null 2>&1
      rm -rf $OUT/prof_$TAG ;;
    pmc|pmc2)   # HBM traffic + MFMA-busy + stall counters of the eager step: separate --pmc passes (never with trace domains other than kernel-trace); pmc2 = the two traffic passes only
      for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
        [ "$step" = pmc2 ] && [ "${SET%% *}" = SQ_VALU_MFMA_BUSY_CYCLES ] && continue
        T=$(echo $SET | cut -d' ' -f1)
        rm -rf $OUT/pmc_$T
        (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $OUT/pmc_$T -o r -- python /root/repo/tools/pmc_probe.py sd21 4 > $OUT/pmc_${T}_$TAG.log 2>&1); note "pmc $T rc=$?"
      done
      python tools/pmc_reduce.py $OUT/hbm_traffic_$TAG.json $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" 2>/dev/null) 2>&1 | tail -n 20
      rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES ;;
    bench:*)   # bench with extra flags, e.g. "bench:--model sdxl-base --latent 96"
      fl=${step#bench:}
      timeout 900 python bench.py $fl > $OUT/bench_${TAG}_$i.log 2> $OUT/bench_${TAG}_$i.err; note "bench[$fl] rc=$?"
      tail -n 1 $OUT/bench_${TAG}_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['metric'], d['ms_per_step'], 'ms', d['value'], 'it/s frac', r['frac'], 'dominant', r.get('dominant_kernel',{}).get('kernel'), r.get('dominant_kernel',{}).get('share'), r.get('dominant_kernel',{}).get('frac'))" 2>&1 | cut -c1-400 ;;
    pytest:*)   # one test file / selection
      sel=${step#pytest:}
      timeout 1700 python -m pytest $sel -m gpu -q -x > $OUT/pytest_${TAG}_$i.log 2>&1; note "pytest[$sel] rc=$? $(tail -n 1 $OUT/pytest_${TAG}_$i.log | cut -c1-200)"
      grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_${TAG}_$i.log | head -n 12 | cut -c1-400 ;;
    epytest:*)   # a test selection under env switches: "epytest:ENV=VAL,ENV2=VAL2@<selection>"
      kv=${step#epytest:}; sel=${kv#*@}; kv=${kv%%@*}
      env ${kv//,/ } timeout 1700 python -m pytest $sel -m gpu -q -x > $OUT/pytest_${TAG}_$i.log 2>&1; note "pytest[$kv $sel] rc=$? $(tail -n 1 $OUT/pytest_${TAG}_$i.log | cut -c1-200)"
      grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_${TAG}_$i.log | head -n 12 | cut -c1-400 ;;
    eprofile:*)   # per-op table under env switches: "eprofile:ENV=VAL,ENV2=VAL2"
      kv=${step#eprofile:}
      env ${kv//,/ } timeout 300 python tools/op_profile.py $OUT/op_profile_${TAG}_$i.json 2 ORIGINAL > $OUT/op_profile_${TAG}_$i.txt 2>&1; note "profile[$kv] rc=$? $(sed -n 2p $OUT/op_profile_${TAG}_$i.txt)"
      grep -E "conv3x3|groupnorm" $OUT/op_profile_${TAG}_$i.txt | head -n 30 | cut -c1-140 ;;
    sh:*)   # a shell command inside the session (e.g. copy the PMC file where bench.py looks for it)
      cmd=${step#sh:}
      bash -c "$cmd"; note "sh[$cmd] rc=$?" ;;
    tune:*)   # end-to-end plan tuner: "tune:<candidates.json> [model] [latent]" -> gpurun_out/tuned_e2e_<tag>.inc
      ar=${step#tune:}
      SD_TUNE=1 timeout 1200 python tools/tune_e2e.py ${ar%% *} $OUT/tuned_e2e_$TAG.inc $OUT/tune_e2e_report_$TAG.json $( [ "$ar" != "${ar#* }" ] && echo ${ar#* } ) > $OUT/tune_e2e_$TAG.log 2>&1; note "tune_e2e rc=$?"
      tail -n 25 $OUT/tune_e2e_$TAG.log | cut -c1-200 ;;
    py:*)
      cmd=${step#py:}
      timeout 900 python tools/$cmd > $OUT/py_${TAG}_$i.log 2>&1; note "py[$cmd] rc=$?"; tail -n 40 $OUT/py_${TAG}_$i.log | cut -c1-200 ;;
    *) note "unknown step $step" ;;
  esac
done
