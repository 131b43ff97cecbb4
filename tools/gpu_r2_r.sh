#!/bin/bash
# round 2, GPU session R: record run - full GPU suite, smoke, benches, kernel trace, per-op profile, PMC passes, other models
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 "Marketing Name.*MI" > $OUT/box_r.log; nproc >> $OUT/box_r.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_r.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_r.log
tail -n 4 $OUT/pytest_r.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_r.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary_r.log
timeout 600 python bench.py > $OUT/bench_r.log 2> $OUT/bench_r.err; echo "bench rc=$?" | tee -a $OUT/summary_r.log
tail -n 1 $OUT/bench_r.log | cut -c1-700
timeout 600 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM > $OUT/bench_r_split.log 2>/dev/null; tail -n 1 $OUT/bench_r_split.log | cut -c1-200
timeout 600 python bench.py --cpu-steps 0 --attention SPLIT_EINSUM_V2 --prompts-per-gpu 2 > $OUT/bench_r_v2_b4.log 2>/dev/null; tail -n 1 $OUT/bench_r_v2_b4.log | cut -c1-200
rm -rf $OUT/prof_r
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_r -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 --repeats 1 > $OUT/rocprof_r.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary_r.log
DB=$(find $OUT/prof_r -name "*.db" | head -n 1)
[ -n "$DB" ] && python tools/timeline.py $DB > $OUT/step_timeline_r.txt 2>&1 && head -n 34 $OUT/step_timeline_r.txt
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats_r.csv > /dev/null 2>&1
timeout 300 python tools/op_profile.py $OUT/op_profile_r.json 2 ORIGINAL > $OUT/op_profile_r.txt 2>&1; head -n 3 $OUT/op_profile_r.txt
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rm -rf $OUT/pmc_$TAG
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $OUT/pmc_$TAG -o r -- python /root/repo/tools/pmc_probe.py sd21 4 > $OUT/pmc_$TAG.log 2>&1); echo "pmc $TAG rc=$?" | tee -a $OUT/summary_r.log
done
python tools/pmc_reduce.py $OUT/r02_hbm_traffic_final.json $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db") 2>&1 | tail -n 25
rm -rf $OUT/pmc_sq
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o r -- python /root/repo/tools/pmc_probe.py sd21 4 > $OUT/pmc_sq.log 2>&1); echo "pmc sq rc=$?" | tee -a $OUT/summary_r.log
python tools/pmc_reduce.py $OUT/r02_sq_counters_final.json $(find $OUT/pmc_sq -name "*.db") 2>&1 | tail -n 16
timeout 900 python tools/model_bench.py > $OUT/model_bench_r.log 2>&1; tail -n 12 $OUT/model_bench_r.log | cut -c1-300
rm -rf $OUT/prof_r $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/pmc_sq
du -sh $OUT | tail -n 1
