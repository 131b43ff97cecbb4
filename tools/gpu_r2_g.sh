#!/bin/bash
# round 2, GPU session G: GroupNorm single-round-trip kernels: tests, profile, bench; SQ stall counters of the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -m gpu -x -q > $OUT/pytest_g.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary_g.log
tail -n 4 $OUT/pytest_g.log | cut -c1-300
timeout 300 python tools/op_profile.py $OUT/op_profile_g.json 2 ORIGINAL > $OUT/op_profile_g.txt 2>&1; head -n 4 $OUT/op_profile_g.txt; grep -E "groupnorm" $OUT/op_profile_g.txt | head
timeout 600 python bench.py --cpu-steps 0 > $OUT/bench_g.log 2> $OUT/bench_g.err; echo "bench rc=$?" | tee -a $OUT/summary_g.log
tail -n 1 $OUT/bench_g.log | cut -c1-300
rm -rf $OUT/pmc_sq
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o r -- python /root/repo/tools/pmc_probe.py sd21 4 > $OUT/pmc_sq.log 2>&1); echo "pmc sq rc=$?" | tee -a $OUT/summary_g.log
python tools/pmc_reduce.py $OUT/r02_sq_counters.json $(find $OUT/pmc_sq -name "*.db") 2>&1 | tail -n 16
