#!/bin/bash
# round 2, GPU session B: remaining GPU tests, CLIP, per-op profile, PMC passes (HBM traffic, MFMA busy)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_round2_gpu.py tests/test_text_encoder_gpu.py tests/test_unet_gpu.py -m gpu -q -s > $OUT/pytest_b.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary_b.log
tail -n 12 $OUT/pytest_b.log | cut -c1-400
timeout 300 python tools/op_profile.py $OUT/op_profile_b2.json 2 ORIGINAL > $OUT/op_profile_b2.txt 2>&1; head -n 60 $OUT/op_profile_b2.txt
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rm -rf $OUT/pmc_$TAG
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace -d $OUT/pmc_$TAG -o r -- python /root/repo/tools/pmc_probe.py sd21 4 > $OUT/pmc_$TAG.log 2>&1); echo "pmc $TAG rc=$?" | tee -a $OUT/summary_b.log
done
python tools/pmc_reduce.py $OUT/r02_hbm_traffic.json $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db") 2>&1 | tail -n 25
du -sh $OUT | tail -n 1
