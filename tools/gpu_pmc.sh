#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT/pmc; export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/pmc/counters.txt 2>&1
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum")
i=0
for S in "${SETS[@]}"; do
  for CFG in "3 1 640 640 32 0 4 30" "3 1 320 320 64 0 1 30" "1 1 320 320 64 0 3 30"; do
    tag=$(echo $CFG | tr ' ' '_')
    rocprofv3 --pmc $S --kernel-trace -d $OUT/pmc/s${i}_$tag -o r -- python /root/repo/tools/pmc_conv.py $CFG > $OUT/pmc/s${i}_$tag.log 2>&1
  done
  i=$((i+1))
done
ls -R $OUT/pmc | head -40
