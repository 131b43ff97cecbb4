#!/bin/bash
# quick iteration: conv/op parity + small UNet parity + bench (+ microbench / full model parity)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
PY="python -u"
run() { local name=$1 to=$2; shift 2; echo "=== $name" | tee -a $OUT/summary.log; timeout $to "$@" > $OUT/$name.log 2>&1; echo "exit $? : $(tail -n 3 $OUT/$name.log | tr '\n' '|' | cut -c1-600)" | tee -a $OUT/summary.log; }
: > $OUT/summary.log
run ops_other 900 $PY -m pytest tests/test_ops_gpu.py -m gpu -q -k "not attention" --timeout 300 -x
run unet_small 1200 $PY -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py -m gpu -q --timeout 600 -x
run psnr 600 $PY tools/psnr_report.py
run bench 1200 $PY bench.py --steps 10 --warmup 2 --cpu-steps 0
run bench_split 600 $PY bench.py --steps 10 --warmup 2 --cpu-steps 0 --attention SPLIT_EINSUM
if [ "${1:-}" = "micro" ]; then run microbench 900 $PY tools/microbench.py; fi
