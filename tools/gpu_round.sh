#!/bin/bash
# One gpurun call = everything we want measured this iteration.  Logs go to gpurun_out/.
# usage: tools/gpu_round.sh [quick]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
{ date; nproc; free -g | head -2; /opt/rocm/bin/rocm-smi --showproductname 2>/dev/null | head -12; } > $OUT/box.log 2>&1
PY="python -u"
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a $OUT/summary.log
  timeout $to "$@" > $OUT/$name.log 2>&1
  echo "exit $? : $(tail -n 3 $OUT/$name.log | tr '\n' '|' | cut -c1-400)" | tee -a $OUT/summary.log
}
: > $OUT/summary.log
run selftest 300 $PY -m pytest tests/test_ops_gpu.py -m gpu -q -k "selftest" --timeout 200
run ops_attention 900 $PY -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" --timeout 300
run ops_other 900 $PY -m pytest tests/test_ops_gpu.py -m gpu -q -k "not attention and not selftest" --timeout 300
run unet_small 1200 $PY -m pytest tests/test_unet_gpu.py -m gpu -q -k "not full" --timeout 600
run unet_full 1200 $PY -m pytest tests/test_unet_gpu.py -m gpu -q -k "full" --timeout 1000
run smoke 600 $PY -c "import __graft_entry__ as g; g.smoke()"
run bench 1200 $PY bench.py --steps 10 --warmup 2 --cpu-steps ${CPU_STEPS:-0}
run bench_split 600 $PY bench.py --steps 10 --warmup 2 --cpu-steps 0 --attention SPLIT_EINSUM
if [ "${1:-}" != "quick" ]; then
  run microbench 900 $PY tools/microbench.py
  cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o bench -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-graph > /root/repo/$OUT/rocprof.log 2>&1
  cd /root/repo
  ls -la $OUT/prof 2>/dev/null | head -20 >> $OUT/summary.log
fi
cat $OUT/summary.log
