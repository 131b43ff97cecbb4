#!/bin/bash
# what the driver does at round end: every GPU test, smoke, bench (+ variants, profile)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2; echo "=== $name" | tee -a $OUT/summary.log; timeout $to "$@" > $OUT/$name.log 2>&1; echo "exit $? : $(tail -n 3 $OUT/$name.log | tr '\n' '|' | cut -c1-700)" | tee -a $OUT/summary.log; }
: > $OUT/summary.log
run pytest_gpu 1800 python -u -m pytest tests -m gpu -q --timeout 900 -x
run smoke 600 python -u -c "import __graft_entry__ as g; g.smoke()"
run bench 1200 python -u bench.py
if [ "${1:-}" = "more" ]; then
  run bench_split 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM
  run bench_v2_b4 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM_V2 --prompts-per-gpu 2
  run bench_split_b8 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM --prompts-per-gpu 4
  run bench_torchrun1 600 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-steps 0
  cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o bench -- python /root/repo/bench.py --steps 5 --warmup 1 --cpu-steps 0 --no-graph > /root/repo/$OUT/rocprof.log 2>&1
  cd /root/repo
fi
