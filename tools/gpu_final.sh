#!/bin/bash
# Round-end record: every GPU test, smoke, the default bench line, the BASELINE configs' variants,
# a rocprofv3 kernel trace of the graph-mode loop (-> per-kernel stats + step timeline).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1 to=$2; shift 2; echo "=== $name" | tee -a $OUT/summary.log; timeout $to "$@" > $OUT/$name.log 2>&1; echo "exit $? : $(tail -n 3 $OUT/$name.log | tr '\n' '|' | cut -c1-900)" | tee -a $OUT/summary.log; }
: > $OUT/summary.log
run pytest_gpu 1800 python -u -m pytest tests -m gpu -q --timeout 900 -x
run smoke 600 python -u -c "import __graft_entry__ as g; g.smoke()"
run bench 1200 python -u bench.py
run bench_split 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM
run bench_v2_b4 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM_V2 --prompts-per-gpu 2
run bench_split_b8 600 python -u bench.py --cpu-steps 0 --attention SPLIT_EINSUM --prompts-per-gpu 4
run bench_torchrun1 600 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-steps 0
rm -rf $OUT/prof_final
(cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof_final -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --cpu-steps 0 > /root/repo/$OUT/rocprof_final.log 2>&1)
DB=$(find $OUT/prof_final -name "*.db" | head -n 1)
python tools/rocpd_stats.py $DB > $OUT/kernel_stats_final.csv 2>/dev/null
python tools/timeline.py $DB > $OUT/step_timeline_final.txt 2>&1
head -n 8 $OUT/step_timeline_final.txt
