#!/bin/bash
# session M: which of the in-sequence candidates hold up end to end (graph replay of the whole step)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { # name, table
  if [ -n "$2" ]; then export SD_PLAN_TABLE=$2; else unset SD_PLAN_TABLE; fi
  timeout 300 python bench.py --cpu-steps 0 --repeats 5 > $OUT/bench_m_$1.log 2> $OUT/bench_m_$1.err
  echo "$1: $(python -c "import json,sys; d=json.loads(open('$OUT/bench_m_$1.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats_ms_per_step'])" 2>&1 | tail -1)"
  tail -n 3 $OUT/bench_m_$1.err | cut -c1-300
}
run base ""
run A tools/tables/tab_A.inc
run B tools/tables/tab_B.inc
run C tools/tables/tab_C.inc
run base2 ""
