#!/bin/bash
# does the HIP graph replay path cost time?  same bench under different launch modes / runtime knobs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python -u bench.py --steps 10 --warmup 2 --cpu-steps 0 $EXTRA 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'event_ms', d['roofline']['launch_ms'])"; }
EXTRA="" run A=1
EXTRA="--no-graph" run A=1
EXTRA="" run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
EXTRA="" run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
EXTRA="" run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
EXTRA="" run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
EXTRA="" run HIP_FORCE_DEV_KERNARG=1
EXTRA="" run GPU_MAX_HW_QUEUES=1
