#!/bin/bash
# per-op profiles under env variants: tools/r4_prof.sh <tag> "<ENV=VAL ...>" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=$1; shift; i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 python tools/op_profile.py $OUT/op_profile_${TAG}_$i.json 2 ORIGINAL > $OUT/op_profile_${TAG}_$i.txt 2>&1
  echo "== [$envs] $(sed -n 2p $OUT/op_profile_${TAG}_$i.txt)"
  grep -E "@8x8|@16x16" $OUT/op_profile_${TAG}_$i.txt | grep -E "conv3x3|groupnorm" | cut -c1-120
done
