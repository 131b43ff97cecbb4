cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/sk5.log; : > $O
for e in "SD_ATTN8_SK_VERBOSE=1" "SD_ATTN8_SK_LOCAL=0" "SD_ATTN8_SK_DBG=1" "SD_ATTN8_SK_DBG=2" "SD_ATTN8_SK_DBG=3"; do
  echo "== $e" >> $O
  env SD_TUNE=1 $e python tools/r6_attn_sk_bench.py 2>&1 | grep -v amdgpu.ids | sort -u | head -24 >> $O
done
