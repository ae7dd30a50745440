#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
for cfgv in "12=9,11=7" "12=9,11=8" "12=9,11=0"; do
  tag=$(echo $cfgv | tr ',=' '__')
  MI355_TUNE=$cfgv B32_STEPS=8 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag --output-format csv -- python tools/exp_b32.py > $OUT/b32_$tag.log 2>&1
  cp $(find /tmp/rp_$tag -name "*kernel_stats.csv" | head -1) $OUT/b32_${tag}_stats.csv
  echo "== $cfgv"; grep "value" $OUT/b32_$tag.log | cut -c1-120
  grep -E "qpg_|qmm_gemm|qmm_epilogue|paged_attn" $OUT/b32_${tag}_stats.csv | cut -c1-150
done
