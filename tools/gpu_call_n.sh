#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
for cfgv in ${MODES:-"13=0" "2=1"}; do
  tag=$(echo $cfgv | tr ',=' '__')
  MI355_TUNE=$cfgv B32_STEPS=6 timeout 300 rocprofv3 --kernel-trace -d /tmp/rp_$tag --output-format csv -- python tools/exp_b32.py > $OUT/b32_$tag.log 2>&1
  f=$(find /tmp/rp_$tag -name "*kernel_trace.csv" | head -1)
  echo "== $cfgv"; grep -o "'value': [0-9.]*" $OUT/b32_$tag.log
  python tools/trace_groups.py $f | grep -v "at::native" | tee $OUT/groups_$tag.txt | head -30
done
