#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_b32 --output-format csv -- python tools/exp_b32.py > $OUT/b32.log 2>&1
cp $(find /tmp/rp_b32 -name "*kernel_stats.csv" | head -1) $OUT/b32_kernel_stats.csv
grep "value" $OUT/b32.log | cut -c1-300
head -16 $OUT/b32_kernel_stats.csv | cut -c1-170
