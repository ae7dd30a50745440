#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c9
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
B32_STEPS=16 B32_AB="44=1;44=5;44=1;44=5" timeout 200 python tools/exp_b32.py 2>&1 | grep "tok/s" > $OUT/b32.log; cat $OUT/b32.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b32
B32_STEPS=6 timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_b32 --output-format csv -- python $R/tools/exp_b32.py > /tmp/b32_trace.log 2>&1
f=$(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_groups.py "$f" | grep -v "at::native" > $OUT/b32_groups.txt 2>&1
head -20 $OUT/b32_groups.txt
