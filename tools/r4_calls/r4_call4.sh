#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c4
mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_dense_model.py -m gpu -q -x -k "wide or chain or loop or moe or batch or tokens or prompt" > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
B32_STEPS=16 B32_AB="49=1;49=0;49=1;49=0" timeout 200 python tools/exp_b32.py 2>&1 | grep "tok/s" > $OUT/b32_fuse.log
cat $OUT/b32_fuse.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b32
B32_STEPS=6 timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_b32 --output-format csv -- python $R/tools/exp_b32.py > /tmp/b32_trace.log 2>&1
f=$(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_groups.py "$f" > $OUT/b32_groups.txt 2>&1
head -24 $OUT/b32_groups.txt
