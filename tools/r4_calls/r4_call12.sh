#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c12
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -k "wide or chain or moe or tokens" 2>&1 | tail -2
for i in 1 2; do B32_STEPS=16 timeout 100 python tools/exp_b32.py 2>&1 | grep "value" | cut -c1-100; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b32
B32_STEPS=6 timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_b32 --output-format csv -- python $R/tools/exp_b32.py > /tmp/b32_trace.log 2>&1
f=$(find /tmp/prof_b32 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_groups.py "$f" | grep "qw1_gemm" 
