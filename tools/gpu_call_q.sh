#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_q
mkdir -p $OUT
cd $R
TUNES=${TUNES:-"15=0;15=20000;15=10000;15=4000;15=0"} PF_BATCHES=${PF_BATCHES:-32} NO_PROBE=1 timeout 600 python tools/exp_decode_sweep.py > $OUT/sweep.log 2>&1
grep "mode=" $OUT/sweep.log
MODES=${MODES:-"15=10000"} bash tools/gpu_call_n.sh | grep -E "^==|value|qmm_gemm"
