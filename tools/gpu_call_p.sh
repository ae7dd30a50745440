#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_p
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -m gpu -k "qmatmul or fused or model or decode or batch" 2>&1 | tail -4
TUNES=${TUNES:-"14=0;14=1;14=1,10=-256;14=0,10=-256;14=1,10=-384;14=1"} PF_BATCHES=32 NO_PROBE=1 timeout 600 python tools/exp_decode_sweep.py > $OUT/sweep.log 2>&1
grep "mode=" $OUT/sweep.log
