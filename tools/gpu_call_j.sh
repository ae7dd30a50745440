#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_j
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "q8_0" > $OUT/q8.log 2>&1
tail -6 $OUT/q8.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_tp2.py -m gpu -q -k "loader" > $OUT/tp.log 2>&1
tail -8 $OUT/tp.log | cut -c1-300
