#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_m
mkdir -p $OUT
cd $R
TUNES="13=0;2=2,13=68;13=802;13=836;2=1,13=836;2=3,13=68;13=0" PF_BATCHES=32 NO_PROBE=1 timeout 600 python tools/exp_decode_sweep.py > $OUT/sweep.log 2>&1
grep "mode=" $OUT/sweep.log
