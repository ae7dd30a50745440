#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_h
mkdir -p $OUT
cd $R
TUNES="0=0;0=2;0=4;0=8;1=1;1=2;1=4;0=8,1=1;0=8,1=2;0=4,1=4;0=2,1=4" PF_BATCHES="1" NO_PROBE=1 timeout 800 python tools/exp_decode_sweep.py > $OUT/sweep.log 2>&1
grep -E "tok/s|rror" $OUT/sweep.log | tail -14
