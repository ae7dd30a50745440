#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_e
mkdir -p $OUT
cd $R
PF_T=2048 PF_MODES=1 PF_QPG=0,7 PF_DBG=0 timeout 400 python tests/bench_prefill.py > $OUT/bench_prefill.log 2>&1
grep prefill $OUT/bench_prefill.log
