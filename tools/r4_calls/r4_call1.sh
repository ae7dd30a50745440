#!/bin/bash
# round 4, call 1: the three never-run kernels + the stream default through the suite, then their A/Bs
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4c1
mkdir -p $OUT
cd $R
MI355_EXPERIMENTS=1 timeout 300 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_ops.py -m gpu -q -k "lds_dma or fused_epilogue or stream" > $OUT/pytest_exp.log 2>&1
tail -15 $OUT/pytest_exp.log
MI355_EXPERIMENTS=1 MI355_TUNING=44:3,5:64 timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest_stream.log 2>&1
tail -15 $OUT/pytest_stream.log
B32_STEPS=16 B32_B1=1 B32_AB="5=0,44=1;5=64,44=3;5=64,44=4;5=0,44=1;5=64,44=3;5=64,44=4" timeout 150 python tools/exp_b32.py 2>&1 | grep "tok/s" > $OUT/b32_stream.log
cat $OUT/b32_stream.log
PF_T=2048 PF_MODES=1 PF_ATTN=0,1,0,1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "tok/s" > $OUT/prefill_attn_ab.log
MI355_TUNING=48:1 PF_T=2048 PF_MODES=1 PF_ATTN=0,1 timeout 200 python tests/bench_prefill.py 2>&1 | grep "tok/s" >> $OUT/prefill_attn_ab.log
cat $OUT/prefill_attn_ab.log
