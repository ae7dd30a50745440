#!/bin/bash
# scratch: TP tests (two processes on one GPU: one-shot peer all-reduce) + model tests
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_b
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_tp2.py -m gpu -q -x -k "one_shot" > $OUT/p2p.log 2>&1
tail -15 $OUT/p2p.log
timeout 600 python -m pytest tests/test_gpu_tp2.py -m gpu -q > $OUT/tp2.log 2>&1
tail -15 $OUT/tp2.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_dense_model.py -m gpu -q > $OUT/model.log 2>&1
tail -8 $OUT/model.log
