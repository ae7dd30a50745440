#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/call_k
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_tp2.py tests/test_gpu_engine.py tests/test_gpu_prefill.py -m gpu -q > $OUT/t.log 2>&1
tail -12 $OUT/t.log | cut -c1-300
