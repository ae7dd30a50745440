#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_dense_model.py tests/test_gpu_linear.py tests/test_gpu_tp2.py -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
LEGS=gptq_qwen2 bash tools/gpu_call_w.sh 2>&1 | grep -E "dense|gptq|rope|cache|rms"
