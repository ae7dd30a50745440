#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -k "moe or Moe or mixtral or expert" 2>&1 | grep -E "passed|failed|rror" | tail -3
LEGS=mixtral_fp8 bash tools/gpu_call_w.sh 2>&1 | grep -E "moe_|mixtral"
